#!/bin/bash
{
python -m pytest tests/test_gpu_kernels.py -x -q -k "notes or heads or row_gru" 2>&1 | tail -5
python -m pytest tests/test_gpu_dead_steps.py tests/test_gpu_model.py -x -q -k "dead or composites_equal or full_config_vs or repeated_backward or two_runs or zero_skip or step_loop or free_running" 2>&1 | tail -5
python -m pytest tests/test_gpu_model_wide.py -x -q 2>&1 | tail -3
} > gpurun_out/r06_t5.txt 2>&1
cat gpurun_out/r06_t5.txt
for r in 1 2; do
  (cd _r5 && python scripts/ab_step.py ZERO_SKIP=True --rounds 2 2>&1 | grep -E "ms/step|rror" | sed 's/^/r5  /')
  python scripts/ab_step.py ZERO_SKIP=True --rounds 2 2>&1 | grep -E "ms/step|rror" | sed 's/^/r6  /'
done > gpurun_out/r06_vs_r05_c.txt
cat gpurun_out/r06_vs_r05_c.txt

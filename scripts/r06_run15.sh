#!/bin/bash
python -m pytest tests/test_gpu_dead_steps.py -x -q > gpurun_out/r06_t15.txt 2>&1
tail -30 gpurun_out/r06_t15.txt | cut -c1-220
python scripts/ab_step.py SORT_DEC_ROWS=False,True --rounds 3 2>&1 | grep -E "ms/step|rror" | tee gpurun_out/r06_ab_sort.txt

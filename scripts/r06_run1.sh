#!/bin/bash
# round-6 GPU check 1: the wgrad batch + folded bias sums: kernel tests, composite bit-identity, parity, then bench + kernel table
mkdir -p gpurun_out
{
python -m pytest tests/test_gpu_kernels.py -x -q -k "wgrad" 2>&1 | tail -5
python -m pytest tests/test_gpu_model.py -x -q -k "composites_equal or full_config_vs or decoder_tf_composite or repeated_backward or zero_skip or sibling" 2>&1 | tail -8
python -m pytest tests/test_gpu_dead_steps.py tests/test_gpu_model_wide.py -x -q 2>&1 | tail -8
} > gpurun_out/r06_t1.txt 2>&1
python bench.py --no-extras --no-cpu-baseline > gpurun_out/r06_b1.json 2> gpurun_out/r06_b1.err
tail -c 400 gpurun_out/r06_b1.err
cat gpurun_out/r06_t1.txt
python - <<'PY'
import json
r=json.load(open('gpurun_out/r06_b1.json'))
print(r['value'], r['ms_per_step'], r['host_enqueue_ms_per_step'])
ro=r['roofline']
print(json.dumps({k:v for k,v in ro.items() if k not in ('also','note','operand_bytes_note')})[:900])
for a in ro['also']: print(a['kernel'][:60], a.get('achieved'), a.get('frac'), a.get('ms_per_step'))
print(json.dumps(r.get('parity',{}).get('benched')))
PY

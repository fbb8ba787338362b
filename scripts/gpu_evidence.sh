#!/bin/bash
# round-6 evidence: bench (full line), kernel tables (headline, free-running, fp32), PMC passes
python bench.py > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err
tail -c 300 gpurun_out/r06_bench.err
python - <<'PY'
import json
r=json.load(open('gpurun_out/r06_bench.json'))
print('VALUE', r['value'], r['ms_per_step'], 'host', r['host_enqueue_ms_per_step'])
ro=r['roofline']; print('ROOF', ro['kernel'][:50], ro['achieved'], ro['frac'], ro['ms_per_step'], ro.get('launches_per_step'))
for a in ro['also']: print('  also', a['kernel'][:50], a.get('achieved'), a.get('frac'), a.get('ms_per_step'))
for k,v in r.get('extra',{}).items(): print(' ', k, json.dumps(v)[:260])
print(json.dumps(r.get('parity',{}).get('benched')))
print(json.dumps(r.get('cpu_baseline',{}))[:300])
PY
bash scripts/gpu_profile.sh r06_c --no-extras > gpurun_out/r06_c_head.txt 2>&1
bash scripts/gpu_profile.sh r06_c_tfr0 --tfr 0 --no-extras > gpurun_out/r06_c_tfr0_head.txt 2>&1
bash scripts/gpu_profile.sh r06_c_fp32 --precision fp32 --no-extras > gpurun_out/r06_c_fp32_head.txt 2>&1
rm -f gpurun_out/r06_c*/r_results.db
PTV_COMMIT=$(cat .commit_stamp 2>/dev/null) bash scripts/gpu_pmc.sh r06_pmc > gpurun_out/r06_pmc_head.txt 2>&1
head -12 gpurun_out/r06_c_head.txt | cut -c1-200; head -30 gpurun_out/r06_c_fp32_head.txt | cut -c1-170

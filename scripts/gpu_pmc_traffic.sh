#!/bin/bash
# HBM traffic of the dominant kernel (notes-GRU forward step) from PMC counters, separate passes
# (FETCH_SIZE and WRITE_SIZE do not fit one pass; MI355X_MICROARCH.md §rocprofv3 PMC slots)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_traffic
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_traffic -o $c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_traffic/$c.log 2>&1
done
python3 - <<'PY'
import csv, glob, json
out = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    f = glob.glob('gpurun_out/pmc_traffic/%s_counter_collection.csv' % c)[0]
    rows = list(csv.DictReader(open(f)))
    if c == 'FETCH_SIZE':
        print(list(rows[0].keys()))
    ker = [r for r in rows if 'gru_fwd_step_kernel<ptv::BF16, 128, 64' in r['Kernel_Name'] and r['Counter_Name'] == c]
    # notes GRU: grid 8 x 128 workgroups of 256 threads
    sel = [r for r in ker if int(r.get('Grid_Size', r.get('Grid_Size_X', 0))) in (8 * 128 * 256, 8 * 256)]
    per = {}
    for r in sel:
        per.setdefault(r['Dispatch_Id'], 0.0)
        per[r['Dispatch_Id']] += float(r['Counter_Value'])
    vals = list(per.values())
    print(c, 'dispatches', len(vals), 'mean', sum(vals) / max(1, len(vals)))
    out[c] = {'dispatches': len(vals), 'mean_counter': sum(vals) / max(1, len(vals))}
json.dump(out, open('gpurun_out/pmc_traffic/summary.json', 'w'), indent=1)
PY

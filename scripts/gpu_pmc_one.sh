#!/bin/bash
# wave-state counters of one micro-benchmark command:  bash scripts/gpu_pmc_one.sh <tag> <kernel-substring> <python script + args...>
tag=$1; pat=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE"; do
  n=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $set --output-format csv -d gpurun_out/$tag -o $n -- python3 "$@" > gpurun_out/$tag/$n.log 2>&1
done
python3 - "$tag" "$pat" <<'PY'
import csv, glob, sys, collections
tag, pat = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob('gpurun_out/%s/*_counter_collection.csv' % tag):
    for r in csv.DictReader(open(f)):
        if pat in r['Kernel_Name']:
            tot[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
disp = max(1, len(set()))
for k in sorted(tot):
    print('%-24s %.4g  (rows %d)' % (k, tot[k], n[k]))
PY
rm -f gpurun_out/$tag/*_counter_collection.csv

#!/bin/bash
# wave-state / instruction-mix / LDS / HBM counters of the kernels matching <kernel-substring> in one command
# usage (GPU box): bash scripts/gpu_pmc_kernel.sh <tag> <kernel-substring> <python script + args...>
tag=$1; pat=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d gpurun_out/$tag -o set$i -- python3 "$@" > gpurun_out/$tag/set$i.log 2>&1
done
python3 - "$tag" "$pat" <<'PY'
import csv, glob, sys, collections, json
tag, pat = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(float); disp = collections.defaultdict(set)
for f in glob.glob('gpurun_out/%s/*_counter_collection.csv' % tag):
    for r in csv.DictReader(open(f)):
        if pat in r['Kernel_Name']:
            tot[r['Counter_Name']] += float(r['Counter_Value']); disp[r['Counter_Name']].add(r['Dispatch_Id'])
out = {k: {'per_launch': tot[k] / max(1, len(disp[k])), 'launches': len(disp[k])} for k in sorted(tot)}
json.dump({'kernel_match': pat, 'counters': out}, open('gpurun_out/%s/pmc_%s.json' % (tag, pat), 'w'), indent=1)
for k in sorted(tot):
    print('%-28s %14.4g per launch  (%d launches)' % (k, out[k]['per_launch'], out[k]['launches']))
PY
rm -f gpurun_out/$tag/*_counter_collection.csv

#!/bin/bash
# HBM traffic (FETCH_SIZE, WRITE_SIZE: separate passes) and MFMA-busy cycles of every kernel of one bench run, grouped by kernel.
# usage (GPU box): bash scripts/gpu_pmc.sh <tag>      -> gpurun_out/<tag>/pmc_by_kernel.json, row_gru_pmc.json
tag=$1
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_MFMA"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $c --output-format csv -d gpurun_out/$tag -o pass$i -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras --no-parity > gpurun_out/$tag/pass$i.log 2>&1
done
python3 scripts/pmc_summary.py gpurun_out/$tag
rm -f gpurun_out/$tag/*_counter_collection.csv

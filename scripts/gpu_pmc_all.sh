#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of EVERY kernel of one bench run, grouped by kernel + grid
# usage (GPU box): bash scripts/gpu_pmc_all.sh <tag>
tag=$1
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c --output-format csv -d gpurun_out/$tag -o $c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/$tag/$c.log 2>&1
done
python3 scripts/pmc_summary.py gpurun_out/$tag
rm -f gpurun_out/$tag/*_counter_collection.csv

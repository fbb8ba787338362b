#!/bin/bash
# usage (on the GPU box, via gpurun): bash scripts/gpu_timeline_eager.sh <tag> [bench args...]
# kernel trace of the EAGER step -> gpurun_out/<tag>/timeline.txt.  The profiler's per-launch host cost makes the B = 512 step
# host-bound (12 ms): pass --batch 1024 to see the eager stream concurrency with the GPU as the bottleneck.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
timeout 400 rocprofv3 --kernel-trace -d gpurun_out/$tag -o r -- python3 bench.py --no-cpu-baseline --no-parity --no-extras --steps 5 --warmup 2 "$@" > gpurun_out/$tag/bench.log 2>&1
grep '"metric"' gpurun_out/$tag/bench.log | cut -c1-200
python3 scripts/timeline.py gpurun_out/$tag/r_results.db --min-us 10 > gpurun_out/$tag/timeline.txt
rm -f gpurun_out/$tag/r_results.db
wc -l gpurun_out/$tag/timeline.txt

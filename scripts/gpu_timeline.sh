#!/bin/bash
# usage (on the GPU box, via gpurun): bash scripts/gpu_timeline.sh <tag> [B]
# kernel trace of the graph-replayed train step -> gpurun_out/<tag>/timeline.txt (one step: per-kernel start / duration / stream)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
timeout 400 rocprofv3 --kernel-trace -d gpurun_out/$tag -o r -- python3 scripts/graph_replay.py "$@" > gpurun_out/$tag/run.log 2>&1
tail -2 gpurun_out/$tag/run.log
python3 scripts/timeline.py gpurun_out/$tag/r_results.db --min-us 10 > gpurun_out/$tag/timeline.txt
rm -f gpurun_out/$tag/r_results.db
wc -l gpurun_out/$tag/timeline.txt

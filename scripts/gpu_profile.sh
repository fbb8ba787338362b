#!/bin/bash
# usage (on the GPU box, via gpurun): bash scripts/gpu_profile.sh <tag> [bench args...]
# kernel trace of bench.py -> gpurun_out/<tag>/ (rocpd db + markdown summary)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
# NSTEPS = steps + warmup (for the per-step figures of the summary); default run: 5 + 2
case " $* " in *" --steps "*) args="$*";; *) args="--steps 5 --warmup 2 $*";; esac
timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/$tag -o r -- python3 bench.py --no-cpu-baseline --no-parity $args > gpurun_out/$tag/bench.log 2>&1
grep '"metric"' gpurun_out/$tag/bench.log | cut -c1-330
python3 scripts/rocpd_summary.py gpurun_out/$tag/r_results.db gpurun_out/$tag/summary.md ${NSTEPS:-7} "$tag" "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-parity $args" | head -34 | cut -c1-200

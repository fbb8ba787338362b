cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06_tl
timeout 400 rocprofv3 --kernel-trace -d gpurun_out/r06_tl -o r -- python3 bench.py --steps 4 --warmup 2 --no-extras --no-cpu-baseline --no-parity > gpurun_out/r06_tl/run.log 2>&1
python3 scripts/timeline.py gpurun_out/r06_tl/r_results.db --min-us 8 > gpurun_out/r06_tl/timeline.txt
rm -f gpurun_out/r06_tl/r_results.db
wc -l gpurun_out/r06_tl/timeline.txt

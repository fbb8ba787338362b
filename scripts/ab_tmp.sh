cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02c
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r02c -o bench -- python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r02c/bench.log 2>&1
grep '"metric"' gpurun_out/r02c/bench.log | cut -c1-1500
rm -f $(find gpurun_out/r02c -name "*kernel_trace.csv")
timeout 900 python bench.py > gpurun_out/r02c/bench_full.log 2>&1
tail -1 gpurun_out/r02c/bench_full.log

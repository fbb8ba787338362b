set -x
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "row_gru" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r02b
cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r02b -o bench -- python bench.py --no-extras --steps 20 --warmup 5 > gpurun_out/r02b/bench.log 2>&1
tail -1 gpurun_out/r02b/bench.log | cut -c1-300
find gpurun_out/r02b -name "*kernel_stats*" | head
rm -f $(find gpurun_out/r02b -name "*kernel_trace.csv") 

cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02f
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r02f -o bench -- python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r02f/bench.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/r02f -o tfr0 -- python bench.py --no-extras --no-cpu-baseline --tfr 0 --steps 4 --warmup 2 > gpurun_out/r02f/tfr0.log 2>&1
rm -f $(find gpurun_out/r02f -name "*kernel_trace.csv") $(find gpurun_out/r02f -name "*_stats.csv")
bash scripts/gpu_pmc_r02.sh r02pmc2 > gpurun_out/r02f/pmc.log 2>&1
timeout 900 python bench.py > gpurun_out/r02f/bench_full.log 2>&1
tail -1 gpurun_out/r02f/bench_full.log | cut -c1-400

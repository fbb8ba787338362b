timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "wgrad" 2>&1 | tail -5
for i in 1 2; do
PTV_WGRAD_BIAS=0 timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 30 --warmup 8 2>&1 | tail -1 | cut -c1-200
PTV_WGRAD_BIAS=1 timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 30 --warmup 8 2>&1 | tail -1 | cut -c1-200
done
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5

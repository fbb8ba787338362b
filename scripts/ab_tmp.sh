timeout 900 python -m pytest tests/test_gpu_model.py -x -q 2>&1 | tail -5
PTV_FREE_REPLAY=0 timeout 300 python bench.py --no-extras --no-cpu-baseline --tfr 0 --steps 6 --warmup 3 2>&1 | tail -1 | cut -c1-200
PTV_FREE_REPLAY=1 timeout 300 python bench.py --no-extras --no-cpu-baseline --tfr 0 --steps 6 --warmup 3 2>&1 | tail -1 | cut -c1-200

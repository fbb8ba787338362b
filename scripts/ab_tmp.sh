timeout 900 python -m pytest tests/test_gpu_model.py -x -q 2>&1 | tail -3
timeout 300 python bench.py --no-extras --no-cpu-baseline --tfr 0 --steps 6 --warmup 3 2>&1 | tail -1 | cut -c1-200
timeout 300 python bench.py --no-extras --no-cpu-baseline --mode decode --batch 2048 --graph --steps 6 --warmup 3 2>&1 | tail -1 | cut -c1-200

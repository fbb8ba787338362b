timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "notes_gru or row_gru" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_model.py -x -q 2>&1 | tail -3
for i in 1 2; do
timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 30 --warmup 8 2>&1 | tail -1 | cut -c1-200
done
timeout 300 python bench.py --no-extras --no-cpu-baseline --tfr 0 --steps 6 --warmup 3 2>&1 | tail -1 | cut -c1-200

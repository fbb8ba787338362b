set -x
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "row_gru or notes_gru" 2>&1 | tail -5
for i in 1 2; do
PTV_ROW_GRU128=0 timeout 300 python bench.py --no-extras --steps 30 --warmup 8 2>&1 | tail -1 | cut -c1-200
PTV_ROW_GRU128=1 timeout 300 python bench.py --no-extras --steps 30 --warmup 8 2>&1 | tail -1 | cut -c1-200
done

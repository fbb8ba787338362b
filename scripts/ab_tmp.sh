timeout 900 python scripts/bench_wgrad.py 2>&1 | tail -21 | cut -c1-100
for i in 1 2; do
timeout 300 python bench.py --no-extras --no-cpu-baseline --steps 30 --warmup 8 2>&1 | tail -1 | cut -c1-200
done

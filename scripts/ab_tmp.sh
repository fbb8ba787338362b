for n in 2 4 2 4; do echo NSET $n; PTV_WGRAD_NSET=$n timeout 900 python scripts/bench_wgrad.py 2>&1 | tail -21 | cut -c1-75; done

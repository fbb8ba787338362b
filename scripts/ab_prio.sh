for pr in 0,0,0,0 0,0,0,1 -1,-1,-1,0 -1,-1,-1,1 0,-1,0,1; do
  echo "== PTV_POOL_PRIO=$pr"; PTV_POOL_PRIO=$pr timeout 200 python scripts/ab_step.py PERSIST_SPLITK=2 --rounds 2 2>&1 | grep ms/step
done

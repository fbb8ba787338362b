#!/bin/bash
# usage: bash scripts/ab_env2.sh "VAR=a VAR2=b" "VAR=c" ...   -- step time under each environment setting (fresh process each)
for e in "$@"; do
  echo "== $e"; env $e timeout 200 python scripts/ab_step.py HEADS_FUSED=True --rounds 2 2>&1 | grep ms/step
done

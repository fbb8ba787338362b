#!/bin/bash
# usage: bash scripts/ab_env.sh VAR v1 v2 ...   -- the B=512 step time under each value of an environment switch (fresh process each)
var=$1; shift
for v in "$@"; do
  echo "== $var=$v"; env $var=$v timeout 200 python scripts/ab_step.py HEADS_FUSED=True --rounds 2 2>&1 | grep ms/step
done

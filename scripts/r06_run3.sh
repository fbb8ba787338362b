#!/bin/bash
# same-box: round-5 tree (_r5/) vs this tree, interleaved
for r in 1 2; do
  (cd _r5 && python scripts/ab_step.py ZERO_SKIP=True --rounds 2 2>&1 | grep -E "ms/step|rror" | sed 's/^/r5  /')
  python scripts/ab_step.py ZERO_SKIP=True --rounds 2 2>&1 | grep -E "ms/step|rror" | sed 's/^/r6  /'
done > gpurun_out/r06_vs_r05.txt
cat gpurun_out/r06_vs_r05.txt

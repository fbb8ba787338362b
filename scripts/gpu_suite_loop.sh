#!/bin/bash
# Runs the GPU suite N times in fresh processes (the driver's exact command) and records the pass counts under gpurun_out/.
N=${1:-5}
mkdir -p gpurun_out
: > gpurun_out/suite_loop.txt
for i in $(seq 1 $N); do
  timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -4 > gpurun_out/suite_run_$i.txt
  echo "run $i: $(tail -1 gpurun_out/suite_run_$i.txt)" >> gpurun_out/suite_loop.txt
done
cat gpurun_out/suite_loop.txt

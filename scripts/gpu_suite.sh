#!/bin/bash
# the full GPU test suite (usage on a GPU box: bash scripts/gpu_suite.sh)
python -m pytest tests -m gpu -x -q > gpurun_out/gpu_suite.txt 2>&1
tail -4 gpurun_out/gpu_suite.txt

#!/bin/bash
python -m pytest tests -m gpu -x -q > gpurun_out/r06_suite1.txt 2>&1
tail -6 gpurun_out/r06_suite1.txt
cat gpurun_out/bf16_horizon.txt

#!/bin/bash
python -m pytest tests -m gpu -x -q > gpurun_out/r06_suite2.txt 2>&1
tail -4 gpurun_out/r06_suite2.txt

#!/bin/bash
PTV_BWD_COMPOSITES=0 python scripts/trace_marks.py > gpurun_out/r06_trace_marks_py.txt 2>&1
python scripts/trace_marks.py > gpurun_out/r06_trace_marks.txt 2>&1
cat gpurun_out/r06_trace_marks.txt | tail -40
bash scripts/gpu_profile.sh r06_b --no-extras > gpurun_out/r06_b_head.txt 2>&1
head -48 gpurun_out/r06_b_head.txt | cut -c1-170

#!/bin/bash
python scripts/trace_calls.py 0 1.7 > gpurun_out/r06_trace_calls_fwd.txt 2>&1
cat gpurun_out/r06_trace_calls_fwd.txt | cut -c1-120
python scripts/trace_marks.py > gpurun_out/r06_trace_marks.txt 2>&1
tail -32 gpurun_out/r06_trace_marks.txt

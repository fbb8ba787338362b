#!/bin/bash
python -m pytest tests/test_gpu_model.py -x -q -k "decoder_free_composite or free_running or step_loop or inference_decode or graph_captured_free or persistent_step_loop or two_runs or config4" > gpurun_out/r06_t10.txt 2>&1
tail -25 gpurun_out/r06_t10.txt
python -m pytest tests/test_gpu_model_wide.py tests/test_gpu_model_large.py -x -q 2>&1 | tail -4

#!/bin/bash
python -m pytest tests/test_gpu_model.py -x -q -k "decoder_free_composite or config4" > gpurun_out/r06_t10.txt 2>&1
tail -30 gpurun_out/r06_t10.txt | cut -c1-200

#!/bin/bash
python scripts/ab_step.py WGRAD_BATCH=0,1,2,3 --rounds 3 2>&1 | grep -E "ms/step|Error|error" > gpurun_out/r06_ab_batch.txt
cat gpurun_out/r06_ab_batch.txt

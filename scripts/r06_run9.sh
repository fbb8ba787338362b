#!/bin/bash
{
python scripts/ab_step.py SHADOW_T_SLOT=7,5,6 --rounds 3 2>&1 | grep -E "ms/step|rror"
python scripts/ab_step.py SUMMARY_AFTER_PERSIST=False,True --module ptvae --rounds 3 2>&1 | grep -E "ms/step|rror"
python scripts/ab_step.py EMBED_MH_SLOT=7,5,6 --rounds 3 2>&1 | grep -E "ms/step|rror"
} > gpurun_out/r06_ab_fwd_sched.txt 2>&1
cat gpurun_out/r06_ab_fwd_sched.txt

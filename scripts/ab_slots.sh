#!/bin/bash
# usage: bash scripts/ab_slots.sh   -- B=512 step time under stream-slot assignments (fresh process each); pool streams 2 and 3 share a hardware queue
run() { echo "== $*"; env "$@" timeout 300 python scripts/ab_step.py HEADS_FUSED=True --rounds 2 2>&1 | grep ms/step; }
run PTV_NOP=1
run PTV_ROW_TURNS=0
run PTV_BIGRU_CHAIN_FIRST=0 PTV_ROW_TURNS=0
run PTV_RHY_ENC_SLOT=4
run PTV_RHY_ENC_SLOT=4 PTV_SUMMARY_SLOT=-1
run PTV_RHY_ENC_SLOT=4 PTV_SUMMARY_SLOT=-1 PTV_EMB_FIRST=1
run PTV_LATE_SLOTS=1,0
run PTV_LATE_SLOTS=3

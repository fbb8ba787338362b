#!/bin/bash
bash scripts/gpu_profile.sh r06_d --no-extras > gpurun_out/r06_d_head.txt 2>&1
grep -E "notes_fwd|notes_bwd|heads_|dur_gru|rows_by_index|rows_by_length|pianotree|sum_steps|gemm_plain_kernel<ptv::BF16, 128, 128, 2, 2, false, false, (true|false), true>|ce_vec|Total kernel|Weight-grad" gpurun_out/r06_d/summary.md | cut -c1-150

// version.hip -- build identification of libptvae_hip.so
#include "../../include/ptvae_hip.h"
extern "C" const char* ptv_arch(void) { return "gfx950"; }
// 2 (round 3): ptv_step_params / ptv_ordered_reductions / ptv_wgrad_mode added; ptv_gemm dtypes bit 3 (column-blocked C); the
// row-partitioned GRU pair takes gc / ext column-blocked and keeps its gate planes unit-blocked
// 3 (round 4): ptv_dur_gru_bwd takes b_hh / tab0 / tab (gates may be NULL: recompute); ptv_txt_conv_relu_pool_*_rows take the arg-max
// map; ptv_*_geom, ptv_heads_*, ptv_pack_mfma_b2 / _multi, ptv_gru_persist_bwd_splitk, ptv_gradnorm_clip_adam_step and the composites
// ptv_decoder_tf_fwd / ptv_chord_decoder_fwd added
// 4 (round 5): ptv_notes_gru_persist_fwd is the wave-role kernel (pairs = 0 packs, gc / gate planes unit-blocked by 16, h0 read only: no
// fp32 states written); ptv_notes_gru_persist_bwd / ptv_row_gru_persist_bwd(H = 512) take the bf16 states; ptv_gemm dtypes bit 4 (C
// column-blocked by 16); ptv_pianotree_targets writes counts[3] (since round 4, unversioned then); ptv_debug_notes_trace added
// 5 (round 5): ptv_row_gru_persist_{fwd,bwd}_perm and ptv_rows_by_length added (panels of rows sorted by length)
// 6 (round 6): the ptv_*_top entry points / PTV_DTF_LIVE_TOP (round 5, unversioned then), the bwd / loss / bigru composites, row limits in
// ptv_wgrad's guarded tail and ptv_dur_out_wgrad, PTV_BGF_D_W_IH_F32; debug / profiling entry points moved to ptvae_hip_debug.h;
// ptv_header_hash added (the loader compares it with the headers it binds from)
extern "C" int ptv_abi_version(void) { return 6; }
#ifndef PTV_HEADER_HASH
#define PTV_HEADER_HASH "unknown"
#endif
extern "C" const char* ptv_header_hash(void) { return PTV_HEADER_HASH; }

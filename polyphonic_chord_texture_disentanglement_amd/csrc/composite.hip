// composite.hip -- whole sub-graphs of the train step behind ONE C entry point: the launch sequence, the persistent launch's turn
// (an event wait / record pair) and every shape decision live here in C++ host code; the caller owns the tensors.
//
//   ptv_decoder_tf_fwd   PtvaeDecoder.decoder, teacher-forced, forward (ptvae.py:430-496 with decode_notes :370-428 and decode_note
//                        :336-368 restructured to 32 + 15 + 5 sequential steps): 15 launches
//
// bf16 precision at the init_model() sizes (the configuration every persistent / fused kernel below is specialised for); anything
// else returns PTV_ERR_UNSUPPORTED BEFORE the first launch and the caller sequences the generic entry points itself.
#include "common.hpp"
#include "../../include/ptvae_hip.h"

namespace {

inline const void* T_(const void* const* t, int i) { return t[i]; }
template <typename X> inline X* M_(const void* const* t, int i) { return reinterpret_cast<X*>(const_cast<void*>(t[i])); }

// dtypes word of ptv_gemm: bit 0 = A bf16, bit 1 = B bf16, bit 2 = C bf16, bit 4 = C column-blocked by 16
constexpr int A16 = 1, B16 = 2, C16 = 4, CBLK16 = 16;

// The parameter-gradient products that become ready at one point of a backward pass, as ONE ptv_wgrad_batch call (one product launch + one
// reduction launch) with every bias gradient taken from the A tiles while they are in LDS (colsum_a) instead of a second pass over the
// gradient matrix.  add(): C[M,N] += A^T B (A [K,M], B [K,N] row-per-sample), csum (or NULL) [M] += column sums of A.  Products the
// weight-gradient kernel does not take (fp32 precision, K < 512: ptv_gemm's own rule) run at once as ptv_gemm + ptv_colsum, as before.
// The Python sequencing (functional.py) makes the same calls one by one: ptv_wgrad with the same colsum_a -- the same bits.
struct WgradGroup {
  ptv_wgrad_job jobs[8]; int n = 0; int P; void* stream;
  WgradGroup(int prec, void* st) : P(prec), stream(st) {}
  int add(int M, int N, long K, const void* A, long lda, int a_bf16, const void* B, long ldb, int b_bf16, float* C, long ldc, float* csum,
          const int* k_top = nullptr, long k_unit = 0, int k_rev = 0, int accumulate = 1, const int* seg_n = nullptr, long seg_unit = 0,
          int seg_period = 0) {
    if (P == PTV_PREC_BF16 && K >= 512) {
      if (n == 8) PTV_TRY(flush());
      jobs[n++] = ptv_wgrad_job{M, N, (int)K, A, lda, B, ldb, C, ldc, 1.f, accumulate, (a_bf16 ? 1 : 0) | (b_bf16 ? 2 : 0), 0, csum, k_top, k_unit, k_rev,
                                seg_n, seg_n ? seg_unit : 0, seg_n ? seg_period : 0};
      return PTV_OK;
    }
    if (k_top) return PTV_ERR_ARG;
    PTV_TRY(ptv_gemm(P, 1, 1, M, N, (int)K, A, lda, B, ldb, C, ldc, nullptr, 1.f, accumulate, 0, 0, (a_bf16 ? A16 : 0) | (b_bf16 ? B16 : 0), stream));
    if (csum) PTV_TRY(ptv_colsum(csum, A, lda, K, M, nullptr, 1, a_bf16, stream));
    return PTV_OK;
  }
  int flush() {
    if (n) { const int q = n; n = 0; return ptv_wgrad_batch(jobs, q, stream); }
    return PTV_OK;
  }
};

}  // namespace

extern "C" int ptv_decoder_tf_supported(const long* d) {
  if (!d) return 0;
  const long B = d[PTV_DTF_D_B], E = d[PTV_DTF_D_E], He = d[PTV_DTF_D_HE], Ht = d[PTV_DTF_D_HT], Hn = d[PTV_DTF_D_HN], Hd = d[PTV_DTF_D_HD],
             NP = d[PTV_DTF_D_NP];
  if (B <= 0 || E != 128 || Hn != 512 || Hd != 64 || NP != 130 || He <= 0 || (He & 7) || (Ht & 7)) return 0;
  return ptv_gru_persist_supported(1, (int)B, (int)Ht) ? 1 : 0;
}

extern "C" int ptv_decoder_tf_fwd(const void* const* t, const long* d, void* stream) {
  if (!t || !d) return PTV_ERR_ARG;
  if (!ptv_decoder_tf_supported(d)) return PTV_ERR_UNSUPPORTED;
  for (int i = 0; i < PTV_DTF_COUNT; i++)
    if (!t[i] && i != PTV_DTF_FORCE_DUR && i != PTV_DTF_GATES_D && i != PTV_DTF_WAIT_EVENT && i != PTV_DTF_RECORD_EVENT && i != PTV_DTF_LIVE_TOP &&
        i != PTV_DTF_PERM && i != PTV_DTF_ROW_LEN && i != PTV_DTF_NS16S && i != PTV_DTF_TOK_S && i != PTV_DTF_SEG_N)
      return PTV_ERR_ARG;
  // rows sorted by length: all four slots or none, and only with a live-step limit (the consumers of the unwritten rows need one)
  const bool sorted = t[PTV_DTF_PERM] != nullptr;
  if (sorted != (t[PTV_DTF_ROW_LEN] != nullptr) || sorted != (t[PTV_DTF_NS16S] != nullptr) || sorted != (t[PTV_DTF_TOK_S] != nullptr) ||
      (sorted && !t[PTV_DTF_LIVE_TOP]))
    return PTV_ERR_ARG;
  const int B = (int)d[PTV_DTF_D_B], E = (int)d[PTV_DTF_D_E], He = (int)d[PTV_DTF_D_HE], Ht = (int)d[PTV_DTF_D_HT], Hn = (int)d[PTV_DTF_D_HN],
            Hd = (int)d[PTV_DTF_D_HD], Zs = (int)d[PTV_DTF_D_ZS], Zi = (int)d[PTV_DTF_D_ZI];
  const long ldp = d[PTV_DTF_D_LDP];
  const int R = 32 * B;                                        // rows (time step, sample) of the notes GRU
  const long M = 15L * R;                                      // rows (note, time step, sample) of the heads / the duration GRU
  const int P = PTV_PREC_BF16;
  hipStream_t s = (hipStream_t)stream;
  float* NS = M_<float>(t, PTV_DTF_NS);
  __bf16* NS16 = M_<__bf16>(t, PTV_DTF_NS16);
  float* TOKS = M_<float>(t, PTV_DTF_TOKS);
  float* HN = M_<float>(t, PTV_DTF_HN);
  __bf16* HN16 = M_<__bf16>(t, PTV_DTF_HN16);
  float* HD = M_<float>(t, PTV_DTF_HD);
  __bf16* HD16 = M_<__bf16>(t, PTV_DTF_HD16);
  const __bf16* w_ih_t = (const __bf16*)T_(t, PTV_DTF_W16_IH_T);      // [3Ht][2He + Zi]
  const __bf16* w_ih_n = (const __bf16*)T_(t, PTV_DTF_W16_IH_N);      // [3Hn][Ht + E]
  const long ld_t = 2L * He + Zi, ld_n = (long)Ht + E;

  // ---- z -> initial time state, z_in (ptvae.py:435-437)
  PTV_TRY(ptv_gemm(P, 0, 0, B, Ht, Zs, T_(t, PTV_DTF_Z), Zs, T_(t, PTV_DTF_W16_ZHID), Zs, NS, Ht, (const float*)T_(t, PTV_DTF_B_ZHID), 1.f, 0, 0, 0,
                   B16, stream));
  PTV_TRY(ptv_gemm(P, 0, 0, B, Zi, Zs, T_(t, PTV_DTF_Z), Zs, T_(t, PTV_DTF_W16_ZIN), Zs, M_<void>(t, PTV_DTF_Z_IN), Zi,
                   (const float*)T_(t, PTV_DTF_B_ZIN), 1.f, 0, 0, 0, B16, stream));
  // ---- time-GRU inputs: token_t = [init ; summary_{t-1}], z_in broadcast over t (ptvae.py:457-462,476-478)
  PTV_TRY(ptv_copy2d(TOKS, 2L * He, (const float*)T_(t, PTV_DTF_INIT_INPUT), 0, B, 2 * He, 1.f, 0, stream));
  PTV_TRY(ptv_copy2d(TOKS + (long)B * 2 * He, 2L * He, (const float*)T_(t, PTV_DTF_XS), 2L * He, R, 2 * He, 1.f, 0, stream));
  PTV_TRY(ptv_gemm(P, 0, 0, R, 3 * Ht, 2 * He, TOKS, 2L * He, w_ih_t, ld_t, M_<void>(t, PTV_DTF_GI_T), 3L * Ht, nullptr, 1.f, 0, 0, -1,
                   B16 | C16, stream));
  PTV_TRY(ptv_gemm(P, 0, 0, B, 3 * Ht, Zi, T_(t, PTV_DTF_Z_IN), Zi, w_ih_t + 2 * He, ld_t, M_<void>(t, PTV_DTF_ZG), 3L * Ht,
                   (const float*)T_(t, PTV_DTF_B_IH_T), 1.f, 0, 0, -1, B16 | C16, stream));
  // ---- the 32 time steps: ONE persistent launch; persistent launches take turns (never two spinning grids half-resident together)
  {
    const void* gi[1] = {T_(t, PTV_DTF_GI_T)}; const long gi_step[1] = {(long)B * 3 * Ht}, gi_ld[1] = {3L * Ht};
    const void* gi2[1] = {T_(t, PTV_DTF_ZG)}; const long gi2_step[1] = {0}, gi2_ld[1] = {3L * Ht};
    const void* w16[1] = {T_(t, PTV_DTF_W16_HH_T)}; const float* bhh[1] = {(const float*)T_(t, PTV_DTF_B_HH_T)};
    float* hall[1] = {NS}; void* hall16[1] = {NS16}; void* gates[1] = {M_<void>(t, PTV_DTF_GATES_T)};
    const int* lengths[1] = {nullptr}; const int rev[1] = {0}; void* xch[1] = {M_<void>(t, PTV_DTF_XCH)};
    if (t[PTV_DTF_WAIT_EVENT] && hipStreamWaitEvent(s, (hipEvent_t)const_cast<void*>(t[PTV_DTF_WAIT_EVENT]), 0) != hipSuccess) return PTV_ERR_LAUNCH;
    PTV_TRY(ptv_gru_persist_fwd(1, B, Ht, 32, gi, gi_step, gi_ld, gi2, gi2_step, gi2_ld, w16, bhh, hall, hall16, gates, lengths, rev, xch,
                                M_<unsigned>(t, PTV_DTF_SYNC), stream));
    if (t[PTV_DTF_RECORD_EVENT] && hipEventRecord((hipEvent_t)const_cast<void*>(t[PTV_DTF_RECORD_EVENT]), s) != hipSuccess) return PTV_ERR_LAUNCH;
  }
  // ---- notes GRU: h0 = dec_time_to_notes_hid(summary_t); input [summary_t | token], summary part hoisted (ptvae.py:374-398)
  const __bf16* nsf = NS16 + (long)B * Ht;                     // the 32 time states as rows (t, b): an MFMA operand as they are
  const float* tok = (const float*)T_(t, PTV_DTF_EMB);
  const int* perm = (const int*)T_(t, PTV_DTF_PERM);
  const int* row_len = (const int*)T_(t, PTV_DTF_ROW_LEN);
  if (sorted) {
    // the decoder's rows from here on in the order PERM (descending number of live note steps): time states and fed tokens are gathered
    // once, everything below is row-wise and does not care which (t, b) a row is
    PTV_TRY(ptv_gather_rows(M_<void>(t, PTV_DTF_NS16S), nsf, perm, R, Ht / 2, 0, 0, 1, stream));
    PTV_TRY(ptv_gather_rows_seg(M_<void>(t, PTV_DTF_TOK_S), tok, perm, R, E, (long)R * E, (long)R * E, 15, (const int*)T_(t, PTV_DTF_SEG_N), stream));
    nsf = (const __bf16*)T_(t, PTV_DTF_NS16S);
    tok = (const float*)T_(t, PTV_DTF_TOK_S);
  }
  PTV_TRY(ptv_gemm(P, 0, 0, R, Hn, Ht, nsf, Ht, T_(t, PTV_DTF_W16_T2N), Ht, HN, Hn, (const float*)T_(t, PTV_DTF_B_T2N), 1.f, 0, 0, 0, A16 | B16,
                   stream));
  PTV_TRY(ptv_gemm(P, 0, 0, R, 3 * Hn, Ht, nsf, Ht, w_ih_n, ld_n, M_<void>(t, PTV_DTF_GC), 3L * Hn, (const float*)T_(t, PTV_DTF_B_IH_N), 1.f, 0, 0,
                   -1, A16 | B16 | C16 | CBLK16, stream));
  // (LIVE_TOP, device int or NULL: the caller wants the outputs of the note steps 0 .. *LIVE_TOP only -- a loss that ignores the padded
  // note slots; the three launches below then leave the later steps' rows of their outputs unwritten)
  const int* live = (const int*)T_(t, PTV_DTF_LIVE_TOP);
  PTV_TRY(ptv_notes_gru_persist_fwd_rows(T_(t, PTV_DTF_PK_NOTES_H), T_(t, PTV_DTF_PK_NOTES_T), (const float*)T_(t, PTV_DTF_B_HH_N), T_(t, PTV_DTF_GC),
                                         tok, HN, HN16, M_<void>(t, PTV_DTF_GATES_N), R, 15 | (sorted && t[PTV_DTF_SEG_N] ? (1 << 24) : 0), live, row_len, stream));
  // ---- pitch head + initial duration state in one pass over the note states (ptvae.py:343-352)
  PTV_TRY(ptv_heads_fwd_rows(HN16 + (long)R * Hn, T_(t, PTV_DTF_PK_WP), T_(t, PTV_DTF_PK_WDH), T_(t, PTV_DTF_PK_WDP), (const float*)T_(t, PTV_DTF_B_P),
                             (const float*)T_(t, PTV_DTF_B_DH), M_<float>(t, PTV_DTF_PITCH), ldp, HD, HD16, M, live, R, row_len, stream));
  // ---- 5-step duration GRU with arg-max feedback; its input is one of three vectors: gate tables (ptvae.py:353-367)
  const int I = 5;
  PTV_TRY(ptv_gemm(PTV_PREC_F32, 0, 0, 1, 3 * Hd, I, T_(t, PTV_DTF_SOS), I, T_(t, PTV_DTF_W_IH_D), I, M_<void>(t, PTV_DTF_TAB0), 3L * Hd,
                   (const float*)T_(t, PTV_DTF_B_IH_D), 1.f, 0, 0, 0, 0, stream));
  PTV_TRY(ptv_gemm(PTV_PREC_F32, 0, 0, 2, 3 * Hd, I, T_(t, PTV_DTF_ONEHOT), I, T_(t, PTV_DTF_W_IH_D), I, M_<void>(t, PTV_DTF_TAB), 3L * Hd,
                   (const float*)T_(t, PTV_DTF_B_IH_D), 1.f, 0, 0, 0, 0, stream));
  PTV_TRY(ptv_dur_gru_fwd_rows(Hd, M, HD, Hd, (const float*)T_(t, PTV_DTF_W_HH_D), (const float*)T_(t, PTV_DTF_B_HH_D), (const float*)T_(t, PTV_DTF_TAB0),
                          (const float*)T_(t, PTV_DTF_TAB), (const float*)T_(t, PTV_DTF_W_OUT_D), (const float*)T_(t, PTV_DTF_B_OUT_D), nullptr,
                          M * Hd, HD16 + M * Hd, M_<void>(t, PTV_DTF_GATES_D), M * Hd, 4 * M * Hd, 1, M_<float>(t, PTV_DTF_DUR), 10,
                              M_<int>(t, PTV_DTF_IDX), M, (const int*)T_(t, PTV_DTF_FORCE_DUR), M, live, R, row_len, stream));
  return PTV_OK;
}

// ---------------------------------------------------------------------------------------------
// ptv_chord_decoder_fwd: RnnDecoder.forward, teacher-forced (ptvae.py:51-87 with tfr = 1: every step is fed the ground-truth chord of the
// step before it): z -> h0 and z_in, tokens [init ; c_0 .. c_{T-2}], the input-side product hoisted out of the T steps, the GRU sequence,
// the three heads.  Any sizes, fp32 or bf16 precision (fp32 master weights either way: the products convert per tile).  11 launches.
// ---------------------------------------------------------------------------------------------
extern "C" int ptv_chord_decoder_fwd(const void* const* t, const long* d, void* stream) {
  if (!t || !d) return PTV_ERR_ARG;
  for (int i = 0; i < PTV_CDF_COUNT; i++)
    if (!t[i]) return PTV_ERR_ARG;
  const int B = (int)d[PTV_CDF_D_B], T = (int)d[PTV_CDF_D_T], H = (int)d[PTV_CDF_D_H], I = (int)d[PTV_CDF_D_I], Z = (int)d[PTV_CDF_D_Z],
            Zi = (int)d[PTV_CDF_D_ZI], P = (int)d[PTV_CDF_D_PREC], gbf = (int)d[PTV_CDF_D_GATES_BF16];
  if (B <= 0 || T <= 0 || H <= 0 || I <= 0 || Z <= 0 || Zi <= 0 || (P != PTV_PREC_F32 && P != PTV_PREC_BF16)) return PTV_ERR_ARG;
  float* hall = M_<float>(t, PTV_CDF_HALL);
  float* toks = M_<float>(t, PTV_CDF_TOKS);
  const float* w_ih = (const float*)T_(t, PTV_CDF_W_IH);
  const long ld_ih = (long)I + Zi;
  const long TB = (long)T * B;
  PTV_TRY(ptv_gemm(P, 0, 0, B, H, Z, T_(t, PTV_CDF_Z), Z, T_(t, PTV_CDF_W_ZHID), Z, hall, H, (const float*)T_(t, PTV_CDF_B_ZHID), 1.f, 0, 0, 0, 0, stream));
  PTV_TRY(ptv_gemm(P, 0, 0, B, Zi, Z, T_(t, PTV_CDF_Z), Z, T_(t, PTV_CDF_W_ZIN), Z, M_<void>(t, PTV_CDF_Z_IN), Zi, (const float*)T_(t, PTV_CDF_B_ZIN), 1.f,
                   0, 0, 0, 0, stream));
  PTV_TRY(ptv_copy2d(toks, I, (const float*)T_(t, PTV_CDF_INIT_INPUT), 0, B, I, 1.f, 0, stream));
  if (T > 1) PTV_TRY(ptv_copy2d(toks + (long)B * I, I, (const float*)T_(t, PTV_CDF_C_SM), I, (long)(T - 1) * B, I, 1.f, 0, stream));
  PTV_TRY(ptv_gemm(P, 0, 0, (int)TB, 3 * H, I, toks, I, w_ih, ld_ih, M_<void>(t, PTV_CDF_GI), 3L * H, nullptr, 1.f, 0, 0, 0, 0, stream));
  PTV_TRY(ptv_gemm(P, 0, 0, B, 3 * H, Zi, T_(t, PTV_CDF_Z_IN), Zi, w_ih + I, ld_ih, M_<void>(t, PTV_CDF_ZG), 3L * H, (const float*)T_(t, PTV_CDF_B_IH),
                   1.f, 0, 0, 0, 0, stream));
  PTV_TRY(ptv_gru_seq_fwd(P, B, H, T, T_(t, PTV_CDF_GI), (long)B * 3 * H, 3L * H, T_(t, PTV_CDF_ZG), 0, 3L * H, T_(t, PTV_CDF_W_HH),
                          (const float*)T_(t, PTV_CDF_B_HH), hall, nullptr, M_<void>(t, PTV_CDF_GATES), nullptr, 0, nullptr, gbf ? 1 : 0, stream));
  const float* hs = hall + (long)B * H;
  const int nr = (int)d[PTV_CDF_D_NROOT], nc = (int)d[PTV_CDF_D_NCHROMA], nb = (int)d[PTV_CDF_D_NBASS];
  PTV_TRY(ptv_gemm(P, 0, 0, (int)TB, nr, H, hs, H, T_(t, PTV_CDF_W_ROOT), H, M_<void>(t, PTV_CDF_ROOT), nr, (const float*)T_(t, PTV_CDF_B_ROOT), 1.f, 0,
                   0, 0, 0, stream));
  PTV_TRY(ptv_gemm(P, 0, 0, (int)TB, nc, H, hs, H, T_(t, PTV_CDF_W_CHROMA), H, M_<void>(t, PTV_CDF_CHROMA), nc, (const float*)T_(t, PTV_CDF_B_CHROMA),
                   1.f, 0, 0, 0, 0, stream));
  PTV_TRY(ptv_gemm(P, 0, 0, (int)TB, nb, H, hs, H, T_(t, PTV_CDF_W_BASS), H, M_<void>(t, PTV_CDF_BASS), nb, (const float*)T_(t, PTV_CDF_B_BASS), 1.f, 0,
                   0, 0, 0, stream));
  return PTV_OK;
}

// ---------------------------------------------------------------------------------------------
// ptv_chord_decoder_bwd: the backward of the above (ChordDecoderTFFn.backward's launch sequence, bit-identical to it)
// ---------------------------------------------------------------------------------------------
extern "C" int ptv_chord_decoder_bwd(const void* const* t, const long* d, void* stream) {
  if (!t || !d) return PTV_ERR_ARG;
  const int B = (int)d[PTV_CDB_D_B], T = (int)d[PTV_CDB_D_T], H = (int)d[PTV_CDB_D_H], I = (int)d[PTV_CDB_D_I], Z = (int)d[PTV_CDB_D_Z],
            Zi = (int)d[PTV_CDB_D_ZI], P = (int)d[PTV_CDB_D_PREC], abf = d[PTV_CDB_D_ACT_BF16] ? 1 : 0, persist = (int)d[PTV_CDB_D_PERSIST],
            S = (int)d[PTV_CDB_D_SPLITK];
  if (B <= 0 || T <= 0 || H <= 0 || I <= 0 || Z <= 0 || Zi <= 0 || (P != PTV_PREC_F32 && P != PTV_PREC_BF16)) return PTV_ERR_ARG;
  for (int i = 0; i < PTV_CDB_COUNT; i++) {
    const bool optional = i == PTV_CDB_WT16_HH || i == PTV_CDB_DROOT || i == PTV_CDB_DCHROMA || i == PTV_CDB_DBASS || i == PTV_CDB_DHZ ||
                          i == PTV_CDB_XCH || i == PTV_CDB_PART || i == PTV_CDB_SYNC || i == PTV_CDB_WAIT_EVENT || i == PTV_CDB_RECORD_EVENT;
    if (!t[i] && !optional) return PTV_ERR_ARG;
  }
  if (persist && (!t[PTV_CDB_WT16_HH] || !t[PTV_CDB_XCH] || !t[PTV_CDB_SYNC] || !abf || (S && !t[PTV_CDB_PART]))) return PTV_ERR_ARG;
  if (!persist && !t[PTV_CDB_DHZ]) return PTV_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const long TB = (long)T * B, ld_ih = (long)I + Zi;
  const float* hall = (const float*)T_(t, PTV_CDB_HALL);
  const float* hs = hall + (long)B * H;
  float* dhs = M_<float>(t, PTV_CDB_DHS);
  // ---- the three heads: dhs = sum_k dlog_k . W_k;  dW_k += dlog_k^T . hs;  db_k += column sums of dlog_k
  const int heads[3][4] = {{PTV_CDB_DROOT, PTV_CDB_W_ROOT, PTV_CDB_G_W_ROOT, PTV_CDB_G_B_ROOT},
                           {PTV_CDB_DCHROMA, PTV_CDB_W_CHROMA, PTV_CDB_G_W_CHROMA, PTV_CDB_G_B_CHROMA},
                           {PTV_CDB_DBASS, PTV_CDB_W_BASS, PTV_CDB_G_W_BASS, PTV_CDB_G_B_BASS}};
  const int ncls[3] = {(int)d[PTV_CDB_D_NROOT], (int)d[PTV_CDB_D_NCHROMA], (int)d[PTV_CDB_D_NBASS]};
  bool have = false;
  WgradGroup wg(P, stream);
  for (int k = 0; k < 3; k++) {
    const void* dl = T_(t, heads[k][0]);
    if (!dl) continue;
    const int n = ncls[k];
    PTV_TRY(ptv_gemm(P, 0, 1, (int)TB, H, n, dl, n, T_(t, heads[k][1]), H, dhs, H, nullptr, 1.f, have ? 1 : 0, 0, 0, 0, stream));
    have = true;
    PTV_TRY(wg.add(n, H, TB, dl, n, 0, hs, H, 0, M_<float>(t, heads[k][2]), H, M_<float>(t, heads[k][3])));
  }
  PTV_TRY(wg.flush());
  if (!have && hipMemsetAsync(dhs, 0, sizeof(float) * TB * H, s) != hipSuccess) return PTV_ERR_LAUNCH;
  // ---- BPTT
  void* dgi = M_<void>(t, PTV_CDB_DGI); void* dgh = M_<void>(t, PTV_CDB_DGH);
  float* dh0 = M_<float>(t, PTV_CDB_DH0);
  if (persist) {
    const float* hall_[1] = {hall}; const void* gates_[1] = {T_(t, PTV_CDB_GATES)}; const void* wt_[1] = {T_(t, PTV_CDB_WT16_HH)};
    const void* ext_[1] = {dhs}; const long ext_step[1] = {(long)B * H}, ext_ld[1] = {(long)H}; const int ext_bf[1] = {0};
    const float* last_[1] = {nullptr}; const long last_ld[1] = {0};
    void* dgi_[1] = {dgi}; void* dgh_[1] = {dgh}; float* dh0_[1] = {dh0}; const int rev[1] = {0};
    void* xch_[1] = {M_<void>(t, PTV_CDB_XCH)}; float* part_[1] = {M_<float>(t, PTV_CDB_PART)};
    if (t[PTV_CDB_WAIT_EVENT] && hipStreamWaitEvent(s, (hipEvent_t)const_cast<void*>(t[PTV_CDB_WAIT_EVENT]), 0) != hipSuccess) return PTV_ERR_LAUNCH;
    if (S) PTV_TRY(ptv_gru_persist_bwd_splitk(S, 1, B, H, T, hall_, gates_, wt_, ext_, ext_step, ext_ld, ext_bf, last_, last_ld, dgi_, dgh_, dh0_, rev,
                                              xch_, part_, M_<unsigned>(t, PTV_CDB_SYNC), stream));
    else PTV_TRY(ptv_gru_persist_bwd(1, B, H, T, hall_, gates_, wt_, ext_, ext_step, ext_ld, ext_bf, last_, last_ld, dgi_, dgh_, dh0_, rev, xch_,
                                     M_<unsigned>(t, PTV_CDB_SYNC), stream));
    if (t[PTV_CDB_RECORD_EVENT] && hipEventRecord((hipEvent_t)const_cast<void*>(t[PTV_CDB_RECORD_EVENT]), s) != hipSuccess) return PTV_ERR_LAUNCH;
  } else {
    const void* w = t[PTV_CDB_WT16_HH] ? T_(t, PTV_CDB_WT16_HH) : T_(t, PTV_CDB_W_HH);
    const int flags = abf | (abf << 3) | ((t[PTV_CDB_WT16_HH] ? 1 : 0) << 4);     // gates bf16, gate gradients bf16, weight operand bf16 (ptv_gru_seq_bwd)
    PTV_TRY(ptv_gru_seq_bwd(P, B, H, T, hall, T_(t, PTV_CDB_GATES), w, dhs, (long)B * H, H, nullptr, 0, nullptr, 0, 0, 0, nullptr, dgi, dgh,
                            M_<float>(t, PTV_CDB_DHZ), dh0, 0, flags, stream));
  }
  // ---- the GRU's parameters and the two z projections: the chain to dz first, then every parameter gradient as one batch (the bias sums
  // ride inside the products)
  const int A = abf ? A16 : 0;
  float* dzg = M_<float>(t, PTV_CDB_DZG);
  PTV_TRY(ptv_sum_steps_top(dzg, dgi, (long)B * 3 * H, T, (long)B * 3 * H, 0, abf, nullptr, stream));
  float* g_ih = M_<float>(t, PTV_CDB_G_W_IH);
  const float* w_ih = (const float*)T_(t, PTV_CDB_W_IH);
  float* dz_in = M_<float>(t, PTV_CDB_DZ_IN); float* dtok0 = M_<float>(t, PTV_CDB_DTOK0);
  PTV_TRY(ptv_gemm(P, 0, 1, B, Zi, 3 * H, dzg, 3L * H, w_ih + I, ld_ih, dz_in, Zi, nullptr, 1.f, 0, 0, 0, 0, stream));
  PTV_TRY(ptv_gemm(P, 0, 1, B, I, 3 * H, dgi, 3L * H, w_ih, ld_ih, dtok0, I, nullptr, 1.f, 0, 0, 0, A, stream));     // only the learned start token
  float* dz = M_<float>(t, PTV_CDB_DZ);
  PTV_TRY(ptv_gemm(P, 0, 1, B, Z, H, dh0, H, T_(t, PTV_CDB_W_ZHID), Z, dz, Z, nullptr, 1.f, 0, 0, 0, 0, stream));
  PTV_TRY(ptv_gemm(P, 0, 1, B, Z, Zi, dz_in, Zi, T_(t, PTV_CDB_W_ZIN), Z, dz, Z, nullptr, 1.f, 1, 0, 0, 0, stream));
  PTV_TRY(wg.add(3 * H, H, TB, dgh, 3L * H, abf, hall, H, 0, M_<float>(t, PTV_CDB_G_W_HH), H, M_<float>(t, PTV_CDB_G_B_HH)));
  PTV_TRY(wg.add(3 * H, Zi, B, dzg, 3L * H, 0, T_(t, PTV_CDB_Z_IN), Zi, 0, g_ih + I, ld_ih, M_<float>(t, PTV_CDB_G_B_IH)));
  PTV_TRY(wg.add(3 * H, I, TB, dgi, 3L * H, abf, T_(t, PTV_CDB_TOKS), I, 0, g_ih, ld_ih, nullptr));
  PTV_TRY(wg.add(H, Z, B, dh0, H, 0, T_(t, PTV_CDB_Z), Z, 0, M_<float>(t, PTV_CDB_G_W_ZHID), Z, M_<float>(t, PTV_CDB_G_B_ZHID)));
  PTV_TRY(wg.add(Zi, Z, B, dz_in, Zi, 0, T_(t, PTV_CDB_Z), Z, 0, M_<float>(t, PTV_CDB_G_W_ZIN), Z, M_<float>(t, PTV_CDB_G_B_ZIN)));
  PTV_TRY(wg.flush());
  PTV_TRY(ptv_colsum(M_<float>(t, PTV_CDB_G_INIT_INPUT), dtok0, I, B, I, nullptr, 1, 0, stream));
  return PTV_OK;
}

// ---------------------------------------------------------------------------------------------
// ptv_decoder_tf_bwd: the backward of ptv_decoder_tf_fwd (functional.decoder_bwd_core's launch sequence on its fused bf16 path, bit-identical
// to it).  Chain on `stream`; the parameter-gradient products on the side stream, forked where the Python sequencing forks them.
// ---------------------------------------------------------------------------------------------
extern "C" int ptv_decoder_tf_bwd(const void* const* t, const long* d, void* stream) {
  if (!t || !d) return PTV_ERR_ARG;
  for (int i = 0; i < PTV_DTB_COUNT; i++)
    if (!t[i] && i != PTV_DTB_TOP_H && i != PTV_DTB_PART_T && i != PTV_DTB_WAIT_EVENT && i != PTV_DTB_RECORD_EVENT && i != PTV_DTB_PERM &&
        i != PTV_DTB_ROW_LEN && i != PTV_DTB_NS16S && i != PTV_DTB_DNS_S && i != PTV_DTB_DTOK_S && i != PTV_DTB_SEG_N)
      return PTV_ERR_ARG;
  // the forward ran on rows sorted by length: every per-row tensor here is in that order; the two gradients that leave the node for
  // row-order-aware consumers (the time states', the fed tokens') are scattered back
  const bool sorted = t[PTV_DTB_PERM] != nullptr;
  if (sorted != (t[PTV_DTB_ROW_LEN] != nullptr) || sorted != (t[PTV_DTB_NS16S] != nullptr) || sorted != (t[PTV_DTB_DNS_S] != nullptr) ||
      sorted != (t[PTV_DTB_DTOK_S] != nullptr) || (sorted && !t[PTV_DTB_TOP_H]))
    return PTV_ERR_ARG;
  const int* perm = (const int*)T_(t, PTV_DTB_PERM);
  const int* row_len = (const int*)T_(t, PTV_DTB_ROW_LEN);

  const int B = (int)d[PTV_DTB_D_B], E = (int)d[PTV_DTB_D_E], He = (int)d[PTV_DTB_D_HE], Ht = (int)d[PTV_DTB_D_HT], Hn = (int)d[PTV_DTB_D_HN],
            Hd = (int)d[PTV_DTB_D_HD], NP = (int)d[PTV_DTB_D_NP], Zs = (int)d[PTV_DTB_D_ZS], Zi = (int)d[PTV_DTB_D_ZI], nblk = (int)d[PTV_DTB_D_NBLK],
            S = (int)d[PTV_DTB_D_SPLITK];
  const long ldp = d[PTV_DTB_D_LDP];
  if (B <= 0 || E != 128 || Hn != 512 || Hd != 64 || NP != 130 || ldp < NP || nblk <= 0 || (S && !t[PTV_DTB_PART_T])) return PTV_ERR_UNSUPPORTED;
  const int R = 32 * B;
  // K segments of the weight-gradient products over (note step, sorted row): the dead blocks of every step are neither read nor multiplied
  const int* seg_n = t[PTV_DTB_PERM] ? (const int*)T_(t, PTV_DTB_SEG_N) : nullptr;
  if (seg_n && !ptv_wgrad_seg_supported(15L * R, R)) return PTV_ERR_ARG;    // (the caller asks ptv_wgrad_seg_supported before it sets the slot)
  const long M = 15L * R;
  const int P = PTV_PREC_BF16;
  hipStream_t s = (hipStream_t)stream, side = (hipStream_t)const_cast<void*>(t[PTV_DTB_SIDE_STREAM]);
  void* sside = (void*)side;
  const int psz = ptv_dur_gru_bwd_part_size();
  const int* top_h = (const int*)T_(t, PTV_DTB_TOP_H);
  const long top_unit = top_h ? R : 0;
  auto fork = [&](int k) -> int {                       // the side stream waits for everything queued on `stream` so far
    hipEvent_t e = (hipEvent_t)const_cast<void*>(t[PTV_DTB_FORK_EVENT0 + k]);
    if (hipEventRecord(e, s) != hipSuccess || hipStreamWaitEvent(side, e, 0) != hipSuccess) return PTV_ERR_LAUNCH;
    return PTV_OK;
  };
  float* G_ = nullptr; (void)G_;
#define GB(i) M_<float>(t, i)

  // ================= duration GRU (5 steps): one kernel; its parameter gradients come back as per-block partials
  float* ddur = GB(PTV_DTB_DDUR); float* dHD0 = GB(PTV_DTB_DHD0); float* part = GB(PTV_DTB_PART);
  const void* HD16 = T_(t, PTV_DTB_HD16);
  ptv_gemm_priority(1);
  PTV_TRY(ptv_dur_gru_bwd(Hd, M, nullptr, M * Hd, 4 * M * Hd, HD16, M * Hd, 1, ddur, 10, (const float*)T_(t, PTV_DTB_W_HH_D),
                          (const float*)T_(t, PTV_DTB_W_OUT_D), (const int*)T_(t, PTV_DTB_IDX), M, dHD0, part, nblk,
                          (const float*)T_(t, PTV_DTB_B_HH_D), (const float*)T_(t, PTV_DTB_TAB0), (const float*)T_(t, PTV_DTB_TAB), stream));
  PTV_TRY(fork(0));
  ptv_gemm_priority(0);
  PTV_TRY(ptv_dur_out_wgrad(ddur, 10, HD16, M * Hd, GB(PTV_DTB_G_W_OUT_D), M, Hd, sside));
  PTV_TRY(ptv_colsum(GB(PTV_DTB_TMP64), ddur, 64, M * 10 / 64, 64, nullptr, 1, 0, sside));      // bias of a 2-column matrix: a 64-column stream, folded
  PTV_TRY(ptv_colsum(GB(PTV_DTB_G_B_OUT_D), GB(PTV_DTB_TMP64), 2, 32, 2, nullptr, 1, 0, sside));
  PTV_TRY(ptv_colsum(GB(PTV_DTB_S), part, psz, nblk, psz, nullptr, 1, 0, sside));
  PTV_TRY(ptv_dur_bwd_finalize(GB(PTV_DTB_S), GB(PTV_DTB_G_W_HH_D), GB(PTV_DTB_G_B_HH_D), GB(PTV_DTB_G_B_IH_D), GB(PTV_DTB_G_W_IH_D), GB(PTV_DTB_G_SOS),
                               (const float*)T_(t, PTV_DTB_W_IH_D), (const float*)T_(t, PTV_DTB_SOS), 5, sside));

  // ================= the two per-note heads: dP += dHD0 . W_dh[:, Hn:], dNSUM = dP . W_p + dHD0 . W_dh[:, :Hn] in one pass
  float* dP = GB(PTV_DTB_DP);
  void* dNSUM = M_<void>(t, PTV_DTB_DNSUM); void* dY16 = M_<void>(t, PTV_DTB_DY16);
  ptv_gemm_priority(1);
  // (blocked = 3 with a limit: dNSUM's dead rows stay unwritten -- the BPTT below gets the same limit as its bound and never reads them)
  PTV_TRY(ptv_heads_bwd_rows(dP, ldp, dHD0, T_(t, PTV_DTB_PK_WDPT), T_(t, PTV_DTB_PK_WCAT), dNSUM, top_h ? 3 : 1, dY16, top_h, top_unit, row_len, M,
                             stream));
  PTV_TRY(fork(1));
  ptv_gemm_priority(0);
  {
    const __bf16* HN16 = (const __bf16*)T_(t, PTV_DTB_HN16);
    const __bf16* nsum = HN16 + (long)R * Hn;                       // the note summaries = states 1 .. 15
    float* tmp = GB(PTV_DTB_TMP200); float* cs = GB(PTV_DTB_CS200);
    WgradGroup wg(P, sside);                                       // both head products in one launch
    PTV_TRY(wg.add(200, Hn, M, dY16, 200, 1, nsum, Hn, 1, tmp, Hn, cs, top_h, top_unit, 0, 0, seg_n, R, 15));
    PTV_TRY(wg.add(Hd, NP, M, dHD0, Hd, 0, T_(t, PTV_DTB_PITCH), ldp, 0, GB(PTV_DTB_G_W_DH) + Hn, (long)Hn + NP, nullptr, top_h, top_unit, 0, 1, seg_n, R, 15));
    PTV_TRY(wg.flush());
    PTV_TRY(ptv_copy2d(GB(PTV_DTB_G_W_P), Hn, tmp, Hn, NP, Hn, 1.f, 1, sside));
    PTV_TRY(ptv_copy2d(GB(PTV_DTB_G_B_P), NP, cs, NP, 1, NP, 1.f, 1, sside));
    PTV_TRY(ptv_copy2d(GB(PTV_DTB_G_W_DH), (long)Hn + NP, tmp + 136L * Hn, Hn, Hd, Hn, 1.f, 1, sside));
    PTV_TRY(ptv_copy2d(GB(PTV_DTB_G_B_DH), Hd, cs + 136, Hd, 1, Hd, 1.f, 1, sside));
  }

  // ================= notes GRU (15 steps x 32 B rows): BPTT, then the gradients of the fed tokens and of the time states
  void* dgi_n = M_<void>(t, PTV_DTB_DGI_N); void* dgh_n = M_<void>(t, PTV_DTB_DGH_N);
  float* dHN0 = GB(PTV_DTB_DHN0); int* top_step = M_<int>(t, PTV_DTB_TOP_STEP);
  float* dGC = GB(PTV_DTB_DGC);
  float* dNS_out = GB(PTV_DTB_DNS); float* dtok_out = GB(PTV_DTB_DTOK);     // what leaves the node, natural row order
  float* dNS = sorted ? GB(PTV_DTB_DNS_S) : dNS_out;                       // what the products below write
  float* dtok = sorted ? GB(PTV_DTB_DTOK_S) : dtok_out;
  const __bf16* wt_ih_n = (const __bf16*)T_(t, PTV_DTB_WT_IH_N);   // [Ht + E, 3Hn]
  ptv_gemm_priority(1);
  PTV_TRY(ptv_notes_gru_persist_bwd_rows(T_(t, PTV_DTB_PK_NOTES_WT), T_(t, PTV_DTB_HN16), T_(t, PTV_DTB_GATES_N), dNSUM, dgi_n, dgh_n, dHN0,
                                         M_<void>(t, PTV_DTB_SCRATCH_N), R, 15 | (seg_n ? 0x10000 : 0), top_h, row_len, top_step, stream));
  PTV_TRY(ptv_sum_steps_seg(dGC, dgi_n, (long)R * 3 * Hn, 15, (long)R * 3 * Hn, 0, 1, top_step, seg_n, 3L * Hn, stream));
  if (hipMemsetAsync(dtok_out + 15L * R * E, 0, sizeof(float) * R * E, s) != hipSuccess) return PTV_ERR_LAUNCH;
  PTV_TRY(ptv_gemm_mtop_seg(P, 0, 0, (int)M, E, 3 * Hn, dgi_n, 3L * Hn, wt_ih_n + (long)Ht * 3 * Hn, 3L * Hn, dtok, E, nullptr, 1.f, 0, 0, 0, A16 | B16,
                            top_step, R, seg_n, R, 15, stream));
  PTV_TRY(ptv_gemm(P, 0, 0, R, Ht, 3 * Hn, dGC, 3L * Hn, wt_ih_n, 3L * Hn, dNS, Ht, nullptr, 1.f, 0, 0, 0, B16, stream));
  PTV_TRY(ptv_gemm(P, 0, 0, R, Ht, Hn, dHN0, Hn, T_(t, PTV_DTB_WT_T2N), Hn, dNS, Ht, nullptr, 1.f, 1, 0, 0, B16, stream));
  if (sorted) {
    PTV_TRY(ptv_scatter_rows(dNS_out, dNS, perm, R, Ht, 0, 0, 1, stream));
    PTV_TRY(ptv_scatter_rows_seg(dtok_out, dtok, perm, R, E, (long)R * E, (long)R * E, 15, seg_n, stream));
  }
  PTV_TRY(fork(2));
  ptv_gemm_priority(0);
  {
    const __bf16* HN16 = (const __bf16*)T_(t, PTV_DTB_HN16);      // states 0 .. 14: the operand of the W_hh gradient
    const __bf16* NSf = sorted ? (const __bf16*)T_(t, PTV_DTB_NS16S) : (const __bf16*)T_(t, PTV_DTB_NS16) + (long)B * Ht;
    float* gw = GB(PTV_DTB_G_W_HH_N); float* gb = GB(PTV_DTB_G_B_HH_N); float* gih = GB(PTV_DTB_G_W_IH_N);
    // the five parameter gradients of the notes GRU and of dec_time_to_notes_hid: ONE product launch + ONE reduction launch; the four bias
    // gradients are column sums taken from the A tiles in LDS (round 5: five products, five reductions, two column-sum passes over dGC / dHN0)
    WgradGroup wg(P, sside);
    PTV_TRY(wg.add(2 * Hn, Hn, M, dgi_n, 3L * Hn, 1, HN16, Hn, 1, gw, Hn, gb, top_step, R, 0, 1, seg_n, R, 15));
    PTV_TRY(wg.add(Hn, Hn, M, dgh_n, Hn, 1, HN16, Hn, 1, gw + 2L * Hn * Hn, Hn, gb + 2 * Hn, top_step, R, 0, 1, seg_n, R, 15));
    PTV_TRY(wg.add(3 * Hn, Ht, R, dGC, 3L * Hn, 0, NSf, Ht, 1, gih, (long)Ht + E, GB(PTV_DTB_G_B_IH_N)));
    PTV_TRY(wg.add(3 * Hn, E, M, dgi_n, 3L * Hn, 1, T_(t, PTV_DTB_TOK_OP), E, 0, gih + Ht, (long)Ht + E, nullptr, top_step, R, 0, 1, seg_n, R, 15));
    PTV_TRY(wg.add(Hn, Ht, R, dHN0, Hn, 0, NSf, Ht, 1, GB(PTV_DTB_G_W_T2N), Ht, GB(PTV_DTB_G_B_T2N)));
    PTV_TRY(wg.flush());
  }

  // ================= time GRU (32 steps x B rows): one persistent launch, split-K teams; then dTOKS and dz
  void* dgi_t = M_<void>(t, PTV_DTB_DGI_T); void* dgh_t = M_<void>(t, PTV_DTB_DGH_T);
  float* dzhid = GB(PTV_DTB_DZHID); float* dZG = GB(PTV_DTB_DZG); float* dz_in = GB(PTV_DTB_DZ_IN); float* dTOKS = GB(PTV_DTB_DTOKS);
  float* dz = GB(PTV_DTB_DZ);
  const __bf16* wt_ih_t = (const __bf16*)T_(t, PTV_DTB_WT_IH_T);   // [2He + Zi, 3Ht]
  ptv_gemm_priority(1);
  {
    const float* hall_[1] = {(const float*)T_(t, PTV_DTB_NS)}; const void* gates_[1] = {T_(t, PTV_DTB_GATES_T)};
    const void* wt_[1] = {T_(t, PTV_DTB_WT_HH_T)};
    const void* ext_[1] = {dNS_out}; const long ext_step[1] = {(long)B * Ht}, ext_ld[1] = {(long)Ht}; const int ext_bf[1] = {0};
    const float* last_[1] = {nullptr}; const long last_ld[1] = {0};
    void* dgi_[1] = {dgi_t}; void* dgh_[1] = {dgh_t}; float* dh0_[1] = {dzhid}; const int rev[1] = {0};
    void* xch_[1] = {M_<void>(t, PTV_DTB_XCH)}; float* part_[1] = {M_<float>(t, PTV_DTB_PART_T)};
    if (t[PTV_DTB_WAIT_EVENT] && hipStreamWaitEvent(s, (hipEvent_t)const_cast<void*>(t[PTV_DTB_WAIT_EVENT]), 0) != hipSuccess) return PTV_ERR_LAUNCH;
    if (S) PTV_TRY(ptv_gru_persist_bwd_splitk(S, 1, B, Ht, 32, hall_, gates_, wt_, ext_, ext_step, ext_ld, ext_bf, last_, last_ld, dgi_, dgh_, dh0_,
                                              rev, xch_, part_, M_<unsigned>(t, PTV_DTB_SYNC), stream));
    else PTV_TRY(ptv_gru_persist_bwd(1, B, Ht, 32, hall_, gates_, wt_, ext_, ext_step, ext_ld, ext_bf, last_, last_ld, dgi_, dgh_, dh0_, rev, xch_,
                                     M_<unsigned>(t, PTV_DTB_SYNC), stream));
    if (t[PTV_DTB_RECORD_EVENT] && hipEventRecord((hipEvent_t)const_cast<void*>(t[PTV_DTB_RECORD_EVENT]), s) != hipSuccess) return PTV_ERR_LAUNCH;
  }
  PTV_TRY(ptv_sum_steps_top(dZG, dgi_t, (long)B * 3 * Ht, 32, (long)B * 3 * Ht, 0, 1, nullptr, stream));
  PTV_TRY(ptv_gemm(P, 0, 0, B, Zi, 3 * Ht, dZG, 3L * Ht, wt_ih_t + 2L * He * 3 * Ht, 3L * Ht, dz_in, Zi, nullptr, 1.f, 0, 0, 0, B16, stream));
  if (hipMemsetAsync(dTOKS + 32L * B * 2 * He, 0, sizeof(float) * B * 2 * He, s) != hipSuccess) return PTV_ERR_LAUNCH;
  PTV_TRY(ptv_gemm(P, 0, 0, R, 2 * He, 3 * Ht, dgi_t, 3L * Ht, wt_ih_t, 3L * Ht, dTOKS, 2L * He, nullptr, 1.f, 0, 0, 0, A16 | B16, stream));
  PTV_TRY(ptv_gemm(P, 0, 0, B, Zs, Ht, dzhid, Ht, T_(t, PTV_DTB_WT_ZHID), Ht, dz, Zs, nullptr, 1.f, 0, 0, 0, B16, stream));
  PTV_TRY(ptv_gemm(P, 0, 0, B, Zs, Zi, dz_in, Zi, T_(t, PTV_DTB_WT_ZIN), Zi, dz, Zs, nullptr, 1.f, 1, 0, 0, B16, stream));
  PTV_TRY(fork(3));
  ptv_gemm_priority(0);
  {
    const __bf16* NS16 = (const __bf16*)T_(t, PTV_DTB_NS16);
    float* g_ih = GB(PTV_DTB_G_W_IH_T); float* gb_ih = GB(PTV_DTB_G_B_IH_T); float* gb_hh = GB(PTV_DTB_G_B_HH_T);
    const long ld_t = 2L * He + Zi;
    // the time GRU's and the two z projections' parameter gradients as one batch.  bias_hh = column sums of dgh over all (t, b) rows, taken
    // inside the W_hh product (round 5: r / z thirds copied from bias_ih, n third summed by a second pass over dgh); bias_ih = column sums
    // of dZG = sum_t dgi_t inside the z_in product.  Only the learned start token's gradient (row block 0 of dTOKS) keeps its own column sum.
    WgradGroup wg(P, sside);
    PTV_TRY(wg.add(3 * Ht, Ht, R, dgh_t, 3L * Ht, 1, NS16, Ht, 1, GB(PTV_DTB_G_W_HH_T), Ht, gb_hh));
    PTV_TRY(wg.add(3 * Ht, Zi, B, dZG, 3L * Ht, 0, T_(t, PTV_DTB_Z_IN), Zi, 0, g_ih + 2 * He, ld_t, gb_ih));
    PTV_TRY(wg.add(3 * Ht, 2 * He, R, dgi_t, 3L * Ht, 1, T_(t, PTV_DTB_TOKS), 2L * He, 0, g_ih, ld_t, nullptr));
    PTV_TRY(wg.add(Ht, Zs, B, dzhid, Ht, 0, T_(t, PTV_DTB_Z), Zs, 0, GB(PTV_DTB_G_W_ZHID), Zs, GB(PTV_DTB_G_B_ZHID)));
    PTV_TRY(wg.add(Zi, Zs, B, dz_in, Zi, 0, T_(t, PTV_DTB_Z), Zs, 0, GB(PTV_DTB_G_W_ZIN), Zs, GB(PTV_DTB_G_B_ZIN)));
    PTV_TRY(wg.flush());
    PTV_TRY(ptv_colsum(GB(PTV_DTB_G_INIT_INPUT), dTOKS, 2L * He, B, 2 * He, nullptr, 1, 0, sside));
  }
  ptv_gemm_priority(1);
#undef GB
  return PTV_OK;
}

// ---------------------------------------------------------------------------------------------
// ptv_bigru_final_bwd: functional._bigru_backward's persistent branch (the two encoders' bi-GRUs) as one call
// ---------------------------------------------------------------------------------------------
extern "C" int ptv_bigru_final_bwd(const void* const* t, const long* d, void* stream) {
  if (!t || !d) return PTV_ERR_ARG;
  const int M = (int)d[PTV_BGB_D_M], T = (int)d[PTV_BGB_D_T], H = (int)d[PTV_BGB_D_H], I = (int)d[PTV_BGB_D_I], xbf = d[PTV_BGB_D_X_BF16] ? 1 : 0,
            dx_acc = d[PTV_BGB_D_DX_ACC] ? 1 : 0, S = (int)d[PTV_BGB_D_SPLITK];
  if (M <= 0 || T < 2 || H <= 0 || I <= 0) return PTV_ERR_ARG;
  const bool want_dx = t[PTV_BGB_DX] != nullptr;
  for (int i = 0; i < PTV_BGB_COUNT; i++) {
    const bool optional = i == PTV_BGB_WT_IH0 || i == PTV_BGB_WT_IH1 || i == PTV_BGB_DX || i == PTV_BGB_PART0 || i == PTV_BGB_PART1 ||
                          i == PTV_BGB_WAIT_EVENT || i == PTV_BGB_RECORD_EVENT;
    if (!t[i] && !optional) return PTV_ERR_ARG;
  }
  if ((want_dx && (!t[PTV_BGB_WT_IH0] || !t[PTV_BGB_WT_IH1])) || (S && (!t[PTV_BGB_PART0] || !t[PTV_BGB_PART1]))) return PTV_ERR_ARG;
  const int P = PTV_PREC_BF16;
  hipStream_t s = (hipStream_t)stream, side = (hipStream_t)const_cast<void*>(t[PTV_BGB_SIDE_STREAM]);
  void* sside = (void*)side;
  const long TM = (long)T * M;
  const float* dout = (const float*)T_(t, PTV_BGB_DOUT);
  void* dgi[2] = {M_<void>(t, PTV_BGB_DGI0), M_<void>(t, PTV_BGB_DGI1)};
  void* dgh[2] = {M_<void>(t, PTV_BGB_DGH0), M_<void>(t, PTV_BGB_DGH1)};
  ptv_gemm_priority(1);
  {
    const float* hall_[2] = {(const float*)T_(t, PTV_BGB_HALL0), (const float*)T_(t, PTV_BGB_HALL1)};
    const void* gates_[2] = {T_(t, PTV_BGB_GATES0), T_(t, PTV_BGB_GATES1)};
    const void* wt_[2] = {T_(t, PTV_BGB_WT_HH0), T_(t, PTV_BGB_WT_HH1)};
    const void* ext_[2] = {nullptr, nullptr}; const long ext_step[2] = {0, 0}, ext_ld[2] = {0, 0}; const int ext_bf[2] = {0, 0};
    const float* last_[2] = {dout, dout + H}; const long last_ld[2] = {2L * H, 2L * H};
    float* dh0_[2] = {nullptr, nullptr}; const int rev[2] = {0, 1};
    void* xch_[2] = {M_<void>(t, PTV_BGB_XCH0), M_<void>(t, PTV_BGB_XCH1)};
    float* part_[2] = {M_<float>(t, PTV_BGB_PART0), M_<float>(t, PTV_BGB_PART1)};
    if (t[PTV_BGB_WAIT_EVENT] && hipStreamWaitEvent(s, (hipEvent_t)const_cast<void*>(t[PTV_BGB_WAIT_EVENT]), 0) != hipSuccess) return PTV_ERR_LAUNCH;
    if (S) PTV_TRY(ptv_gru_persist_bwd_splitk(S, 2, M, H, T, hall_, gates_, wt_, ext_, ext_step, ext_ld, ext_bf, last_, last_ld, dgi, dgh, dh0_, rev,
                                              xch_, part_, M_<unsigned>(t, PTV_BGB_SYNC), stream));
    else PTV_TRY(ptv_gru_persist_bwd(2, M, H, T, hall_, gates_, wt_, ext_, ext_step, ext_ld, ext_bf, last_, last_ld, dgi, dgh, dh0_, rev, xch_,
                                     M_<unsigned>(t, PTV_BGB_SYNC), stream));
    if (t[PTV_BGB_RECORD_EVENT] && hipEventRecord((hipEvent_t)const_cast<void*>(t[PTV_BGB_RECORD_EVENT]), s) != hipSuccess) return PTV_ERR_LAUNCH;
  }
  const void* x = T_(t, PTV_BGB_X);
  auto products = [&](int dir, void* st) -> int {
    const int o = dir ? 4 : 0;
    WgradGroup wg(P, st);                                          // a direction's W_ih / W_hh gradients (and both bias sums): one launch
    PTV_TRY(wg.add(3 * H, I, TM, dgi[dir], 3L * H, 1, x, I, xbf, M_<float>(t, PTV_BGB_G_W_IH0 + o), I, M_<float>(t, PTV_BGB_G_B_IH0 + o)));
    PTV_TRY(wg.add(3 * H, H, TM, dgh[dir], 3L * H, 1, T_(t, dir ? PTV_BGB_H16_1 : PTV_BGB_H16_0), H, 1, M_<float>(t, PTV_BGB_G_W_HH0 + o), H,
                   M_<float>(t, PTV_BGB_G_B_HH0 + o), nullptr, 0, dir ? T : 0));
    return wg.flush();
  };
  // the reversed direction's products on the side stream ...
  hipEvent_t ef = (hipEvent_t)const_cast<void*>(t[PTV_BGB_FORK_EVENT]), ej = (hipEvent_t)const_cast<void*>(t[PTV_BGB_JOIN_EVENT]);
  if (hipEventRecord(ef, s) != hipSuccess || hipStreamWaitEvent(side, ef, 0) != hipSuccess) return PTV_ERR_LAUNCH;
  ptv_gemm_priority(0);
  PTV_TRY(products(1, sside));
  // ... the forward direction's (and its share of dx) on this one
  ptv_gemm_priority(1);
  PTV_TRY(products(0, stream));
  float* dx = M_<float>(t, PTV_BGB_DX);
  if (want_dx) PTV_TRY(ptv_gemm(P, 0, 0, (int)TM, I, 3 * H, dgi[0], 3L * H, T_(t, PTV_BGB_WT_IH0), 3L * H, dx, I, nullptr, 1.f, dx_acc, 0, 0, A16 | B16, stream));
  if (hipEventRecord(ej, side) != hipSuccess || hipStreamWaitEvent(s, ej, 0) != hipSuccess) return PTV_ERR_LAUNCH;
  if (want_dx) PTV_TRY(ptv_gemm(P, 0, 0, (int)TM, I, 3 * H, dgi[1], 3L * H, T_(t, PTV_BGB_WT_IH1), 3L * H, dx, I, nullptr, 1.f, 1, 0, 0, A16 | B16, stream));
  return PTV_OK;
}

// ---------------------------------------------------------------------------------------------
// ptv_bigru_final_fwd: functional._bigru_forward's persistent branch as one call
// ---------------------------------------------------------------------------------------------
extern "C" int ptv_bigru_final_fwd(const void* const* t, const long* d, void* stream) {
  if (!t || !d) return PTV_ERR_ARG;
  const int M = (int)d[PTV_BGF_D_M], T = (int)d[PTV_BGF_D_T], H = (int)d[PTV_BGF_D_H], I = (int)d[PTV_BGF_D_I], xbf = d[PTV_BGF_D_X_BF16] ? 1 : 0;
  if (M <= 0 || T < 2 || H <= 0 || I <= 0) return PTV_ERR_ARG;
  for (int i = 0; i < PTV_BGF_COUNT; i++)
    if (!t[i] && i != PTV_BGF_LENGTHS && i != PTV_BGF_WAIT_EVENT && i != PTV_BGF_RECORD_EVENT) return PTV_ERR_ARG;
  const int P = PTV_PREC_BF16;
  hipStream_t s = (hipStream_t)stream;
  const long TM = (long)T * M;
  const int gi_slot[2] = {PTV_BGF_GI0, PTV_BGF_GI1}, hall_slot[2] = {PTV_BGF_HALL0, PTV_BGF_HALL1};
  const int wih_slot[2] = {PTV_BGF_W16_IH0, PTV_BGF_W16_IH1}, bih_slot[2] = {PTV_BGF_B_IH0, PTV_BGF_B_IH1};
  for (int dir = 0; dir < 2; dir++) {
    PTV_TRY(ptv_gemm(P, 0, 0, (int)TM, 3 * H, I, T_(t, PTV_BGF_X), I, T_(t, wih_slot[dir]), I, M_<void>(t, gi_slot[dir]), 3L * H,
                     (const float*)T_(t, bih_slot[dir]), 1.f, 0, 0, -1, (xbf ? A16 : 0) | (d[PTV_BGF_D_WIH_F32] ? 0 : B16) | C16, stream));
    if (hipMemsetAsync(M_<void>(t, hall_slot[dir]), 0, sizeof(float) * M * H, s) != hipSuccess) return PTV_ERR_LAUNCH;
  }
  {
    const void* gi[2] = {T_(t, PTV_BGF_GI0), T_(t, PTV_BGF_GI1)}; const long gi_step[2] = {(long)M * 3 * H, (long)M * 3 * H}, gi_ld[2] = {3L * H, 3L * H};
    const void* gi2[2] = {nullptr, nullptr}; const long z2[2] = {0, 0};
    const void* w16[2] = {T_(t, PTV_BGF_W16_HH0), T_(t, PTV_BGF_W16_HH1)};
    const float* bhh[2] = {(const float*)T_(t, PTV_BGF_B_HH0), (const float*)T_(t, PTV_BGF_B_HH1)};
    float* hall[2] = {M_<float>(t, PTV_BGF_HALL0), M_<float>(t, PTV_BGF_HALL1)};
    void* h16[2] = {M_<void>(t, PTV_BGF_H16_0), M_<void>(t, PTV_BGF_H16_1)};
    void* gates[2] = {M_<void>(t, PTV_BGF_GATES0), M_<void>(t, PTV_BGF_GATES1)};
    const int* len[2] = {(const int*)T_(t, PTV_BGF_LENGTHS), (const int*)T_(t, PTV_BGF_LENGTHS)}; const int rev[2] = {0, 1};
    void* xch[2] = {M_<void>(t, PTV_BGF_XCH0), M_<void>(t, PTV_BGF_XCH1)};
    if (t[PTV_BGF_WAIT_EVENT] && hipStreamWaitEvent(s, (hipEvent_t)const_cast<void*>(t[PTV_BGF_WAIT_EVENT]), 0) != hipSuccess) return PTV_ERR_LAUNCH;
    PTV_TRY(ptv_gru_persist_fwd(2, M, H, T, gi, gi_step, gi_ld, gi2, z2, z2, w16, bhh, hall, h16, gates, len, rev, xch, M_<unsigned>(t, PTV_BGF_SYNC),
                                stream));
    if (t[PTV_BGF_RECORD_EVENT] && hipEventRecord((hipEvent_t)const_cast<void*>(t[PTV_BGF_RECORD_EVENT]), s) != hipSuccess) return PTV_ERR_LAUNCH;
  }
  float* out = M_<float>(t, PTV_BGF_OUT);
  for (int dir = 0; dir < 2; dir++)
    PTV_TRY(ptv_copy2d(out + (long)dir * H, 2L * H, M_<float>(t, hall_slot[dir]) + (long)T * M * H, H, M, H, 1.f, 0, stream));
  return PTV_OK;
}

// ---------------------------------------------------------------------------------------------
// ptv_bigru_rows_fwd / ptv_bigru_rows_bwd: functional._bigru_forward / _bigru_backward's row-kernel branch (the note-summary bi-GRU)
// ---------------------------------------------------------------------------------------------
extern "C" int ptv_bigru_rows_fwd(const void* const* t, const long* d, void* stream) {
  if (!t || !d) return PTV_ERR_ARG;
  const int M = (int)d[PTV_BRF_D_M], T = (int)d[PTV_BRF_D_T], H = (int)d[PTV_BRF_D_H], I = (int)d[PTV_BRF_D_I];
  if (M <= 0 || T <= 0 || H != 128 || I != 128) return PTV_ERR_UNSUPPORTED;
  for (int i = 0; i < PTV_BRF_COUNT; i++)
    if (!t[i] && i != PTV_BRF_LENGTHS && i != PTV_BRF_PERM) return PTV_ERR_ARG;
  hipStream_t s = (hipStream_t)stream, side = (hipStream_t)const_cast<void*>(t[PTV_BRF_SIDE_STREAM]);
  hipEvent_t ef = (hipEvent_t)const_cast<void*>(t[PTV_BRF_FORK_EVENT]), ej = (hipEvent_t)const_cast<void*>(t[PTV_BRF_JOIN_EVENT]);
  float* out = M_<float>(t, PTV_BRF_OUT);
  auto rows = [&](int dir, hipStream_t st) -> int {
    const int o = dir ? 4 : 0, q = dir ? 3 : 0;
    float* hall = M_<float>(t, PTV_BRF_HALL0 + q);
    if (hipMemsetAsync(hall, 0, sizeof(float) * M * H, st) != hipSuccess) return PTV_ERR_LAUNCH;
    return ptv_row_gru_persist_fwd_perm(H, T_(t, PTV_BRF_PK_WG_H0 + o), T_(t, PTV_BRF_PK_WG_T0 + o), (const float*)T_(t, PTV_BRF_B_HH0 + o),
                                        (const float*)T_(t, PTV_BRF_B_IH0 + o), nullptr, (const float*)T_(t, PTV_BRF_X), (long)M * I,
                                        (const int*)T_(t, PTV_BRF_LENGTHS), (const int*)T_(t, PTV_BRF_PERM), hall, M_<void>(t, PTV_BRF_H16_0 + q),
                                        M_<void>(t, PTV_BRF_GATES0 + q), out + (long)dir * H, 2L * H, M, T, dir, (void*)st);
  };
  if (hipEventRecord(ef, s) != hipSuccess || hipStreamWaitEvent(side, ef, 0) != hipSuccess) return PTV_ERR_LAUNCH;
  PTV_TRY(rows(1, side));
  PTV_TRY(rows(0, s));
  if (hipEventRecord(ej, side) != hipSuccess || hipStreamWaitEvent(s, ej, 0) != hipSuccess) return PTV_ERR_LAUNCH;
  return PTV_OK;
}

extern "C" int ptv_bigru_rows_bwd(const void* const* t, const long* d, void* stream) {
  if (!t || !d) return PTV_ERR_ARG;
  const int M = (int)d[PTV_BRB_D_M], T = (int)d[PTV_BRB_D_T], H = (int)d[PTV_BRB_D_H], I = (int)d[PTV_BRB_D_I], dx_acc = d[PTV_BRB_D_DX_ACC] ? 1 : 0;
  const long dout_ld = d[PTV_BRB_D_DOUT_LD];
  if (M <= 0 || T <= 0 || H != 128 || I != 128) return PTV_ERR_UNSUPPORTED;
  const bool want_dx = t[PTV_BRB_DX] != nullptr;
  for (int i = 0; i < PTV_BRB_COUNT; i++) {
    const bool optional = i == PTV_BRB_LENGTHS || i == PTV_BRB_PERM || i == PTV_BRB_WT_IH0 || i == PTV_BRB_WT_IH1 || i == PTV_BRB_DX ||
                          i == PTV_BRB_TOP0 || i == PTV_BRB_TOP1 || i == PTV_BRB_SEG;
    if (!t[i] && !optional) return PTV_ERR_ARG;
  }
  if (want_dx && (!t[PTV_BRB_WT_IH0] || !t[PTV_BRB_WT_IH1])) return PTV_ERR_ARG;
  const int P = PTV_PREC_BF16;
  hipStream_t s = (hipStream_t)stream, side = (hipStream_t)const_cast<void*>(t[PTV_BRB_SIDE_STREAM]);
  hipEvent_t ef = (hipEvent_t)const_cast<void*>(t[PTV_BRB_FORK_EVENT]), ej = (hipEvent_t)const_cast<void*>(t[PTV_BRB_JOIN_EVENT]);
  const long TM = (long)T * M;
  const float* dout = (const float*)T_(t, PTV_BRB_DOUT);
  const void* x = T_(t, PTV_BRB_X);
  float* dx = M_<float>(t, PTV_BRB_DX);
  auto rows = [&](int dir, hipStream_t st) -> int {
    const int o = dir ? 5 : 0, g = dir ? 4 : 0, q = dir ? 4 : 0;
    void* dgi = M_<void>(t, PTV_BRB_DGI0 + q); void* dgh = M_<void>(t, PTV_BRB_DGH0 + q);
    int* top = M_<int>(t, PTV_BRB_TOP0 + q);
    PTV_TRY(ptv_row_gru_persist_bwd_perm(H, T_(t, PTV_BRB_PK_WT0 + o), T_(t, PTV_BRB_HALL0 + o), T_(t, PTV_BRB_GATES0 + o), nullptr, dout + (long)dir * H,
                                         dout_ld, (const int*)T_(t, PTV_BRB_LENGTHS), (const int*)T_(t, PTV_BRB_PERM), dgi, dgh, nullptr,
                                         M_<void>(t, PTV_BRB_SCRATCH0 + q), M, T, dir, top, (void*)st));
    WgradGroup wg(P, (void*)st);
    PTV_TRY(wg.add(3 * H, I, TM, dgi, 3L * H, 1, x, I, 0, M_<float>(t, PTV_BRB_G_W_IH0 + g), I, M_<float>(t, PTV_BRB_G_B_IH0 + g), top, top ? M : 0, 0));
    // (rows sorted by length: dgh and the states are indexed by POSITION -- the dead 128-row blocks of every step are skipped; the reversed
    // direction indexes its steps by processing order, so its segments run backwards)
    const int* seg = t[PTV_BRB_PERM] && top ? (const int*)T_(t, PTV_BRB_SEG) : nullptr;
    PTV_TRY(wg.add(3 * H, H, TM, dgh, 3L * H, 1, T_(t, PTV_BRB_H16_0 + o), H, 1, M_<float>(t, PTV_BRB_G_W_HH0 + g), H,
                   M_<float>(t, PTV_BRB_G_B_HH0 + g), top, top ? M : 0, dir ? T : 0, 1, seg, M, dir ? -T : T));
    return wg.flush();
  };
  auto dx_of = [&](int dir, int acc) -> int {
    const int o = dir ? 5 : 0, q = dir ? 4 : 0;
    const int* top = (const int*)T_(t, PTV_BRB_TOP0 + q);
    if (top) return ptv_gemm_mtop(P, 0, 0, (int)TM, I, 3 * H, T_(t, PTV_BRB_DGI0 + q), 3L * H, T_(t, PTV_BRB_WT_IH0 + o), 3L * H, dx, I, nullptr, 1.f, acc,
                                  0, 0, A16 | B16, top, M, stream);
    return ptv_gemm(P, 0, 0, (int)TM, I, 3 * H, T_(t, PTV_BRB_DGI0 + q), 3L * H, T_(t, PTV_BRB_WT_IH0 + o), 3L * H, dx, I, nullptr, 1.f, acc, 0, 0,
                    A16 | B16, stream);
  };
  if (hipEventRecord(ef, s) != hipSuccess || hipStreamWaitEvent(side, ef, 0) != hipSuccess) return PTV_ERR_LAUNCH;
  ptv_gemm_priority(0);
  PTV_TRY(rows(1, side));
  ptv_gemm_priority(1);
  PTV_TRY(rows(0, s));
  if (want_dx) PTV_TRY(dx_of(0, dx_acc));
  if (hipEventRecord(ej, side) != hipSuccess || hipStreamWaitEvent(s, ej, 0) != hipSuccess) return PTV_ERR_LAUNCH;
  if (want_dx) PTV_TRY(dx_of(1, 1));
  return PTV_OK;
}

// ---------------------------------------------------------------------------------------------
// ptv_decoder_free_fwd: functional_free.DecoderStepFn.forward's persistent path as one call (include/ptvae_hip.h: SURVEY.md 8b's
// decoder_free_fwd).  The same launches with the same arguments in the same order as the Python sequencing -- the same bits; the two
// per-time-step scratch matrices (gi, H0GC) are reused by every step (one stream: a step's consumers are queued before the next step's
// producer).
// ---------------------------------------------------------------------------------------------
extern "C" int ptv_decoder_free_fwd(const void* const* t, const long* d, const void* const* wl, const void* const* io, const void* const* wr,
                                    const void* const* ior, const unsigned* note_mask, const unsigned char* time_coin, void* stream) {
  if (!t || !d || !wl || !io || !note_mask || !time_coin) return PTV_ERR_ARG;
  const int B = (int)d[PTV_DFF_D_B], Zs = (int)d[PTV_DFF_D_ZS], Zi = (int)d[PTV_DFF_D_ZI], He = (int)d[PTV_DFF_D_HE], Ht = (int)d[PTV_DFF_D_HT],
            Hn = (int)d[PTV_DFF_D_HN], Hd = (int)d[PTV_DFF_D_HD], E = (int)d[PTV_DFF_D_E], NP = (int)d[PTV_DFF_D_NP];
  const long ldp = d[PTV_DFF_D_LDP];
  const bool train = d[PTV_DFF_D_TRAIN] != 0, replay = d[PTV_DFF_D_REPLAY] != 0, inference = d[PTV_DFF_D_INFERENCE] != 0;
  if (B <= 0 || E != 128 || He != 128 || Hn != 512 || Hd != 64 || NP != 130 || Ht <= 0 || (Ht & 7) || Zs <= 0 || Zi <= 0 || ldp < NP)
    return PTV_ERR_UNSUPPORTED;
  if (replay && !train) return PTV_ERR_ARG;
  bool need_resum = inference;
  for (int i = 0; i < 31 && !need_resum; i++) need_resum = !time_coin[i];
  {
    const int always[] = {PTV_DFF_Z, PTV_DFF_TOK0_SRC, PTV_DFF_W_ZHID, PTV_DFF_B_ZHID, PTV_DFF_W_ZIN, PTV_DFF_B_ZIN, PTV_DFF_W_IH_T, PTV_DFF_B_IH_T,
                          PTV_DFF_INIT_INPUT, PTV_DFF_B_HH_T, PTV_DFF_W_IH_T_OP, PTV_DFF_W_HH_T_OP, PTV_DFF_W_CAT, PTV_DFF_B_CAT, PTV_DFF_NS, PTV_DFF_NS16,
                          PTV_DFF_Z_IN, PTV_DFF_ZG, PTV_DFF_TOKS, PTV_DFF_GI, PTV_DFF_H0GC, PTV_DFF_TOK, PTV_DFF_PRED};
    for (int i : always) if (!t[i]) return PTV_ERR_ARG;
    if (train && !t[PTV_DFF_GATES_T]) return PTV_ERR_ARG;
    if (need_resum && (!wr || !ior)) return PTV_ERR_ARG;
    if (replay) {
      const int rp[] = {PTV_DFF_W_IH_N, PTV_DFF_B_IH_N, PTV_DFF_B_HH_N, PTV_DFF_W_DH, PTV_DFF_B_DH, PTV_DFF_W_HH_D, PTV_DFF_B_HH_D, PTV_DFF_TAB0, PTV_DFF_TAB,
                        PTV_DFF_W_OUT_D, PTV_DFF_B_OUT_D, PTV_DFF_PK_NOTES_H, PTV_DFF_PK_NOTES_T, PTV_DFF_PITCH, PTV_DFF_HN, PTV_DFF_HN16, PTV_DFF_GATES_N,
                        PTV_DFF_HD, PTV_DFF_HD16, PTV_DFF_IDX, PTV_DFF_GC16, PTV_DFF_DUR_SCR, PTV_DFF_IDX_SCR};
      for (int i : rp) if (!t[i]) return PTV_ERR_ARG;
      if (need_resum)
        for (int i : {PTV_DFF_PK_E_H0, PTV_DFF_PK_E_T0, PTV_DFF_B_HH_E0, PTV_DFF_B_IH_E0, PTV_DFF_PK_E_H1, PTV_DFF_PK_E_T1, PTV_DFF_B_HH_E1, PTV_DFF_B_IH_E1,
                      PTV_DFF_PLEN, PTV_DFF_XH0, PTV_DFF_XH1, PTV_DFF_XH16_0, PTV_DFF_XH16_1, PTV_DFF_XG0, PTV_DFF_XG1})
          if (!t[i]) return PTV_ERR_ARG;
    }
  }
  const int P = PTV_PREC_BF16;
  hipStream_t s = (hipStream_t)stream;
  const int R = 32 * B;
  const long M = 15L * R;
  float* NS = M_<float>(t, PTV_DFF_NS);
  __bf16* NS16 = M_<__bf16>(t, PTV_DFF_NS16);
  float* TOKS = M_<float>(t, PTV_DFF_TOKS);
  float* TOK = M_<float>(t, PTV_DFF_TOK);
  float* PRED = M_<float>(t, PTV_DFF_PRED);
  float* zg = M_<float>(t, PTV_DFF_ZG);
  const long ld_t = 2L * He + Zi;
  const float* w_ih_t = (const float*)T_(t, PTV_DFF_W_IH_T);
  // ---- prologue (ptvae.py:435-437,457-462)
  PTV_TRY(ptv_gemm(P, 0, 0, B, Ht, Zs, T_(t, PTV_DFF_Z), Zs, T_(t, PTV_DFF_W_ZHID), Zs, NS, Ht, (const float*)T_(t, PTV_DFF_B_ZHID), 1.f, 0, 0, 0, 0, stream));
  PTV_TRY(ptv_gemm(P, 0, 0, B, Zi, Zs, T_(t, PTV_DFF_Z), Zs, T_(t, PTV_DFF_W_ZIN), Zs, M_<void>(t, PTV_DFF_Z_IN), Zi, (const float*)T_(t, PTV_DFF_B_ZIN), 1.f, 0,
                   0, 0, 0, stream));
  PTV_TRY(ptv_gemm(P, 0, 0, B, 3 * Ht, Zi, T_(t, PTV_DFF_Z_IN), Zi, w_ih_t + 2 * He, ld_t, zg, 3L * Ht, (const float*)T_(t, PTV_DFF_B_IH_T), 1.f, 0, 0, 0, 0,
                   stream));
  PTV_TRY(ptv_copy2d(TOKS, 2L * He, (const float*)T_(t, PTV_DFF_INIT_INPUT), 0, B, 2 * He, 1.f, 0, stream));
  PTV_TRY(ptv_cast_bf16(NS, NS16, (long)B * Ht, stream));
  // the first note token of every time step: the <sos> embedding / the embedded ground-truth slot 0 (ptvae.py:388-392)
  PTV_TRY(ptv_copy2d(TOK, E, (const float*)T_(t, PTV_DFF_TOK0_SRC), d[PTV_DFF_D_TOK0_LDS], R, E, 1.f, 0, stream));
  PTV_TRY(ptv_copy2d(PRED, E, TOK, E, R, E, 1.f, 0, stream));
  // ---- the step loop
  const int wih_bf = d[PTV_DFF_D_W_IH_T_BF16] ? 1 : 0, whh_bf = d[PTV_DFF_D_W_HH_T_BF16] ? 1 : 0;
  float* gi = M_<float>(t, PTV_DFF_GI);
  float* h0gc = M_<float>(t, PTV_DFF_H0GC);
  if (io[18] != (const void*)h0gc) return PTV_ERR_ARG;
  __bf16* gates_t = M_<__bf16>(t, PTV_DFF_GATES_T);
  const int loop_flags = (int)d[PTV_DFF_D_LOOP_FLAGS], cluster = (int)d[PTV_DFF_D_CLUSTER];
  const void* ior_[7];
  if (need_resum) for (int i = 0; i < 7; i++) ior_[i] = ior[i];
  if (cluster && t[PTV_DFF_WAIT_EVENT] && hipStreamWaitEvent(s, (hipEvent_t)const_cast<void*>(t[PTV_DFF_WAIT_EVENT]), 0) != hipSuccess) return PTV_ERR_LAUNCH;
  if (cluster) {
    // the members' exchange words carry (time step, note step) tags that repeat from one forward pass to the next: whatever the caller did with
    // the buffer, this pass starts from tag 0 (64 KB per panel; the arrival counters likewise)
    if (!io[19] || !io[20]) return PTV_ERR_ARG;
    const long panels = (B + 15) / 16;
    if (hipMemsetAsync(const_cast<void*>(io[19]), 0, (size_t)panels * 65536, s) != hipSuccess ||
        hipMemsetAsync(const_cast<void*>(io[20]), 0, sizeof(unsigned) * (size_t)(panels + 1), s) != hipSuccess)
      return PTV_ERR_LAUNCH;
  }
  for (int ts = 0; ts < 32; ts++) {
    const float* tok_t = TOKS + (long)ts * B * 2 * He;
    PTV_TRY(ptv_gemm(P, 0, 0, B, 3 * Ht, 2 * He, tok_t, 2L * He, T_(t, PTV_DFF_W_IH_T_OP), ld_t, gi, 3L * Ht, nullptr, 1.f, 0, 0, 0, wih_bf ? B16 : 0, stream));
    // flags of ptv_gru_step_fwd: bit 0 gates bf16, bit 2 gi2 (zg: fp32 here), bit 4 weight bf16
    PTV_TRY(ptv_gru_step_fwd(P, B, Ht, NS + (long)ts * B * Ht, Ht, NS16 + (long)ts * B * Ht, NS16 + (long)(ts + 1) * B * Ht, gi, 3L * Ht, zg, 3L * Ht,
                             T_(t, PTV_DFF_W_HH_T_OP), (const float*)T_(t, PTV_DFF_B_HH_T), NS + (long)(ts + 1) * B * Ht, Ht,
                             train ? (void*)(gates_t + (long)ts * 4 * B * Ht) : nullptr, (long)B * Ht, nullptr, 0, nullptr,
                             (train ? 1 : 0) | (whh_bf << 4), stream));
    PTV_TRY(ptv_gemm(P, 0, 0, B, 4 * Hn, Ht, NS16 + (long)(ts + 1) * B * Ht, Ht, T_(t, PTV_DFF_W_CAT), Ht, h0gc, 4L * Hn, (const float*)T_(t, PTV_DFF_B_CAT), 1.f,
                     0, 0, 0, A16 | B16, stream));
    PTV_TRY(ptv_free_note_loop(wl, io, ldp, B, ts, inference ? 0u : note_mask[ts], loop_flags, stream));
    if (ts == 31) break;
    float* tok_next = TOKS + (long)(ts + 1) * B * 2 * He;
    if (!inference && time_coin[ts]) {
      if (!t[PTV_DFF_XS]) return PTV_ERR_ARG;
      PTV_TRY(ptv_copy2d(tok_next, 2L * He, (const float*)T_(t, PTV_DFF_XS) + (long)ts * B * 2 * He, 2L * He, B, 2 * He, 1.f, 0, stream));
    } else {
      ior_[6] = tok_next;
      PTV_TRY(ptv_free_resummarize(wr, ior_, B, ts, (int)d[PTV_DFF_D_RESUM_TRAIN], stream));
    }
  }
  if (cluster && t[PTV_DFF_RECORD_EVENT] && hipEventRecord((hipEvent_t)const_cast<void*>(t[PTV_DFF_RECORD_EVENT]), s) != hipSuccess) return PTV_ERR_LAUNCH;
  if (!replay) return PTV_OK;
  // ---- recompute what the backward reads, batched over all rows (the step loop stored decisions and tokens only)
  const float* w_ih_n = (const float*)T_(t, PTV_DFF_W_IH_N);
  const float* w_dh = (const float*)T_(t, PTV_DFF_W_DH);
  __bf16* HN16 = M_<__bf16>(t, PTV_DFF_HN16);
  float* HD = M_<float>(t, PTV_DFF_HD);
  __bf16* HD16 = M_<__bf16>(t, PTV_DFF_HD16);
  PTV_TRY(ptv_gemm(P, 0, 0, R, 3 * Hn, Ht, NS16 + (long)B * Ht, Ht, w_ih_n, (long)Ht + E, M_<void>(t, PTV_DFF_GC16), 3L * Hn, (const float*)T_(t, PTV_DFF_B_IH_N),
                   1.f, 0, 0, -1, A16 | C16 | CBLK16, stream));
  PTV_TRY(ptv_notes_gru_persist_fwd(T_(t, PTV_DFF_PK_NOTES_H), T_(t, PTV_DFF_PK_NOTES_T), (const float*)T_(t, PTV_DFF_B_HH_N), T_(t, PTV_DFF_GC16), TOK,
                                    M_<float>(t, PTV_DFF_HN), HN16, M_<void>(t, PTV_DFF_GATES_N), R, 15, stream));
  PTV_TRY(ptv_gemm(P, 0, 0, (int)M, Hd, Hn, HN16 + (long)R * Hn, Hn, w_dh, (long)Hn + NP, HD, Hd, (const float*)T_(t, PTV_DFF_B_DH), 1.f, 0, 0, 0, A16, stream));
  PTV_TRY(ptv_gemm(P, 0, 0, (int)M, Hd, NP, T_(t, PTV_DFF_PITCH), ldp, w_dh + Hn, (long)Hn + NP, HD, Hd, nullptr, 1.f, 1, 0, 0, 0, stream));
  PTV_TRY(ptv_dur_gru_fwd(Hd, M, HD, Hd, (const float*)T_(t, PTV_DFF_W_HH_D), (const float*)T_(t, PTV_DFF_B_HH_D), (const float*)T_(t, PTV_DFF_TAB0),
                          (const float*)T_(t, PTV_DFF_TAB), (const float*)T_(t, PTV_DFF_W_OUT_D), (const float*)T_(t, PTV_DFF_B_OUT_D), nullptr, M * Hd,
                          HD16 + M * Hd, M_<void>(t, PTV_DFF_GATES_D), M * Hd, 4 * M * Hd, 1, M_<float>(t, PTV_DFF_DUR_SCR), 10, M_<int>(t, PTV_DFF_IDX_SCR), M,
                          (const int*)T_(t, PTV_DFF_IDX), M, stream));
  PTV_TRY(ptv_cast_bf16(HD, HD16, M * Hd, stream));
  if (need_resum) {
    for (int dir = 0; dir < 2; dir++) {
      const int o = dir ? 4 : 0;
      PTV_TRY(ptv_row_gru_persist_fwd(He, T_(t, PTV_DFF_PK_E_H0 + o), T_(t, PTV_DFF_PK_E_T0 + o), (const float*)T_(t, PTV_DFF_B_HH_E0 + o),
                                      (const float*)T_(t, PTV_DFF_B_IH_E0 + o), nullptr, PRED, (long)R * E, (const int*)T_(t, PTV_DFF_PLEN),
                                      M_<float>(t, dir ? PTV_DFF_XH1 : PTV_DFF_XH0), M_<void>(t, dir ? PTV_DFF_XH16_1 : PTV_DFF_XH16_0),
                                      M_<void>(t, dir ? PTV_DFF_XG1 : PTV_DFF_XG0), nullptr, 0, R, 16, dir, stream));
    }
  }
  return PTV_OK;
}

// ---------------------------------------------------------------------------------------------
// ptv_decoder_free_bwd: functional_free.DecoderStepFn.backward as one call (include/ptvae_hip.h)
// ---------------------------------------------------------------------------------------------
extern "C" int ptv_decoder_free_bwd(const void* const* t_tf, const long* d_tf, const void* const* t_rows, const long* d_rows, const void* const* t,
                                    const long* d, void* stream) {
  if (!t_tf || !d_tf || !t || !d || (t_rows && !d_rows)) return PTV_ERR_ARG;
  const int B = (int)d[PTV_DFB_D_B], E = (int)d[PTV_DFB_D_E], He = (int)d[PTV_DFB_D_HE];
  if (B <= 0 || E <= 0 || He <= 0) return PTV_ERR_ARG;
  for (int i = 0; i < PTV_DFB_COUNT; i++)
    if (!t[i] && i != PTV_DFB_DX_PRED) return PTV_ERR_ARG;
  if ((t_rows != nullptr) != (t[PTV_DFB_DX_PRED] != nullptr)) return PTV_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const long R = 32L * B;
  float* demb = M_<float>(t, PTV_DFB_DEMB); float* dPRED = M_<float>(t, PTV_DFB_DPRED);
  // ---- duration GRU, heads, notes GRU, time GRU: the batched BPTT of the teacher-forced path on the recorded fed tokens
  PTV_TRY(ptv_decoder_tf_bwd(t_tf, d_tf, stream));
  // ---- route the token gradients: ground-truth embedding (coin set / slot 0) vs predicted tokens; time tokens likewise
  PTV_TRY(ptv_route_slices((const float*)T_(t, PTV_DFB_DTOK), demb, dPRED, (const int*)T_(t, PTV_DFB_MASK_TOK), (long)B * E, 15 * 32, 0, stream));
  PTV_TRY(ptv_route_slices((const float*)T_(t, PTV_DFB_DTOKS) + (long)B * 2 * He, M_<float>(t, PTV_DFB_DXS), M_<float>(t, PTV_DFB_DXSP),
                           (const int*)T_(t, PTV_DFB_MASK_TIME), (long)B * 2 * He, 32, 0, stream));
  if (t_rows) {
    PTV_TRY(ptv_bigru_rows_bwd(t_rows, d_rows, stream));
    PTV_TRY(ptv_copy2d(dPRED, E, (const float*)T_(t, PTV_DFB_DX_PRED), E, 16 * R, E, 1.f, 1, stream));
  }
  // ---- slot 0 of the predicted tokens is the ground-truth <sos> embedding; the rest -> note_embedding
  PTV_TRY(ptv_copy2d(demb, E, dPRED, E, R, E, 1.f, 1, stream));
  if (hipMemsetAsync(dPRED, 0, sizeof(float) * R * E, s) != hipSuccess) return PTV_ERR_LAUNCH;
  float* mh = M_<float>(t, PTV_DFB_MH);
  PTV_TRY(ptv_multihot((const long*)T_(t, PTV_DFB_XHAT), mh, 136, B, stream));
  WgradGroup wg(PTV_PREC_BF16, stream);
  PTV_TRY(wg.add(E, 135, 16 * R, dPRED, E, 0, mh, 136, 0, M_<float>(t, PTV_DFB_G_W_EMB), 135, M_<float>(t, PTV_DFB_G_B_EMB)));
  return wg.flush();
}

// ---------------------------------------------------------------------------------------------
// ptv_vae_loss_fwd / ptv_vae_loss_bwd: functional.VaeLossFn's two launch sequences
// ---------------------------------------------------------------------------------------------
extern "C" int ptv_vae_loss_fwd(const void* const* t, const long* d, const double* sc, void* stream) {
  if (!t || !d || !sc) return PTV_ERR_ARG;
  const int B = (int)d[PTV_VL_D_B], Z = (int)d[PTV_VL_D_Z], NP = (int)d[PTV_VL_D_NP], sm_p = (int)d[PTV_VL_D_SM_P], sm_c = (int)d[PTV_VL_D_SM_C];
  const long ldp = d[PTV_VL_D_LDP];
  const int need[] = {PTV_VL_X, PTV_VL_C, PTV_VL_PITCH, PTV_VL_DUR, PTV_VL_MU_C, PTV_VL_SD_C, PTV_VL_MU_R, PTV_VL_SD_R, PTV_VL_ROOT, PTV_VL_CHROMA,
                      PTV_VL_BASS, PTV_VL_PITCH_T, PTV_VL_DUR_T, PTV_VL_COUNTS, PTV_VL_ROOT_T, PTV_VL_CHROMA_T, PTV_VL_BASS_T, PTV_VL_SUMS, PTV_VL_OUT};
  for (int i : need) if (!t[i]) return PTV_ERR_ARG;
  if (B <= 0 || Z <= 0 || NP <= 0 || ldp < NP) return PTV_ERR_ARG;
  const long rows = (long)B * 480;
  float* sums = M_<float>(t, PTV_VL_SUMS);
  int* pitch_t = M_<int>(t, PTV_VL_PITCH_T); int* dur_t = M_<int>(t, PTV_VL_DUR_T); int* counts = M_<int>(t, PTV_VL_COUNTS);
  if (!d[PTV_VL_D_HAVE_TARGETS]) PTV_TRY(ptv_pianotree_targets((const long*)T_(t, PTV_VL_X), B, sm_p, pitch_t, dur_t, counts, stream));
  PTV_TRY(ptv_ce_fwd((const float*)T_(t, PTV_VL_PITCH), ldp, pitch_t, rows, NP, 130, sums + 0, stream));
  PTV_TRY(ptv_ce_fwd((const float*)T_(t, PTV_VL_DUR), 2, dur_t, rows * 5, 2, 2, sums + 1, stream));
  int* root_t = M_<int>(t, PTV_VL_ROOT_T); int* chroma_t = M_<int>(t, PTV_VL_CHROMA_T); int* bass_t = M_<int>(t, PTV_VL_BASS_T);
  PTV_TRY(ptv_chord_targets((const float*)T_(t, PTV_VL_C), B, sm_c, root_t, chroma_t, bass_t, stream));
  PTV_TRY(ptv_kl_fwd((const float*)T_(t, PTV_VL_MU_C), (const float*)T_(t, PTV_VL_SD_C), (long)B * Z, sums + 2, stream));
  PTV_TRY(ptv_kl_fwd((const float*)T_(t, PTV_VL_MU_R), (const float*)T_(t, PTV_VL_SD_R), (long)B * Z, sums + 3, stream));
  PTV_TRY(ptv_ce_fwd((const float*)T_(t, PTV_VL_ROOT), 12, root_t, (long)B * 8, 12, -1, sums + 4, stream));
  PTV_TRY(ptv_ce_fwd((const float*)T_(t, PTV_VL_CHROMA), 2, chroma_t, (long)B * 96, 2, -1, sums + 5, stream));
  PTV_TRY(ptv_ce_fwd((const float*)T_(t, PTV_VL_BASS), 12, bass_t, (long)B * 8, 12, -1, sums + 6, stream));
  return ptv_loss_finalize(sums, counts, (float)sc[0], (float)sc[1], (float)sc[2], (float)sc[3], (float)sc[4], (float)sc[5], M_<float>(t, PTV_VL_OUT),
                           stream);
}

extern "C" int ptv_vae_loss_bwd(const void* const* t, const long* d, const double* sc, void* stream) {
  if (!t || !d || !sc) return PTV_ERR_ARG;
  const int B = (int)d[PTV_VL_D_B], Z = (int)d[PTV_VL_D_Z], NP = (int)d[PTV_VL_D_NP];
  const long ldp = d[PTV_VL_D_LDP];
  const int need[] = {PTV_VL_PITCH, PTV_VL_DUR, PTV_VL_MU_C, PTV_VL_SD_C, PTV_VL_MU_R, PTV_VL_SD_R, PTV_VL_ROOT, PTV_VL_CHROMA, PTV_VL_BASS, PTV_VL_PITCH_T,
                      PTV_VL_DUR_T, PTV_VL_COUNTS, PTV_VL_ROOT_T, PTV_VL_CHROMA_T, PTV_VL_BASS_T, PTV_VL_GOUT, PTV_VL_GS, PTV_VL_DPITCH, PTV_VL_DDUR,
                      PTV_VL_DMU_C, PTV_VL_DSD_C, PTV_VL_DMU_R, PTV_VL_DSD_R, PTV_VL_DROOT, PTV_VL_DCHROMA, PTV_VL_DBASS};
  for (int i : need) if (!t[i]) return PTV_ERR_ARG;
  if (B <= 0 || Z <= 0 || NP <= 0 || ldp < NP) return PTV_ERR_ARG;
  const long rows = (long)B * 480;
  float* gs = M_<float>(t, PTV_VL_GS);
  PTV_TRY(ptv_loss_bwd_scales((const float*)T_(t, PTV_VL_GOUT), (const int*)T_(t, PTV_VL_COUNTS), (float)sc[0], (float)sc[1], (float)sc[2], (float)sc[3],
                              (float)sc[4], (float)sc[5], gs, stream));
  PTV_TRY(ptv_ce_bwd((const float*)T_(t, PTV_VL_PITCH), ldp, (const int*)T_(t, PTV_VL_PITCH_T), rows, NP, 130, gs + 0, M_<float>(t, PTV_VL_DPITCH), ldp,
                     stream));
  PTV_TRY(ptv_ce_bwd((const float*)T_(t, PTV_VL_DUR), 2, (const int*)T_(t, PTV_VL_DUR_T), rows * 5, 2, 2, gs + 1, M_<float>(t, PTV_VL_DDUR), 2, stream));
  PTV_TRY(ptv_kl_bwd((const float*)T_(t, PTV_VL_MU_C), (const float*)T_(t, PTV_VL_SD_C), (long)B * Z, gs + 2, M_<float>(t, PTV_VL_DMU_C),
                     M_<float>(t, PTV_VL_DSD_C), stream));
  PTV_TRY(ptv_kl_bwd((const float*)T_(t, PTV_VL_MU_R), (const float*)T_(t, PTV_VL_SD_R), (long)B * Z, gs + 3, M_<float>(t, PTV_VL_DMU_R),
                     M_<float>(t, PTV_VL_DSD_R), stream));
  PTV_TRY(ptv_ce_bwd((const float*)T_(t, PTV_VL_ROOT), 12, (const int*)T_(t, PTV_VL_ROOT_T), (long)B * 8, 12, -1, gs + 4, M_<float>(t, PTV_VL_DROOT), 12,
                     stream));
  PTV_TRY(ptv_ce_bwd((const float*)T_(t, PTV_VL_CHROMA), 2, (const int*)T_(t, PTV_VL_CHROMA_T), (long)B * 96, 2, -1, gs + 5, M_<float>(t, PTV_VL_DCHROMA),
                     2, stream));
  return ptv_ce_bwd((const float*)T_(t, PTV_VL_BASS), 12, (const int*)T_(t, PTV_VL_BASS_T), (long)B * 8, 12, -1, gs + 6, M_<float>(t, PTV_VL_DBASS), 12,
                    stream);
}

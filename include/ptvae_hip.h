/* ptvae_hip.h -- C ABI of libptvae_hip.so: the MI355X (gfx950) kernels behind the polyphonic-VAE
 * training step (reference: ZZWaang/polyphonic-chord-texture-disentanglement, ptvae.py / model.py).
 *
 * The reference has no native layer of its own: its "FFI" for this path is PyTorch's ATen
 * (nn.GRU / nn.Linear / nn.Conv2d / CrossEntropyLoss / Normal / Adam).  Each entry point below
 * names the reference call site(s) whose arithmetic it replaces.  INTEGRATION.md shows the
 * ctypes binding the Python host uses.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (torch tensors on the Python side) unless an entry point
 *     says "HOST array"; matrices are row-major with explicit leading dimensions (in elements)
 *   - element types: fp32 by default.  In bf16 precision the tensors that only ever feed MFMA operands or epilogues may be
 *     held as bf16 in HBM -- the caller says which through the dtype bits below (PTV_A/B/C_BF16 for ptv_gemm, PTV_GRU_*_BF16
 *     for the GRU entry points); parameters, optimiser state, recurrent state h, logits and all reductions stay fp32
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued on it, nothing synchronises
 *   - return value: 0 = ok, <0 = error (-1 bad argument, -2 launch failure, -3 unsupported shape for a
 *     specialised kernel: use the generic entry point); no exceptions cross the ABI
 *   - `prec`: PTV_PREC_F32 (0) = v_mfma_f32_16x16x4_f32, exact fp32 (parity path; every tensor fp32);
 *             PTV_PREC_BF16 (1) = v_mfma_f32_16x16x32_bf16, bf16 operands / fp32 accumulate (fp32 tensors are converted
 *             while staged, tensors flagged bf16 are read as they are)
 */
#ifndef PTVAE_HIP_H
#define PTVAE_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define PTV_PREC_F32 0
#define PTV_PREC_BF16 1

/* dtype bits of ptv_gemm: operands / result held as bf16 in HBM (bf16 precision only).  In bf16 mode
 * tensors that only ever feed MFMA operands are kept as bf16: half the traffic, no conversion. */
#define PTV_A_BF16 1
#define PTV_B_BF16 2
#define PTV_C_BF16 4
/* dtype bits of the GRU entry points */
#define PTV_GRU_GATES_BF16 1   /* saved gate planes */
#define PTV_GRU_GI_BF16 2      /* gi */
#define PTV_GRU_GI2_BF16 4     /* gi2 */
#define PTV_GRU_DG_BF16 8      /* dgi / dgh (backward) */
#define PTV_GRU_W_BF16 16       /* w_hh points at a bf16 copy of the weight (needs hall16 / DG_BF16);
                                   forward: W_hh [3H,H]; backward: the TRANSPOSED copy W_hh^T [H,3H] */
#define PTV_GRU_EXT_BF16 64     /* dh_ext (backward) holds bf16 */
#define PTV_GRU_SKIP_CAST0 32  /* hall16 slot 0 is already valid (chained single-step calls) */

/* Library / build identification ("gfx950"; ABI version 2 since round 3). */
const char* ptv_arch(void);
int ptv_abi_version(void);
/* first 16 hex digits of sha256(ptvae_hip.h || ptvae_hip_debug.h) as they were when the library was built (csrc/Makefile); a loader that
 * binds from these headers compares it with its own hash of them and refuses a stale library */
const char* ptv_header_hash(void);

/* ------------------------------------------------------------------------------------------------
 * Dense product  C[M,N] = act(alpha * A.B^T + bias) (+ C)      -- every nn.Linear forward
 * (ptvae.py:16-17,37-38,100-101,264-267,286-293), its input gradient (transB) and weight gradient
 * (transA+transB, split-K with atomic accumulation).
 *   transA = 0: A[m*lda + k]   1: A[k*lda + m]
 *   transB = 0: B[n*ldb + k] (nn.Linear weight layout)   1: B[k*ldb + n]
 *   act: 0 none, 1 exp (linear_var(...).exp_(), ptvae.py:27,120)
 *   splitk: 0 auto, >0 forced number of K splits, <0 never split
 *   dtypes: bit 0 / 1 / 2 = A / B / C hold bf16 (bf16 precision only); bit 3 (8) / bit 4 (16) = C is COLUMN-BLOCKED by w = 32 / 16:
 *           element (m, n) lives at ((n / w) * M + m) * w + n % w (ldc unused, N a multiple of w, no K split) -- the layout in which the
 *           row-partitioned recurrences (ptv_notes_gru_persist_fwd: gc, w = 16; ptv_notes_gru_persist_bwd: ext, w = 32) read their
 *           per-row operands: a wave touches one contiguous kilobyte instead of 16 half cache lines
 */
int ptv_gemm(int prec, int transA, int transB, int M, int N, int K,
             const void* A, long lda, const void* B, long ldb,
             void* C, long ldc, const float* bias, float alpha,
             int accumulate, int act, int splitk, int dtypes, void* stream);
/* the same with a row limit on A (transA = 0): the rows of A from (*m_top + 1) * m_unit on are known to be zero (device int written
 * by the kernel that produced A, see ptv_notes_gru_persist_bwd): those row tiles skip the product, C gets bias / stays as it is */
int ptv_gemm_mtop(int prec, int transA, int transB, int M, int N, int K, const void* A, long lda, const void* B, long ldb,
                  void* C, long ldc, const float* bias, float alpha, int accumulate, int act, int splitk, int dtypes,
                  const int* m_top, long m_unit, void* stream);
/* ... and with ROW SEGMENTS of A (round 6; seg_n or NULL): the rows of A are units of seg_unit rows (a multiple of 128: a note step's decoder rows
 * in length order) of which only the first seg_n[unit % seg_period] (device ints, multiples of 128) hold anything -- the other row tiles are
 * treated like the ones beyond m_top (no K loop; C = bias / zero / unchanged) */
int ptv_gemm_mtop_seg(int prec, int transA, int transB, int M, int N, int K, const void* A, long lda, const void* B, long ldb,
                      void* C, long ldc, const float* bias, float alpha, int accumulate, int act, int splitk, int dtypes,
                      const int* m_top, long m_unit, const int* seg_n, long seg_unit, int seg_period, void* stream);
/* plain products enqueued after this call raise their wave priority (p != 0) or run at the default one (0): host-side marker for the
 * launches of a latency chain that share the GPU with weight-gradient products on sibling streams.  Process-wide, read at enqueue time. */
int ptv_gemm_priority(int p);

/* ------------------------------------------------------------------------------------------------
 * GRU recurrence over T steps for M independent rows (torch.nn.GRU cell semantics; replaces the
 * per-step `self.gru(...)` / `dec_*_gru(...)` calls at ptvae.py:23,64-65,116,360,396-398,450,461-462,484).
 *   gi  [T] x [M,3H]   input-side pre-activations W_i x + b_i (gate order r,z,n), produced by ptv_gemm
 *   gi2 optional second addend with its own strides (may be NULL)
 *   hall [T+1][M][H]   slot 0 = initial state (written by the caller), slot s+1 = state after step s
 *   hall16 [T+1][M][H] optional bf16 copy of hall (bf16 precision): read as the MFMA operand of each step,
 *                 written next to hall, and reused by the caller as a bf16 operand of later products
 *   gates [T][4][M][H] saved r,z,n,(W_hn h + b_hn) for the backward pass, or NULL (inference)
 *   lengths[M] or NULL: packed-sequence masking (row m updated at time t iff t < lengths[m]),
 *                 the pack_padded_sequence semantics of ptvae.py:446-453,480-486
 *   reverse: processing step s consumes time index t = T-1-s (the *_reverse direction)
 *   gi_idx[M] or NULL: row m reads gi row gi_idx[m] instead of m (the duration GRU's input is a
 *                 one-hot token, so W_i x + b_i is a 2-row table indexed by the previous argmax)
 */
int ptv_gru_seq_fwd(int prec, int M, int H, int T,
                    const void* gi, long gi_step_stride, long gi_ld,
                    const void* gi2, long gi2_step_stride, long gi2_ld,
                    const void* w_hh, const float* b_hh,
                    float* hall, void* hall16, void* gates,
                    const int* lengths, int reverse, const int* gi_idx, int flags, void* stream);

/* One GRU cell step with explicit strides (same kernel as ptv_gru_seq_fwd, T = 1): the free-running
 * decoder (ptvae.py:395-424,460-486) advances a [B]-row window of the step-major buffers per step. */
int ptv_gru_step_fwd(int prec, int M, int H,
                     const float* hprev, long ld_hprev, const void* hprev16, void* hout16,
                     const void* gi, long gi_ld, const void* gi2, long gi2_ld,
                     const void* w_hh, const float* b_hh,
                     float* hout, long ld_hout,
                     void* gates, long gates_plane,
                     const int* lengths, int t, const int* gi_idx, int flags, void* stream);

/* BPTT through ptv_gru_seq_fwd (replaces autograd through the same call sites).
 *   dh_ext [T] x [M,H]  gradient arriving at the state after processing step s (may be NULL)
 *   dh_last [M,H]       gradient arriving at the final state only (may be NULL)
 *   lr_a/lr_b           optional low-rank external gradient: dh_s += lr_a_s[M,k] . lr_b[k,H]
 *                       (the 2-wide dur_out_linear feeding back into the duration GRU state)
 *   dgi [T][M][3H] (indexed by TIME t), dgh [T][M][3H] (indexed by processing step s)
 *   dhz scratch [2][M][H];  dh0 [M,H] gradient w.r.t. the initial state (may be NULL)
 * Weight gradients follow with ptv_gemm(transA=1,transB=1): dW_hh += dgh^T.hall[0:T], dW_ih += dgi^T.x
 */
int ptv_gru_seq_bwd(int prec, int M, int H, int T,
                    const float* hall, const void* gates, const void* w_hh,
                    const void* dh_ext, long ext_step_stride, long ext_ld,
                    const float* dh_last, long last_ld,
                    const float* lr_a, long lr_step_stride, long lr_lda, int lr_k, const float* lr_b,
                    void* dgi, void* dgh, float* dhz, float* dh0,
                    int reverse, int flags, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Persistent, weight-stationary form of ptv_gru_seq_fwd / ptv_gru_seq_bwd for the small-M chains (the 32-step
 * dec_time_gru ptvae.py:461-462, the encoder bi-GRUs ptvae.py:23,116, the chord decoder ptvae.py:64-65):
 * ONE launch walks all T steps of up to 4 independent chains (e.g. both directions of a bi-GRU).  A workgroup
 * keeps its W_hh slice in LDS and the state of its own cells in registers for the whole sequence; per step only
 * the bf16 MFMA operand (h_s, or dgh_s in the BPTT) is exchanged between the workgroups of one row group
 * (write-through stores + arrival counter + L1-bypassing loads).  bf16 precision / bf16 storage only:
 *   gi, gi2, gates, dgi, dgh are bf16; w_hh16 = bf16 W_hh [3H,H]; w_t16 = bf16 W_hh^T [H,3H]; hall fp32 + hall16.
 * Same tensor layouts and semantics as ptv_gru_seq_fwd / ptv_gru_seq_bwd (dgi by time, dgh by processing step).
 * All array arguments are HOST arrays of NC entries (device pointers / strides per chain).
 * `xch`: per chain a bf16 scratch tensor for the exchanged operand, (T+1)*M*H elements (forward) / T*M*3H (backward),
 *   held K-blocked [step][k/8][row][8] so that the consumers' MFMA-fragment loads are contiguous across lanes.
 * `sync`: 16 * (1 + 32) device words (the host allocates 16 * (1 + 32 + 128): see the split-K form below) ZEROED by the caller on `stream` before the call: word 0 = error flag (non-zero
 *   after the launch = a bounded spin gave up, results invalid), word 16*(1+g) = arrival counter of row group g.
 * Returns PTV_ERR_UNSUPPORTED (-3) when the shape does not fit one workgroup per CU (H % 256, H <= 1024,
 * rows per workgroup <= 256): the caller then uses the per-step entry points.  At most ONE persistent launch may
 * be in flight on the device at a time (chain them with events across streams).
 */
int ptv_gru_persist_supported(int NC, int M, int H);
/* CUs the persistent grids leave free (default 0 = size to every CU): a data-parallel run may set it so that RCCL's channel kernels
 * always find a CU without a 96-KB workgroup on it (dist.GradSync, PTV_PERSIST_CU_RESERVE); the grids are sized from CUs - reserve */
int ptv_gru_persist_cu_reserve(int cus);
/* how consumers read the exchanged operand: 0 = sc1 loads, 1 = nt loads, 2 = plain loads behind one agent acquire per step */
int ptv_gru_persist_load_policy(int lp);
int ptv_gru_persist_fwd(int NC, int M, int H, int T,
                        const void* const* gi, const long* gi_step, const long* gi_ld,
                        const void* const* gi2, const long* gi2_step, const long* gi2_ld,
                        const void* const* w_hh16, const float* const* b_hh,
                        float* const* hall, void* const* hall16, void* const* gates,
                        const int* const* lengths, const int* reverse, void* const* xch, unsigned* sync, void* stream);
int ptv_gru_persist_bwd(int NC, int M, int H, int T,
                        const float* const* hall, const void* const* gates, const void* const* w_t16,
                        const void* const* dh_ext, const long* ext_step, const long* ext_ld, const int* ext_bf16,
                        const float* const* dh_last, const long* last_ld,
                        void* const* dgi, void* const* dgh, float* const* dh0,
                        const int* reverse, void* const* xch, unsigned* sync, void* stream);
/* The BPTT with split-K TEAMS (round 4): S = 2 or 4 consecutive workgroups share 16*S hidden units and split K = 3H, so a workgroup
 * reads 1/S of the exchanged operand per step (the classic kernel is bound by exactly that read: 64 unit groups x M x 3H bf16 per
 * step); the team adds its fp32 partial tiles in a fixed order (bit-reproducible).  Same arguments and results as
 * ptv_gru_persist_bwd (summation order differs) plus
 *   part: per chain ptv_gru_persist_part_elems(NC, M, H, S) floats of scratch (partial tiles, ring of two steps),
 *   sync: 16 * (1 + 32 + 128) ZEROED words (row-group counters as above, then one counter per team from word 16*33 on).
 * Up to 512 rows per workgroup: the four chains of two bi-GRUs at M = 512, H = 1024 fit ONE launch. */
int ptv_gru_persist_splitk_supported(int NC, int M, int H, int S);
long ptv_gru_persist_part_elems(int NC, int M, int H, int S);
int ptv_gru_persist_bwd_splitk(int S, int NC, int M, int H, int T,
                               const float* const* hall, const void* const* gates, const void* const* w_t16,
                               const void* const* dh_ext, const long* ext_step, const long* ext_ld, const int* ext_bf16,
                               const float* const* dh_last, const long* last_ld,
                               void* const* dgi, void* const* dgh, float* const* dh0,
                               const int* reverse, void* const* xch, float* const* part, unsigned* sync, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Streaming helpers (layout shuffles and reductions that torch would do with cat/transpose/sum).
 */
/* dst[r*ldd + c] = (accumulate ? dst : 0) + alpha*src[r*lds + c]   (lds = 0 broadcasts one row) */
int ptv_copy2d(float* dst, long ldd, const float* src, long lds, long rows, int cols, float alpha,
               int accumulate, void* stream);
/* [D0,D1,W] -> [D1,D0,W]  (batch-major API tensors <-> step-major internal layout) */
int ptv_transpose01(float* dst, const float* src, int D0, int D1, int W, void* stream);
/* out[i] = (accumulate ? out[i] : 0) + sum_t in[t*stride + i] */
int ptv_sum_steps(float* out, const void* in, long n, int T, long stride, int accumulate, int in_bf16, void* stream);
/* the same over the planes 0 .. *t_top only (device int; the later planes are known to be zero, see ptv_notes_gru_persist_bwd) */
/* *top = max(*top, index of the last `unit`-row block of x [rows, cols] (fp32, row stride ld) that holds a non-zero): which trailing
 * note steps of a gradient received nothing (the loss ignores padded slots) -- the limit handed to ptv_gemm_mtop / ptv_wgrad */
/* process-wide switch (default 1): the backward kernels pass over work whose result is exactly zero -- note steps / tiles at which no
 * gradient arrives (tested on the arriving gradient), panel steps beyond the longest packed sequence.  0 = run everything dense. */
int ptv_zero_skip(int enable);
int ptv_last_nonzero_unit(const float* x, long rows, int cols, long ld, long unit, int* top, void* stream);
int ptv_sum_steps_top(float* out, const void* in, long n, int T, long stride, int accumulate, int in_bf16, const int* t_top,
                      void* stream);
/* ... and with ROW SEGMENTS (round 6; seg_n or NULL): the planes are note steps over length-sorted rows of row_elems elements each, plane t holds
 * something in its first seg_n[t] rows only (device ints, not growing with t: ptv_rows_seg_counts) -- the rest is zero and is not read */
int ptv_sum_steps_seg(float* out, const void* in, long n, int T, long stride, int accumulate, int in_bf16, const int* t_top,
                      const int* seg_n, long row_elems, void* stream);
/* out[g*N + n] += sum over rows r with (sel ? sel[r] : 0) == g of A[r*lda + n]   (bias gradients;
 * with sel: the duration GRU's W_ih gradient, whose inputs are one-hot tokens) */
int ptv_colsum(float* out, const void* A, long lda, long rows, int N, const int* sel, int G, int a_bf16, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Note embedding as a gather: PtvaeDecoder.emb_x (ptvae.py:531-535) = get_len_index_tensor
 * (ptvae.py:292-297) + index_tensor_to_multihot_tensor (ptvae.py:299-313) + note_embedding Linear.
 *   x [B,32,16,6] int64 -> emb STEP-MAJOR [16][32][B][E], lengths [32][B] int32 (lengths may be NULL)
 * bwd: ptv_multihot materialises the [16*32*B, 135] multi-hot matrix (same row order, row stride ld) so
 *      that dW[E,135] = demb^T . multihot runs as a split-K ptv_gemm; dbias = ptv_colsum(demb).
 */
int ptv_embed_fwd(const long* x, const float* W, const float* bias, float* emb, int* lengths, int B, int E, void* stream);
int ptv_multihot(const long* x, float* out, long ld, int B, void* stream);
/* get_len_index_tensor alone (ptvae.py:292-297): lengths [32][B] int32 = 16 - number of <pad> pitches of each (step, sample) */
int ptv_grid_lengths(const long* x, int* lengths, int B, void* stream);
/* the same rows as bf16 (exact), ld >= 136 with column 135 zeroed: operand of the bf16-precision note_embedding weight gradient */
int ptv_multihot_bf16(const long* x, void* out, long ld, int B, void* stream);
/* The same three for ANY grid geometry the reference's constructors accept (ptvae.py:127-147 PtvaeEncoder, :220-241 PtvaeDecoder):
 * S = num_step, N = max_simu_note, P = pitch_range = max_pitch - min_pitch + 3, D = dur_width (<= 8), pad = pitch_pad.
 *   x [B,S,N,1+D] int64 -> emb step-major [N][S][B][E], lengths [S][B], multihot rows [N*S*B][P+D] (bf16 != 0: bf16 rows zero-filled to
 *   the 8-column granule, ld >= that).  A pitch index outside [0, P) contributes no pitch column (the reference drops column P, the
 *   <pad> column: ptvae.py:186,311).  train.py:32 constructs PtvaeEncoder(max_pitch=31): P = 34. */
int ptv_embed_fwd_geom(const long* x, const float* W, const float* bias, float* emb, int* lengths, int B, int E,
                       int S, int N, int P, int D, int pad, void* stream);
int ptv_grid_lengths_geom(const long* x, int* lengths, int B, int S, int N, int D, int pad, void* stream);
int ptv_multihot_geom(const long* x, void* out, long ld, int B, int S, int N, int P, int D, int bf16, void* stream);

/* ------------------------------------------------------------------------------------------------
 * TextureEncoder front end (ptvae.py:95-99,112-114): Conv2d(1,C,(4,12),stride(4,1)) + ReLU +
 * MaxPool2d((1,4)) fused; pr_mat [B,32,128] -> pooled [B,C,8,29].  bwd recomputes the conv and
 * accumulates dW[C,48], dbias[C] (pr_mat is an input: no dX).
 */
int ptv_txt_conv_relu_pool_fwd(const float* pr_mat, const float* w, const float* bias, float* pooled, int B, int C, void* stream);
int ptv_txt_conv_relu_pool_bwd(const float* pr_mat, const float* w, const float* bias, const float* dpooled,
                               float* dw, float* dbias, int B, int C, void* stream);
/* the same with the pooled map held as the rows of the reference's raw view (ptvae.py:114): feat [B*8][ld], ld >= C*29 -- element
 * (b, ch, beat, pp) at row b*8 + f / (C*29), column f % (C*29), f = (ch*8 + beat)*29 + pp.  A 16-byte-multiple ld keeps the rows of fc1's
 * operand aligned for the MFMA loaders; ld = C*29 is the plain tensor of the entry points above. */
int ptv_txt_conv_relu_pool_fwd_rows(const float* pr_mat, const float* w, const float* bias, float* feat, long ld, int B, int C,
                                    signed char* arg, void* stream);
/* arg (may be NULL): int8 [B*8][C*29], written by the forward -- which of the 4 pooled positions won (-1: ReLU cut all four).  Given to the
 * backward, the convolution is not recomputed (w / bias may then be NULL): same decisions, same gradients, a fifth of the arithmetic
 * (this kernel is the LAST launch of the backward pass's longest chain) */
int ptv_txt_conv_relu_pool_bwd_rows(const float* pr_mat, const float* w, const float* bias, const float* dfeat, long ld,
                                    float* dw, float* dbias, int B, int C, const signed char* arg, void* stream);

/* ------------------------------------------------------------------------------------------------
 * reparameterize(): get_zs_from_dists / Normal.rsample (amc_dl/torch_plus/train_utils.py:33-34)
 *   z[b*ldz + j] = mu + sd*eps (eps NULL -> z = mu); kl_sum += sum(-log sd + (sd^2+mu^2)/2 - 1/2)
 * bwd: dmu = dz + klw*mu + dmu_ext ; dsd = dz*eps + klw*(sd - 1/sd) + dsd_ext ; dlv = dsd*sd
 *      (lv = log sd is the linear_var output, ptvae.py:27,120); any of dz/eps/ext may be NULL;
 *      mul_sd = 0 returns dsd itself in `dlv`.
 */
/* eps for the reparameterisation as a pure function of (seed, stream_id, GLOBAL sample index row_offset + r, column):
 * Philox4x32-10 + Box-Muller.  The same batch gives the same noise however it is sharded over ranks (SURVEY.md 8 d/e). */
int ptv_philox_normal(float* out, long rows, int Z, unsigned long long seed, unsigned long long stream_id, long row_offset,
                      void* stream);
int ptv_reparam_kl_fwd(const float* mu, const float* sd, const float* eps, float* z, long ldz, float* kl_sum, int B, int Z, void* stream);
int ptv_reparam_kl_bwd(const float* mu, const float* sd, const float* eps, const float* dz, long lddz,
                       const float* dmu_ext, const float* dsd_ext, float klw, int mul_sd, float* dmu, float* dlv, int B, int Z, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Losses: DisentangleVAE.loss_function (model.py:57-68), PtvaeDecoder.recon_loss (ptvae.py:498-511),
 * chord_loss (model.py:70-83), kl_loss (model.py:85-90) + kl_with_normal (train_utils.py:45-49).
 *   targets: int32 arrays in the row order of the logits (step_major = 1: [15][32][B], else [B][32][15];
 *            chord: [8][B] or [B][8]); counts[0/1] += number of non-ignored pitch / duration targets; counts[2] (three zero-initialised
 *            ints) = max(counts[2], last note step 0..14 that holds any non-ignored target): an upper bound of where the logits' gradient
 *            can be non-zero, the decoder backward's zero-skip limit (round 4; replaces two scans of the 134-MB gradient)
 *   ptv_ce_fwd: nll_sum += sum over non-ignored rows of -log softmax(logits)[target]
 *   ptv_ce_bwd: dlogits = gscale[0] * (softmax - onehot), 0 on ignored rows (gscale is a DEVICE scalar)
 *   ptv_loss_finalize: 7 sums + 2 counts -> the 11 scalars in train.py:54-55 order
 *   ptv_loss_bwd_scales: upstream grads of the 11 scalars -> 7 per-component scale factors (device)
 */
int ptv_pianotree_targets(const long* x, int B, int step_major, int* pitch_t, int* dur_t, int* counts, void* stream);
/* the same + row_live (int32 [32 B], ZEROED by the caller, or NULL): row (t, b)'s number of live note steps = 1 + the last note step at
 * which it holds a target (what the teacher-forced decoder needs of that row when only the loss consumes it) */
int ptv_pianotree_targets_rows(const long* x, int B, int step_major, int* pitch_t, int* dur_t, int* counts, int* row_live, void* stream);
/* rows by index, `planes` planes of `rows` rows of row_words 4-byte words: gather dst[pl][p] = src[pl][idx[p]], scatter dst[pl][idx[p]] =
 * src[pl][p] (plane strides in words).  The decoder's rows in length-sorted order and back. */
int ptv_gather_rows(void* dst, const void* src, const int* idx, long rows, int row_words, long src_plane_words, long dst_plane_words, int planes,
                    void* stream);
int ptv_scatter_rows(void* dst, const void* src, const int* idx, long rows, int row_words, long src_plane_words, long dst_plane_words, int planes,
                     void* stream);
/* ... with ROW SEGMENTS (round 6; seg_n int32 [planes] or NULL = the calls above): of plane q only the first seg_n[q] rows IN SORTED ORDER hold
 * anything (ptv_rows_seg_counts).  The gather leaves the other rows of dst unwritten (its readers clip to the same segments); the scatter
 * writes zeros to their places in dst without reading src. */
int ptv_gather_rows_seg(void* dst, const void* src, const int* idx, long rows, int row_words, long src_plane_words, long dst_plane_words,
                        int planes, const int* seg_n, void* stream);
int ptv_scatter_rows_seg(void* dst, const void* src, const int* idx, long rows, int row_words, long src_plane_words, long dst_plane_words,
                         int planes, const int* seg_n, void* stream);
int ptv_chord_targets(const float* c, int B, int step_major, int* root_t, int* chroma_t, int* bass_t, void* stream);
int ptv_ce_fwd(const float* logits, long ld, const int* targets, long rows, int C, int ignore_index, float* nll_sum, void* stream);
int ptv_ce_bwd(const float* logits, long ld, const int* targets, long rows, int C, int ignore_index, const float* gscale,
               float* dlogits, long ldd, void* stream);
/* weighted duration loss (recon_loss(..., weighted_dur=True), ptvae.py:512-527): the 5 duration bit positions are 5 separate
 * CrossEntropyLoss(ignore_index = 2) means combined with weights (1, .6, .4, .3, .3).  Row r of the [rows*5, 2] logits
 * belongs to group r % G: per-group nll sums / valid counts (fwd), per-group gradient scales (bwd).
 * ptv_wdur_finalize folds them into (sums1 = dl, counts1 = 1) so ptv_loss_finalize / ptv_loss_bwd_scales apply unchanged;
 * ptv_wdur_scales turns the upstream scale of dl into the 5 per-group scales gs1 * w[d] / cnt[d]. */
int ptv_ce_group_fwd(const float* logits, long ld, const int* targets, long rows, int C, int ignore_index, int G,
                     float* nll_sum, int* count, void* stream);
int ptv_ce_group_bwd(const float* logits, long ld, const int* targets, long rows, int C, int ignore_index, int G,
                     const float* gscale, float* dlogits, long ldd, void* stream);
int ptv_wdur_finalize(const float* gsum5, const int* gcnt5, float w0, float w1, float w2, float w3, float w4, float* sums1,
                      int* counts1, void* stream);
int ptv_wdur_scales(const float* gs1, const int* gcnt5, float w0, float w1, float w2, float w3, float w4, float* out5,
                    void* stream);
int ptv_kl_fwd(const float* mu, const float* sd, long n, float* kl_sum, void* stream);
int ptv_kl_bwd(const float* mu, const float* sd, long n, const float* gscale, float* dmu, float* dsd, void* stream);
int ptv_loss_finalize(const float* sums, const int* counts, float beta, float w0, float w1, float n_kl, float n_root,
                      float n_chroma, float* out11, void* stream);
int ptv_loss_bwd_scales(const float* gout11, const int* counts, float beta, float w0, float w1, float n_kl, float n_root,
                        float n_chroma, float* gs7, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Duration head (ptvae.py:361-367): dur_out[r*ld_out + 0..1] = dur_out_linear(h[r]); idx[r] = argmax
 * (force_idx, if given, overrides the argmax: replay mode for parity checks).
 */
int ptv_dur_out_token(const float* h, int H, const float* w_out, const float* b_out, float* dur_out, long ld_out,
                      int* idx, const int* force_idx, long rows, void* stream);
/* weight gradient of dur_out_linear (autograd of ptvae.py:361-362 over the 5 duration steps) in one pass over the bf16 state
 * planes: gw[c*64 + u] += sum_d sum_m ddur[m*ld_dd + 2d + c] * hall16[(d+1)*plane_h + m*64 + u]   (H = 64) */
int ptv_dur_out_wgrad(const float* ddur, long ld_dd, const void* hall16, long plane_h, float* gw, long rows, int H, void* stream);

/* ------------------------------------------------------------------------------------------------
 * The whole 5-step duration GRU of decode_note (ptvae.py:353-367) in ONE kernel (bf16 precision, H = 64):
 * W_hh and the 3-row gate table stay in LDS, the state in registers, each wave owns 16 rows for all 5
 * steps; est_dur and the argmax feedback are computed in the same kernel.  Returns PTV_ERR_ARG for other
 * H (callers then use ptv_gru_step_fwd + ptv_dur_out_token per step).
 *   hall / hall16: state after step d at base + d*plane_h + row*64 (either may be NULL)
 *   gates: plane p of step d at base + d*step_g + p*plane_g + row*64 (NULL = inference)
 *   dur_out[row*ld_out + 2d..2d+1], idx[d*idx_stride + row]; force (replay) may be NULL
 */
int ptv_dur_gru_fwd(int H, long M, const float* h0, long ld_h0, const float* w_hh, const float* b_hh,
                    const float* tab0, const float* tab, const float* w_out, const float* b_out,
                    float* hall, long plane_h, void* hall16, void* gates, long plane_g, long step_g, int gates_bf16,
                    float* dur_out, long ld_out, int* idx, long idx_stride, const int* force, long force_stride,
                    void* stream);
/* the same with a LIVE-ROW LIMIT: m_top (device int32, or NULL = all rows) and m_unit (a multiple of 16) -- only the rows below
 * (*m_top + 1) * m_unit are computed; the other rows of hall / hall16 / gates / dur_out / idx stay unwritten.  For callers whose loss
 * ignores the padded note slots (ptvae.py:498-511): rows are ordered [note step][32 B], m_unit = 32 B, *m_top = the last note step with a
 * target. */
int ptv_dur_gru_fwd_top(int H, long M, const float* h0, long ld_h0, const float* w_hh, const float* b_hh,
                        const float* tab0, const float* tab, const float* w_out, const float* b_out,
                        float* hall, long plane_h, void* hall16, void* gates, long plane_g, long step_g, int gates_bf16,
                        float* dur_out, long ld_out, int* idx, long idx_stride, const int* force, long force_stride,
                        const int* m_top, long m_unit, void* stream);
/* ... and with the rows of every note step sorted by descending length (row_len int32 [m_unit]): 16-row tiles whose first row has no target
 * at their note step are passed over (outputs unwritten) */
int ptv_dur_gru_fwd_rows(int H, long M, const float* h0, long ld_h0, const float* w_hh, const float* b_hh,
                         const float* tab0, const float* tab, const float* w_out, const float* b_out,
                         float* hall, long plane_h, void* hall16, void* gates, long plane_g, long step_g, int gates_bf16,
                         float* dur_out, long ld_out, int* idx, long idx_stride, const int* force, long force_stride,
                         const int* m_top, long m_unit, const int* row_len, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Backward of the 5-step duration GRU (autograd of ptvae.py:353-367) in ONE kernel (bf16 gates, H = 64): dh
 * stays in registers over the 5 steps, dgh_d . W_hh runs against W_hh^T resident in LDS, and the parameter
 * gradients are accumulated in-kernel from LDS-transposed row tiles.  Writes dh0 [M,64] (gradient reaching
 * dur_hid_linear's output) and one [256 x 80] fp32 partial per block into part[nblocks][ptv_dur_gru_bwd_part_size()]:
 *   rows 0..191 = (dr, dz, dn*r) gate units, rows 192..255 = dn;  columns 0..63 = . h_d (dW_hh),
 *   columns 64..66 = sums over the rows whose step input was <sos> / one-hot token 0 / token 1.
 * ptv_dur_bwd_finalize folds the column-summed partial S [256 x 80] into the gradients of weight_hh, bias_hh,
 * bias_ih, weight_ih [192, I] and the <sos> token [I] (all +=).
 *   gates/hall as written by ptv_dur_gru_fwd (hall: the fp32 states or, h_bf16 = 1, their bf16 copies hall16); ddur [M, 10] = d loss / d est_dur; idx[d*idx_stride + row].
 *   gates == NULL: RECOMPUTE mode -- the forward was called with gates = NULL (it then writes a third of the bytes) and the backward
 *   rebuilds r, z, n, hn of each step from h_{d-1}, W_hh, b_hh and the gate tables tab0 / tab the forward was given (the forward's own
 *   MFMA product and fp32 expressions, hence the forward's values); otherwise b_hh / tab0 / tab may be NULL.
 */
int ptv_dur_gru_bwd_part_size(void);
int ptv_dur_gru_bwd(int H, long M, const void* gates, long plane_g, long step_g, const void* hall, long plane_h, int h_bf16,
                    const float* ddur, long ld_dd, const float* w_hh, const float* w_out,
                    const int* idx, long idx_stride, float* dh0, float* part, int nblocks,
                    const float* b_hh, const float* tab0, const float* tab, void* stream);
int ptv_dur_bwd_finalize(const float* S, float* g_whh, float* g_bhh, float* g_bih, float* g_wih, float* g_sos,
                         const float* w_ih, const float* sos, int I, void* stream);

/* ------------------------------------------------------------------------------------------------
 * The per-note heads fused (csrc/heads.hip, bf16 precision, init_model() geometry Hn = 512 / 130 classes / Hd = 64): one pass over the
 * note summaries for decode_note's two Linears (ptvae.py:343-352) and one for their input gradients.
 *   fwd: pitch [M][ldp] = hn16 . W_p^T + b_p (fp32);  hd0 [M][64] = hn16 . W_dh[:, :512]^T + pitch . W_dh[:, 512:]^T + b_dh
 *        (+ bf16 copy hd16, may be NULL).  wp / wdh / wdp: ptv_pack_mfma_b of W_p [130,512], W_dh[:, :512] [64,512], W_dh[:, 512:] [64,130].
 *   bwd: dp [M][ldp] += dhd0 . W_dh[:, 512:] (in place);  dnsum [M][512] bf16 = dp . W_p + dhd0 . W_dh[:, :512], row-major or
 *        (blocked != 0) column-blocked by 32 ([16][M][32], what ptv_notes_gru_persist_bwd reads).  wdpT: pack of W_dh[:, 512:]^T
 *        [130,64];  wcat: PAIR-interleaved pack of the [512][224] matrix [W_p^T (130 columns, zero-padded to 160) | W_dh[:, :512]^T].
 *        Rows from (*m_top + 1) * m_unit on (device int, or NULL) are known to be zero: their dnsum is written as zeros -- or, with bit 1 of
 *        `blocked` set (blocked = 3), left unwritten: the consumer is ptv_notes_gru_persist_bwd_top with the same limit as its bound.
 *        dy16 (or NULL): [M][200] bf16 = [dp (130) | 0 (6) | dhd0 (64)] of the live rows -- ONE ptv_wgrad operand for the weight gradients
 *        of both Linears over the note summaries (one pass over the [M][512] summaries instead of two).
 */
int ptv_heads_fwd(const void* hn16, const void* wp_packed, const void* wdh_packed, const void* wdp_packed, const float* b_p,
                  const float* b_dh, float* pitch, long ldp, float* hd0, void* hd16, long M, void* stream);
/* the same with the live-row limit of ptv_dur_gru_fwd_top (m_unit a multiple of 128): later rows of pitch / hd0 / hd16 stay unwritten */
int ptv_heads_fwd_top(const void* hn16, const void* wp_packed, const void* wdh_packed, const void* wdp_packed, const float* b_p,
                      const float* b_dh, float* pitch, long ldp, float* hd0, void* hd16, long M, const int* m_top, long m_unit,
                      void* stream);
int ptv_heads_bwd(float* dp, long ldp, const float* dhd0, const void* wdpT_packed, const void* wcat_packed, void* dnsum16,
                  int blocked, void* dy16, const int* m_top, long m_unit, long M, void* stream);
/* ... with the rows of every note step sorted by descending length (row_len int32 [m_unit], needs m_top): 128-row blocks whose first row
 * has no target at their note step are passed over -- forward: their logits are written as zeros (finite operands of a weight-gradient
 * product), hd0 / hd16 stay unwritten; backward: their dy16 rows are zeros, their dnsum rows unwritten */
int ptv_heads_fwd_rows(const void* hn16, const void* wp_packed, const void* wdh_packed, const void* wdp_packed, const float* b_p,
                       const float* b_dh, float* pitch, long ldp, float* hd0, void* hd16, long M, const int* m_top, long m_unit,
                       const int* row_len, void* stream);
int ptv_heads_bwd_rows(float* dp, long ldp, const float* dhd0, const void* wdpT_packed, const void* wcat_packed, void* dnsum16,
                       int blocked, void* dy16, const int* m_top, long m_unit, const int* row_len, long M, void* stream);

/* ------------------------------------------------------------------------------------------------
 * COMPOSITE: PtvaeDecoder.decoder, teacher-forced, FORWARD -- ptvae.py:430-496 with decode_notes (:370-428) and decode_note (:336-368)
 * restructured to 32 + 15 + 5 sequential steps -- behind one call: the 15 launches, the persistent launch's turn (event wait /
 * record) and every shape decision are C++ host code (csrc/composite.hip); the caller owns the tensors and hands them over as one
 * pointer table t[PTV_DTF_COUNT] and one dimension table d[PTV_DTF_D_COUNT].  bf16 precision at the init_model() sizes only
 * (E = 128, Hn = 512, Hd = 64, 130 pitch classes, Ht with a persistent plan): ptv_decoder_tf_supported(d) == 0 / PTV_ERR_UNSUPPORTED
 * otherwise, before anything is launched -- the caller then sequences the entry points above itself.
 *   rows: R = 32*B (time step, sample), M = 15*R (note, time step, sample).  Layouts as documented at the entry points it calls.
 */
enum PtvDtfTensor {
  /* inputs */
  PTV_DTF_Z = 0,          /* [B, Zs] fp32 */
  PTV_DTF_EMB,            /* [16, R, E] fp32: the embedded ground-truth notes, step-major (ptv_embed_fwd) */
  PTV_DTF_XS,             /* [R, 2He] fp32: ground-truth note summaries (ptvae.py:446-453) */
  PTV_DTF_FORCE_DUR,      /* [5, M] int32 or NULL: replayed duration decisions (tests) */
  /* fp32 parameters */
  PTV_DTF_B_ZHID, PTV_DTF_B_ZIN, PTV_DTF_INIT_INPUT, PTV_DTF_B_IH_T, PTV_DTF_B_HH_T, PTV_DTF_B_T2N, PTV_DTF_B_IH_N, PTV_DTF_B_HH_N,
  PTV_DTF_B_P, PTV_DTF_B_DH, PTV_DTF_W_HH_D, PTV_DTF_B_HH_D, PTV_DTF_W_IH_D, PTV_DTF_B_IH_D, PTV_DTF_SOS,
  PTV_DTF_ONEHOT,         /* [2, 5] fp32 constant: rows onehot(0), onehot(1) */
  PTV_DTF_W_OUT_D, PTV_DTF_B_OUT_D,
  /* bf16 operand copies of the weights, [out, in] row-major */
  PTV_DTF_W16_ZHID, PTV_DTF_W16_ZIN, PTV_DTF_W16_IH_T, PTV_DTF_W16_HH_T, PTV_DTF_W16_T2N, PTV_DTF_W16_IH_N,
  /* MFMA-fragment packs (ptv_pack_mfma_multi): notes GRU W_hh / token part of W_ih; pitch_out, dur_hid[:, :Hn], dur_hid[:, Hn:] */
  PTV_DTF_PK_NOTES_H, PTV_DTF_PK_NOTES_T, PTV_DTF_PK_WP, PTV_DTF_PK_WDH, PTV_DTF_PK_WDP,
  /* outputs and what the backward keeps */
  PTV_DTF_NS,             /* [33, B, Ht] fp32 time states (slot 0 = z2dec_hid(z)) */
  PTV_DTF_NS16,           /* the same, bf16 */
  PTV_DTF_Z_IN,           /* [B, Zi] fp32 */
  PTV_DTF_TOKS,           /* [33, B, 2He] fp32 time-step tokens */
  PTV_DTF_GI_T,           /* [R, 3Ht] bf16 */
  PTV_DTF_ZG,             /* [B, 3Ht] bf16 */
  PTV_DTF_GATES_T,        /* [32, 4, B, Ht] bf16 */
  PTV_DTF_HN,             /* [16, R, Hn] fp32 note states */
  PTV_DTF_HN16,           /* the same, bf16 */
  PTV_DTF_GC,             /* [R, 3Hn] bf16, column-blocked by 32 */
  PTV_DTF_GATES_N,        /* [15, 4, R, Hn] bf16 */
  PTV_DTF_PITCH,          /* [M, ldp] fp32 pitch logits (130 used) */
  PTV_DTF_HD,             /* [6, M, Hd] fp32 (slot 0 written) */
  PTV_DTF_HD16,           /* [6, M, Hd] bf16 */
  PTV_DTF_TAB0, PTV_DTF_TAB,    /* [1, 3Hd], [2, 3Hd] fp32 gate tables of the duration GRU */
  PTV_DTF_GATES_D,        /* [5, 4, M, Hd] bf16, or NULL (the backward then recomputes them: ptv_dur_gru_bwd) */
  PTV_DTF_DUR,            /* [M, 10] fp32 duration logits */
  PTV_DTF_IDX,            /* [5, M] int32 duration decisions */
  /* workspaces and the persistent launch's turn */
  PTV_DTF_XCH,            /* 33*B*Ht bf16 exchange buffer of ptv_gru_persist_fwd */
  PTV_DTF_SYNC,           /* its zeroed sync words */
  PTV_DTF_WAIT_EVENT,     /* hipEvent_t recorded after the previous persistent launch of the process, or NULL */
  PTV_DTF_RECORD_EVENT,   /* hipEvent_t to record after this one, or NULL */
  PTV_DTF_LIVE_TOP,       /* device int32 or NULL: the caller uses the outputs of the note steps 0 .. *LIVE_TOP only (a loss that ignores the
                           * padded note slots, ptvae.py:498-511): the notes GRU, the heads and the duration GRU leave the later steps' rows of
                           * HN16 / GATES_N / PITCH / HD / HD16 / GATES_D / DUR / IDX unwritten */
  /* rows sorted by length (round 6: per-row dead work; all four or none).  With them the notes GRU, the heads and the duration GRU work on
   * the decoder's rows in the order PERM -- row p of their tensors is row PERM[p] of the time states / tokens -- and pass over the
   * (note step, 64-row panel) pairs whose longest row has no target there; the loss must be given its targets in the same order. */
  PTV_DTF_PERM,           /* int32 [R]: rows by descending ROW_LEN (ptv_rows_by_length), or NULL */
  PTV_DTF_ROW_LEN,        /* int32 [R]: live note steps of the row at position p (ptv_pianotree_targets_rows, gathered by PERM) */
  PTV_DTF_NS16S,          /* out [R, Ht] bf16: the time states NS16[1:] gathered by PERM (operand of the hoisted products, kept for the backward) */
  PTV_DTF_TOK_S,          /* out [15, R, E] fp32: the fed tokens EMB[:15] gathered by PERM */
  PTV_DTF_SEG_N,          /* int32 [15] (ptv_rows_seg_counts) or NULL, sorted mode only: TOK_S is gathered for the live blocks of every note step only -- the
                             backward must then be given PTV_DTB_SEG_N (its products never read the other rows) */
  PTV_DTF_COUNT
};
enum PtvDtfDim { PTV_DTF_D_B = 0, PTV_DTF_D_E, PTV_DTF_D_HE, PTV_DTF_D_HT, PTV_DTF_D_HN, PTV_DTF_D_HD, PTV_DTF_D_NP, PTV_DTF_D_ZS, PTV_DTF_D_ZI,
                 PTV_DTF_D_LDP, PTV_DTF_D_COUNT };
int ptv_decoder_tf_supported(const long* d);
int ptv_decoder_tf_fwd(const void* const* t, const long* d, void* stream);

/* COMPOSITE: RnnDecoder.forward, teacher-forced (ptvae.py:51-87 with tfr = 1), FORWARD in one call: z -> h0 / z_in, the tokens
 * [init ; c_0 .. c_{T-2}], the hoisted input product, the T-step GRU, the three heads.  Any sizes; d[PTV_CDF_D_PREC] = 0 (fp32) / 1 (bf16
 * MFMA operands, fp32 masters converted per tile); all tensors fp32 except GATES when d[PTV_CDF_D_GATES_BF16]. */
enum PtvCdfTensor {
  PTV_CDF_Z = 0,          /* [B, Z] */
  PTV_CDF_C_SM,           /* [T, B, I] the chord steps, step-major */
  PTV_CDF_W_ZHID, PTV_CDF_B_ZHID, PTV_CDF_W_ZIN, PTV_CDF_B_ZIN, PTV_CDF_INIT_INPUT,
  PTV_CDF_W_IH,           /* [3H, I + Zi] */
  PTV_CDF_B_IH, PTV_CDF_W_HH, PTV_CDF_B_HH, PTV_CDF_W_ROOT, PTV_CDF_B_ROOT, PTV_CDF_W_CHROMA, PTV_CDF_B_CHROMA, PTV_CDF_W_BASS, PTV_CDF_B_BASS,
  PTV_CDF_HALL,           /* out [T+1, B, H] states (slot 0 = z2dec_hid(z)) */
  PTV_CDF_Z_IN,           /* out [B, Zi] */
  PTV_CDF_TOKS,           /* out [T, B, I] */
  PTV_CDF_GI,             /* out [T*B, 3H] */
  PTV_CDF_ZG,             /* out [B, 3H] */
  PTV_CDF_GATES,          /* out [T, 4, B, H] fp32 or bf16 */
  PTV_CDF_ROOT, PTV_CDF_CHROMA, PTV_CDF_BASS,   /* out [T*B, 12 / 24 / 12] logits */
  PTV_CDF_COUNT
};
enum PtvCdfDim { PTV_CDF_D_B = 0, PTV_CDF_D_T, PTV_CDF_D_H, PTV_CDF_D_I, PTV_CDF_D_Z, PTV_CDF_D_ZI, PTV_CDF_D_PREC, PTV_CDF_D_GATES_BF16,
                 PTV_CDF_D_NROOT, PTV_CDF_D_NCHROMA, PTV_CDF_D_NBASS, PTV_CDF_D_COUNT };
int ptv_chord_decoder_fwd(const void* const* t, const long* d, void* stream);

/* ptv_chord_decoder_bwd: autograd through RnnDecoder.forward, teacher-forced (ptvae.py:51-87) -- what ChordDecoderTFFn.backward sequences:
 * the three heads' input / weight / bias gradients, the GRU's BPTT (one persistent launch with its event turn when d[PTV_CDB_D_PERSIST],
 * S = d[PTV_CDB_D_SPLITK] teams; else the per-step kernels), the weight / bias gradients of the GRU, of the start token and of the two
 * z projections, and dz.  26 launches.  Every G_* slot is the parameter's gradient buffer and is ACCUMULATED into (the caller zeroes it
 * once per step); a NULL D* slot = that head received no gradient.  fp32 tensors except GATES / DGI / DGH (bf16 when d[PTV_CDB_D_ACT_BF16])
 * and WT16 / XCH.  Same arithmetic, launch by launch, as the Python sequencing it replaces (bit-identical results). */
enum PtvCdbTensor {
  PTV_CDB_Z = 0,          /* [B, Z] */
  PTV_CDB_W_ZHID, PTV_CDB_W_ZIN, PTV_CDB_W_IH, PTV_CDB_W_HH, PTV_CDB_W_ROOT, PTV_CDB_W_CHROMA, PTV_CDB_W_BASS,   /* fp32 masters */
  PTV_CDB_WT16_HH,        /* bf16 W_hh^T [H, 3H] (persistent BPTT and the bf16 per-step kernels), or NULL: fp32 W_hh */
  PTV_CDB_HALL, PTV_CDB_GATES, PTV_CDB_TOKS, PTV_CDB_Z_IN,                                 /* saved by the forward */
  PTV_CDB_DROOT, PTV_CDB_DCHROMA, PTV_CDB_DBASS,                                           /* in: [T*B, n] or NULL */
  PTV_CDB_DZ,             /* out [B, Z] */
  PTV_CDB_G_INIT_INPUT, PTV_CDB_G_W_ZHID, PTV_CDB_G_B_ZHID, PTV_CDB_G_W_ZIN, PTV_CDB_G_B_ZIN, PTV_CDB_G_W_IH, PTV_CDB_G_B_IH, PTV_CDB_G_W_HH,
  PTV_CDB_G_B_HH, PTV_CDB_G_W_ROOT, PTV_CDB_G_B_ROOT, PTV_CDB_G_W_CHROMA, PTV_CDB_G_B_CHROMA, PTV_CDB_G_W_BASS, PTV_CDB_G_B_BASS,
  PTV_CDB_DHS,            /* scratch [T*B, H] fp32 */
  PTV_CDB_DGI, PTV_CDB_DGH,      /* scratch [T, B, 3H] */
  PTV_CDB_DHZ,            /* scratch [2, B, H] fp32 (per-step kernels) or NULL */
  PTV_CDB_DH0,            /* scratch [B, H] */
  PTV_CDB_DZG,            /* scratch [B, 3H] fp32 */
  PTV_CDB_DZ_IN,          /* scratch [B, Zi] */
  PTV_CDB_DTOK0,          /* scratch [B, I] */
  PTV_CDB_XCH, PTV_CDB_PART, PTV_CDB_SYNC,   /* persistent launch: exchange buffer T*B*3H bf16, split-K partials or NULL, zeroed sync words */
  PTV_CDB_WAIT_EVENT, PTV_CDB_RECORD_EVENT,  /* hipEvent_t of the previous persistent launch of the process / to record after this one, or NULL */
  PTV_CDB_COUNT
};
enum PtvCdbDim { PTV_CDB_D_B = 0, PTV_CDB_D_T, PTV_CDB_D_H, PTV_CDB_D_I, PTV_CDB_D_Z, PTV_CDB_D_ZI, PTV_CDB_D_PREC, PTV_CDB_D_ACT_BF16,
                 PTV_CDB_D_NROOT, PTV_CDB_D_NCHROMA, PTV_CDB_D_NBASS, PTV_CDB_D_PERSIST, PTV_CDB_D_SPLITK, PTV_CDB_D_COUNT };
int ptv_chord_decoder_bwd(const void* const* tensors, const long* dims, void* stream);

/* ptv_decoder_tf_bwd: autograd through PtvaeDecoder.decoder, teacher-forced (ptvae.py:430-496) -- what functional.decoder_bwd_core sequences
 * on its fused bf16 path at the init_model() sizes: the chain  duration-GRU BPTT -> heads -> notes-GRU BPTT -> note-token / time-state
 * gradients -> time-GRU BPTT (persistent, split-K teams, event turn) -> dTOKS, dz  on `stream`, and the four groups of weight / bias
 * gradient products on the SIDE stream, each forked (event record / wait) where the Python sequencing forks it.  The caller joins the side
 * stream (or defers the join) and keeps every tensor of the table alive until then.  ~50 launches, the same arithmetic launch by launch
 * (bit-identical results).  Every G_* slot is the parameter's gradient buffer and is ACCUMULATED into.  TOP_H: device int, the last note
 * step whose gradient is non-zero (zero-skip limit of the head / duration products), or NULL = dense. */
enum PtvDtbTensor {
  /* inputs */
  PTV_DTB_Z = 0,          /* [B, Zs] fp32 */
  PTV_DTB_TOK_OP,         /* [15 R, E] fp32: the note tokens fed to the notes GRU */
  PTV_DTB_DP,             /* [M, ldp] fp32: gradient of the pitch logits; ACCUMULATED into (dP += dHD0 . W_dh[:, Hn:]) */
  PTV_DTB_DDUR,           /* [M, 10] fp32 */
  PTV_DTB_TOP_H,
  /* parameters (fp32 masters) and their packed / transposed bf16 forms */
  PTV_DTB_W_HH_D, PTV_DTB_B_HH_D, PTV_DTB_W_IH_D, PTV_DTB_W_OUT_D, PTV_DTB_SOS,
  PTV_DTB_PK_WDPT, PTV_DTB_PK_WCAT,                      /* heads_packs: 'wdpT', 'wcat' */
  PTV_DTB_PK_NOTES_WT,                                   /* notes_packs: 'wt' */
  PTV_DTB_WT_IH_N, PTV_DTB_WT_T2N, PTV_DTB_WT_IH_T, PTV_DTB_WT_HH_T, PTV_DTB_WT_ZHID, PTV_DTB_WT_ZIN,   /* transposed bf16 shadows [in, out] */
  /* saved by the forward */
  PTV_DTB_NS, PTV_DTB_NS16, PTV_DTB_Z_IN, PTV_DTB_TOKS, PTV_DTB_GATES_T, PTV_DTB_HN16, PTV_DTB_GATES_N, PTV_DTB_PITCH, PTV_DTB_HD16,
  PTV_DTB_TAB0, PTV_DTB_TAB, PTV_DTB_IDX,
  /* outputs */
  PTV_DTB_DZ,             /* [B, Zs] */
  PTV_DTB_DTOK,           /* [16, R, E] (slot 15 zeroed here) */
  PTV_DTB_DTOKS,          /* [33, B, 2He] (slot 32 zeroed here) */
  /* gradient buffers, DEC_PARAM_NAMES order */
  PTV_DTB_G_W_ZHID, PTV_DTB_G_B_ZHID, PTV_DTB_G_W_ZIN, PTV_DTB_G_B_ZIN, PTV_DTB_G_INIT_INPUT,
  PTV_DTB_G_W_IH_T, PTV_DTB_G_W_HH_T, PTV_DTB_G_B_IH_T, PTV_DTB_G_B_HH_T,
  PTV_DTB_G_W_T2N, PTV_DTB_G_B_T2N,
  PTV_DTB_G_W_IH_N, PTV_DTB_G_W_HH_N, PTV_DTB_G_B_IH_N, PTV_DTB_G_B_HH_N,
  PTV_DTB_G_W_P, PTV_DTB_G_B_P, PTV_DTB_G_W_DH, PTV_DTB_G_B_DH, PTV_DTB_G_W_OUT_D, PTV_DTB_G_B_OUT_D,
  PTV_DTB_G_W_IH_D, PTV_DTB_G_W_HH_D, PTV_DTB_G_B_IH_D, PTV_DTB_G_B_HH_D, PTV_DTB_G_SOS,
  /* scratch */
  PTV_DTB_DHD0,           /* [M, Hd] fp32 */
  PTV_DTB_PART,           /* [NBLK, ptv_dur_gru_bwd_part_size()] fp32 */
  PTV_DTB_S,              /* [part_size] fp32, ZEROED by the caller */
  PTV_DTB_TMP64,          /* [64] fp32, ZEROED */
  PTV_DTB_DNSUM,          /* [M, Hn] bf16 */
  PTV_DTB_DY16,           /* [M, 200] bf16 */
  PTV_DTB_TMP200,         /* [200, Hn] fp32 */
  PTV_DTB_CS200,          /* [200] fp32, ZEROED */
  PTV_DTB_DGI_N, PTV_DTB_DGH_N, PTV_DTB_DHN0, PTV_DTB_SCRATCH_N,
  PTV_DTB_TOP_STEP,       /* device int initialised to -1 */
  PTV_DTB_DGC, PTV_DTB_DNS,
  PTV_DTB_DGI_T, PTV_DTB_DGH_T, PTV_DTB_DZHID, PTV_DTB_DZG, PTV_DTB_DZ_IN,
  PTV_DTB_XCH, PTV_DTB_PART_T, PTV_DTB_SYNC,             /* the time GRU's persistent BPTT */
  PTV_DTB_WAIT_EVENT, PTV_DTB_RECORD_EVENT,              /* hipEvent_t: persistent launches take turns */
  PTV_DTB_SIDE_STREAM,    /* hipStream_t of the sibling stream */
  PTV_DTB_FORK_EVENT0, PTV_DTB_FORK_EVENT1, PTV_DTB_FORK_EVENT2, PTV_DTB_FORK_EVENT3,   /* hipEvent_t, one per fork */
  /* the forward ran on rows sorted by length (PTV_DTF_PERM ...): all or none.  NS16 / TOK_OP are then the gathered copies the forward left
   * (NS16: slot 0 unused, rows from NS16S - B*Ht ... the table holds NS16S itself in PTV_DTB_NS16S), DNS / DTOK come out in natural row order */
  PTV_DTB_PERM, PTV_DTB_ROW_LEN,
  PTV_DTB_NS16S,          /* [R, Ht] bf16 gathered time states (operand of the weight_ih / time_to_notes gradients) */
  PTV_DTB_DNS_S,          /* scratch [R, Ht] fp32: the gradient of the time states in sorted row order, scattered into DNS */
  PTV_DTB_DTOK_S,         /* scratch [15, R, E] fp32: the token gradient in sorted row order, scattered into DTOK */
  PTV_DTB_SEG_N,          /* int32 [15] (ptv_rows_seg_counts of ROW_LEN) or NULL: the weight-gradient products over (note step, row) skip the dead
                             blocks of every step (sorted mode only) */
  PTV_DTB_COUNT
};
enum PtvDtbDim { PTV_DTB_D_B = 0, PTV_DTB_D_E, PTV_DTB_D_HE, PTV_DTB_D_HT, PTV_DTB_D_HN, PTV_DTB_D_HD, PTV_DTB_D_NP, PTV_DTB_D_ZS, PTV_DTB_D_ZI,
                 PTV_DTB_D_LDP, PTV_DTB_D_NBLK, PTV_DTB_D_SPLITK, PTV_DTB_D_COUNT };
int ptv_decoder_tf_bwd(const void* const* tensors, const long* dims, void* stream);

/* ptv_bigru_final_bwd: autograd through a bidirectional GRU whose final states are its output (RnnEncoder / TextureEncoder, ptvae.py:23-31,
 * 116-122), bf16 precision, both directions' BPTT in ONE persistent launch (event turn in here), then per direction the weight / bias
 * gradients dW_ih += dgi^T . x, dW_hh += dgh^T . h (bias sums fused) -- the reversed direction's on the side stream, forked and joined in
 * here -- and, with DX, the input gradient dx (+)= dgi_0 . W_ih_0 + dgi_1 . W_ih_1.  What functional._bigru_backward sequences on its
 * persistent branch, bit-identical to it.  G_* slots are accumulated into. */
enum PtvBgbTensor {
  PTV_BGB_X = 0,          /* [T*M, I] the forward's input rows (fp32, or bf16 with d[PTV_BGB_D_X_BF16]) */
  PTV_BGB_DOUT,           /* [M, 2H] fp32: gradient of the two final states */
  PTV_BGB_HALL0, PTV_BGB_H16_0, PTV_BGB_GATES0, PTV_BGB_WT_HH0, PTV_BGB_WT_IH0,   /* forward direction: fp32 / bf16 states [T+1, M, H], gate planes,
                                                                                   * bf16 W_hh^T [H, 3H], bf16 W_ih^T [I, 3H] (NULL without DX) */
  PTV_BGB_HALL1, PTV_BGB_H16_1, PTV_BGB_GATES1, PTV_BGB_WT_HH1, PTV_BGB_WT_IH1,   /* reversed direction */
  PTV_BGB_G_W_IH0, PTV_BGB_G_W_HH0, PTV_BGB_G_B_IH0, PTV_BGB_G_B_HH0, PTV_BGB_G_W_IH1, PTV_BGB_G_W_HH1, PTV_BGB_G_B_IH1, PTV_BGB_G_B_HH1,
  PTV_BGB_DX,             /* out [T*M, I] fp32, or NULL: no input gradient wanted */
  PTV_BGB_DGI0, PTV_BGB_DGH0, PTV_BGB_DGI1, PTV_BGB_DGH1,                        /* scratch [T, M, 3H] bf16 */
  PTV_BGB_XCH0, PTV_BGB_XCH1, PTV_BGB_PART0, PTV_BGB_PART1, PTV_BGB_SYNC,        /* persistent launch (PART*: split-K partials or NULL) */
  PTV_BGB_WAIT_EVENT, PTV_BGB_RECORD_EVENT, PTV_BGB_SIDE_STREAM, PTV_BGB_FORK_EVENT, PTV_BGB_JOIN_EVENT,
  PTV_BGB_COUNT
};
enum PtvBgbDim { PTV_BGB_D_M = 0, PTV_BGB_D_T, PTV_BGB_D_H, PTV_BGB_D_I, PTV_BGB_D_X_BF16, PTV_BGB_D_DX_ACC, PTV_BGB_D_SPLITK, PTV_BGB_D_COUNT };
int ptv_bigru_final_bwd(const void* const* tensors, const long* dims, void* stream);

/* ptv_bigru_final_fwd: the forward of the above (functional._bigru_forward's persistent branch): per direction the input product
 * gi = x . W_ih^T + b_ih (bf16 out), the zero initial state, both recurrences in ONE persistent launch (event turn in here), the two final
 * states copied into OUT [M, 2H].  Saved for the backward: HALL / H16 / GATES.  bf16 precision; lengths (packed sequences) or NULL. */
enum PtvBgfTensor {
  PTV_BGF_X = 0,          /* [T*M, I] fp32 (or bf16 with d[PTV_BGF_D_X_BF16]) */
  PTV_BGF_LENGTHS,        /* [M] int32 or NULL */
  PTV_BGF_W16_IH0, PTV_BGF_B_IH0, PTV_BGF_W16_HH0, PTV_BGF_B_HH0, PTV_BGF_W16_IH1, PTV_BGF_B_IH1, PTV_BGF_W16_HH1, PTV_BGF_B_HH1,   /* bf16 weight shadows, fp32 biases */
  PTV_BGF_OUT,            /* out [M, 2H] fp32 */
  PTV_BGF_GI0, PTV_BGF_HALL0, PTV_BGF_H16_0, PTV_BGF_GATES0, PTV_BGF_GI1, PTV_BGF_HALL1, PTV_BGF_H16_1, PTV_BGF_GATES1,
  PTV_BGF_XCH0, PTV_BGF_XCH1, PTV_BGF_SYNC, PTV_BGF_WAIT_EVENT, PTV_BGF_RECORD_EVENT,
  PTV_BGF_COUNT
};
enum PtvBgfDim { PTV_BGF_D_M = 0, PTV_BGF_D_T, PTV_BGF_D_H, PTV_BGF_D_I, PTV_BGF_D_X_BF16,
                 PTV_BGF_D_WIH_F32,   /* W16_IH* are the fp32 masters (an input width that is no multiple of 8 has no bf16 shadow) */
                 PTV_BGF_D_COUNT };
int ptv_bigru_final_fwd(const void* const* tensors, const long* dims, void* stream);

/* ptv_bigru_rows_fwd / ptv_bigru_rows_bwd: the same pair for a bi-GRU over MANY short independent rows (dec_notes_emb_gru, the ground-truth
 * note summaries, ptvae.py:446-453: 32 B rows x 16 notes, H = 128) on the row-partitioned kernels -- one launch per direction for the whole
 * sequence, the input product fused, the reversed direction on the side stream (forked and joined in here).  What functional._bigru_forward
 * / _bigru_backward sequence on their row-kernel branch, bit-identical to it.  LENGTHS / PERM as ptv_row_gru_persist_fwd_perm; TOP0 / TOP1:
 * device ints initialised to -1 (the BPTT reports the last live time index; the products stop there), NULL = no limit. */
enum PtvBrfTensor {
  PTV_BRF_X = 0,          /* [T, M, I] fp32 */
  PTV_BRF_LENGTHS, PTV_BRF_PERM,           /* int32 [M] or NULL */
  PTV_BRF_PK_WG_H0, PTV_BRF_PK_WG_T0, PTV_BRF_B_HH0, PTV_BRF_B_IH0, PTV_BRF_PK_WG_H1, PTV_BRF_PK_WG_T1, PTV_BRF_B_HH1, PTV_BRF_B_IH1,
  PTV_BRF_OUT,            /* out [M, 2H] */
  PTV_BRF_HALL0, PTV_BRF_H16_0, PTV_BRF_GATES0, PTV_BRF_HALL1, PTV_BRF_H16_1, PTV_BRF_GATES1,
  PTV_BRF_SIDE_STREAM, PTV_BRF_FORK_EVENT, PTV_BRF_JOIN_EVENT,
  PTV_BRF_COUNT
};
enum PtvBrfDim { PTV_BRF_D_M = 0, PTV_BRF_D_T, PTV_BRF_D_H, PTV_BRF_D_I, PTV_BRF_D_COUNT };
int ptv_bigru_rows_fwd(const void* const* tensors, const long* dims, void* stream);
enum PtvBrbTensor {
  PTV_BRB_X = 0, PTV_BRB_DOUT, PTV_BRB_LENGTHS, PTV_BRB_PERM,
  PTV_BRB_PK_WT0, PTV_BRB_HALL0, PTV_BRB_H16_0, PTV_BRB_GATES0, PTV_BRB_WT_IH0,       /* WT_IH*: bf16 W_ih^T [I, 3H], NULL without DX */
  PTV_BRB_PK_WT1, PTV_BRB_HALL1, PTV_BRB_H16_1, PTV_BRB_GATES1, PTV_BRB_WT_IH1,
  PTV_BRB_G_W_IH0, PTV_BRB_G_W_HH0, PTV_BRB_G_B_IH0, PTV_BRB_G_B_HH0, PTV_BRB_G_W_IH1, PTV_BRB_G_W_HH1, PTV_BRB_G_B_IH1, PTV_BRB_G_B_HH1,
  PTV_BRB_DX,             /* out [T*M, I] fp32 or NULL */
  PTV_BRB_DGI0, PTV_BRB_DGH0, PTV_BRB_SCRATCH0, PTV_BRB_TOP0, PTV_BRB_DGI1, PTV_BRB_DGH1, PTV_BRB_SCRATCH1, PTV_BRB_TOP1,
  PTV_BRB_SIDE_STREAM, PTV_BRB_FORK_EVENT, PTV_BRB_JOIN_EVENT,
  PTV_BRB_SEG,            /* int32 [T] or NULL, with PERM = ptv_rows_by_length(LENGTHS) only: ptv_rows_seg_counts of the lengths in that order -- the
                             weight_hh products (operands indexed by position) skip the dead 128-row blocks of every step */
  PTV_BRB_COUNT
};
enum PtvBrbDim { PTV_BRB_D_M = 0, PTV_BRB_D_T, PTV_BRB_D_H, PTV_BRB_D_I, PTV_BRB_D_DX_ACC, PTV_BRB_D_DOUT_LD, PTV_BRB_D_COUNT };
int ptv_bigru_rows_bwd(const void* const* tensors, const long* dims, void* stream);

/* ptv_vae_loss_fwd / ptv_vae_loss_bwd: DisentangleVAE.loss_function (model.py:57-68: PianoTree reconstruction CE x2 with ignore_index, the
 * two KL terms, the three chord CEs, the 11 output scalars) and its backward, one call each -- what functional.VaeLossFn sequences (12 / 9
 * launches).  scal = {beta, w0, w1, B * Z, 8 B, 96 B} as doubles.  PITCH_T / DUR_T / COUNTS: the targets of ptv_pianotree_targets; with
 * d[PTV_VL_D_HAVE_TARGETS] they were computed by the caller already (DisentangleVAE.loss() does, before the decoder). */
enum PtvVlTensor {
  PTV_VL_X = 0, PTV_VL_C,                                    /* int64 [B,32,16,6], fp32 [B,8,36] */
  PTV_VL_PITCH, PTV_VL_DUR,                                  /* logits in memory order: [480 B, ldp], [2400 B, 2] */
  PTV_VL_MU_C, PTV_VL_SD_C, PTV_VL_MU_R, PTV_VL_SD_R,        /* [B, Z] */
  PTV_VL_ROOT, PTV_VL_CHROMA, PTV_VL_BASS,                   /* [8 B, 12], [96 B, 2], [8 B, 12] */
  PTV_VL_PITCH_T, PTV_VL_DUR_T, PTV_VL_COUNTS, PTV_VL_ROOT_T, PTV_VL_CHROMA_T, PTV_VL_BASS_T,   /* int32 targets / counts[3] */
  PTV_VL_SUMS,            /* fwd: [8] fp32 ZEROED */
  PTV_VL_OUT,             /* fwd out: [11] */
  PTV_VL_GOUT,            /* bwd in: [11] */
  PTV_VL_GS,              /* bwd scratch: [8] */
  PTV_VL_DPITCH, PTV_VL_DDUR, PTV_VL_DMU_C, PTV_VL_DSD_C, PTV_VL_DMU_R, PTV_VL_DSD_R, PTV_VL_DROOT, PTV_VL_DCHROMA, PTV_VL_DBASS,   /* bwd out */
  PTV_VL_COUNT
};
enum PtvVlDim { PTV_VL_D_B = 0, PTV_VL_D_Z, PTV_VL_D_NP, PTV_VL_D_LDP, PTV_VL_D_SM_P, PTV_VL_D_SM_C, PTV_VL_D_HAVE_TARGETS, PTV_VL_D_COUNT };
int ptv_vae_loss_fwd(const void* const* tensors, const long* dims, const double* scal, void* stream);
int ptv_vae_loss_bwd(const void* const* tensors, const long* dims, const double* scal, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Free-running tokens.
 * ptv_note_token (ptvae.py:408-416 + pitch_dur_ind_to_note_token :328-334): per row pitch argmax (first
 *   maximal index), predicted token pred[r] = note_embedding(onehot(pitch) | 5 duration bits) with
 *   dur_idx[d*dur_stride + r] the duration argmaxes, predicted grid row xhat[r] = (pitch, bits) int64,
 *   running predicted length plen[r] (first <eos> step n; the last step fills 15).
 * ptv_chord_token (ptvae.py:72-78): next chord-decoder token [B,36]; root/bass parts are the UNION over
 *   the batch of the rows' argmax one-hots (the reference's index-broadcast quirk), chroma = per-row bits.
 * ptv_route_slices: dst(mask[s]) (+)= src for nslices consecutive slices of slice_elems floats -- routes
 *   token gradients to the ground-truth embedding (mask 1) or the predicted-token buffer (mask 0).
 */
int ptv_note_token(const float* pitch, long ld_pitch, const int* dur_idx, long dur_stride, const float* W, const float* bias, int E,
                   float* pred, long ld_pred, long* xhat, long xhat_stride, int* plen, int n, int last,
                   const int* force_pitch, int M, void* stream);
int ptv_chord_token(const float* root, const float* chroma, const float* bass, unsigned* masks2, float* token, int B, void* stream);
int ptv_route_slices(const float* src, float* dstA, float* dstB, const int* mask, long slice_elems, int nslices, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Data contract on device: what ArrangementDataset.__getitem__ (dataset.py:88-112) does per item
 * with converter.py:65-68 (augment_pr), :78-113 (pr_to_onehot_pr + piano_roll_to_target), :116-147 (target_to_3dtarget
 * with the arguments of dataset.py:98-104) and :150-164 (expand_chord), for a whole batch in one launch.
 *   pr      [N,32,128] uint8 accompaniment piano-rolls (2 onset, 1 sustain, 0 silence; converter.py:35-47)
 *   chord14 [N,8,14] float  raw chords [root, 12 chroma bits, bass]
 *   index[B] (NULL = identity) picks the item of sample b, shift[B] (NULL = 0) its transposition in semitones
 *   -> pr_mat [B,32,128] f32, x [B,32,16,6] int64, c [B,8,36] f32;  *err |= 1 if a step held more than 14 onsets
 *      (the reference raises IndexError there; the kernel keeps the lowest 14)
 */
int ptv_batch_transform(const unsigned char* pr, const float* chord14, const int* index, const int* shift,
                        float* pr_mat, long* x, float* c, int* err, int B, void* stream);
/* interp_path (model.py:216-242): out[b, i, :] = slerp between z1[b] and z2[b] at i/(n-1), norms interpolated geometrically */
int ptv_slerp_path(const float* z1, const float* z2, float* out, int B, int D, int n, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Free-running / scheduled-sampling decoder as row-partitioned persistent kernels (csrc/freerun.hip): PtvaeDecoder.decode_notes
 * + decode_note (ptvae.py:336-428) for ONE time step t and ALL 15 note steps in one launch, a workgroup per panel of 16 samples
 * (state in LDS / registers, weights streamed from L2 in MFMA-fragment-major packing), and the re-summarisation of the
 * predicted notes (bi-GRU with packed-sequence masking, ptvae.py:476-486) that yields the next time-step token.
 * bf16 MFMA operands, fp32 state / logits; init_model() geometry only (E = He = 128, Hn = 512, Hd = 64, 130 pitches).
 *   ptv_pack_mfma_b: W fp32 [N][ld] (K columns from the given base) -> bf16 [ceil(N/16)][ceil(K/32)][64 lanes][8]: the B fragment
 *     of 16 rows x 32 k as one contiguous 1-KB wave load (zero padded); ptv_pack_mfma_b_size = elements of `out`.
 *     pairs = 1 (N % 32 == 0) interleaves the rows of every pair of tiles so that lane (row, quad) of the MFMA C layout owns the 8
 *     consecutive output columns 32t + 8*quad .. +7 across the pair's two accumulators (no cross-lane movement in the epilogue).
 *   ptv_free_note_loop: w / io are HOST arrays of 16 / 17 device pointers:
 *     w  = { pack(dec_notes_gru.weight_hh), pack(dec_notes_gru.weight_ih[:, Ht:]), pack(pitch_out_linear.weight),
 *            pack(dur_hid_linear.weight[:, :512]), pack(dur_hid_linear.weight[:, 512:]), pack(dec_dur_gru.weight_hh),
 *            dec_notes_gru.bias_hh, pitch_out_linear.bias, dur_hid_linear.bias, dec_dur_gru.bias_hh, tab0 [192] = W_ih_d sos + b,
 *            tab [2][192] = W_ih_d onehot(0/1) + b, dur_out_linear.weight, .bias, note_embedding.weight^T [135][128], .bias }
 *     io = { gc [B][1536] (W_ih_n[:, :Ht] ns + b_ih for this t), emb [16][R][128] ground-truth embedding or NULL,
 *            HN [16][R][512] (slot 0 rows of t = initial state, written by the caller), gates_n [15][4][R][512] bf16,
 *            pitch [M][ld_pitch], HD [6][M][64], gates_d [5][4][M][64] bf16, dur [M][10], idx [5][M] int32,
 *            TOK [15][R][128] (slot 0 rows of t = first token, written by the caller), PRED [16][R][128],
 *            xhat [B][32][16][6] int64, plen [R] int32 (zeroed), force_pitch [15][R] or NULL, force_dur [5][M] or NULL,
 *            HN16 [16][R][512] bf16 or NULL, HD16 [6][M][64] bf16 or NULL (bf16 state copies for the backward; HD16 replaces HD[1..5]) }
 *     io[17] = NULL (timing experiments), io[18] = NULL or fp32 [B][2048] = [initial state | gc] of this time step (then io[0] is
 *     ignored and slot 0 of HN is written by the kernel); io[19] = xch, 8-byte words [ceil(B/16)][2][16][256], io[20] = cnt, uint32
 *     [ceil(B/16) + 1] ZEROED by the caller before the launch of t = 0 (cluster mode below; else NULL): io has 21 entries;
 *     with R = 32*B, M = 15*R; the rows of time step t are [t*B, (t+1)*B).  coin_mask bit n = feed the ground-truth note n+1
 *     (teacher-forcing coin, ptvae.py:420).  train = 0 skips what only the backward reads (HN, gates, HD, TOK); train = 2 stores only
 *     the fed tokens TOK: the caller then recomputes states and gates for ALL rows with the batched kernels (ptv_notes_gru_persist_fwd,
 *     ptv_dur_gru_fwd with the stored decisions forced), cheaper than 16-row panels streaming them out note step by note step.
 *     Bits 16 / 17 of train force the 4-wave kernel / the 8-wave kernel whose producer waves stream the next note step's state
 *     products under the head phases of the current one (default: by panel count).
 *     Bits 18-20 of train = S in {2, 4}, or bit 22 = eight members: cluster mode of the 4-wave kernel -- S workgroups (co-resident: ceil(B/16) * S <= the CU count, else
 *     PTV_ERR_UNSUPPORTED) share a panel: each streams 1/S of the notes-GRU gate weights (the product bound by one CU's L2 port), the
 *     new bf16 state is all-gathered through xch once per note step as 8-byte words {2 units, step tag} that the readers poll
 *     (agent-scope stores / loads; xch = ceil(B/16) x 2 x 16 x 256 words = 64 KB per panel, ZEROED by the caller before t = 0; cnt counts
 *     arrivals for diagnostics; the launches of t = 0..31 must follow each other in order on one stream), heads / duration GRU /
 *     embedding are computed redundantly by all members and written by member 0.  cnt[ceil(B/16)] != 0 afterwards: a member gave up
 *     waiting (results void).
 *     Bit 21 of train: the 4-wave kernel streams the head weights from L2 every note step instead of keeping them in registers / LDS
 *     for the whole launch (the kernel before round 6; timing comparisons, bit-identical results).
 *   ptv_free_resummarize: w = { pack(W_ih), pack(W_hh), pack(W_ih_reverse), pack(W_hh_reverse), b_ih, b_hh, b_ih_r, b_hh_r } of
 *     dec_notes_emb_gru; io = { PRED, plen, XH fwd [17][R][128] (slot 0 zero), XH bwd, XG fwd [16][4][R][128] bf16, XG bwd,
 *     tok_next = TOKS[t+1] [B][256] }.  train bit 0: save states and gates (XH, XG) for the backward; bit 1: stream the weights from L2
 *     every step instead of keeping a wave's 48 fragments in registers for the launch (timing comparisons, bit-identical).
 */
long ptv_pack_mfma_b_size(int N, int K);
int ptv_pack_mfma_b(const float* W, long ld, int N, int K, void* out, int pairs, void* stream);
/* the same for a TRANSPOSED source (trans != 0: element (n, k) at W[k*ld + n]) and / or into the k-block sub-range [kb0, kb0 + ceil(K/32)) of
 * a packed buffer holding NT tiles x KBtot k-blocks (two sources side by side along K; rows n >= N and columns k >= K are zero) */
int ptv_pack_mfma_b2(const float* W, long ld, int N, int K, void* out, int pairs, int trans, int NT, int kb0, int KBtot, void* stream);
/* up to 8 such packs in ONE launch: jobs = n rows of 10 longs {W, ld, N, K, out, pairs, trans, NT, kb0, KBtot} (host array) */
int ptv_pack_mfma_multi(const long* jobs, int n, void* stream);
int ptv_free_note_loop(const void* const* w, const void* const* io, long ld_pitch, int B, int t, unsigned coin_mask, int train,
                       void* stream);
int ptv_free_resummarize(const void* const* w, const void* const* io, int B, int t, int train, void* stream);

/* ------------------------------------------------------------------------------------------------
 * ptv_decoder_free_fwd (COMPOSITE, round 6; SURVEY.md 8b's decoder_free_fwd): the forward of the free-running / scheduled-sampling
 * PianoTree decoder as ONE call -- PtvaeDecoder.decoder with teacher-forcing coins (ptvae.py:430-496), decode_notes / decode_note
 * (:336-428) and the re-summarisation of the predicted notes (:476-486): what train.py's schedule runs from its third batch on
 * (scheduler.py:48-54, train.py:23-24,59-63) and what inference_decode always runs (model.py:124-131).  The launch sequence of
 * functional_free.DecoderStepFn.forward's persistent path, bit-identical to it:
 *   prologue   z -> initial time state, z_in, its gate contribution; first time token; first note token
 *   32 x       time-GRU input product, time-GRU cell (ptv_gru_step_fwd), [initial notes state | hoisted input part] in one product,
 *              ptv_free_note_loop (all 15 note steps of the time step), then the next time token: the ground-truth summary (time coin) or
 *              ptv_free_resummarize over the predicted notes
 *   recompute  (training, D_REPLAY) what the batched backward reads, for all 480*B rows at once by the teacher-forced kernels on the
 *              recorded tokens with the stored duration decisions forced: hoisted input part, notes GRU (ptv_notes_gru_persist_fwd),
 *              dur_hid, the duration GRU (ptv_dur_gru_fwd), and the re-summarisation bi-GRU (ptv_row_gru_persist_fwd x 2)
 * bf16 precision at the init_model() sizes (E = 128, He = 128, Hn = 512, Hd = 64, 130 pitches): anything else returns
 * PTV_ERR_UNSUPPORTED before the first launch.  t: pointer table (enum PtvDffTensor; the caller owns every tensor), d: dimension table
 * (enum PtvDffDim).  wl / io: the HOST pointer arrays of ptv_free_note_loop (io[18] = the H0GC scratch), wr / ior: those of
 * ptv_free_resummarize (ior[6] is overwritten per time step with TOKS[t + 1]).  note_mask: HOST [32] -- bit n of word t = note coin n of
 * time step t (ptvae.py:420); time_coin: HOST [31] bytes -- time step t + 1 is fed the ground-truth summary (ptvae.py:476).
 * The autograd backward of this node is ptv_decoder_tf_bwd on the recorded tokens plus the routing of the token gradients
 * (ptv_route_slices) and ptv_bigru_rows_bwd over the predicted notes: functional_free.DecoderStepFn.backward.
 */
enum PtvDffTensor {
  PTV_DFF_Z = 0,            /* [B, Zs] fp32 */
  PTV_DFF_XS,               /* [32 B, 2He] fp32 ground-truth note summaries, or NULL (inference / no time coin set) */
  PTV_DFF_TOK0_SRC,         /* first note token of every time step: [R, E] rows (the embedded ground-truth slot 0) or ONE row (<sos>, D_TOK0_LDS = 0) */
  /* parameters (fp32) and their operand copies */
  PTV_DFF_W_ZHID, PTV_DFF_B_ZHID, PTV_DFF_W_ZIN, PTV_DFF_B_ZIN, PTV_DFF_W_IH_T, PTV_DFF_B_IH_T, PTV_DFF_INIT_INPUT, PTV_DFF_B_HH_T,
  PTV_DFF_W_IH_T_OP,        /* dec_time_gru.weight_ih_l0 as the per-step product's operand: bf16 shadow (D_W_IH_T_BF16) or the fp32 weight */
  PTV_DFF_W_HH_T_OP,        /* dec_time_gru.weight_hh_l0 likewise (D_W_HH_T_BF16) */
  PTV_DFF_W_CAT, PTV_DFF_B_CAT,   /* [dec_time_to_notes_hid ; dec_notes_gru.weight_ih[:, :Ht]] bf16 [Hn + 3Hn, Ht] and its bias fp32 */
  /* recompute block (D_REPLAY): fp32 weights + packs */
  PTV_DFF_W_IH_N, PTV_DFF_B_IH_N, PTV_DFF_B_HH_N, PTV_DFF_W_DH, PTV_DFF_B_DH, PTV_DFF_W_HH_D, PTV_DFF_B_HH_D, PTV_DFF_TAB0, PTV_DFF_TAB,
  PTV_DFF_W_OUT_D, PTV_DFF_B_OUT_D, PTV_DFF_PK_NOTES_H, PTV_DFF_PK_NOTES_T,
  PTV_DFF_PK_E_H0, PTV_DFF_PK_E_T0, PTV_DFF_B_HH_E0, PTV_DFF_B_IH_E0, PTV_DFF_PK_E_H1, PTV_DFF_PK_E_T1, PTV_DFF_B_HH_E1, PTV_DFF_B_IH_E1,
  /* state / outputs */
  PTV_DFF_NS, PTV_DFF_NS16, PTV_DFF_Z_IN, PTV_DFF_ZG, PTV_DFF_TOKS, PTV_DFF_GATES_T,
  PTV_DFF_GI, PTV_DFF_H0GC,          /* scratch [B, 3Ht] / [B, Hn + 3Hn] fp32, reused by every time step */
  PTV_DFF_TOK, PTV_DFF_PRED, PTV_DFF_PITCH, PTV_DFF_HN, PTV_DFF_HN16, PTV_DFF_GATES_N, PTV_DFF_HD, PTV_DFF_HD16, PTV_DFF_GATES_D,
  PTV_DFF_IDX, PTV_DFF_PLEN, PTV_DFF_GC16, PTV_DFF_DUR_SCR, PTV_DFF_IDX_SCR,
  PTV_DFF_XH0, PTV_DFF_XH1, PTV_DFF_XH16_0, PTV_DFF_XH16_1, PTV_DFF_XG0, PTV_DFF_XG1,
  PTV_DFF_WAIT_EVENT, PTV_DFF_RECORD_EVENT,   /* hipEvent_t or NULL: the persistent-launch turn around the cluster-mode note loops */
  PTV_DFF_COUNT
};
enum PtvDffDim {
  PTV_DFF_D_B = 0, PTV_DFF_D_ZS, PTV_DFF_D_ZI, PTV_DFF_D_HE, PTV_DFF_D_HT, PTV_DFF_D_HN, PTV_DFF_D_HD, PTV_DFF_D_E, PTV_DFF_D_NP,
  PTV_DFF_D_LDP,            /* row stride of the pitch logits */
  PTV_DFF_D_TRAIN, PTV_DFF_D_REPLAY, PTV_DFF_D_INFERENCE,
  PTV_DFF_D_LOOP_FLAGS,     /* the `train` word of ptv_free_note_loop (mode, kernel choice, cluster size); in cluster mode the call zeroes io[19] / io[20] itself */
  PTV_DFF_D_CLUSTER,        /* != 0: the note loops take the persistent-launch turn (WAIT / RECORD events) */
  PTV_DFF_D_RESUM_TRAIN,    /* the `train` word of ptv_free_resummarize */
  PTV_DFF_D_TOK0_LDS,       /* row stride of TOK0_SRC (0 = one row for all) */
  PTV_DFF_D_W_IH_T_BF16, PTV_DFF_D_W_HH_T_BF16,
  PTV_DFF_D_COUNT
};
int ptv_decoder_free_fwd(const void* const* t, const long* d, const void* const* wl, const void* const* io, const void* const* wr,
                         const void* const* ior, const unsigned* note_mask, const unsigned char* time_coin, void* stream);

/* ptv_decoder_free_bwd (COMPOSITE, round 6; SURVEY.md 8b's decoder_free_bwd): autograd through the node above (argmax is not
 * differentiable: with the fed tokens recorded every chain is the batched BPTT of the teacher-forced path) as ONE call --
 *   ptv_decoder_tf_bwd(t_tf, d_tf)            the decoder's whole backward on the recorded tokens (tables of that entry point)
 *   ptv_route_slices x 2                      token gradients to the ground-truth embedding (coin set / slot 0) or to the predicted
 *                                             tokens; time-token gradients to the ground-truth summaries or to the re-summarised ones
 *   ptv_bigru_rows_bwd(t_rows, d_rows)        (t_rows != NULL) BPTT of the re-summarisation bi-GRU over the predicted notes
 *                                             (ptvae.py:480-486), its input gradient added to the predicted tokens' gradient
 *   slot 0 of the predicted tokens is the ground-truth <sos> embedding: its gradient moves over; then the predicted tokens ->
 *   note_embedding: multi-hot operand of the predicted grid (ptv_multihot) and grad_W / grad_b in one ptv_wgrad pass
 * -- functional_free.DecoderStepFn.backward's launch sequence, bit-identical to it.  t: enum PtvDfbTensor, d: enum PtvDfbDim. */
enum PtvDfbTensor {
  PTV_DFB_DTOK = 0,         /* [15, R, E] fp32 (slot 15 of the [16, R, E] buffer unused): gradient w.r.t. the fed note tokens (out of t_tf) */
  PTV_DFB_DTOKS,            /* [33, B, 2He] fp32: gradient w.r.t. the time-step tokens (out of t_tf) */
  PTV_DFB_DEMB, PTV_DFB_DPRED,   /* [16, R, E] fp32, zero on entry: gradient of the ground-truth embedding / of the predicted tokens */
  PTV_DFB_DXS, PTV_DFB_DXSP,     /* [32, B, 2He] fp32, zero on entry: gradient of the ground-truth / re-summarised time tokens */
  PTV_DFB_MASK_TOK, PTV_DFB_MASK_TIME,   /* int32 [15 * 32] / [32] routing masks of the coins */
  PTV_DFB_DX_PRED,          /* [16 R, E] fp32: the input gradient t_rows produces (its PTV_BRB_DX), or NULL without t_rows */
  PTV_DFB_XHAT,             /* [B, 32, 16, 6] int64 predicted grid */
  PTV_DFB_MH,               /* [16 R, 136] fp32 scratch: its multi-hot rows */
  PTV_DFB_G_W_EMB, PTV_DFB_G_B_EMB,     /* note_embedding gradients [E, 135] / [E], accumulated into */
  PTV_DFB_COUNT
};
enum PtvDfbDim { PTV_DFB_D_B = 0, PTV_DFB_D_E, PTV_DFB_D_HE, PTV_DFB_D_COUNT };
int ptv_decoder_free_bwd(const void* const* t_tf, const long* d_tf, const void* const* t_rows, const long* d_rows, const void* const* t,
                         const long* d, void* stream);

/* ------------------------------------------------------------------------------------------------
 * The teacher-forced notes GRU (dec_notes_gru over 15 note steps x 32*B rows, ptvae.py:395-398 restructured per SURVEY.md 7.1)
 * as row-partitioned persistent kernels: ONE launch for the whole sequence, a workgroup owns 64 rows, the state stays on the CU, W_hh
 * streams from L2 in ptv_pack_mfma_b packing, the token product is fused.  bf16 precision, Hn = 512, E = 128.
 *   fwd (csrc/notes_roles.hip, round 5): 8 waves per workgroup with ROLES -- four product waves stream the weights L2 -> registers and
 *        issue the MFMAs, four cell waves move every HBM operand / result and run the gate arithmetic with the fp32 state of their
 *        cells in registers for all T steps; accumulators pass through 16-KB LDS slots.
 *        wg_h = pack(W_hh [1536,512]), wg_t = pack(W_ih[:, Ht:] [1536,128]), both with pairs = 0; gc bf16 = W_ih[:, :Ht] ns + b_ih,
 *        the [R][1536] matrix stored COLUMN-BLOCKED by 16 ([96][R][16]: ptv_gemm dtypes bit 4 writes it that way); emb fp32 [T][R][128]
 *        fed tokens; h0 fp32 [R][512] the initial state (read only: the fp32 states of the later steps never leave the CU); HN16 bf16
 *        [T+1][R][512] every state, slot 0 included (the operand of the heads, the weight-gradient products and the BPTT); gates bf16
 *        [T][4] planes (r, z, n, W_hn h + b_hn) or NULL -- PRIVATE to this forward / BPTT pair: each plane is unit-blocked by 16,
 *        plane[u / 16][row][u % 16] (whole-kilobyte wave accesses), not the [R][512] of ptv_gru_seq_fwd.
 *        T: bits 0-7 = steps; bits 8-15 = debug flags and bits 16-23 = weight-ring depth of the timing scripts (0 = defaults).
 *   bwd: wt = pack(W_hh^T [512,1536]) (pairs = 1); HN16 / gates as the forward left them (the previous state enters the gate
 *        gradients at bf16 precision); ext bf16 = gradient arriving at the state after step s, the [T*R][512] matrix of the heads'
 *        input-gradient products stored COLUMN-BLOCKED by 32 ([16][T*R][32], ptv_gemm dtypes bit 3);
 *        dgi bf16 [T][R][1536]; dgh bf16 [T][R][512] = the n third (dn * r) only -- the r and z thirds of dgh are dgi's, so
 *        grad W_hh[0:1024] = dgi[:, 0:1024]^T . h and grad W_hh[1024:] = dgh^T . h; dh0 fp32 [R][512] or NULL; scratch:
 *        ptv_notes_gru_persist_scratch_elems(R) bf16 elements.  The BPTT passes over the late steps of a 64-row panel at which no gradient
 *        arrives (zero rows of ext: the padded note slots the loss ignores) and writes zero rows for them; top_step (or NULL; device int,
 *        initialised to -1 by the caller) receives the last step at which anything arrived for any panel, so that the products over dgi /
 *        dgh can stop after that step's rows (ptv_wgrad k_top).
 */
int ptv_notes_gru_persist_fwd(const void* wg_h, const void* wg_t, const float* b_hh, const void* gc, const float* emb,
                              const float* h0, void* HN16, void* gates, long R, int T, void* stream);
/* the same with a live-step limit: live_top (device int32, or NULL) -- only the note steps 0 .. *live_top run; the later slots of HN16 and
 * planes of gates stay unwritten (the BPTT must then be given a top_step limit <= *live_top: ptv_notes_gru_persist_bwd) */
int ptv_notes_gru_persist_fwd_top(const void* wg_h, const void* wg_t, const float* b_hh, const void* gc, const float* emb,
                                  const float* h0, void* HN16, void* gates, long R, int T, const int* live_top, void* stream);
long ptv_notes_gru_persist_scratch_elems(long R);
int ptv_notes_gru_persist_bwd(const void* wt, const void* HN16, const void* gates, const void* ext, void* dgi, void* dgh,
                              float* dh0, void* scratch, long R, int T, int* top_step, void* stream);
/* the same with a caller-given bound (device int, or NULL = the call above): no gradient arrives after note step *bound -- the forward
 * stopped there (ptv_notes_gru_persist_fwd_top) or the loss says so -- and every consumer of dgi / dgh stops at top_step (<= *bound,
 * required non-NULL then).  The note steps beyond the bound are not touched: neither is `ext` read there (it may be unwritten) nor are
 * zero rows of dgi / dgh written (without the bound: 64 KB read and 256 KB of zeros written per 64-row panel and dead step). */
int ptv_notes_gru_persist_bwd_top(const void* wt, const void* HN16, const void* gates, const void* ext, void* dgi, void* dgh,
                                  float* dh0, void* scratch, long R, int T, const int* bound, int* top_step, void* stream);
/* both with the rows sorted by descending length (row_len int32 [R] = live note steps of the row at each position, or NULL): a 64-row panel
 * runs the steps its first (= longest) row has.  Forward: the HN16 slots of a panel's dead steps up to *live_top are zero-filled (not with
 * bit 24 of T: the caller's products clip to the same segments), their gate planes unwritten.  Backward (needs bound): `ext` is not read at a panel's dead steps; dgi / dgh get zero rows there up to *bound -- unless
 * bit 16 of T is set (round 6): the caller's consumers clip to the same 128-row segments (ptv_wgrad_job.seg_n, ptv_sum_steps_seg,
 * ptv_gemm_mtop_seg) and the rows may stay unwritten. */
int ptv_notes_gru_persist_fwd_rows(const void* wg_h, const void* wg_t, const float* b_hh, const void* gc, const float* emb,
                                   const float* h0, void* HN16, void* gates, long R, int T, const int* live_top, const int* row_len,
                                   void* stream);
int ptv_notes_gru_persist_bwd_rows(const void* wt, const void* HN16, const void* gates, const void* ext, void* dgi, void* dgh,
                                   float* dh0, void* scratch, long R, int T, const int* bound, const int* row_len, int* top_step,
                                   void* stream);
/* which kernel ptv_notes_gru_persist_bwd runs: 1 (default) = 8 waves per workgroup, the A operand LDS-resident, the carry dh (x) z in registers
 * (csrc/notes_roles.hip), 0 = the 4-wave kernel of rounds 2-4 (csrc/notes_persist.hip); same arguments, same results to rounding.
 * Process-wide. */
int ptv_notes_bwd_variant(int eight_waves);

/* The same kernels for any GRU whose rows are many and independent; H = 512 (above) or H = 128 with 128 inputs, which is one
 * direction of dec_notes_emb_gru, the note-summary bi-GRU over the 16 notes of each of the 32*B steps (ptvae.py:446-453,480-486).
 *   fwd: w_hh = pack(W_hh [3H,H]), w_x = pack(W_ih [3H,128]) (pairs = 1); b_ih NULL when folded into gc; gc bf16 [R][3H] or NULL;
 *        x fp32, row m of step t at x + t*x_step + m*128; lengths int32 [R] or NULL (row m is updated at time t iff t < lengths[m],
 *        as nn.utils.rnn.pack_padded_sequence does); reverse = 1 walks time T-1..0; out (or NULL) receives the final state,
 *        row m at out + m*out_ld (out_ld % 4 == 0).
 *        The two instances are specialised: H = 512 needs gc and takes no b_ih / lengths / reverse / out; H = 128 needs b_ih and no
 *        gc (PTV_ERR_UNSUPPORTED otherwise).
 *   bwd: ext bf16 [T][R][H] (H = 512 only, required there); dh_last fp32 (row stride last_ld) = gradient of the final state (H = 128
 *        only), or NULL; lengths: the lengths the forward ran with, or NULL -- with lengths both kernels pass over the steps that lie
 *        beyond the longest row of a 64-row panel (identity for the whole panel, as pack_padded_sequence leaves them out;
 *        the forward then leaves that step's gates unwritten, so the backward MUST be given the same lengths); H = 512 writes only the n third of dgh ([T][R][512], see above); dgi is indexed by
 *        TIME, dgh by processing step (so dgh pairs with HN16[:T] and dgi with x in the weight-gradient products).
 */
int ptv_row_gru_persist_fwd(int H, const void* w_hh, const void* w_x, const float* b_hh, const float* b_ih, const void* gc,
                            const float* x, long x_step, const int* lengths, float* HN, void* HN16, void* gates,
                            float* out, long out_ld, long R, int T, int reverse, void* stream);
long ptv_row_gru_persist_scratch_elems(int H, long R);
int ptv_row_gru_persist_bwd(int H, const void* wt, const void* HN, const void* gates, const void* ext,
                            const float* dh_last, long last_ld, const int* lengths, void* dgi, void* dgh, float* dh0,
                            void* scratch, long R, int T, int reverse, int* top_step, void* stream);
/* The same with a ROW PERMUTATION (H = 128): panel position p works on row perm[p] of x / lengths / out (forward) and of dh_last / lengths
 * / dgi (backward); the kernels' private tensors HN / HN16 / gates and dgh are indexed by position (dgh pairs with HN16[:T], dgi with x in
 * the weight-gradient products, as before).  With perm = ptv_rows_by_length(lengths) a 64-row panel holds rows of (almost) one length, so
 * the passing-over of a panel's masked steps -- what pack_padded_sequence does for the reference, ptvae.py:446-453 -- removes the masked
 * work itself instead of only the steps beyond the LONGEST of 64 unrelated rows.  perm NULL = the entry points above.  dh0 unsupported. */
int ptv_row_gru_persist_fwd_perm(int H, const void* w_hh, const void* w_x, const float* b_hh, const float* b_ih, const void* gc,
                                 const float* x, long x_step, const int* lengths, const int* perm, float* HN, void* HN16,
                                 void* gates, float* out, long out_ld, long R, int T, int reverse, void* stream);
int ptv_row_gru_persist_bwd_perm(int H, const void* wt, const void* HN, const void* gates, const void* ext,
                                 const float* dh_last, long last_ld, const int* lengths, const int* perm, void* dgi, void* dgh,
                                 float* dh0, void* scratch, long R, int T, int reverse, int* top_step, void* stream);
/* perm [R] = the rows in order of DESCENDING lengths[row] (0 <= length <= max_len <= 38), ties in row order: a stable counting sort in one
 * workgroup (deterministic: the order of the K rows of the weight-gradient products depends on it) */
int ptv_rows_by_length(const int* lengths, int* perm, long R, int max_len, void* stream);

/* ------------------------------------------------------------------------------------------------
 * clip_grad_norm_ (module.py:142-143) + torch.optim.Adam.step (train.py:50, scheduler.py:69-74) over
 * flat fp32 buffers: sumsq = |g|^2 (device scalar), then p,m,v updated with g*gscale clipped to `clip`.
 */
/* dst[i] = bf16(src[i]): refreshes the bf16 shadow of the flat parameter buffer once per step */
int ptv_cast_bf16(const float* src, void* dst, long n, void* stream);
/* dst[c*rows + r] = bf16(src[r*cols + c]): transposed bf16 copy of a weight matrix (operand of the dX products) */
int ptv_transpose_cast_bf16(const float* src, void* dst, int rows, int cols, void* stream);
/* the same for every matrix of the flat parameter buffer in one launch: desc[i] = {offset, rows, cols, first 32x32 tile}
 * (device array of nmat x 4 longs, tiles numbered consecutively; ntiles = their total); flat_t[offset + c*rows + r] */
int ptv_transpose_cast_bf16_batched(const float* flat, void* flat_t, const long* desc, int nmat, long ntiles, void* stream);
int ptv_grad_sumsq(const float* g, long n, float* sumsq, void* stream);
int ptv_clip_adam_step(float* p, const float* g, float* m, float* v, long n, const float* sumsq, float gscale, float clip,
                       float lr, float beta1, float beta2, float eps, int step, void* stream);
/* the same step that also writes the bf16 operand copy p16[i] = bf16(p[i]) of the updated parameters (p16 may be NULL) */
int ptv_clip_adam_step_shadow(float* p, const float* g, float* m, float* v, long n, const float* sumsq, float gscale, float clip,
                              float lr, float beta1, float beta2, float eps, int step, void* p16, void* stream);
/* SURVEY 8b's gradnorm_clip_adam_step: torch.nn.utils.clip_grad_norm_(params, clip) + Adam.step() (module.py:142-144, train.py:50) over
 * the flat buffers in ONE call -- sumsq[0] = sum g^2 (left on the device: the pre-clip norm is sqrt(sumsq) * gscale), then the clipped
 * update of ptv_clip_adam_step_shadow.  gscale: 1/world under data parallelism (the bucket holds a SUM over ranks). */
int ptv_gradnorm_clip_adam_step(float* p, const float* g, float* m, float* v, long n, float* sumsq, float gscale, float clip,
                                float lr, float beta1, float beta2, float eps, int step, void* p16, void* stream);

/* Reproducible reductions (default ON; environment PTV_WGRAD_ORDERED=0 or ptv_ordered_reductions(0) turn them off).  The
 * reference's CPU path is run-to-run deterministic.  With the switch on, nothing on the train step ends in an fp32
 * atomicAdd whose order depends on arrival: the K slabs of ptv_wgrad and the K splits of ptv_gemm store partial tiles into a
 * per-stream workspace and one more launch adds them in slab order; the grid reductions (ptv_grad_sumsq, ptv_kl_fwd,
 * ptv_reparam_kl_fwd, ptv_ce_fwd, ptv_colsum, ptv_dur_out_wgrad, ptv_txt_conv_relu_pool_bwd) park one partial per block and the last
 * block to arrive adds them in block order.  Two runs of the same step then produce the same bits.  (Still atomic: the grouped
 * cross-entropy of the weighted duration loss, ptv_ce_group_fwd.)  ptv_wgrad_mode switches the two product paths only. */
int ptv_ordered_reductions(int on);
int ptv_wgrad_mode(int ordered);
/* bf16 x bf16 weight-gradient products through the LDS-DMA kernel (global_load_lds staging, three stage buffers; default 0 = register-staged:
 * faster standalone, slower beside the persistent recurrences -- csrc/wgrad.hip); process-wide, PTV_WGRAD_DMA sets the initial value */
int ptv_wgrad_dma(int enable);
/* reductions that ran on fp32 atomics although ordered mode is on (no workspace: first use inside a capture, > 64 streams, a need
 * beyond the scratch); 0 after any step is what makes the step bit-reproducible.  reset != 0 clears the counter. */
long ptv_ordered_fallbacks(int reset);

/* Per-step scalars on the device (graph-replayed train steps, graph_step.py): while dev4 is set, ptv_loss_finalize /
 * ptv_loss_bwd_scales read beta = dev4[0] (the KL weight of train.py:56-58's schedule; only where the call's own beta is non-zero) and
 * ptv_clip_adam_step* read lr = dev4[1], 1 - beta1^t = dev4[2], sqrt(1 - beta2^t) = dev4[3] (Adam's bias corrections,
 * scheduler.py:69-74 + train.py:50) instead of their by-value arguments.  NULL restores by-value behaviour. */
int ptv_step_params(const float* dev4);

/* ------------------------------------------------------------------------------------------------
 * Weight-gradient product (csrc/wgrad.hip): C[M,N] (fp32, row stride ldc) (+)= alpha * sum_k A[k*lda + m] * B[k*ldb + n], i.e.
 * grad_W = grad_out^T . input as autograd forms it for every nn.Linear / nn.GRU weight (ptvae.py:16-17,23,64,116,360,396,450,461),
 * with both operands stored row-per-sample.  bf16 MFMA, fp32 accumulate; dtypes bit 0 / 1 = A / B already bf16 in HBM (fp32
 * otherwise, rounded to bf16 on the way in).  K is cut into slabs that reduce into C with fp32 atomics; slabs = 0 picks the count.
 * colsum_a (or NULL): fp32 [M] += sum_k A[k*lda + m], the bias gradient that belongs to the same layer (grad_b = column sums of
 * grad_out), taken from the A tiles while they are in LDS instead of a second pass over A (always accumulates).
 * k_top (or NULL): device int; the rows of A from (*k_top + 1) * k_unit on are zero (the kernel that wrote A says so: the notes BPTT
 * reports the last note step at which any gradient arrived) -- the product stops there.  k_unit must be a multiple of 32.
 * k_rev > 0: A holds its k_rev units in reversed order (the *_reverse direction of a GRU, gradients indexed by processing step):
 * the zero part is then the rows BEFORE (k_rev - *k_top - 1) * k_unit and the product starts there.
 * ptv_gemm(prec = bf16, transA = transB = 1) routes here.
 */
int ptv_wgrad(int M, int N, int K, const void* A, long lda, const void* B, long ldb, float* C, long ldc, float alpha,
              int accumulate, int dtypes, int slabs, float* colsum_a, const int* k_top, long k_unit, int k_rev, void* stream);
/* the same for a gradient matrix that lives in TWO arrays: C[M1 + M2, N] (+)= alpha * [A1 | A2]^T . B, rows 0 .. M1-1 of C from the columns
 * of A1, the rest from A2 (same dtype, M1 a multiple of 128) -- one pass over B instead of two (the notes GRU's weight_hh gradient:
 * [dgi[:, :1024] | dgh]^T . h, ptv_notes_gru_persist_bwd) */
int ptv_wgrad_cat(int M1, const void* A1, long lda1, int M2, const void* A2, long lda2, int N, int K, const void* B, long ldb,
                  float* C, long ldc, float alpha, int accumulate, int dtypes, int slabs, float* colsum_a, const int* k_top,
                  long k_unit, int k_rev, void* stream);
/* Several weight-gradient products in ONE launch (plus one launch for their ordered reductions): the parameter gradients that become ready
 * at the same point of a backward pass -- a GRU's W_ih / W_hh with their bias sums, a decoder's z projections (ptvae.py:16-17,23,64,116,
 * 360,396,450,461 as above).  Each job has exactly the meaning of one ptv_wgrad call with the same fields, and gives the same bits as that
 * call would (same slab plan, same reduction order); the jobs' outputs (C, colsum_a) must not overlap.  HOST array of jobs. */
typedef struct ptv_wgrad_job {
  int M, N, K;
  const void* A; long lda;
  const void* B; long ldb;
  float* C; long ldc;
  float alpha;
  int accumulate, dtypes, slabs;
  float* colsum_a;
  const int* k_top; long k_unit; int k_rev;
  /* round 6, K SEGMENTS (or seg_n = NULL): K runs over units of seg_unit rows (a note step's R decoder rows in length order); of unit q only the
   * first seg_n[q % seg_period] rows (seg_period < 0: seg_n[|seg_period| - 1 - q % |seg_period|], units in reversed order; device ints,
   * multiples of 32) hold anything -- the rest of A is zero and the rest of B may never have
   * been written.  The product skips them (slabs never straddle a unit: ptv_wgrad_seg_supported(K, seg_unit)), and so does
   * the ordered reduction; against the same product without segments the result is bit-identical when the skipped rows of A are zero. */
  const int* seg_n; long seg_unit; int seg_period;
} ptv_wgrad_job;
int ptv_wgrad_batch(const ptv_wgrad_job* jobs, int njobs, void* stream);
/* 1 if a product of depth K can take segments of seg_unit rows: seg_unit a multiple of 128 that divides K, and K cut into at most 248 slabs of
 * the largest power of two that divides seg_unit (B = 512: 16384-row units; 3 * 2^k units work too; 128 * 129 rows do not) */
int ptv_wgrad_seg_supported(long K, long seg_unit);
/* seg_n[s] = 128 * #{128-row blocks whose FIRST row has row_len > s}, s < steps: the live prefix of every note step when the decoder's rows run
 * in descending length order (ptv_rows_by_length) and a block is dead beyond its first row's length -- what the *_rows kernels of the decoder
 * skip and ptv_wgrad_batch's segments clip.  R a multiple of 128. */
int ptv_rows_seg_counts(const int* row_len, long R, int steps, int* seg_n, void* stream);


#ifdef __cplusplus
}
#endif
#endif /* PTVAE_HIP_H */

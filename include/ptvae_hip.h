/* ptvae_hip.h -- C ABI of libptvae_hip.so: the MI355X (gfx950) kernels behind the polyphonic-VAE
 * training step (reference: ZZWaang/polyphonic-chord-texture-disentanglement, ptvae.py / model.py).
 *
 * The reference has no native layer of its own: its "FFI" for this path is PyTorch's ATen
 * (nn.GRU / nn.Linear / nn.Conv2d / CrossEntropyLoss / Normal / Adam).  Each entry point below
 * names the reference call site(s) whose arithmetic it replaces.  INTEGRATION.md shows the
 * ctypes binding the Python host uses.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (torch tensors on the Python side);
 *     all matrices are fp32 row-major with explicit leading dimensions (in elements)
 *   - `stream` is a hipStream_t passed as void*; kernels are enqueued on it, nothing synchronises
 *   - return value: 0 = ok, <0 = error (PTV_ERR_*); no exceptions cross the ABI
 *   - `prec`: PTV_PREC_F32 (0) = v_mfma_f32_16x16x4_f32, exact fp32 (parity path);
 *             PTV_PREC_BF16 (1) = v_mfma_f32_16x16x32_bf16, bf16 operands / fp32 accumulate
 *     (state, activations, gradients and weights stay fp32 in HBM in both modes)
 */
#ifndef PTVAE_HIP_H
#define PTVAE_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define PTV_PREC_F32 0
#define PTV_PREC_BF16 1

/* Library / build identification ("gfx950"). */
const char* ptv_arch(void);
int ptv_abi_version(void);

/* ------------------------------------------------------------------------------------------------
 * Dense product  C[M,N] = act(alpha * A.B^T + bias) (+ C)      -- every nn.Linear forward
 * (ptvae.py:16-17,37-38,100-101,264-267,286-293), its input gradient (transB) and weight gradient
 * (transA+transB, split-K with atomic accumulation).
 *   transA = 0: A[m*lda + k]   1: A[k*lda + m]
 *   transB = 0: B[n*ldb + k] (nn.Linear weight layout)   1: B[k*ldb + n]
 *   act: 0 none, 1 exp (linear_var(...).exp_(), ptvae.py:27,120)
 *   splitk: 0 auto, >0 forced number of K splits, <0 never split
 */
int ptv_gemm(int prec, int transA, int transB, int M, int N, int K,
             const float* A, long lda, const float* B, long ldb,
             float* C, long ldc, const float* bias, float alpha,
             int accumulate, int act, int splitk, void* stream);

/* ------------------------------------------------------------------------------------------------
 * GRU recurrence over T steps for M independent rows (torch.nn.GRU cell semantics; replaces the
 * per-step `self.gru(...)` / `dec_*_gru(...)` calls at ptvae.py:23,64-65,116,360,396-398,450,461-462,484).
 *   gi  [T] x [M,3H]   input-side pre-activations W_i x + b_i (gate order r,z,n), produced by ptv_gemm
 *   gi2 optional second addend with its own strides (may be NULL)
 *   hall [T+1][M][H]   slot 0 = initial state (written by the caller), slot s+1 = state after step s
 *   gates [T][4][M][H] saved r,z,n,(W_hn h + b_hn) for the backward pass, or NULL (inference)
 *   lengths[M] or NULL: packed-sequence masking (row m updated at time t iff t < lengths[m]),
 *                 the pack_padded_sequence semantics of ptvae.py:446-453,480-486
 *   reverse: processing step s consumes time index t = T-1-s (the *_reverse direction)
 */
int ptv_gru_seq_fwd(int prec, int M, int H, int T,
                    const float* gi, long gi_step_stride, long gi_ld,
                    const float* gi2, long gi2_step_stride, long gi2_ld,
                    const float* w_hh, const float* b_hh,
                    float* hall, float* gates,
                    const int* lengths, int reverse, void* stream);

/* BPTT through ptv_gru_seq_fwd (replaces autograd through the same call sites).
 *   dh_ext [T] x [M,H]  gradient arriving at the state after processing step s (may be NULL)
 *   dh_last [M,H]       gradient arriving at the final state only (may be NULL)
 *   dgi [T][M][3H] (indexed by TIME t), dgh [T][M][3H] (indexed by processing step s)
 *   dhz scratch [2][M][H];  dh0 [M,H] gradient w.r.t. the initial state (may be NULL)
 * Weight gradients follow with ptv_gemm(transA=1,transB=1): dW_hh += dgh^T.hall[0:T], dW_ih += dgi^T.x
 */
int ptv_gru_seq_bwd(int prec, int M, int H, int T,
                    const float* hall, const float* gates, const float* w_hh,
                    const float* dh_ext, long ext_step_stride, long ext_ld,
                    const float* dh_last, long last_ld,
                    float* dgi, float* dgh, float* dhz, float* dh0,
                    int reverse, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* PTVAE_HIP_H */

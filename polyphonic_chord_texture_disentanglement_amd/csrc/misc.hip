// misc.hip -- duration-head token kernel, chord-decoder token kernel, fused clip + Adam.
#include <mutex>
#include <stdlib.h>
#include "common.hpp"
#include "gemm_core.hpp"
#include "../../include/ptvae_hip.h"
#include "../../include/ptvae_hip_debug.h"

namespace ptv {
int g_zero_skip = 1;

// est_dur = dur_out_linear(h) (ptvae.py:361-362), next token index = argmax (ptvae.py:365-367).
// 16 lanes per row (float4 each covers H <= 64 ... loops for larger H), 4 rows per wave.
__global__ void dur_out_token_kernel(const float* __restrict__ h, int H, const float* __restrict__ w_out, const float* __restrict__ b_out,
                                     float* __restrict__ dur_out, long ld_out, int* __restrict__ idx, const int* __restrict__ force_idx, long rows) {
  const int sub = threadIdx.x & 15;
  const long rpb = blockDim.x / 16;
  for (long r = (long)blockIdx.x * rpb + threadIdx.x / 16; r < rows; r += (long)gridDim.x * rpb) {
    float s0 = 0.f, s1 = 0.f;
    for (int j = sub; j < H; j += 16) { float v = h[r * H + j]; s0 += v * w_out[j]; s1 += v * w_out[H + j]; }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) { s0 += __shfl_xor(s0, o, 16); s1 += __shfl_xor(s1, o, 16); }
    if (sub == 0) {
      s0 += b_out[0]; s1 += b_out[1];
      dur_out[r * ld_out + 0] = s0; dur_out[r * ld_out + 1] = s1;
      if (idx) idx[r] = force_idx ? force_idx[r] : (s1 > s0 ? 1 : 0);      // first max wins ties (torch.max)
    }
  }
}

// ---------------------------------------------------------------------------------------------
// gradient global norm + clip_grad_norm_(.,clip) + Adam (module.py:142-144, train.py:50) over the
// flat parameter / gradient buffers.  Two launches, no host sync: the norm stays on the device.
// ---------------------------------------------------------------------------------------------
__global__ void cast_flat_bf16_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(src)[i];
    st4f(dst, i * 4, true, v.x, v.y, v.z, v.w);
  }
}

// 32x32 LDS-tiled transpose + cast (coalesced both ways)
__global__ void transpose_cast_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, int rows, int cols) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < rows && c < cols) ? src[(long)r * cols + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;
    if (c < cols && r < rows) dst[(long)c * rows + r] = (__bf16)tile[tx][i];
  }
}

// every weight matrix of the flat buffer in ONE launch: desc[i] = (offset, rows, cols, first tile); a block finds its
// matrix by a short scan of the tile prefix (the ~40 per-matrix launches of the per-step shadow refresh were a
// 0.3 ms serial prologue of 5-microsecond kernels)
__global__ void transpose_cast_batched_kernel(const float* __restrict__ flat, __bf16* __restrict__ flat_t,
                                              const long* __restrict__ desc, int nmat) {
  __shared__ float tile[32][33];
  int mi = 0;
  while (mi + 1 < nmat && desc[4 * (mi + 1) + 3] <= (long)blockIdx.x) mi++;
  const long off = desc[4 * mi];
  const int rows = (int)desc[4 * mi + 1], cols = (int)desc[4 * mi + 2];
  const int t = (int)(blockIdx.x - desc[4 * mi + 3]);
  const int tc = (cols + 31) / 32;
  const int c0 = (t % tc) * 32, r0 = (t / tc) * 32;
  const float* src = flat + off;
  __bf16* dst = flat_t + off;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    const int r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < rows && c < cols) ? src[(long)r * cols + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, r = r0 + tx;
    if (c < cols && r < rows) dst[(long)c * rows + r] = (__bf16)tile[tx][i];
  }
}

__global__ void sumsq_kernel(const float* __restrict__ g, long n, float* __restrict__ out, OrdScratch sc) {
  __shared__ float red[4];
  float s = 0.f;
  const long n4 = n / 4;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    float4 v = g4[i]; s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  for (long i = n4 * 4 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) s += g[i] * g[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) red[0] = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  ordered_commit(out, red, 1, sc, 0, blockIdx.x, gridDim.x);
}

__global__ void clip_adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n,
                                 const float* __restrict__ sumsq, float gscale, float clip, float lr, float b1, float b2, float eps,
                                 float bc1, float bc2_sqrt, __bf16* __restrict__ p16, const float* __restrict__ sp) {
  // sp (ptv_step_params): the per-step scalars of a graph-replayed step live on the device -- [1] lr, [2] 1 - b1^t, [3] sqrt(1 - b2^t)
  if (sp) { lr = sp[1]; bc1 = sp[2]; bc2_sqrt = sp[3]; }
  // grads are first scaled by gscale (1/world_size after a sum all-reduce); sumsq is of the UNSCALED buffer
  const float norm = sqrtf(sumsq[0]) * gscale;
  float coef = clip > 0.f ? clip / (norm + 1e-6f) : 1.f;
  coef = fminf(coef, 1.f) * gscale;
  const float step = lr / bc1;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float gi = g[i] * coef;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    const float pn = p[i] - step * mi / (sqrtf(vi) / bc2_sqrt + eps);
    p[i] = pn;
    if (p16) p16[i] = (__bf16)pn;               // the bf16 operand copy of the parameters, refreshed in the same pass (else: a 109-MB cast next step)
  }
}


// ---------------------------------------------------------------------------------------------
// free-running decoder tokens (ptvae.py:408-416,328-334): per row, pitch argmax over the 130 logits,
// predicted note token = note_embedding(onehot(pitch) | 5 duration argmax bits), predicted grid row
// (pitch, bits) and the running predicted length (first <eos> position; 15 if none by the last step).
// One wave per row.
// ---------------------------------------------------------------------------------------------
__global__ void note_token_kernel(const float* __restrict__ pitch, long ld_pitch, const int* __restrict__ dur_idx, long dur_stride,
                                  const float* __restrict__ W, const float* __restrict__ bias, int E,
                                  float* __restrict__ pred, long ld_pred, long* __restrict__ xhat, long xhat_stride,
                                  int* __restrict__ plen, int n, int last, const int* __restrict__ force_pitch, int M) {
  const int lane = threadIdx.x & 63;
  for (int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); r < M; r += gridDim.x * (blockDim.x >> 6)) {
    const float* lr = pitch + (long)r * ld_pitch;
    float best = -INFINITY; int bi = 0x7fffffff;
    for (int c = lane; c < 130; c += 64) { float v = lr[c]; if (v > best) { best = v; bi = c; } }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      float ov = __shfl_xor(best, o, 64); int oi = __shfl_xor(bi, o, 64);
      if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }       // first maximal index (torch.max)
    }
    if (force_pitch) bi = force_pitch[r];
    int bits[5];
#pragma unroll
    for (int d = 0; d < 5; d++) bits[d] = dur_idx[(long)d * dur_stride + r];
    for (int e = lane; e < E; e += 64) {
      float v = bias[e] + W[e * 135 + bi];
#pragma unroll
      for (int d = 0; d < 5; d++) v += W[e * 135 + 130 + d] * (float)bits[d];
      pred[(long)r * ld_pred + e] = v;
    }
    if (lane == 0) {
      long* xr = xhat + (long)r * xhat_stride;
      xr[0] = bi;
#pragma unroll
      for (int d = 0; d < 5; d++) xr[1 + d] = bits[d];
      int L = plen[r];
      if (L == 0 && bi == 129) L = n;                 // lengths[eos & (lengths == 0)] = t      (ptvae.py:415-416)
      if (last && L == 0) L = n;                      // lengths[lengths == 0] = t              (ptvae.py:425)
      plen[r] = L;
    }
  }
}

// chord-decoder free-running token (ptvae.py:72-78).  QUIRK reproduced: the reference's index broadcast
// makes the root / bass part of EVERY row the union over the batch of all rows' argmax one-hots.
__global__ void chord_argmax_kernel(const float* __restrict__ root, const float* __restrict__ chroma, const float* __restrict__ bass,
                                    unsigned* __restrict__ masks, float* __restrict__ token, int B) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  int ar = 0, ab = 0;
  for (int k = 1; k < 12; k++) { if (root[b * 12 + k] > root[b * 12 + ar]) ar = k; if (bass[b * 12 + k] > bass[b * 12 + ab]) ab = k; }
  atomicOr(masks + 0, 1u << ar);
  atomicOr(masks + 1, 1u << ab);
  for (int k = 0; k < 12; k++) token[b * 36 + 12 + k] = chroma[b * 24 + 2 * k + 1] > chroma[b * 24 + 2 * k] ? 1.f : 0.f;
}
__global__ void chord_union_kernel(const unsigned* __restrict__ masks, float* __restrict__ token, int B) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * 12) return;
  int b = i / 12, k = i % 12;
  token[b * 36 + k] = (masks[0] >> k) & 1u ? 1.f : 0.f;
  token[b * 36 + 24 + k] = (masks[1] >> k) & 1u ? 1.f : 0.f;
}

// dst_sel[s][row, :] (+)= src[s][row, :] where sel = mask[s] picks dstA (mask != 0) or dstB (may be null)
__global__ void route_slices_kernel(const float* __restrict__ src, float* __restrict__ dstA, float* __restrict__ dstB,
                                    const int* __restrict__ mask, long slice_elems, int nslices, int accumulate) {
  const long total = slice_elems * nslices;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int s = (int)(i / slice_elems);
    float* d = mask[s] ? dstA : dstB;
    if (d) d[i] = accumulate ? d[i] + src[i] : src[i];
  }
}

}  // namespace ptv

using namespace ptv;

extern "C" int ptv_dur_out_token(const float* h, int H, const float* w_out, const float* b_out, float* dur_out, long ld_out,
                                 int* idx, const int* force_idx, long rows, void* stream) {
  if (!h || !w_out || !b_out || !dur_out || rows <= 0 || H <= 0) return PTV_ERR_ARG;
  long nb = (rows + 15) / 16; if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(dur_out_token_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, h, H, w_out, b_out, dur_out, ld_out, idx, force_idx, rows);
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

// weight gradient of dur_out_linear over all 5 duration steps in one pass:  gW[c][u] += sum_d sum_m ddur[m][2d+c] * h_{d+1}[m][u]
// (5 separate [2 x 64] split-K products each re-launch and re-read; this reads the 5 bf16 state planes once).
// block: 256 threads = 8 unit-octets (16-byte loads of 8 bf16) x 32 row lanes; LDS tree over the row lanes, atomics per block.
__global__ void dur_out_wgrad_kernel(const float* __restrict__ ddur, long ld_dd, const __bf16* __restrict__ hall16, long plane_h,
                                     float* __restrict__ gw, long rows, int H, OrdScratch sc) {
  __shared__ float red[4][2][64];                                  // (2.5 KB, not 16.9: the launch fits beside the chain's 148-158-KB workgroups)
  __shared__ float tot[128];
  const int uo = threadIdx.x & 7, rlane = threadIdx.x >> 3;
  float a0[8], a1[8];
#pragma unroll
  for (int e = 0; e < 8; e++) a0[e] = a1[e] = 0.f;
  for (long r = (long)blockIdx.x * 32 + rlane; r < rows; r += (long)gridDim.x * 32) {
#pragma unroll
    for (int d = 0; d < 5; d++) {
      const float2 g = *reinterpret_cast<const float2*>(ddur + r * ld_dd + 2 * d);
      // an ignored target's gradient is exactly zero (more than half of the rows): its state is not read -- and need not have been
      // written (a forward that stopped at the batch's last live note step leaves the later rows of hall16 as they were)
      if (g.x == 0.f && g.y == 0.f) continue;
      const bf16x8 h = *reinterpret_cast<const bf16x8*>(hall16 + (d + 1) * plane_h + r * H + uo * 8);
#pragma unroll
      for (int e = 0; e < 8; e++) { const float hv = (float)h[e]; a0[e] += g.x * hv; a1[e] += g.y * hv; }
    }
  }
  // a wave holds 8 row lanes x 8 unit octets (lane = (rlane & 7) * 8 + uo): the row lanes meet by shuffles, the 4 waves through LDS
#pragma unroll
  for (int e = 0; e < 8; e++) {
    float v0 = a0[e], v1 = a1[e];
#pragma unroll
    for (int d = 8; d < 64; d <<= 1) { v0 += __shfl_xor(v0, d, 64); v1 += __shfl_xor(v1, d, 64); }
    if ((threadIdx.x & 63) < 8) { red[threadIdx.x >> 6][0][uo * 8 + e] = v0; red[threadIdx.x >> 6][1][uo * 8 + e] = v1; }
  }
  __syncthreads();
  if (threadIdx.x < 128) {
    const int c = threadIdx.x >> 6, u = threadIdx.x & 63;
    float s = 0.f;
    for (int q = 0; q < 4; q++) s += red[q][c][u];
    tot[c * 64 + u] = s;                                          // (H = 64: gw[c * H + u])
  }
  __syncthreads();
  ordered_commit(gw, tot, 128, sc, 0, blockIdx.x, gridDim.x);
}

extern "C" int ptv_dur_out_wgrad(const float* ddur, long ld_dd, const void* hall16, long plane_h, float* gw, long rows, int H, void* stream) {
  if (!ddur || !hall16 || !gw || rows <= 0 || H != 64 || (ld_dd & 1) || (plane_h & 7)) return PTV_ERR_ARG;
  long nb = (rows + 31) / 32; if (nb > 2048) nb = 2048;
  OrdScratch sc = ord_scratch((hipStream_t)stream, 256L * 128, 1);
  if (sc.slots && nb > 256) nb = 256;                                   // (the last block adds nb partials per output)
  hipLaunchKernelGGL(dur_out_wgrad_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, ddur, ld_dd, (const __bf16*)hall16, plane_h, gw, rows, H, sc);
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

extern "C" int ptv_grad_sumsq(const float* g, long n, float* sumsq, void* stream) {
  if (!g || !sumsq || n <= 0) return PTV_ERR_ARG;
  if (reinterpret_cast<uintptr_t>(g) & 15) return PTV_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(sumsq, 0, sizeof(float), s) != hipSuccess) return PTV_ERR_LAUNCH;
  long nb = (n / 4 + 255) / 256; if (nb > 512) nb = 512; if (nb < 1) nb = 1;       // (one atomicAdd on *sumsq per block: keep the queue on that address short)
  hipLaunchKernelGGL(sumsq_kernel, dim3((int)nb), dim3(256), 0, s, g, n, sumsq, ord_scratch(s, nb, 1));
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

extern "C" int ptv_clip_adam_step_shadow(float* p, const float* g, float* m, float* v, long n, const float* sumsq, float gscale, float clip,
                                         float lr, float beta1, float beta2, float eps, int step, void* p16, void* stream) {
  if (!p || !g || !m || !v || !sumsq || n <= 0 || step < 1) return PTV_ERR_ARG;
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  long nb = (n + 255) / 256; if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(clip_adam_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, sumsq, gscale, clip, lr, beta1, beta2, eps,
                     (float)bc1, (float)sqrt(bc2), (__bf16*)p16, ptv::g_step_params);
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

// clip_grad_norm_(parameters, clip) + Adam.step() of the reference's loop body (module.py:142-144, train.py:50) as ONE entry point: the
// gradient's sum of squares and the update that reads it, back to back on `stream` (gradnorm_clip_adam_step)
extern "C" int ptv_gradnorm_clip_adam_step(float* p, const float* g, float* m, float* v, long n, float* sumsq, float gscale, float clip,
                                           float lr, float beta1, float beta2, float eps, int step, void* p16, void* stream) {
  PTV_TRY(ptv_grad_sumsq(g, n, sumsq, stream));
  return ptv_clip_adam_step_shadow(p, g, m, v, n, sumsq, gscale, clip, lr, beta1, beta2, eps, step, p16, stream);
}

extern "C" int ptv_clip_adam_step(float* p, const float* g, float* m, float* v, long n, const float* sumsq, float gscale, float clip,
                                  float lr, float beta1, float beta2, float eps, int step, void* stream) {
  return ptv_clip_adam_step_shadow(p, g, m, v, n, sumsq, gscale, clip, lr, beta1, beta2, eps, step, nullptr, stream);
}

extern "C" int ptv_note_token(const float* pitch, long ld_pitch, const int* dur_idx, long dur_stride, const float* W, const float* bias, int E,
                              float* pred, long ld_pred, long* xhat, long xhat_stride, int* plen, int n, int last,
                              const int* force_pitch, int M, void* stream) {
  if (!pitch || !dur_idx || !W || !bias || !pred || !xhat || !plen || M <= 0 || E <= 0) return PTV_ERR_ARG;
  int nb = (M + 3) / 4; if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(note_token_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, pitch, ld_pitch, dur_idx, dur_stride, W, bias, E,
                     pred, ld_pred, xhat, xhat_stride, plen, n, last, force_pitch, M);
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

extern "C" int ptv_chord_token(const float* root, const float* chroma, const float* bass, unsigned* masks2, float* token, int B, void* stream) {
  if (!root || !chroma || !bass || !masks2 || !token || B <= 0) return PTV_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(masks2, 0, 2 * sizeof(unsigned), s) != hipSuccess) return PTV_ERR_LAUNCH;
  hipLaunchKernelGGL(chord_argmax_kernel, dim3(cdiv(B, 256)), dim3(256), 0, s, root, chroma, bass, masks2, token, B);
  hipLaunchKernelGGL(chord_union_kernel, dim3(cdiv((long)B * 12, 256)), dim3(256), 0, s, masks2, token, B);
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

extern "C" int ptv_route_slices(const float* src, float* dstA, float* dstB, const int* mask, long slice_elems, int nslices, int accumulate, void* stream) {
  if (!src || !mask || slice_elems <= 0 || nslices <= 0) return PTV_ERR_ARG;
  long nb = (slice_elems * nslices + 255) / 256; if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(route_slices_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, src, dstA, dstB, mask, slice_elems, nslices, accumulate);
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

extern "C" int ptv_cast_bf16(const float* src, void* dst, long n, void* stream) {
  if (!src || !dst || n <= 0 || (n & 3)) return PTV_ERR_ARG;
  long nb = (n / 4 + 255) / 256; if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(cast_flat_bf16_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, src, (__bf16*)dst, n / 4);
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

extern "C" int ptv_transpose_cast_bf16_batched(const float* flat, void* flat_t, const long* desc, int nmat, long ntiles, void* stream) {
  if (!flat || !flat_t || !desc || nmat <= 0 || ntiles <= 0) return PTV_ERR_ARG;
  hipLaunchKernelGGL(transpose_cast_batched_kernel, dim3((unsigned)ntiles), dim3(256), 0, (hipStream_t)stream, flat, (__bf16*)flat_t, desc, nmat);
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

extern "C" int ptv_transpose_cast_bf16(const float* src, void* dst, int rows, int cols, void* stream) {
  if (!src || !dst || rows <= 0 || cols <= 0) return PTV_ERR_ARG;
  hipLaunchKernelGGL(transpose_cast_kernel, dim3(cdiv(cols, 32), cdiv(rows, 32)), dim3(256), 0, (hipStream_t)stream, src, (__bf16*)dst, rows, cols);
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

namespace ptv {
const float* g_step_params = nullptr;
int g_ordered = [] { const char* e = getenv("PTV_WGRAD_ORDERED"); return (e && e[0] == '0') ? 0 : 1; }();

// scratch of the ordered grid reductions (common.hpp): one allocation per stream, made on the stream's first use -- outside any
// capture when the captured step was warmed up first; inside a capture without a buffer the kernels fall back to atomics
int g_ord_fallbacks = 0;
OrdScratch ord_scratch(hipStream_t s, long need_floats, int need_counters) {
  OrdScratch none{nullptr, nullptr};
  if (!g_ordered) return none;
  static OrdScratch pool[64]; static hipStream_t keys[64]; static int n = 0; static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  if (need_floats > ORD_SLOT_FLOATS || need_counters > ORD_COUNTERS) { g_ord_fallbacks++; return none; }
  int i = 0;
  for (; i < n; i++) if (keys[i] == s) break;
  if (i < n) return pool[i];
  if (n == 64) { g_ord_fallbacks++; return none; }
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &st) == hipSuccess && st != hipStreamCaptureStatusNone) { g_ord_fallbacks++; return none; }
  char* p = nullptr;
  const size_t bytes = ORD_SLOT_FLOATS * sizeof(float) + ORD_COUNTERS * sizeof(unsigned);
  if (hipMalloc(reinterpret_cast<void**>(&p), bytes) != hipSuccess) { g_ord_fallbacks++; return none; }
  if (hipMemset(p + ORD_SLOT_FLOATS * sizeof(float), 0, ORD_COUNTERS * sizeof(unsigned)) != hipSuccess) { (void)hipFree(p); g_ord_fallbacks++; return none; }
  keys[n] = s; pool[n] = OrdScratch{reinterpret_cast<float*>(p), reinterpret_cast<unsigned*>(p + ORD_SLOT_FLOATS * sizeof(float))};
  return pool[n++];
}
}  // namespace ptv
extern "C" int ptv_wgrad_mode(int ordered);
extern "C" int ptv_ordered_reductions(int on) { ptv::g_ordered = on ? 1 : 0; return ptv_wgrad_mode(on); }
// number of reductions that ran on fp32 atomics although ordered mode was on (no workspace available); reset = 1 clears it
extern "C" long ptv_ordered_fallbacks(int reset) { const long n = ptv::g_ord_fallbacks; if (reset) ptv::g_ord_fallbacks = 0; return n; }
// Launches made from now on read their per-step scalars from this device array instead of their by-value arguments (NULL: by value
// again): [0] beta (ptv_loss_finalize / ptv_loss_bwd_scales), [1] lr, [2] 1 - beta1^t, [3] sqrt(1 - beta2^t) (ptv_clip_adam_step*).
// A hipGraph captured while it is set replays with whatever the host has written there since (graph_step.GraphedTrainStep).
extern "C" int ptv_step_params(const float* dev4) { ptv::g_step_params = dev4; return PTV_OK; }

extern "C" int ptv_zero_skip(int enable) { ptv::g_zero_skip = enable ? 1 : 0; return PTV_OK; }

// Test / diagnosis aid (tests/test_gpu_zz_dist.py): nwg workgroups that each hold `lds_bytes` of LDS and 256 threads for `usec`
// microseconds and do nothing -- what a collective kernel of another library looks like to the persistent recurrences when it sits
// on the CUs they were sized for (include/ptvae_hip.h: at most one workgroup per CU, gru_persist.hip plan()).
namespace ptv {
__global__ void pin_cus_kernel(long long ticks, int* sink) {
  extern __shared__ char pin_smem[];
  pin_smem[threadIdx.x] = (char)threadIdx.x;
  __syncthreads();
  const long long t0 = wall_clock64();                          // constant 100 MHz counter
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(16);
  if (sink && pin_smem[(threadIdx.x + 1) & 255] == 77 && ticks < 0) sink[0] = 1;      // (keeps the LDS allocation alive)
}
}  // namespace ptv
extern "C" int ptv_debug_pin_cus(int nwg, int lds_bytes, int usec, void* stream) {
  if (nwg <= 0 || lds_bytes < 256 || lds_bytes > 160 * 1024 || usec < 0 || usec > 100000) return PTV_ERR_ARG;
  if (lds_bytes > 64 * 1024 &&
      hipFuncSetAttribute((const void*)ptv::pin_cus_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) return PTV_ERR_LAUNCH;
  hipLaunchKernelGGL(ptv::pin_cus_kernel, dim3(nwg), dim3(256), lds_bytes, (hipStream_t)stream, (long long)usec * 100, (int*)nullptr);
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

// texture.hip -- TextureEncoder front end (ptvae.py:95-99,112-114):
//   Conv2d(1, C, kernel (4,12), stride (4,1)) -> ReLU -> MaxPool2d((1,4)) over pr_mat [B,32,128],
// fused into one kernel: the [B,32,128] piano-roll is read once (coalesced, 4 rows of 128 floats
// per beat staged in LDS), the [B,C,8,117] conv map never exists in HBM.
// Output pooled [B,C,8,29]; its raw reinterpretation as [B*8, C*29/... ] rows of 290 (the
// reference's .view(bs, 8, -1), ptvae.py:114) is done by the caller with a plain pointer cast.
#include "common.hpp"
#include "../../include/ptvae_hip.h"

namespace ptv {

constexpr int TX_MAXC = 16;

// element (b, ch, beat, pp) of the pooled map [B,C,8,29] inside the reference's raw view [B*8 rows][C*29] (ptvae.py:114) held with row
// stride ld >= C*29.  ld = C*29 is the plain contiguous tensor; a 16-byte multiple keeps the rows of fc1's operand aligned (the weight
// gradient of fc1 ran 334 us on the element-wise path with 290-float rows, round 4)
__device__ __forceinline__ long tx_index(int b, int ch, int beat, int pp, int C, long ld) {
  const int W = C * 29, f = (ch * 8 + beat) * 29 + pp;
  return ((long)b * 8 + f / W) * ld + f % W;
}

// one block iteration = one (b, beat); thread -> (ch, pp)
// arg (optional): [B*8][C*29] int8 -- which of the 4 pooled positions won (first max, as MaxPool2d), -1 where ReLU cut all four: what the
// backward needs of the recomputed convolution
__global__ void txt_conv_fwd_kernel(const float* __restrict__ pr, const float* __restrict__ w, const float* __restrict__ bias,
                                    float* __restrict__ pooled, int B, int C, long ld, signed char* __restrict__ arg) {
  __shared__ float rows[4][128];
  __shared__ float ws[TX_MAXC * 48 + TX_MAXC];
  for (int i = threadIdx.x; i < C * 48; i += blockDim.x) ws[i] = w[i];
  for (int i = threadIdx.x; i < C; i += blockDim.x) ws[C * 48 + i] = bias[i];
  const int nout = C * 29;
  for (long it = blockIdx.x; it < (long)B * 8; it += gridDim.x) {
    const int b = (int)(it / 8), beat = (int)(it % 8);
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += blockDim.x) rows[i >> 7][i & 127] = pr[((long)b * 32 + beat * 4) * 128 + i];
    __syncthreads();
    for (int o = threadIdx.x; o < nout; o += blockDim.x) {
      const int ch = o / 29, pp = o % 29;
      const float* wc = ws + ch * 48;
      float best = 0.f;                       // ReLU floor: max(relu(v_q)) = max(0, max v_q)
      int bq = -1;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int p0 = pp * 4 + q;
        float v = ws[C * 48 + ch];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
          for (int j = 0; j < 12; j++) v += wc[i * 12 + j] * rows[i][p0 + j];
        if (v > best) { best = v; bq = q; }      // first max wins ties, as MaxPool2d does
      }
      pooled[tx_index(b, ch, beat, pp, C, ld)] = best;
      if (arg) arg[it * nout + o] = (signed char)bq;
    }
    if (beat == 0 && ld > nout) {                                  // the row padding of this sample's 8 rows: finite (zero), never data
      const int padw = (int)(ld - nout);
      for (int i = threadIdx.x; i < 8 * padw; i += blockDim.x) pooled[((long)b * 8 + i / padw) * ld + nout + i % padw] = 0.f;
    }
  }
}

__device__ __forceinline__ void ordered_commit_tx(float* dw, float* dbias, const float* acc, int C, const OrdScratch& sc) {
  __shared__ int s_last;
  const int L = C * 49, n = gridDim.x;
  for (int i = threadIdx.x; i < L; i += blockDim.x) __hip_atomic_store(sc.slots + (long)blockIdx.x * L + i, acc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(sc.counters, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(n - 1);
  __syncthreads();
  if (!s_last) return;
  for (int i = threadIdx.x; i < L; i += blockDim.x) {
    float t = 0.f;
    int b = 0;
    for (; b + 8 <= n; b += 8) {                                    // eight partials in flight, added in block order
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; q++) v[q] = __hip_atomic_load(sc.slots + (long)(b + q) * L + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
      for (int q = 0; q < 8; q++) t += v[q];
    }
    for (; b < n; b++) t += __hip_atomic_load(sc.slots + (long)b * L + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int c2 = i / 49, k = i % 49;
    if (k < 48) dw[c2 * 48 + k] += t; else dbias[c2] += t;
  }
  if (threadIdx.x == 0) __hip_atomic_store(sc.counters, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// dW[ch,i,j] += sum dpool * [conv>0 at the arg-max q] * pr ; dbias likewise.  Conv is recomputed.
__global__ void txt_conv_bwd_kernel(const float* __restrict__ pr, const float* __restrict__ w, const float* __restrict__ bias,
                                    const float* __restrict__ dpooled, float* __restrict__ dw, float* __restrict__ dbias, int B, int C, long ld, OrdScratch sc,
                                    const signed char* __restrict__ arg) {
  __shared__ float stage[TX_MAXC * 29][7];
  constexpr int NB = 4;                                           // (sample, beat) items per trip: their 2-KB row loads are in flight together
  __shared__ float rows[NB][4][128];
  __shared__ float ws[TX_MAXC * 48 + TX_MAXC];
  __shared__ float acc[TX_MAXC * 49];
  if (!arg) {
    for (int i = threadIdx.x; i < C * 48; i += blockDim.x) ws[i] = w[i];
    for (int i = threadIdx.x; i < C; i += blockDim.x) ws[C * 48 + i] = bias[i];
  }
  for (int i = threadIdx.x; i < C * 49; i += blockDim.x) acc[i] = 0.f;
  const int nout = C * 29;
  // thread o = threadIdx.x (< nout) keeps a private gradient for its channel across all iterations
  float g[49];
#pragma unroll
  for (int k = 0; k < 49; k++) g[k] = 0.f;
  const int o = threadIdx.x;
  const int ch = o / 29, pp = o % 29;
  // (round 4: one item per trip was a chain of 32 load -> barrier -> compute round trips per block, 207 us for a 480-weight convolution)
  const long total = (long)B * 8;
  for (long base = (long)blockIdx.x * NB; base < total; base += (long)gridDim.x * NB) {
    __syncthreads();
    for (int i = threadIdx.x; i < NB * 512; i += blockDim.x) {
      const long it = base + (i >> 9);
      if (it < total) {
        const int b = (int)(it / 8), beat = (int)(it % 8);
        rows[i >> 9][(i >> 7) & 3][i & 127] = pr[((long)b * 32 + beat * 4) * 128 + (i & 511)];
      }
    }
    // this thread's pooled gradients of the NB items: requested before the barrier, consumed after the recomputed convolutions
    float dn[NB];
    int an[NB];
#pragma unroll
    for (int n = 0; n < NB; n++) {
      const long it = base + n;
      dn[n] = (o < nout && it < total) ? dpooled[tx_index((int)(it / 8), ch, (int)(it % 8), pp, C, ld)] : 0.f;
      an[n] = (arg && o < nout && it < total) ? (int)arg[it * nout + o] : -1;
    }
    __syncthreads();
    if (o < nout) {
      const float* wc = ws + ch * 48;
#pragma unroll
      for (int n = 0; n < NB; n++) {
        if (base + n >= total) break;
        float best = 0.f; int bq = -1;
        if (arg) bq = an[n];                       // the forward's decision (same arithmetic, same result -- without the 192 MACs)
        else {
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const int p0 = pp * 4 + q;
            float v = ws[C * 48 + ch];
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
              for (int j = 0; j < 12; j++) v += wc[i * 12 + j] * rows[n][i][p0 + j];
            if (v > best) { best = v; bq = q; }      // first max wins ties, as MaxPool2d does
          }
        }
        if (bq >= 0) {
          const float d = dn[n];
          const int p0 = pp * 4 + bq;
#pragma unroll
          for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 12; j++) g[i * 12 + j] += d * rows[n][i][p0 + j];
          g[48] += d;
        }
      }
    }
  }
  // the 29 pooled positions of a channel are 29 threads: their private gradients meet in LDS, 7 taps at a time, and ONE thread per
  // (channel, tap) adds them in position order (LDS float atomics would add in arrival order)
#pragma unroll
  for (int kk = 0; kk < 7; kk++) {
    __syncthreads();
    if (o < nout) {
#pragma unroll
      for (int j = 0; j < 7; j++) stage[o][j] = g[kk * 7 + j];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * 7; i += blockDim.x) {
      const int c2 = i / 7, j = i % 7;
      float t = 0.f;
      for (int q = 0; q < 29; q++) t += stage[c2 * 29 + q][j];
      acc[c2 * 49 + kk * 7 + j] = t;
    }
  }
  __syncthreads();
  if (sc.slots) {                                                // acc [C][49] -> dw [C][48] | dbias [C]: one ordered vector of C*49
    ordered_commit_tx(dw, dbias, acc, C, sc);
    return;
  }
  for (int i = threadIdx.x; i < C * 49; i += blockDim.x) {
    float v = acc[i];
    if (v == 0.f) continue;
    int c2 = i / 49, k = i % 49;
    if (k < 48) atomicAdd(dw + c2 * 48 + k, v); else atomicAdd(dbias + c2, v);
  }
}

}  // namespace ptv

using namespace ptv;

extern "C" int ptv_txt_conv_relu_pool_fwd_rows(const float* pr_mat, const float* w, const float* bias, float* feat, long ld, int B, int C,
                                               signed char* arg, void* stream) {
  if (!pr_mat || !w || !bias || !feat || B <= 0 || C <= 0 || C > TX_MAXC || ld < C * 29) return PTV_ERR_ARG;
  int grid = B * 8 < 2048 ? B * 8 : 2048;
  hipLaunchKernelGGL(txt_conv_fwd_kernel, dim3(grid), dim3(320), 0, (hipStream_t)stream, pr_mat, w, bias, feat, B, C, ld, arg);
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}
extern "C" int ptv_txt_conv_relu_pool_fwd(const float* pr_mat, const float* w, const float* bias, float* pooled, int B, int C, void* stream) {
  return ptv_txt_conv_relu_pool_fwd_rows(pr_mat, w, bias, pooled, (long)C * 29, B, C, nullptr, stream);
}

extern "C" int ptv_txt_conv_relu_pool_bwd_rows(const float* pr_mat, const float* w, const float* bias, const float* dfeat, long ld,
                                               float* dw, float* dbias, int B, int C, const signed char* arg, void* stream) {
  if (!pr_mat || !dfeat || !dw || !dbias || B <= 0 || C <= 0 || C > TX_MAXC || ld < C * 29) return PTV_ERR_ARG;
  if (!arg && (!w || !bias)) return PTV_ERR_ARG;                  // without the forward's arg-max map the convolution is recomputed
  int nthreads = ((C * 29 + 63) / 64) * 64;       // one thread per (ch, pp)
  int grid = (B * 8 + 3) / 4 < 512 ? (B * 8 + 3) / 4 : 512;       // four items per trip
  OrdScratch sc = ord_scratch((hipStream_t)stream, 256L * C * 49, 1);
  if (sc.slots && grid > 256) grid = 256;                         // (the last block adds `grid` partials per tap)
  hipLaunchKernelGGL(txt_conv_bwd_kernel, dim3(grid), dim3(nthreads), 0, (hipStream_t)stream, pr_mat, w, bias, dfeat, dw, dbias, B, C, ld, sc, arg);
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}
extern "C" int ptv_txt_conv_relu_pool_bwd(const float* pr_mat, const float* w, const float* bias, const float* dpooled,
                                          float* dw, float* dbias, int B, int C, void* stream) {
  return ptv_txt_conv_relu_pool_bwd_rows(pr_mat, w, bias, dpooled, (long)C * 29, dw, dbias, B, C, nullptr, stream);
}

// elementwise.hip -- the HBM-bound kernels of the train step: layout shuffles, reductions over
// rows / steps, note-embedding gather + its gradient, reparameterisation + KL.
// All are simple streaming kernels: coalesced 4-byte/16-byte accesses, grid-stride loops,
// wave-shuffle + LDS reductions, one atomic per block where a cross-block sum is needed.
#include "common.hpp"
#include "gemm_core.hpp"
#include "../../include/ptvae_hip.h"

namespace ptv {

static inline int grid_for(long n, int block = 256, int cap = 4096) {
  long b = (n + block - 1) / block;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return (int)b;
}

// dst[r*ldd + c] = (accumulate ? dst : 0) + alpha * src[r*lds + c]      (lds may be 0: row broadcast)
__global__ void copy2d_kernel(float* dst, long ldd, const float* src, long lds, long rows, int cols, float alpha, int accumulate) {
  long total = rows * cols;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long r = i / cols; int c = (int)(i % cols);
    float v = alpha * src[r * lds + c];
    float* d = dst + r * ldd + c;
    *d = accumulate ? *d + v : v;
  }
}

// [D0, D1, W] -> [D1, D0, W]
__global__ void transpose01_kernel(float* dst, const float* src, int D0, int D1, int W) {
  long total = (long)D0 * D1 * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int w = (int)(i % W); long r = i / W; int d0 = (int)(r % D0); int d1 = (int)(r / D0);
    dst[i] = src[((long)d0 * D1 + d1) * W + w];
  }
}

// out[i] = (accumulate? out[i] : 0) + sum_t in[t*stride + i]
__global__ void sum_steps_kernel(float* out, const void* in, long n, int T, long stride, int accumulate, int in_bf16, const int* t_top) {
  if (t_top) T = min(T, *t_top + 1);                              // the planes after *t_top are known to be zero
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float s = accumulate ? out[i] : 0.f;
    for (int t = 0; t < T; t++) s += ld1f(in, t * stride + i, in_bf16);
    out[i] = s;
  }
}

// vector form: 16-byte loads (8 bf16 / 4 fp32 per thread), the T planes of one element group requested together
template <bool BF>
__global__ void sum_steps_vec_kernel(float* __restrict__ out, const void* __restrict__ in, long nvec, int T_, long stride, int accumulate,
                                     const int* __restrict__ t_top, const int* __restrict__ seg_n, int row_elems) {
  __builtin_amdgcn_s_setprio(3);                                         // always part of a latency chain
  constexpr int E = BF ? 8 : 4;
  if (t_top) T_ = min(T_, *t_top + 1);
  for (long v = (long)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (long)gridDim.x * blockDim.x) {
    const long i = v * E;
    // row segments (ptv_sum_steps_seg): plane t holds something in its first seg_n[t] rows only, and seg_n does not grow with t -- the planes
    // that hold this element's row are a prefix
    int T = T_;
    if (seg_n) {
      const int row = (int)(i / row_elems);
      T = 0;
      while (T < T_ && seg_n[T] > row) T++;
    }
    float s[E];
#pragma unroll
    for (int e = 0; e < E; e++) s[e] = accumulate ? out[i + e] : 0.f;
    int t = 0;
    for (; t + 4 <= T; t += 4) {
      if constexpr (BF) {
        bf16x8 q[4];
#pragma unroll
        for (int u = 0; u < 4; u++) q[u] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const __bf16*>(in) + (t + u) * stride + i);
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
          for (int e = 0; e < E; e++) s[e] += (float)q[u][e];
      } else {
        float4 q[4];
#pragma unroll
        for (int u = 0; u < 4; u++) q[u] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(in) + (t + u) * stride + i);
#pragma unroll
        for (int u = 0; u < 4; u++) { s[0] += q[u].x; s[1] += q[u].y; s[2] += q[u].z; s[3] += q[u].w; }
      }
    }
    for (; t < T; t++) {
#pragma unroll
      for (int e = 0; e < E; e++) s[e] += ld1f(in, t * stride + i + e, BF);
    }
#pragma unroll
    for (int e = 0; e < E; e += 4) *reinterpret_cast<float4*>(out + i + e) = make_float4(s[e], s[e + 1], s[e + 2], s[e + 3]);
  }
}

// out[g*N + n] += sum_{rows r with (sel ? sel[r] : 0) == g} A[r*lda + n]   (G <= 2; block partial + atomics)
// block = 256 threads = 16 column quads (64 columns, 16-byte loads) x 16 row lanes
template <bool VEC>
__global__ void colsum_kernel(float* out, const void* __restrict__ A, long lda, long rows, int N, const int* __restrict__ sel, int G, int bf, OrdScratch sc) {
  // (2.6 KB of LDS, not 8.7: a bias-gradient launch on a sibling stream then fits beside the 148-158-KB workgroups of the chain's row / head
  // kernels instead of waiting for a CU to drain -- round 5 saw such launches take 100-650 us for microseconds of work)
  __shared__ float red[4][2][64];
  __shared__ float tot[2 * 64];
  const int cq = threadIdx.x & 15, ry = threadIdx.x >> 4;
  const int n = blockIdx.x * 64 + cq * 4;
  const long rows_per_block = (rows + gridDim.y - 1) / gridDim.y;
  const long r0 = blockIdx.y * rows_per_block;
  const long r1 = min(rows, r0 + rows_per_block);
  float s[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  if (n < N) {
    for (long r = r0 + ry; r < r1; r += 16) {
      const int g = sel ? sel[r] : 0;
      float4 v;
      if (VEC) v = ld4f(A, r * lda + n, bf);
      else {
        v.x = ld1f(A, r * lda + n, bf); v.y = n + 1 < N ? ld1f(A, r * lda + n + 1, bf) : 0.f;
        v.z = n + 2 < N ? ld1f(A, r * lda + n + 2, bf) : 0.f; v.w = n + 3 < N ? ld1f(A, r * lda + n + 3, bf) : 0.f;
      }
      if (g == 0) { s[0][0] += v.x; s[0][1] += v.y; s[0][2] += v.z; s[0][3] += v.w; }
      else if (g == 1) { s[1][0] += v.x; s[1][1] += v.y; s[1][2] += v.z; s[1][3] += v.w; }
    }
  }
  // a wave holds 4 row lanes x 16 column quads (lane = (ry & 3) * 16 + cq): the row lanes meet by shuffles, the 4 waves through LDS
#pragma unroll
  for (int g = 0; g < 2; g++)
#pragma unroll
    for (int e = 0; e < 4; e++) {
      float v = s[g][e];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      s[g][e] = v;
    }
  if ((threadIdx.x & 63) < 16)
    for (int g = 0; g < G; g++)
#pragma unroll
      for (int e = 0; e < 4; e++) red[threadIdx.x >> 6][g][cq * 4 + e] = s[g][e];
  __syncthreads();
  for (int i = threadIdx.x; i < G * 64; i += blockDim.x) {
    const int g = i / 64, c = i % 64;
    float t = 0.f;
#pragma unroll
    for (int y = 0; y < 4; y++) t += red[y][g][c];
    if (sc.slots) tot[i] = t;
    else if (blockIdx.x * 64 + c < N && t != 0.f) atomicAdd(out + (long)g * N + blockIdx.x * 64 + c, t);
  }
  if (sc.slots) {
    // ordered: this column block's row blocks park their [G][64] partials; the last one adds them in row-block order
    __syncthreads();
    __shared__ int s_last;
    const int L = G * 64, n = gridDim.y;
    float* base = sc.slots + (long)blockIdx.x * n * L;
    for (int i = threadIdx.x; i < L; i += blockDim.x) __hip_atomic_store(base + (long)blockIdx.y * L + i, tot[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(sc.counters + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(n - 1);
    __syncthreads();
    if (s_last) {
      // the partials [n][G][64] are summed the way the rows were: row lane ry takes partials ry, ry + 16, ... (float4 per column quad),
      // then the 16 lanes meet (shuffles inside a wave, LDS across the waves) -- a fixed association, 16 loads in flight per column quad
      for (int g = 0; g < G; g++) {
        float a4[4] = {0.f, 0.f, 0.f, 0.f};
        for (int b = ry; b < n; b += 16) {
          const float* q = base + (long)b * L + g * 64 + cq * 4;      // (agent-scope loads: other XCDs' L2s wrote these)
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; e++) v[e] = __hip_atomic_load(q + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
          for (int e = 0; e < 4; e++) a4[e] += v[e];
        }
#pragma unroll
        for (int e = 0; e < 4; e++) {                                 // the wave's 4 row lanes by shuffles, the waves through LDS (as above)
          float v = a4[e];
          v += __shfl_xor(v, 16, 64);
          v += __shfl_xor(v, 32, 64);
          if ((threadIdx.x & 63) < 16) red[threadIdx.x >> 6][g][cq * 4 + e] = v;
        }
      }
      __syncthreads();
      for (int i = threadIdx.x; i < L; i += blockDim.x) {
        const int g = i / 64, c = i % 64;
        if (blockIdx.x * 64 + c >= N) continue;
        float s2 = 0.f;
#pragma unroll
        for (int y = 0; y < 4; y++) s2 += red[y][g][c];
        out[(long)g * N + blockIdx.x * 64 + c] += s2;
      }
      if (threadIdx.x == 0) __hip_atomic_store(sc.counters + blockIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// note embedding as a gather (ptvae.py:299-313 builds a dense multi-hot and multiplies by the
// (P+D) x E weight; one-hot . W is a column gather + D duration columns)
//   x [B,S,N,1+D] int64 -> emb step-major [N][S][B][E], lengths [S][B] int32
// Grid geometry (ptvae.py:127-147 / :220-241): S = num_step, N = max_simu_note, P = pitch_range, D = dur_width, pad = pitch_pad.
// DEF = the 32 x 16 x (130+5) grid of init_model() with every bound a compile-time constant; the other instantiation takes them at
// run time (train.py:32 builds PtvaeEncoder(max_pitch=31): P = 34).
// ---------------------------------------------------------------------------------------------
struct GridGeom { int S, N, P, D, pad; };
constexpr int GEOM_DMAX = 8;

template <bool DEF>
__global__ void embed_fwd_kernel(const long* __restrict__ x, const float* __restrict__ W, const float* __restrict__ bias,
                                 float* __restrict__ emb, int B, int E, GridGeom g) {
  __builtin_amdgcn_s_setprio(3);                                         // always part of a latency chain
  const int S = DEF ? 32 : g.S, N = DEF ? 16 : g.N, P = DEF ? 130 : g.P, D = DEF ? 5 : g.D;
  constexpr int DM = DEF ? 5 : GEOM_DMAX;
  const int C = P + D;
  extern __shared__ __attribute__((aligned(16))) float wt[];     // [P+D][E] transposed weight
  for (int i = threadIdx.x; i < C * E; i += blockDim.x) { int e = i / C, p = i % C; wt[p * E + e] = W[i]; }     // (coalesced reads; the transposing side is the LDS)
  __syncthreads();
  const long notes = (long)B * S * N;
  if ((E & 3) == 0 && blockDim.x >= E / 4) {
    // 4 features per thread: 16-byte LDS reads and stores, blockDim / (E/4) notes in flight per block
    const int tpn = E / 4, per = blockDim.x / tpn;
    const int e = (threadIdx.x % tpn) * 4, sub = threadIdx.x / tpn;
    if (sub >= per) return;
    const float4 bv = *reinterpret_cast<const float4*>(bias + e);
    // four notes per thread and trip: their index rows (1+D x int64 each, one cache line apart from every neighbour) are requested
    // together -- with one note per trip the loop was a chain of exposed load latencies (8 notes in flight per block)
    const long stride = (long)gridDim.x * per;
    for (long i0 = (long)blockIdx.x * per + sub; i0 < notes; i0 += 4 * stride) {
      long xi[4][1 + DM];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const long i = min(i0 + u * stride, notes - 1);
        const int b = (int)(i % B); const long q = i / B; const int t = (int)(q % S), n = (int)(q / S);
        const long* xr = x + (((long)b * S + t) * N + n) * (1 + D);
#pragma unroll
        for (int d = 0; d < 1 + DM; d++) xi[u][d] = d <= D ? xr[d] : 0;
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const long i = i0 + u * stride;
        if (i >= notes) break;
        const int p = (int)xi[u][0];
        float4 v = bv;
        if (p >= 0 && p < P) { const float4 w = *reinterpret_cast<const float4*>(wt + p * E + e); v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w; }
#pragma unroll
        for (int d = 0; d < DM; d++) {
          if (d >= D) break;
          const float f = (float)xi[u][1 + d];
          const float4 w = *reinterpret_cast<const float4*>(wt + (P + d) * E + e);
          v.x += w.x * f; v.y += w.y * f; v.z += w.z * f; v.w += w.w * f;
        }
        *reinterpret_cast<float4*>(emb + i * E + e) = v;
      }
    }
    return;
  }
  const int per = blockDim.x / E > 0 ? blockDim.x / E : 1;       // notes in flight per block
  const int e = threadIdx.x % E, sub = threadIdx.x / E;
  if (sub >= per) return;
  for (long i = (long)blockIdx.x * per + sub; i < notes; i += (long)gridDim.x * per) {
    // i indexes the OUTPUT row (n, t, b)
    const int b = (int)(i % B); const long q = i / B; const int t = (int)(q % S), n = (int)(q / S);
    const long* xr = x + (((long)b * S + t) * N + n) * (1 + D);
    const int p = (int)xr[0];
    float v = bias[e];
    if (p >= 0 && p < P) v += wt[p * E + e];
    for (int d = 0; d < D; d++) v += wt[(P + d) * E + e] * (float)xr[1 + d];
    emb[i * E + e] = v;
  }
}

__global__ void lengths_kernel(const long* __restrict__ x, int* __restrict__ lengths, int B, GridGeom g) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;       // output index t*B + b
  if (i >= (long)B * g.S) return;
  int b = (int)(i % B), t = (int)(i / B);
  const long* xr = x + ((long)b * g.S + t) * g.N * (1 + g.D);
  int pad = 0;
  for (int n = 0; n < g.N; n++) pad += (xr[n * (1 + g.D)] == g.pad);
  lengths[i] = g.N - pad;
}

// multihot [N*S*B, P+D] (row stride ld) in the embedding's step-major row order: the matrix the
// reference multiplies by note_embedding.weight (ptvae.py:299-313).  Only the BACKWARD uses it here:
// dW = demb^T . multihot is a K = 262144-deep product that belongs on the MFMA (split-K) path instead of
// on atomics.  T = float, or bf16 (0, 1 and 2 are exact: half the bytes for the bf16-precision weight-gradient product; its rows are
// zero-filled up to `cols` >= P+D)
template <typename T>
__global__ void multihot_kernel(const long* __restrict__ x, T* __restrict__ out, long ld, int B, int cols, GridGeom g) {
  const long notes = (long)B * g.S * g.N;
  const int C = g.P + g.D;
  for (long i = (long)blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6); i < notes; i += (long)gridDim.x * (blockDim.x / 64)) {
    const int b = (int)(i % B); const long q = i / B; const int t = (int)(q % g.S), n = (int)(q / g.S);
    const long* xr = x + (((long)b * g.S + t) * g.N + n) * (1 + g.D);
    const int p = (int)xr[0];
    T* o = out + i * ld;
    for (int c = threadIdx.x & 63; c < cols; c += 64) o[c] = (T)(c < g.P ? (c == p ? 1.f : 0.f) : (c < C ? (float)xr[1 + c - g.P] : 0.f));
  }
}

// ---------------------------------------------------------------------------------------------
// reparameterisation + KL (train_utils.py:33-34,45-49):  z = mu + std*eps ; kl = mean(-log std + (std^2+mu^2)/2 - 1/2)
// ---------------------------------------------------------------------------------------------
__global__ void reparam_kl_fwd_kernel(const float* __restrict__ mu, const float* __restrict__ sd, const float* __restrict__ eps,
                                      float* __restrict__ z, long ldz, float* __restrict__ kl_sum, int B, int Z, OrdScratch sc) {
  __shared__ float red[4];
  long total = (long)B * Z;
  float s = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int b = (int)(i / Z), j = (int)(i % Z);
    float m = mu[i], d = sd[i];
    z[(long)b * ldz + j] = m + d * (eps ? eps[i] : 0.f);
    s += -logf(d) + (d * d + m * m) * 0.5f - 0.5f;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) red[0] = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  ordered_commit(kl_sum, red, 1, sc, 0, blockIdx.x, gridDim.x);
}

// dmu = dz + klw*mu ; dlv = (dz*eps + klw*(std - 1/std)) * std      (lv = log std is the Linear output)
__global__ void reparam_kl_bwd_kernel(const float* __restrict__ mu, const float* __restrict__ sd, const float* __restrict__ eps,
                                      const float* __restrict__ dz, long lddz, const float* __restrict__ dmu_ext,
                                      const float* __restrict__ dsd_ext, float klw, int mul_sd,
                                      float* __restrict__ dmu, float* __restrict__ dlv, int B, int Z) {
  long total = (long)B * Z;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int b = (int)(i / Z), j = (int)(i % Z);
    float m = mu[i], d = sd[i];
    float g = dz ? dz[(long)b * lddz + j] : 0.f;
    float gm = g + klw * m + (dmu_ext ? dmu_ext[i] : 0.f);
    float gs = g * (eps ? eps[i] : 0.f) + klw * (d - 1.0f / d) + (dsd_ext ? dsd_ext[i] : 0.f);
    dmu[i] = gm;
    dlv[i] = mul_sd ? gs * d : gs;
  }
}

}  // namespace ptv

using namespace ptv;

extern "C" int ptv_copy2d(float* dst, long ldd, const float* src, long lds, long rows, int cols, float alpha, int accumulate, void* stream) {
  if (!dst || !src || rows < 0 || cols < 0) return PTV_ERR_ARG;
  if (rows == 0 || cols == 0) return PTV_OK;
  hipLaunchKernelGGL(copy2d_kernel, dim3(grid_for(rows * cols)), dim3(256), 0, (hipStream_t)stream, dst, ldd, src, lds, rows, cols, alpha, accumulate);
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

extern "C" int ptv_transpose01(float* dst, const float* src, int D0, int D1, int W, void* stream) {
  if (!dst || !src || D0 <= 0 || D1 <= 0 || W <= 0) return PTV_ERR_ARG;
  hipLaunchKernelGGL(transpose01_kernel, dim3(grid_for((long)D0 * D1 * W)), dim3(256), 0, (hipStream_t)stream, dst, src, D0, D1, W);
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

// top = max(top, (last row block that holds a non-zero) / unit): which trailing part of a gradient matrix is all zero
__global__ void last_nonzero_kernel(const float* __restrict__ x, long rows, int cols, long ld, long unit, int* __restrict__ top, int vec) {
  __builtin_amdgcn_s_setprio(3);                                         // always part of a latency chain
  const long chunks = (rows + 63) / 64;
  __shared__ int s_top;
  for (long c = chunks - 1 - blockIdx.x; c >= 0; c -= gridDim.x) {
    const long r1 = min(rows, (c + 1) * 64);
    // other blocks raise *top concurrently: ONE thread samples it per trip so that the whole block takes the same branch (a
    // per-thread read could send some waves out of the loop and the rest into the barrier below)
    __syncthreads();                                                               // (the previous trip's readers of s_top are done)
    if (threadIdx.x == 0) s_top = *reinterpret_cast<volatile int*>(top);
    __syncthreads();
    if ((r1 - 1) / unit <= s_top) break;                                           // the chunk's LAST row is already covered by a report
    bool nz = false;
    if (vec) {
      // the chunk's rows with their padding are one contiguous span: 16-byte loads.  (Padding that is not zero can only make `top` too
      // large, i.e. less is skipped -- never wrong.)
      const float4* p4 = reinterpret_cast<const float4*>(x + c * 64 * ld);
      const long n4 = (r1 - c * 64) * ld / 4;
      for (long i = threadIdx.x; i < n4; i += blockDim.x) {
        const float4 v = p4[i];
        nz |= (v.x != 0.f) | (v.y != 0.f) | (v.z != 0.f) | (v.w != 0.f);
      }
    } else
    for (long i = c * 64 * (long)cols + threadIdx.x; i < r1 * cols; i += blockDim.x) {
      const long r = i / cols; const int q = (int)(i % cols);
      nz |= x[r * ld + q] != 0.f;
    }
    if (__syncthreads_or(nz)) {
      if (threadIdx.x == 0) atomicMax(top, (int)((r1 - 1) / unit));
      break;                                                                       // every earlier chunk of this block is below it
    }
  }
}

extern "C" int ptv_sum_steps_top(float* out, const void* in, long n, int T, long stride, int accumulate, int in_bf16, const int* t_top,
                                 void* stream) {
  return ptv_sum_steps_seg(out, in, n, T, stride, accumulate, in_bf16, t_top, nullptr, 0, stream);
}

extern "C" int ptv_sum_steps_seg(float* out, const void* in, long n, int T, long stride, int accumulate, int in_bf16, const int* t_top,
                                 const int* seg_n, long row_elems, void* stream) {
  if (!out || !in || n <= 0 || T <= 0) return PTV_ERR_ARG;
  const int E = in_bf16 ? 8 : 4;
  const bool vec = (n % E) == 0 && (stride % E) == 0 && (reinterpret_cast<uintptr_t>(in) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0;
  if (seg_n && (!vec || row_elems <= 0 || (row_elems % E))) return PTV_ERR_ARG;
  if (vec && in_bf16) hipLaunchKernelGGL(sum_steps_vec_kernel<true>, dim3(grid_for(n / E)), dim3(256), 0, (hipStream_t)stream, out, in, n / E, T, stride, accumulate, t_top, seg_n, (int)row_elems);
  else if (vec) hipLaunchKernelGGL(sum_steps_vec_kernel<false>, dim3(grid_for(n / E)), dim3(256), 0, (hipStream_t)stream, out, in, n / E, T, stride, accumulate, t_top, seg_n, (int)row_elems);
  else hipLaunchKernelGGL(sum_steps_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, out, in, n, T, stride, accumulate, in_bf16, t_top);
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

extern "C" int ptv_sum_steps(float* out, const void* in, long n, int T, long stride, int accumulate, int in_bf16, void* stream) {
  return ptv_sum_steps_top(out, in, n, T, stride, accumulate, in_bf16, nullptr, stream);
}

extern "C" int ptv_colsum(float* out, const void* A, long lda, long rows, int N, const int* sel, int G, int a_bf16, void* stream) {
  if (!out || !A || rows < 0 || N <= 0 || G <= 0 || G > 2) return PTV_ERR_ARG;
  if (rows == 0) return PTV_OK;
  int gx = cdiv(N, 64);
  long want = 2048 / gx; if (want < 1) want = 1;                 // ~2048 blocks in flight
  long gy = (rows + 63) / 64; if (gy > want) gy = want; if (gy < 1) gy = 1;
  // 16-byte loads need whole column quads inside the row: N a multiple of 4, or rows padded up to one (the 130 pitch logits live in
  // 136-float rows; whatever the padding holds only meets sums that are never stored)
  const bool vec = ((lda & 3) == 0) && (lda >= ((N + 3) & ~3)) && ((reinterpret_cast<uintptr_t>(A) & (a_bf16 ? 7 : 15)) == 0);
  OrdScratch sc = ord_scratch((hipStream_t)stream, (long)gx * gy * G * 64, gx);
  if (vec) hipLaunchKernelGGL((colsum_kernel<true>), dim3(gx, (int)gy), dim3(256), 0, (hipStream_t)stream, out, A, lda, rows, N, sel, G, a_bf16, sc);
  else hipLaunchKernelGGL((colsum_kernel<false>), dim3(gx, (int)gy), dim3(256), 0, (hipStream_t)stream, out, A, lda, rows, N, sel, G, a_bf16, sc);
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

static inline bool geom_ok(int S, int N, int P, int D) { return S > 0 && N > 0 && P > 0 && D >= 0 && D <= GEOM_DMAX; }

extern "C" int ptv_embed_fwd_geom(const long* x, const float* W, const float* bias, float* emb, int* lengths, int B, int E,
                                  int S, int N, int P, int D, int pad, void* stream) {
  if (!x || !W || !bias || !emb || B <= 0 || E <= 0 || E > 256 || !geom_ok(S, N, P, D)) return PTV_ERR_ARG;
  if ((size_t)(P + D) * E * sizeof(float) > 150 * 1024) return PTV_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const GridGeom g{S, N, P, D, pad};
  const bool def = S == 32 && N == 16 && P == 130 && D == 5;
  const size_t lds = (size_t)(P + D) * E * sizeof(float);
  const int grid = grid_for((long)B * S * N, 8, 512);                     // (default grid: two 69-KB blocks per CU: one round)
  if (def) hipLaunchKernelGGL((embed_fwd_kernel<true>), dim3(grid), dim3(256), lds, s, x, W, bias, emb, B, E, g);
  else {
    static bool attr = false;
    if (!attr) { if (hipFuncSetAttribute((const void*)embed_fwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) return PTV_ERR_LAUNCH; attr = true; }
    hipLaunchKernelGGL((embed_fwd_kernel<false>), dim3(grid), dim3(256), lds, s, x, W, bias, emb, B, E, g);
  }
  if (lengths) hipLaunchKernelGGL(lengths_kernel, dim3(cdiv((long)B * S, 256)), dim3(256), 0, s, x, lengths, B, g);
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

extern "C" int ptv_embed_fwd(const long* x, const float* W, const float* bias, float* emb, int* lengths, int B, int E, void* stream) {
  return ptv_embed_fwd_geom(x, W, bias, emb, lengths, B, E, 32, 16, 130, 5, 130, stream);
}

extern "C" int ptv_grid_lengths_geom(const long* x, int* lengths, int B, int S, int N, int D, int pad, void* stream) {
  if (!x || !lengths || B <= 0 || !geom_ok(S, N, 1, D)) return PTV_ERR_ARG;
  hipLaunchKernelGGL(lengths_kernel, dim3(cdiv((long)B * S, 256)), dim3(256), 0, (hipStream_t)stream, x, lengths, B, GridGeom{S, N, 1, D, pad});
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

extern "C" int ptv_grid_lengths(const long* x, int* lengths, int B, void* stream) {
  return ptv_grid_lengths_geom(x, lengths, B, 32, 16, 5, 130, stream);
}

extern "C" int ptv_multihot_geom(const long* x, void* out, long ld, int B, int S, int N, int P, int D, int bf16, void* stream) {
  if (!x || !out || B <= 0 || !geom_ok(S, N, P, D) || ld < P + D) return PTV_ERR_ARG;
  long nb = ((long)B * S * N + 3) / 4; if (nb > 8192) nb = 8192;
  const GridGeom g{S, N, P, D, 0};
  if (bf16) {
    const int cols = (int)(ld < (long)((P + D + 7) / 8 * 8) ? ld : (P + D + 7) / 8 * 8);         // zero-filled up to the 16-byte row granule
    hipLaunchKernelGGL((multihot_kernel<__bf16>), dim3((int)nb), dim3(256), 0, (hipStream_t)stream, x, (__bf16*)out, ld, B, cols, g);
  } else hipLaunchKernelGGL((multihot_kernel<float>), dim3((int)nb), dim3(256), 0, (hipStream_t)stream, x, (float*)out, ld, B, P + D, g);
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

extern "C" int ptv_multihot(const long* x, float* out, long ld, int B, void* stream) {
  return ptv_multihot_geom(x, out, ld, B, 32, 16, 130, 5, 0, stream);
}

extern "C" int ptv_multihot_bf16(const long* x, void* out, long ld, int B, void* stream) {
  if (ld < 136) return PTV_ERR_ARG;
  return ptv_multihot_geom(x, out, ld, B, 32, 16, 130, 5, 1, stream);
}

extern "C" int ptv_reparam_kl_fwd(const float* mu, const float* sd, const float* eps, float* z, long ldz, float* kl_sum, int B, int Z, void* stream) {
  if (!mu || !sd || !z || !kl_sum || B <= 0 || Z <= 0) return PTV_ERR_ARG;
  hipLaunchKernelGGL(reparam_kl_fwd_kernel, dim3(grid_for((long)B * Z, 256, 256)), dim3(256), 0, (hipStream_t)stream, mu, sd, eps, z, ldz, kl_sum, B, Z, kl_sum ? ord_scratch((hipStream_t)stream, 256, 1) : OrdScratch{nullptr, nullptr});
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

extern "C" int ptv_reparam_kl_bwd(const float* mu, const float* sd, const float* eps, const float* dz, long lddz,
                                  const float* dmu_ext, const float* dsd_ext, float klw, int mul_sd, float* dmu, float* dlv, int B, int Z, void* stream) {
  if (!mu || !sd || !dmu || !dlv || B <= 0 || Z <= 0) return PTV_ERR_ARG;
  hipLaunchKernelGGL(reparam_kl_bwd_kernel, dim3(grid_for((long)B * Z)), dim3(256), 0, (hipStream_t)stream, mu, sd, eps, dz, lddz, dmu_ext, dsd_ext, klw, mul_sd, dmu, dlv, B, Z);
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

extern "C" int ptv_last_nonzero_unit(const float* x, long rows, int cols, long ld, long unit, int* top, void* stream) {
  if (!x || !top || rows <= 0 || cols <= 0 || ld < cols || unit <= 0) return PTV_ERR_ARG;
  long nb = (rows + 63) / 64; if (nb > 2048) nb = 2048;
  // whole 64-row chunks as 16-byte loads when every chunk is a 16-byte aligned span whose length is a multiple of 4 floats
  const int vec = (reinterpret_cast<uintptr_t>(x) & 15) == 0 && ((64 * ld) % 4) == 0 && ((rows * ld) % 4) == 0 && (ld % 4 == 0 || ld == cols);
  hipLaunchKernelGGL(last_nonzero_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, x, rows, cols, ld, unit, top, vec);
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

// ---------------------------------------------------------------------------------------------
// rows by index: dst[plane][p][:] = src[plane][idx[p]][:] (gather) / dst[plane][idx[p]][:] = src[plane][p][:] (scatter), rows of `w4`
// 4-byte words (16-byte pieces when w4 % 4 == 0 and everything is aligned).  The decoder's rows in length-sorted order (round 6:
// per-row dead work): time states and fed tokens are gathered into it, the gradients of both scattered back.
// ---------------------------------------------------------------------------------------------
template <bool SCATTER, bool VEC>
__global__ void rows_by_index_kernel(unsigned* __restrict__ dst, const unsigned* __restrict__ src, const int* __restrict__ idx, long rows, int w4,
                                     long src_plane, long dst_plane, int planes, const int* __restrict__ seg) {
  const int per = VEC ? w4 / 4 : w4;                                     // pieces per row
  const long total = rows * per * planes;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % per);
    const long q = i / per;
    const long p = q % rows;
    const int pl = (int)(q / rows);
    const long r = idx[p];
    const long so = (long)pl * src_plane + (SCATTER ? p : r) * w4, dof = (long)pl * dst_plane + (SCATTER ? r : p) * w4;
    // row segments (ptv_gather_rows_seg / ptv_scatter_rows_seg): of plane pl only the first seg[pl] SORTED rows hold anything -- a gather
    // leaves the others unwritten (nobody reads them), a scatter writes zeros without reading the source
    if (seg && p >= seg[pl]) {
      if (SCATTER) { if (VEC) reinterpret_cast<uint4*>(dst + dof)[c] = make_uint4(0u, 0u, 0u, 0u); else dst[dof + c] = 0u; }
      continue;
    }
    if (VEC) reinterpret_cast<uint4*>(dst + dof)[c] = reinterpret_cast<const uint4*>(src + so)[c];
    else dst[dof + c] = src[so + c];
  }
}
static int rows_by_index(bool scatter, void* dst, const void* src, const int* idx, long rows, int w4, long src_plane, long dst_plane, int planes, const int* seg,
                         void* stream) {
  if (!dst || !src || !idx || rows < 0 || w4 <= 0 || planes <= 0) return PTV_ERR_ARG;
  if (rows == 0) return PTV_OK;
  const bool vec = (w4 % 4 == 0) && (src_plane % 4 == 0) && (dst_plane % 4 == 0) && ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & 15) == 0;
  const long total = rows * (vec ? w4 / 4 : w4) * planes;
  int nb = (int)((total + 255) / 256); if (nb > 8192) nb = 8192; if (nb < 1) nb = 1;
  unsigned* d_ = (unsigned*)dst; const unsigned* s_ = (const unsigned*)src;
  if (scatter) { if (vec) hipLaunchKernelGGL((rows_by_index_kernel<true, true>), dim3(nb), dim3(256), 0, (hipStream_t)stream, d_, s_, idx, rows, w4, src_plane, dst_plane, planes, seg);
                 else hipLaunchKernelGGL((rows_by_index_kernel<true, false>), dim3(nb), dim3(256), 0, (hipStream_t)stream, d_, s_, idx, rows, w4, src_plane, dst_plane, planes, seg); }
  else { if (vec) hipLaunchKernelGGL((rows_by_index_kernel<false, true>), dim3(nb), dim3(256), 0, (hipStream_t)stream, d_, s_, idx, rows, w4, src_plane, dst_plane, planes, seg);
         else hipLaunchKernelGGL((rows_by_index_kernel<false, false>), dim3(nb), dim3(256), 0, (hipStream_t)stream, d_, s_, idx, rows, w4, src_plane, dst_plane, planes, seg); }
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}
// live prefix of every note step when the rows run in descending length order and deadness goes by 128-row blocks (include/ptvae_hip.h)
__global__ void rows_seg_counts_kernel(const int* __restrict__ row_len, int nblk, int steps, int* __restrict__ seg_n) {
  __shared__ int cnt[64];
  for (int i = threadIdx.x; i < steps; i += blockDim.x) cnt[i] = 0;
  __syncthreads();
  for (int b = threadIdx.x; b < nblk; b += blockDim.x) {
    const int len = row_len[(long)b * 128];
    for (int s_ = 0; s_ < steps && s_ < len; s_++) atomicAdd(&cnt[s_], 128);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < steps; i += blockDim.x) seg_n[i] = cnt[i];
}
extern "C" int ptv_rows_seg_counts(const int* row_len, long R, int steps, int* seg_n, void* stream) {
  if (!row_len || !seg_n || R <= 0 || (R & 127) || steps <= 0 || steps > 64) return PTV_ERR_ARG;
  hipLaunchKernelGGL(rows_seg_counts_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, row_len, (int)(R >> 7), steps, seg_n);
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

extern "C" int ptv_gather_rows(void* dst, const void* src, const int* idx, long rows, int row_words, long src_plane_words, long dst_plane_words,
                               int planes, void* stream) {
  return rows_by_index(false, dst, src, idx, rows, row_words, src_plane_words, dst_plane_words, planes, nullptr, stream);
}
extern "C" int ptv_gather_rows_seg(void* dst, const void* src, const int* idx, long rows, int row_words, long src_plane_words, long dst_plane_words,
                                   int planes, const int* seg_n, void* stream) {
  return rows_by_index(false, dst, src, idx, rows, row_words, src_plane_words, dst_plane_words, planes, seg_n, stream);
}
extern "C" int ptv_scatter_rows_seg(void* dst, const void* src, const int* idx, long rows, int row_words, long src_plane_words, long dst_plane_words,
                                    int planes, const int* seg_n, void* stream) {
  return rows_by_index(true, dst, src, idx, rows, row_words, src_plane_words, dst_plane_words, planes, seg_n, stream);
}
extern "C" int ptv_scatter_rows(void* dst, const void* src, const int* idx, long rows, int row_words, long src_plane_words, long dst_plane_words,
                                int planes, void* stream) {
  return rows_by_index(true, dst, src, idx, rows, row_words, src_plane_words, dst_plane_words, planes, nullptr, stream);
}

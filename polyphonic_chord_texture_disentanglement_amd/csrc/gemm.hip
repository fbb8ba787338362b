// gemm.hip -- plain dense products on the MFMA core: Linear forward (NT), dX (NN), dW (TN, split-K).
// C-ABI entry: ptv_gemm (include/ptvae_hip.h).
#include <stdlib.h>
#include <mutex>
#include <type_traits>
#include "common.hpp"
#include "gemm_core.hpp"
#include "../../include/ptvae_hip.h"

namespace ptv {

// C[m,n] = act(alpha*acc + bias[n]) (+ C[m,n] when accumulate); atomic adds when split-K
struct EpiPlain {
  struct Params {
    void* C; long ldc;
    const float* bias;
    float alpha;
    int accumulate;   // C += ...
    int act;          // 0 none, 1 exp
    int atomic;       // split-K partial sums: atomicAdd into C (C pre-initialised, fp32 only)
    int c_bf16;       // C holds bf16
    int c_blocked;    // C is COLUMN-BLOCKED by w = 32 or 16 (log2 here, 0 = row-major): element (m, n) at ((n / w) * c_rows + m) * w + n % w (ldc unused) -- the layout the
    int c_rows;       // row-partitioned recurrences read their per-row operands in (one contiguous kilobyte per wave access)
    long split_stride;   // ordered split-K: split z stores its partial at C + z * split_stride floats (C = a workspace, ldc = N);
                         // splitk_reduce_kernel adds the partials in split order
  };
  static __device__ __forceinline__ long coff(const Params& p, int m, int n) {
    return p.c_blocked ? (((long)(n >> p.c_blocked) * p.c_rows + m) << p.c_blocked) + (n & ((1 << p.c_blocked) - 1)) : (long)m * p.ldc + n;
  }
  // a row tile that lies in the declared-zero part of A (GemmArgs::m_top): accumulating or atomically adding zero changes nothing
  static __device__ __forceinline__ bool dead_is_noop(const Params& p) { return (p.accumulate || p.atomic) && p.act == 0 && p.bias == nullptr; }
  template <int FM, int FN, int NG> struct Pre {};
  template <int FM, int FN, int NG>
  static __device__ __forceinline__ void prefetch(const Params&, Pre<FM, FN, NG>&, int, int, int, int) {}
  // wave tiles 32 / 64 columns wide: accumulators go through RowStage so C is written in whole row runs
  template <int FN> static constexpr bool staged() { return FN == 2 || FN == 4; }
  template <int FM, int FN, int NG> static constexpr int lds_bytes() { return staged<FN>() ? 4 * RowStage<1, (staged<FN>() ? FN * 16 : 64)>::WAVE_BYTES : 0; }
  template <int FM, int FN, int NG>
  static __device__ __forceinline__ void apply(const Params& p_in, f32x4 (&acc)[FM][NG * FN], const Pre<FM, FN, NG>&,
                                               int m0, int n0, int M, int N, int split, char* lds) {
    Params q = p_in;
    if (q.split_stride) q.C = reinterpret_cast<float*>(q.C) + (long)split * q.split_stride;
    const Params& p = q;
    const bool bf = p.c_bf16 != 0;
    const bool vec = (p.c_blocked || (p.ldc & 3) == 0) && ((reinterpret_cast<uintptr_t>(p.C) & (bf ? 7 : 15)) == 0);
    const bool use_bias = p.bias != nullptr && split == 0;
    if constexpr (staged<FN>()) {
      using RS = RowStage<1, FN * 16>;
      float* st = RS::base(lds);
      const int u = RS::unit();
      // interior wave tiles with aligned C: one of four straight-line epilogues (the generic cell below carries
      // every runtime option and costs ~100 instructions per cell -- for short-K products that, not memory,
      // set the block lifetime)
      const bool bias_ok = !use_bias || (reinterpret_cast<uintptr_t>(p.bias) & 15) == 0;
      if (vec && bias_ok && p.act == 0 && m0 + 16 * FM <= M && n0 + 16 * FN <= N && !(bf && p.atomic)) {
        const int mode = p.atomic ? 3 : (p.accumulate ? (bf ? 4 : 2) : (bf ? 1 : 0));
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (use_bias) b4 = *reinterpret_cast<const float4*>(p.bias + n0 + u);
        if (mode == 0) fast_rows<FM, FN, 0>(p, acc, st, m0, n0 + u, b4);
        else if (mode == 1) fast_rows<FM, FN, 1>(p, acc, st, m0, n0 + u, b4);
        else if (mode == 2) fast_rows<FM, FN, 2>(p, acc, st, m0, n0 + u, b4);
        else if (mode == 4) fast_rows<FM, FN, 4>(p, acc, st, m0, n0 + u, b4);
        else fast_rows<FM, FN, 3>(p, acc, st, m0, n0 + u, b4);
        return;
      }
#pragma unroll
      for (int i = 0; i < FM; i++) {
        RS::put(st, acc[i]);
#pragma unroll
        for (int c = 0; c < RS::PASSES; c++) {
          const int row = RS::row(c), m = m0 + i * 16 + row, n = n0 + u;
          if (m < M && n < N) cell(p, m, n, N, RS::get(st, row, u), bf, vec, use_bias);
        }
        __builtin_amdgcn_wave_barrier();
      }
    } else {
      const int lane = threadIdx.x & 63;
#pragma unroll
      for (int i = 0; i < FM; i++)
#pragma unroll
        for (int j = 0; j < FN; j++) {
          const int m = m0 + i * 16 + (lane & 15), n = n0 + j * 16 + (lane >> 4) * 4;
          if (m < M && n < N) cell(p, m, n, N, acc[i][j], bf, vec, use_bias);
        }
    }
  }
  // MODE 0: store fp32, 1: store bf16, 2: C += (fp32), 3: atomicAdd (fp32 split-K partials), 4: C += (bf16)
  template <int FM, int FN, int MODE>
  static __device__ __forceinline__ void fast_rows(const Params& p, f32x4 (&acc)[FM][FN], float* st, int m0, int n, const float4& b4) {
    using RS = RowStage<1, FN * 16>;
    const int u = RS::unit();
#pragma unroll
    for (int i = 0; i < FM; i++) {
      RS::put(st, acc[i]);
#pragma unroll
      for (int c = 0; c < RS::PASSES; c++) {
        const int row = RS::row(c);
        const f32x4 a = RS::get(st, row, u);
        const long off = coff(p, m0 + i * 16 + row, n);
        float v0 = p.alpha * a[0] + b4.x, v1 = p.alpha * a[1] + b4.y, v2 = p.alpha * a[2] + b4.z, v3 = p.alpha * a[3] + b4.w;
        if constexpr (MODE == 1 || MODE == 4) {
          if constexpr (MODE == 4) {
            const bf16x4 q = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(p.C) + off);
            v0 += (float)q[0]; v1 += (float)q[1]; v2 += (float)q[2]; v3 += (float)q[3];
          }
          bf16x4 o; o[0] = (__bf16)v0; o[1] = (__bf16)v1; o[2] = (__bf16)v2; o[3] = (__bf16)v3;
          *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(p.C) + off) = o;
        } else if constexpr (MODE == 3) {
          float* cp = reinterpret_cast<float*>(p.C) + off;
          atomicAdd(cp, v0); atomicAdd(cp + 1, v1); atomicAdd(cp + 2, v2); atomicAdd(cp + 3, v3);
        } else {
          float4* cp = reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + off);
          if constexpr (MODE == 2) { const float4 q = *cp; v0 += q.x; v1 += q.y; v2 += q.z; v3 += q.w; }
          *cp = make_float4(v0, v1, v2, v3);
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  static __device__ __forceinline__ void cell(const Params& p, int m, int n, int N, const f32x4& a, bool bf, bool vec, bool use_bias) {
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
      v[e] = p.alpha * a[e];
      if (use_bias && n + e < N) v[e] += p.bias[n + e];
      if (p.act == 1) v[e] = expf(v[e]);
    }
    const long off = coff(p, m, n);
    if (p.atomic) {
      float* c = reinterpret_cast<float*>(p.C) + off;
#pragma unroll
      for (int e = 0; e < 4; e++) if (n + e < N) atomicAdd(c + e, v[e]);
    } else if (vec && n + 3 < N) {
      if (p.accumulate) { const float4 q = ld4f(p.C, off, bf); v[0] += q.x; v[1] += q.y; v[2] += q.z; v[3] += q.w; }
      st4f(p.C, off, bf, v[0], v[1], v[2], v[3]);
    } else {
#pragma unroll
      for (int e = 0; e < 4; e++) if (n + e < N) st1f(p.C, off + e, bf, p.accumulate ? ld1f(p.C, off + e, bf) + v[e] : v[e]);
    }
  }
};

#ifndef PTV_PF_TN
#define PTV_PF_TN 3
#endif
#ifndef PTV_PF_NT
#define PTV_PF_NT 2
#endif
// register prefetch depth of the 128x128 tile: bf16 sources stage 16 B per thread per k row, cheap enough to go deeper
#define PLAIN_PF_BIG(KA, KB, SA, SB) ((SA) && (SB) ? ((KA) ? PTV_PF_TN : PTV_PF_NT) : 1)

template <class CT, int BM, int BN, int WGM, int WGN, bool KA, bool KB, bool SA, bool SB>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_plain_kernel(GemmArgs g, EpiPlain::Params ep) {
  if (g.prio) __builtin_amdgcn_s_setprio(3);
  gemm_body<CT, BM, BN, WGM, WGN, 1, KA, KB, EpiPlain, SA, SB, (BM * BN <= 64 * 64 ? 2 : PLAIN_PF_BIG(KA, KB, SA, SB))>(g, ep);
}

// C[m][n] (+)= ws[0][m][n] + ws[1][m][n] + ... in split order (alpha and the bias are already inside the partials)
__global__ void splitk_reduce_kernel(float* __restrict__ C, long ldc, const float* __restrict__ ws, int M, int N, int splits, int accumulate) {
  const long total = (long)M * N;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    float sum = 0.f;
    for (int z = 0; z < splits; z++) sum += ws[(long)z * total + i];
    float* cp = C + (i / N) * ldc + (i % N);
    *cp = (accumulate ? *cp : 0.f) + sum;
  }
}

__global__ void fill_rows_kernel(float* C, long ldc, int M, int N, float v) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  long total = (long)M * N;
  for (; i < total; i += (long)gridDim.x * blockDim.x) C[(i / N) * ldc + (i % N)] = v;
}

template <class CT, int BM, int BN, bool KA, bool KB, bool SA, bool SB>
static void launch_plain(const GemmArgs& g, const EpiPlain::Params& ep, int splits, hipStream_t s) {
  dim3 grid(cdiv(g.N, BN), cdiv(g.M, BM), splits);
  hipLaunchKernelGGL((gemm_plain_kernel<CT, BM, BN, 2, 2, KA, KB, SA, SB>), grid, dim3(NTHREADS), 0, s, g, ep);
}

// 1 (default): split-K partials through a per-stream workspace + ordered reduction; 0: fp32 atomics into C (PTV_WGRAD_ORDERED=0,
// ptv_wgrad_mode)
int g_splitk_ordered = [] { const char* e = getenv("PTV_WGRAD_ORDERED"); return (e && e[0] == '0') ? 0 : 1; }();

static float* splitk_workspace(hipStream_t s, size_t bytes) {
  struct Buf { float* p; size_t bytes; };
  static Buf pool[64]; static hipStream_t keys[64]; static int n = 0; static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  int i = 0;
  for (; i < n; i++) if (keys[i] == s) break;
  if (i == n) { if (n == 64) return nullptr; keys[n++] = s; pool[i] = Buf{nullptr, 0}; }
  Buf& b = pool[i];
  if (b.bytes < bytes) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) == hipSuccess && st != hipStreamCaptureStatusNone) return nullptr;   // cannot grow inside a capture
    // (the outgrown buffer is not freed: a captured hipGraph may hold its address)
    const size_t want = bytes + bytes / 4;
    if (hipMalloc(reinterpret_cast<void**>(&b.p), want) != hipSuccess) { b = Buf{nullptr, 0}; return nullptr; }
    b.bytes = want;
  }
  return b.p;
}

template <class CT, bool SA, bool SB>
static int gemm_dispatch(int transA, int transB, GemmArgs g, EpiPlain::Params ep, int splitk, hipStream_t s) {
  // tile choice: 128x128 when it still yields >= ~1 block per CU, else 64x64.  Weight-gradient
  // products (transA: K = rows x steps is huge, M x N small) fill the chip with K splits; the
  // thresholds / block targets below are the measured optima of a round-1 sweep on MI355X (scripts/bench_wgrad.py covers the same shapes).
  const long blocks_big = (long)cdiv(g.M, 128) * cdiv(g.N, 128);
  const bool deepk = transA && g.K >= 4096;
  const bool big = blocks_big >= 192 || (deepk && g.M >= 256 && g.N >= 256);
  const int bm = big ? 128 : 64;
  long blocks = (long)cdiv(g.M, bm) * cdiv(g.N, bm);
  int splits = 1;
  if (splitk > 0) splits = splitk;
  else if (splitk == 0 && blocks < 256 && g.K >= 8 * CT::BK && ep.act == 0 && !ep.c_bf16) {
    const long target = deepk ? (big ? 640 : 1536) : 512;
    splits = (int)((target + blocks - 1) / blocks);
    int maxs = g.K / (4 * CT::BK);
    if (splits > maxs) splits = maxs;
    if (splits > 192) splits = 192;
    if (splits < 1) splits = 1;
  }
  int kper = g.K;
  if (splits > 1) {
    if (splits >= 8) splits = (splits + 7) / 8 * 8;            // whole splits per XCD (gemm_body's block map)
    kper = cdiv(cdiv(g.K, splits), CT::BK) * CT::BK;
    const int s2 = cdiv(g.K, kper);
    if (s2 != splits && (s2 & 7) != 0 && s2 >= 8) {            // keep a multiple of 8 after rounding kper up
      splits = s2 / 8 * 8;
      kper = cdiv(cdiv(g.K, splits), CT::BK) * CT::BK;
      splits = cdiv(g.K, kper);
    } else splits = s2;
  }
  g.k_per_split = kper;
  float* ws = nullptr;
  const EpiPlain::Params ep_user = ep;
  if (splits > 1 && g_splitk_ordered) {
    // ordered reduction: every split stores its partial tile-by-tile into a per-stream workspace, one more launch adds them in
    // split order -- bit-reproducible, no fp32 atomics.  (Dead row tiles of an m_top product store zeros.)
    ws = splitk_workspace(s, (size_t)splits * g.M * g.N * sizeof(float));
    if (!ws) g_ord_fallbacks++;
  }
  if (ws) {
    ep.C = ws; ep.ldc = g.N; ep.accumulate = 0; ep.atomic = 0; ep.split_stride = (long)g.M * g.N;
  } else if (splits > 1) {
    ep.atomic = 1;
    if (!ep.accumulate) {
      long total = (long)g.M * g.N;
      int nb = (int)((total + 255) / 256); if (nb > 2048) nb = 2048;
      hipLaunchKernelGGL(fill_rows_kernel, dim3(nb), dim3(256), 0, s, reinterpret_cast<float*>(ep.C), ep.ldc, g.M, g.N, 0.f);
    }
  }
#define PTV_LAUNCH(KA, KB)                                              \
  do {                                                                  \
    if (big) launch_plain<CT, 128, 128, KA, KB, SA, SB>(g, ep, splits, s);      \
    else launch_plain<CT, 64, 64, KA, KB, SA, SB>(g, ep, splits, s);            \
  } while (0)
  if (!transA && !transB) PTV_LAUNCH(false, false);
  else if (!transA && transB) PTV_LAUNCH(false, true);
  else if (transA && transB) PTV_LAUNCH(true, true);
  else PTV_LAUNCH(true, false);
#undef PTV_LAUNCH
  if (ws) {
    const long total = (long)g.M * g.N;
    int nb = (int)((total + 255) / 256); if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(nb), dim3(256), 0, s, reinterpret_cast<float*>(ep_user.C), ep_user.ldc, ws, g.M, g.N, splits, ep_user.accumulate);
  }
  return PTV_OK;
}

}  // namespace ptv

namespace ptv { int g_gemm_prio = 0; }
// products launched from now on raise their wave priority (p != 0) / run at the default priority (0): the host marks the launches of
// its latency chain so that they win instruction issue against the weight-gradient products on sibling streams
extern "C" int ptv_gemm_priority(int p) { ptv::g_gemm_prio = p ? 1 : 0; return PTV_OK; }

extern "C" int ptv_gemm_mtop(int prec, int transA, int transB, int M, int N, int K,
                             const void* A, long lda, const void* B, long ldb,
                             void* C, long ldc, const float* bias, float alpha,
                             int accumulate, int act, int splitk, int dtypes, const int* m_top, long m_unit, void* stream) {
  return ptv_gemm_mtop_seg(prec, transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, alpha, accumulate, act, splitk, dtypes, m_top, m_unit, nullptr, 0, 0,
                           stream);
}

extern "C" int ptv_gemm_mtop_seg(int prec, int transA, int transB, int M, int N, int K,
                                 const void* A, long lda, const void* B, long ldb,
                                 void* C, long ldc, const float* bias, float alpha,
                                 int accumulate, int act, int splitk, int dtypes, const int* m_top, long m_unit,
                                 const int* seg_n, long seg_unit, int seg_period, void* stream) {
  if (M < 0 || N < 0 || K < 0 || !A || !B || !C) return PTV_ERR_ARG;
  if (M == 0 || N == 0) return PTV_OK;
  const bool sa = dtypes & 1, sb = dtypes & 2, sc = dtypes & 4;
  if ((sa || sb) && prec != PTV_PREC_BF16) return PTV_ERR_ARG;       // bf16 operands feed the bf16 MFMA path only
  if (sc && splitk > 1) return PTV_ERR_ARG;                          // split-K accumulates with fp32 atomics
  if ((dtypes & 8) && (dtypes & 16)) return PTV_ERR_ARG;
  if ((dtypes & 8) && ((N & 31) || splitk > 1)) return PTV_ERR_ARG;  // column-blocked C: whole 32- / 16-column blocks, one writer per element
  if ((dtypes & 16) && ((N & 15) || splitk > 1)) return PTV_ERR_ARG;
  if ((dtypes & 24) && splitk == 0) splitk = 1;
  // weight gradients (both operands row-per-sample): the transposing-LDS-read kernel of wgrad.hip
  if (prec == PTV_PREC_BF16 && transA && transB && !sc && !(dtypes & 24) && !bias && act == 0 && K >= 512 && splitk <= 0)
    return ptv_wgrad(M, N, K, A, lda, B, ldb, reinterpret_cast<float*>(C), ldc, alpha, accumulate, dtypes & 3, 0, nullptr, nullptr, 0, 0, stream);
  if (m_top && (transA || m_unit <= 0)) return PTV_ERR_ARG;            // a row limit on A: A must be row-per-sample
  if (seg_n && (transA || seg_unit <= 0 || (seg_unit & 127) || seg_period <= 0)) return PTV_ERR_ARG;   // (row tiles are 64 or 128 rows: whole tiles dead or live)
  ptv::GemmArgs g{A, lda, B, ldb, M, N, K, K, 0, m_top, m_unit, ptv::g_gemm_prio, seg_n, (int)seg_unit, seg_period};
  ptv::EpiPlain::Params ep{C, ldc, bias, alpha, accumulate, act, 0, sc ? 1 : 0, (dtypes & 8) ? 5 : ((dtypes & 16) ? 4 : 0), M, 0};
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if (prec != PTV_PREC_BF16) rc = ptv::gemm_dispatch<ptv::F32, false, false>(transA, transB, g, ep, splitk, s);
  else if (sa && sb) rc = ptv::gemm_dispatch<ptv::BF16, true, true>(transA, transB, g, ep, splitk, s);
  else if (sa) rc = ptv::gemm_dispatch<ptv::BF16, true, false>(transA, transB, g, ep, splitk, s);
  else if (sb) rc = ptv::gemm_dispatch<ptv::BF16, false, true>(transA, transB, g, ep, splitk, s);
  else rc = ptv::gemm_dispatch<ptv::BF16, false, false>(transA, transB, g, ep, splitk, s);
  if (rc != PTV_OK) return rc;
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

extern "C" int ptv_gemm(int prec, int transA, int transB, int M, int N, int K,
                        const void* A, long lda, const void* B, long ldb,
                        void* C, long ldc, const float* bias, float alpha,
                        int accumulate, int act, int splitk, int dtypes, void* stream) {
  return ptv_gemm_mtop(prec, transA, transB, M, N, K, A, lda, B, ldb, C, ldc, bias, alpha, accumulate, act, splitk, dtypes, nullptr, 0, stream);
}

// wgrad.hip -- weight-gradient products dW[M,N] += alpha * A^T B on the MFMA core, A [K,M] and B [K,N] both K-major (rows = the
// 32*B*T samples of the batch, thousands to hundreds of thousands of them; M, N = a layer's output / input widths).
//
// These are what autograd computes for every nn.Linear / nn.GRU weight of the reference (ptvae.py:16-17,23,64,116,360,396,450,
// 461): grad_W = grad_out^T . input.  Both operands arrive row-per-sample, i.e. k-strided for an MFMA fragment.  The generic
// GEMM (gemm_core.hpp) transposes them in registers on the way into LDS (eight 8-byte LDS stores per thread per tile); here the
// tile goes into LDS exactly as it lies in HBM ([k][columns], 16-byte copies) and the fragments come out through gfx950's
// transposing LDS read (ds_read_b64_tr_b16: a 16-lane group reads a [4 k][16 columns] block, lane i receives column i).
//
//   block  = 128 x 128 outputs, 4 waves as 2 x 2 (64 x 64 each: 4 x 4 accumulator fragments), one slab of K
//   stage  = 32 rows of A and of B (2 x 9 KB with the row padding); two LDS buffers, two register sets of prefetch
//   k order inside a 32-row stage: lane group g takes rows 4g..4g+3 and 16+4g..16+4g+3 (the same permutation for both operands,
//           so the product is unchanged) -- the four groups of a wave then read 16 consecutive rows: conflict-free at the
//           288-byte row stride
//   slabs  reduce into C with fp32 atomics (every caller accumulates into a gradient buffer), blocks of one XCD share panels
//           through that XCD's L2 (tile ranges or whole slabs per XCD)
//
// fp32 sources are rounded to bf16 on the way into LDS (the bf16 precision mode's MFMA operands, as in gemm_core.hpp).
// Entry: ptv_wgrad (include/ptvae_hip.h); ptv_gemm routes its bf16 transA && transB calls here.
#include <stdlib.h>
#include <mutex>
#include "common.hpp"
#include "prof.hpp"
#include "../../include/ptvae_hip.h"

namespace ptv {

extern int g_gemm_prio;            // gemm.hip: set by ptv_gemm_priority

typedef __bf16 wbf16x8 __attribute__((ext_vector_type(8)));
typedef short ws16x4 __attribute__((ext_vector_type(4)));
typedef short ws16x8 __attribute__((ext_vector_type(8)));
typedef float wf32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) ws16x4 lds_s16x4;

constexpr int WBM = 128, WBN = 128, WBK = 32, WLD = 144;       // tile, stage depth, LDS row stride in bf16 (288 B)
constexpr int WSTAGE = WBK * WLD;                              // elements of one operand's stage

struct WgArgs {
  const void* A; long lda;
  const void* B; long ldb;
  float* C; long ldc;
  int M, N, K, kper;             // kper: rows of K per slab (multiple of 32)
  int tiles_m, tiles_n, nslab;
  int map;                       // 0: slab = b / tiles; 1: whole slabs per XCD; 2: tile ranges per XCD
  float alpha;
  float* csum;                   // [M] += column sums of A (the bias gradient that goes with this weight gradient), or null
  const int* k_top; long k_unit;  // rows from (*k_top + 1) * k_unit on are known to be zero in A (written by the kernel that produced A), or null
  int k_rev;                     // > 0: A is stored in REVERSED unit order (k_rev units): the zero part is the rows BEFORE (k_rev - *k_top - 1) * k_unit
  int prio;                      // the launch belongs to a latency chain (ptv_gemm_priority): raised wave priority
  float* ws;                     // ordered reduction (ptv_wgrad_mode 1): slab partials go to ws[(slab0 + slab) * tiles + tile][128][128] with plain
  float* ws_csum;                // stores (ws_csum[(slab0 + slab) * M + m] for the column sums) and wgrad_reduce_kernel adds them up in slab
  int slab0;                     // order; null: fp32 atomics into C (run-to-run rounding differs)
  // second source of A (ptv_wgrad_cat): output rows m >= split (a multiple of 128) are the columns m - split of A2 -- two gradient
  // matrices that meet the same B (the notes GRU's dgi[:, :1024] and dgh: one pass over the states instead of two), or null
  const void* A2; long lda2; int split;
  int k_base;                    // this launch's first row in the numbering of the k_top limits (the guarded tail launch starts at kfast)
};

// which part of a slab's K range survives the k_top limits: shared by the product kernel and the ordered reduction (a slab that is
// empty writes no partial and must not be read)
__device__ __forceinline__ bool slab_range(const WgArgs& g, int slab, int& k_begin, int& k_end) {
  k_begin = slab * g.kper;
  int k_lim = g.K, k_from = 0;
  if (g.k_top) {                                                  // (multiples of the 32-row stage whenever k_unit is)
    if (g.k_rev > 0) {
      const long lo = ((long)g.k_rev - *g.k_top - 1) * g.k_unit - g.k_base;
      if (lo > 0) k_from = (int)min((long)g.K, lo);
    } else {
      const long lim = max(((long)*g.k_top + 1) * g.k_unit - g.k_base, 0L);
      if (lim < k_lim) k_lim = (int)lim;
    }
  }
  if (k_begin >= k_lim || k_begin + g.kper <= k_from) return false;
  k_end = min(k_lim, k_begin + g.kper);
  if (k_from > k_begin) k_begin = k_from;                         // (k_from is a multiple of 32 like every slab start)
  return true;
}

// one thread's share of a stage of one operand: 2 chunks of 8 columns (chunk c: row c / 16, columns (c % 16) * 8).
// Columns at or beyond ncols only ever meet outputs that are not stored, so a chunk that straddles ncols may be read whole as long as
// the bytes exist (every row but the matrix's last: the overread lands in the next row); the element-wise path serves that last
// row and operands whose rows are not 16-byte aligned.
template <bool F32SRC> struct WSrc { using T = __bf16; };
template <> struct WSrc<true> { using T = float; };

template <bool F32SRC>
__device__ __attribute__((noinline)) wbf16x8 wload_slow(const void* p, long off, int nvalid) {
  const typename WSrc<F32SRC>::T* q = reinterpret_cast<const typename WSrc<F32SRC>::T*>(p) + off;
  wbf16x8 x;
#pragma unroll
  for (int e = 0; e < 8; e++) x[e] = e < nvalid ? (__bf16)q[e] : (__bf16)0.f;
  return x;
}

template <bool F32SRC> struct WStage;
template <> struct WStage<false> {
  wbf16x8 v[2];
  // no guards and no select on the loaded data (which would pin the wait for it right behind the load): 16-byte aligned rows,
  // every row of the stage in range; a chunk at or beyond ncols reads column 0's bytes instead -- whatever it holds only meets
  // outputs that are never stored
  __device__ __forceinline__ void load_fast(const __bf16* p, long ld) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      v[i] = *reinterpret_cast<const wbf16x8*>(p + (long)(i * 16) * ld);
    }
  }
  __device__ __forceinline__ void load(const void* p, long ld, int k0, int kend, int klast, int col, int ncols, bool vec) {
    const __bf16* s = reinterpret_cast<const __bf16*>(p);
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int k = k0 + (threadIdx.x >> 4) + i * 16;
      wbf16x8 x;
#pragma unroll
      for (int e = 0; e < 8; e++) x[e] = (__bf16)0.f;
      if (k < kend && col < ncols) {
        if (vec && (col + 8 <= ncols || k < klast)) x = *reinterpret_cast<const wbf16x8*>(s + (long)k * ld + col);
        else x = wload_slow<false>(p, (long)k * ld + col, ncols - col);
      }
      v[i] = x;
    }
  }
  __device__ __forceinline__ void store(__bf16* st) const {
#pragma unroll
    for (int i = 0; i < 2; i++) *reinterpret_cast<wbf16x8*>(st + ((threadIdx.x >> 4) + i * 16) * WLD + (threadIdx.x & 15) * 8) = v[i];
  }
};
template <> struct WStage<true> {
  wf32x4 v[2][2];
  __device__ __forceinline__ void load_fast(const float* p, long ld) {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const float* q = p + (long)(i * 16) * ld;
      v[i][0] = *reinterpret_cast<const wf32x4*>(q); v[i][1] = *reinterpret_cast<const wf32x4*>(q + 4);
    }
  }
  __device__ __forceinline__ void load(const void* p, long ld, int k0, int kend, int klast, int col, int ncols, bool vec) {
    const float* s = reinterpret_cast<const float*>(p);
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const int k = k0 + (threadIdx.x >> 4) + i * 16;
      wf32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
      if (k < kend && col < ncols) {
        if (vec && (col + 8 <= ncols || k < klast)) {
          const float* q = s + (long)k * ld + col;
          lo = *reinterpret_cast<const wf32x4*>(q);
          hi = *reinterpret_cast<const wf32x4*>(q + 4);
        } else {
          const wbf16x8 x = wload_slow<true>(p, (long)k * ld + col, ncols - col);
#pragma unroll
          for (int e = 0; e < 4; e++) { lo[e] = (float)x[e]; hi[e] = (float)x[4 + e]; }
        }
      }
      v[i][0] = lo; v[i][1] = hi;
    }
  }
  __device__ __forceinline__ void store(__bf16* st) const {
#pragma unroll
    for (int i = 0; i < 2; i++) {
      wbf16x8 x;
#pragma unroll
      for (int e = 0; e < 4; e++) { x[e] = (__bf16)v[i][0][e]; x[4 + e] = (__bf16)v[i][1][e]; }
      *reinterpret_cast<wbf16x8*>(st + ((threadIdx.x >> 4) + i * 16) * WLD + (threadIdx.x & 15) * 8) = x;
    }
  }
};

// MFMA operand of 16 columns starting at `col`: lane (i = lane & 15, g = lane >> 4) receives rows 4g..4g+3 and 16+4g..16+4g+3 of
// column col + i.  Supplier lane s of a 16-lane group hands in the address of row s >> 2, columns 4 (s & 3) .. +3 of the block.
template <int LD>
__device__ __forceinline__ wbf16x8 tr_frag_ld(const __bf16* st, int col) {
  const int lane = threadIdx.x & 63, s = lane & 15, g = lane >> 4;
  const __bf16* p = st + (g * 4 + (s >> 2)) * LD + col + (s & 3) * 4;
  const ws16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
  const ws16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 16 * LD));
  const ws16x8 w = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(wbf16x8, w);
}
__device__ __forceinline__ wbf16x8 tr_frag(const __bf16* st, int col) { return tr_frag_ld<WLD>(st, col); }

// (Round 4's in-kernel fix-up -- the LAST block to deliver a partial of an output tile adds the tile's partials up in slab order, no separate
// reduction launch -- measured 9.26 ms per step against 8.16 with the 39 reduction launches (write-through hand-off: 64 four-byte sc1 stores
// per lane, and the last block of each tile reading S x 64 KB past the caches) and was removed in round 5.)

// what a block does with its finished 128 x 128 partial (shared by the register-staged and the LDS-DMA kernel)
__device__ __forceinline__ void wgrad_store(const WgArgs& g, wf32x4 (&acc)[4][4], wf32x4 (&accs)[4], bool do_sum, int slab, int tile, int tiles,
                                            int m_blk, int n_blk, int wm, int wn, int lane) {
  // acc[i][j]: lane holds C[m = wm + 16 i + 4 (lane >> 4) + r][n = wn + 16 j + (lane & 15)], r = 0..3: one atomic instruction of the
  // wave covers 4 rows x 16 consecutive columns (4 cache lines; the other operand order would touch 16)
  if (do_sum && (lane & 15) == 0) {                              // every column of accs holds the same sums
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int m = m_blk + wm + i * 16 + (lane >> 4) * 4 + r;
        if (m < g.M) {
          if (g.ws_csum) __hip_atomic_store(g.ws_csum + (long)(g.slab0 + slab) * g.M + m, accs[i][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          else atomicAdd(g.csum + m, accs[i][r]);
        }
      }
  }
  if (g.ws) {                                                    // ordered reduction: this slab's partial tile, plain stores
    float* wt = g.ws + ((long)(g.slab0 + slab) * tiles + tile) * (WBM * WBN);
    const int mrem = g.M - m_blk - wm - (lane >> 4) * 4, nrem = g.N - n_blk - wn - (lane & 15);      // rows / columns of C left from this lane's first cell
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int r = 0; r < 4; r++)
          if (i * 16 + r < mrem && j * 16 < nrem) {                // (cells outside C are never read back)
            float* q = wt + (wm + i * 16 + (lane >> 4) * 4 + r) * WBN + wn + j * 16 + (lane & 15);
            *q = acc[i][j][r];
          }
    return;
  }
  const bool single = g.nslab == 1;
  const bool inner = m_blk + WBM <= g.M && n_blk + WBN <= g.N;
#pragma unroll
  for (int i = 0; i < 4; i++) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int n = n_blk + wn + j * 16 + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int m = m_blk + wm + i * 16 + (lane >> 4) * 4 + r;
        if (!inner && (m >= g.M || n >= g.N)) continue;
        float* cp = g.C + (long)m * g.ldc + n;
        if (single) *cp += g.alpha * acc[i][j][r];              // the only writer of this element: plain read-modify-write
        else atomicAdd(cp, g.alpha * acc[i][j][r]);
      }
    }
  }
}

// GUARD = false: rows 16-byte aligned, K a multiple of 32 and not reaching the operands' last row (see ptv_wgrad)
template <bool AF32, bool BF32, bool GUARD, int NSET>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(WgArgs g) {
  if (g.prio) __builtin_amdgcn_s_setprio(3);
  __shared__ __attribute__((aligned(16))) __bf16 As[2 * WSTAGE];
  __shared__ __attribute__((aligned(16))) __bf16 Bs[2 * WSTAGE];
  const int tiles = g.tiles_m * g.tiles_n;
  int b = blockIdx.x, slab, tile;
  if (g.map == 1) { const int q = b >> 3; slab = (q / tiles) * 8 + (b & 7); tile = q % tiles; }
  else if (g.map == 2) { const int tx = tiles >> 3, q = b >> 3; tile = (b & 7) * tx + q % tx; slab = q / tx; }
  else { slab = b / tiles; tile = b % tiles; }
  int k_begin, k_end;
  if (!slab_range(g, slab, k_begin, k_end)) return;
  const int m_blk = (tile / g.tiles_n) * WBM, n_blk = (tile % g.tiles_n) * WBN;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const bool second = g.A2 != nullptr && m_blk >= g.split;
  const void* const srcA = second ? g.A2 : g.A;
  const long lda_ = second ? g.lda2 : g.lda;
  const int ma0 = second ? m_blk - g.split : m_blk;               // first column of this tile row inside its source
  const int Ma = g.A2 ? (second ? g.M - g.split : g.split) : g.M;  // columns of that source
  const bool veca = (lda_ % (AF32 ? 4 : 8) == 0) && ((reinterpret_cast<uintptr_t>(srcA) & 15) == 0);
  const bool vecb = (g.ldb % (BF32 ? 4 : 8) == 0) && ((reinterpret_cast<uintptr_t>(g.B) & 15) == 0);

  wf32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = wf32x4{0.f, 0.f, 0.f, 0.f};

  // bias gradient: sum_k A[k][m] = A^T . 1 -- four more MFMAs per stage against a fragment of ones, in the waves that own the first
  // 64 columns of the first column tile; the A tile is already in LDS, a separate column-sum kernel would read A from HBM again
  const bool do_sum = g.csum != nullptr && n_blk == 0 && wn == 0;
  wf32x4 accs[4];
#pragma unroll
  for (int i = 0; i < 4; i++) accs[i] = wf32x4{0.f, 0.f, 0.f, 0.f};
  wbf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; e++) ones[e] = (__bf16)1.0f;
  const int nst = (k_end - k_begin + WBK - 1) / WBK;
  // NSET register sets of prefetch: stage t lives in set t % NSET until iteration t-1 moves it into LDS buffer t & 1; its loads
  // were issued NSET iterations before that.  bf16 sources cost 4 VGPRs per chunk, fp32 sources 8.
  WStage<AF32> ra[NSET];
  WStage<BF32> rb[NSET];
  const int cola = ma0 + (threadIdx.x & 15) * 8, colb = n_blk + (threadIdx.x & 15) * 8;
  const bool coka = cola < Ma, cokb = colb < g.N;
  const typename WSrc<AF32>::T* pa = reinterpret_cast<const typename WSrc<AF32>::T*>(srcA) + (long)(k_begin + (threadIdx.x >> 4)) * lda_ + (coka ? cola : 0);
  const typename WSrc<BF32>::T* pb = reinterpret_cast<const typename WSrc<BF32>::T*>(g.B) + (long)(k_begin + (threadIdx.x >> 4)) * g.ldb + (cokb ? colb : 0);
#define WG_FETCH(set, t)                                                                            \
  do {                                                                                              \
    if constexpr (GUARD) {                                                                          \
      ra[set].load(srcA, lda_, k_begin + (t) * WBK, k_end, g.K - 1, cola, Ma, veca);                \
      rb[set].load(g.B, g.ldb, k_begin + (t) * WBK, k_end, g.K - 1, colb, g.N, vecb);               \
    } else {                                                                                        \
      ra[set].load_fast(pa + (long)(t) * WBK * lda_, lda_);                                         \
      rb[set].load_fast(pb + (long)(t) * WBK * g.ldb, g.ldb);                                       \
    }                                                                                               \
  } while (0)
  // every fetch is unconditional (the stage index is clamped to the slab's last stage; a redundant stage is never consumed): no
  // control flow between a load and the LDS store that consumes it, so the compiler's s_waitcnt vmcnt() are exact counts instead of
  // the vmcnt(0) it falls back to across branches -- which would serialise the whole prefetch
  const int last = nst - 1;
#pragma unroll
  for (int u = 0; u < NSET; u++) WG_FETCH(u, min(u, last));
  ra[0].store(As); rb[0].store(Bs);
  WG_FETCH(0, min(NSET, last));
  __syncthreads();
#define WG_BODY(u, t)                                                                                   \
  do {                                                                                                  \
    const __bf16* as = As + ((u) & 1) * WSTAGE;                                                         \
    const __bf16* bs = Bs + ((u) & 1) * WSTAGE;                                                         \
    wbf16x8 fa[4], fb[4];                                                                               \
    _Pragma("unroll") for (int i = 0; i < 4; i++) fa[i] = tr_frag(as, wm + i * 16);                     \
    _Pragma("unroll") for (int j = 0; j < 4; j++) fb[j] = tr_frag(bs, wn + j * 16);                     \
    /* the other buffer was last read before the barrier that ended the previous stage */              \
    constexpr int nu = ((u) + 1) % NSET;                                                                \
    ra[nu].store(As + (((u) + 1) & 1) * WSTAGE); rb[nu].store(Bs + (((u) + 1) & 1) * WSTAGE);           \
    WG_FETCH(nu, min((t) + 1 + NSET, last));                                                            \
    _Pragma("unroll") for (int i = 0; i < 4; i++)                                                       \
      _Pragma("unroll") for (int j = 0; j < 4; j++)                                                     \
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);          \
    if (do_sum) {                                                                                       \
      _Pragma("unroll") for (int i = 0; i < 4; i++)                                                     \
        accs[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], ones, accs[i], 0, 0, 0);               \
    }                                                                                                   \
    __syncthreads();                                                                                    \
  } while (0)
  int t0 = 0;
  for (; t0 + NSET <= nst; t0 += NSET) {
    WG_BODY(0, t0);
    WG_BODY(1, t0 + 1);
    if constexpr (NSET == 4) { WG_BODY(2, t0 + 2); WG_BODY(3, t0 + 3); }
  }
  if (t0 < nst) {
    WG_BODY(0, t0);
    if constexpr (NSET == 4) {
      if (t0 + 1 < nst) { WG_BODY(1, t0 + 1); if (t0 + 2 < nst) WG_BODY(2, t0 + 2); }
    }
  }
#undef WG_BODY
#undef WG_FETCH
  wgrad_store(g, acc, accs, do_sum, slab, tile, tiles, m_blk, n_blk, wm, wn, lane);
}

// ---------------------------------------------------------------------------------------------
// LDS-DMA variant (round 5) for bf16 x bf16 operands on the unguarded fast path: the 32-row stages go HBM -> LDS directly
// (global_load_lds_dwordx4: no staging registers, no ds_write pass), three stage buffers, two stages in flight across ONE raw barrier per
// stage with counted vmcnt -- the register-staged kernel waits for its prefetch at every __syncthreads.  The deep products of the step
// (1024 / 512 / 200 x 512 x 245760, 3072 x 1024 x 16384) sat at 0.27 of the MFMA peak and 0.3 of the CU's 64 B/clk load path: latency.
// A DMA instruction writes 64 lanes x 16 bytes CONTIGUOUSLY (four 256-byte rows), so rows cannot be padded; the 16-byte chunks of a row are
// XOR-swizzled instead -- on the SOURCE address, the LDS image stays linear -- such that the eight rows a 32-lane transposing read touches
// land in eight different 32-byte bank groups.
// ---------------------------------------------------------------------------------------------
constexpr int DSTAGE = WBK * WBM;                                // elements of one operand's stage: 32 rows x 128 columns, unpadded
constexpr int DNBUF = 3;
__device__ __forceinline__ int dma_swz(int row) { return ((row & 3) << 1) ^ (((row >> 2) & 1) << 3); }    // chunk index XOR of a row

// MFMA operand of 16 columns starting at `col` (a multiple of 16) out of a swizzled, unpadded stage: as tr_frag_ld
__device__ __forceinline__ wbf16x8 tr_frag_swz(const __bf16* st, int col) {
  const int lane = threadIdx.x & 63, s = lane & 15, g = lane >> 4;
  const int row = g * 4 + (s >> 2);
  const int chunk = ((col >> 3) + ((s & 3) >> 1)) ^ dma_swz(row);            // (row + 16 has the same swizzle)
  const __bf16* p = st + row * WBM + chunk * 8 + (s & 1) * 4;
  const ws16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
  const ws16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 16 * WBM));
  const ws16x8 w = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(wbf16x8, w);
}

__global__ __launch_bounds__(256, 2) void wgrad_dma_kernel(WgArgs g) {
  if (g.prio) __builtin_amdgcn_s_setprio(3);
  __shared__ __attribute__((aligned(1024))) __bf16 S[2 * DNBUF * DSTAGE];        // [buffer][A | B][32][128]: ONE array (a second LDS object makes
  const int tiles = g.tiles_m * g.tiles_n;                                        // hipcc wait vmcnt(0) in front of every ds_read)
  int b = blockIdx.x, slab, tile;
  if (g.map == 1) { const int q = b >> 3; slab = (q / tiles) * 8 + (b & 7); tile = q % tiles; }
  else if (g.map == 2) { const int tx = tiles >> 3, q = b >> 3; tile = (b & 7) * tx + q % tx; slab = q / tx; }
  else { slab = b / tiles; tile = b % tiles; }
  int k_begin, k_end;
  if (!slab_range(g, slab, k_begin, k_end)) return;
  const int m_blk = (tile / g.tiles_n) * WBM, n_blk = (tile % g.tiles_n) * WBN;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  const bool second = g.A2 != nullptr && m_blk >= g.split;
  const __bf16* const srcA = reinterpret_cast<const __bf16*>(second ? g.A2 : g.A);
  const long lda_ = second ? g.lda2 : g.lda;
  const int ma0 = second ? m_blk - g.split : m_blk;
  const int Ma = g.A2 ? (second ? g.M - g.split : g.split) : g.M;

  wf32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = wf32x4{0.f, 0.f, 0.f, 0.f};
  const bool do_sum = g.csum != nullptr && n_blk == 0 && wn == 0;
  wf32x4 accs[4];
#pragma unroll
  for (int i = 0; i < 4; i++) accs[i] = wf32x4{0.f, 0.f, 0.f, 0.f};
  wbf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; e++) ones[e] = (__bf16)1.0f;
  const int nst = (k_end - k_begin + WBK - 1) / WBK, last = nst - 1;

  // this wave's share of a stage: DMA instructions 2 wave, 2 wave + 1 of each operand (instruction j = rows 4j .. 4j+3).  Lane l lands at
  // LDS offset l * 16 of the instruction's kilobyte = (row 4j + l / 16, chunk position l % 16), and fetches the chunk that belongs there
  const __bf16* pa[2]; const __bf16* pb[2];
#pragma unroll
  for (int q = 0; q < 2; q++) {
    const int row = (2 * wave + q) * 4 + (lane >> 4), chunk = (lane & 15) ^ dma_swz(row);
    const int ca = ma0 + chunk * 8, cb = n_blk + chunk * 8;
    pa[q] = srcA + (long)(k_begin + row) * lda_ + (ca < Ma ? ca : 0);                      // (columns beyond M / N only meet outputs that are never stored)
    pb[q] = reinterpret_cast<const __bf16*>(g.B) + (long)(k_begin + row) * g.ldb + (cb < g.N ? cb : 0);
  }
  // (inline asm, not __builtin_amdgcn_global_load_lds: hipcc knows the builtin writes LDS and puts s_waitcnt vmcnt(0) in front of the
  // stage's first ds_read -- the very wait this kernel exists to avoid.  The asm is invisible to its counters: the waits are the explicit
  // ones below.  M0 carries the wave-uniform LDS address and is restored in the same statement.)
  typedef __attribute__((address_space(3))) const void* lptr_t;
  const unsigned lds0 = (unsigned)(uintptr_t)(lptr_t)S;
  auto glds = [&](const __bf16* src, unsigned dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
  };
  auto fetch = [&](int t) {                                                               // stage t -> buffer t % 3 (clamped: a redundant stage is never read)
    const int tt = t < last ? t : last;
    const unsigned buf = lds0 + (unsigned)((t % DNBUF) * (2 * DSTAGE) * 2);
#pragma unroll
    for (int q = 0; q < 2; q++) {
      glds(pa[q] + (long)tt * WBK * lda_, buf + (unsigned)((2 * wave + q) * 1024));
      glds(pb[q] + (long)tt * WBK * g.ldb, buf + (unsigned)(DSTAGE * 2 + (2 * wave + q) * 1024));
    }
  };
  fetch(0); fetch(1);
  for (int t = 0; t < nst; t++) {
    // stage t has landed for THIS wave's requests (two stages = 8 requests are outstanding at most; the 4 of stage t + 1 may stay) ...
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                  // ... and for everyone's; and every wave is done reading buffer (t - 1) % 3
    asm volatile("" ::: "memory");
    fetch(t + 2);                                                  // (into the buffer the barrier has just freed)
    const __bf16* as = S + (t % DNBUF) * (2 * DSTAGE);
    const __bf16* bs = as + DSTAGE;
    wbf16x8 fa[4], fb[4];
#pragma unroll
    for (int i = 0; i < 4; i++) fa[i] = tr_frag_swz(as, wm + i * 16);
#pragma unroll
    for (int j = 0; j < 4; j++) fb[j] = tr_frag_swz(bs, wn + j * 16);
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    if (do_sum) {
#pragma unroll
      for (int i = 0; i < 4; i++) accs[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], ones, accs[i], 0, 0, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // (the two redundant stages: nothing may land in LDS after the block has left)
  wgrad_store(g, acc, accs, do_sum, slab, tile, tiles, m_blk, n_blk, wm, wn, lane);
}

// (Round 4's 256 x 128 block tile -- two vertically adjacent output tiles per block sharing the B stage, 12 fragment reads per 32 MFMAs
// instead of 8 per 16 -- measured SLOWER and was removed in round 5: 1536 x 512 x 245760 516 us at its best slab count against 467, the
// step's 20 shapes 2306 against 1903 us; its 249 registers left two blocks per CU, and the third block was hiding more latency than the
// LDS relief bought.  profiles/r04_wgrad_xcd_map.txt.)

// ordered reduction of the slab partials: C[m][n] (+)= alpha * (p_0 + p_1 + ...) in slab order -- the same bits on every run.
// ga: the fast launch's arguments (slabs 0 .. ga.nslab-1), gb: the guarded tail launch (one slab, number ga.nslab), if has_b.
__global__ void wgrad_reduce_kernel(WgArgs ga, WgArgs gb, int has_a, int has_b, int accumulate) {
  const int tiles = ga.tiles_m * ga.tiles_n;
  // the live slabs of the fast launch are one contiguous range (the k_top limits cut K at one end): [s0, s1)
  int s0 = 0, s1 = 0;
  if (has_a) {
    bool any = false;
    for (int sl = 0; sl < ga.nslab; sl++) {
      int kb, ke;
      if (slab_range(ga, sl, kb, ke)) { if (!any) s0 = sl; s1 = sl + 1; any = true; }
    }
  }
  bool live_b = false;
  if (has_b) { int kb, ke; live_b = slab_range(gb, 0, kb, ke); }
  const long tstride = (long)tiles * (WBM * WBN);
  const float* wb = has_b ? gb.ws + (long)gb.slab0 * tstride : nullptr;
  const bool vec = (ga.N & 3) == 0 && (ga.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(ga.C) & 15) == 0;
  if (vec) {
    const int N4 = ga.N >> 2;
    const long total4 = (long)ga.M * N4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
      const int m = (int)(i / N4), n = (int)(i % N4) * 4;
      const long off = (long)((m / WBM) * ga.tiles_n + n / WBN) * (WBM * WBN) + (long)(m % WBM) * WBN + (n % WBN);
      const float* p = ga.ws + off;
      float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
      int sl = s0;
      for (; sl + 8 <= s1; sl += 8) {                               // eight partials in flight, added in slab order
        float4 v[8];
#pragma unroll
        for (int q = 0; q < 8; q++) v[q] = *reinterpret_cast<const float4*>(p + (long)(sl + q) * tstride);
#pragma unroll
        for (int q = 0; q < 8; q++) { sum.x += v[q].x; sum.y += v[q].y; sum.z += v[q].z; sum.w += v[q].w; }
      }
      for (; sl < s1; sl++) {
        const float4 v = *reinterpret_cast<const float4*>(p + (long)sl * tstride);
        sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
      }
      if (live_b) { const float4 v = *reinterpret_cast<const float4*>(wb + off); sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w; }
      float4* cp = reinterpret_cast<float4*>(ga.C + (long)m * ga.ldc + n);
      float4 c = accumulate ? *cp : make_float4(0.f, 0.f, 0.f, 0.f);
      c.x += ga.alpha * sum.x; c.y += ga.alpha * sum.y; c.z += ga.alpha * sum.z; c.w += ga.alpha * sum.w;
      *cp = c;
    }
  } else {
    const long total = (long)ga.M * ga.N;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
      const int m = (int)(i / ga.N), n = (int)(i % ga.N);
      const long off = (long)((m / WBM) * ga.tiles_n + n / WBN) * (WBM * WBN) + (long)(m % WBM) * WBN + (n % WBN);
      float sum = 0.f;
      const float* p = ga.ws + off;
      int sl = s0;
      for (; sl + 8 <= s1; sl += 8) {
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; q++) v[q] = p[(long)(sl + q) * tstride];
#pragma unroll
        for (int q = 0; q < 8; q++) sum += v[q];
      }
      for (; sl < s1; sl++) sum += p[(long)sl * tstride];
      if (live_b) sum += wb[off];
      float* cp = ga.C + (long)m * ga.ldc + n;
      *cp = (accumulate ? *cp : 0.f) + ga.alpha * sum;
    }
  }
  if (ga.ws_csum) {                                                 // the bias gradient (always accumulates)
    for (long m = (long)blockIdx.x * blockDim.x + threadIdx.x; m < ga.M; m += (long)gridDim.x * blockDim.x) {
      float sum = 0.f;
      for (int sl = s0; sl < s1; sl++) sum += ga.ws_csum[(long)sl * ga.M + m];
      if (live_b) sum += gb.ws_csum[(long)gb.slab0 * ga.M + m];
      ga.csum[m] += sum;
    }
  }
}

__global__ void wgrad_zero_kernel(float* C, long ldc, int M, int N) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)M * N;
  for (; i < total; i += (long)gridDim.x * blockDim.x) C[(i / N) * ldc + (i % N)] = 0.f;
}

}  // namespace ptv

using namespace ptv;

namespace ptv {
// 1 (default): slab partials through a workspace + ordered reduction (bit-reproducible, and no fp32 atomics: they retire at
// ~1e11 elements/s, a fifth of the time of the deep products); 0: atomics into C
static int g_wgrad_dma = [] { const char* e = getenv("PTV_WGRAD_DMA"); return e ? atoi(e) : 0; }();
static int g_wgrad_mode = [] { const char* e = getenv("PTV_WGRAD_ORDERED"); return (e && e[0] == '0') ? 0 : 1; }();

// grow-only workspace per stream: launches on one stream are ordered, so the next product's partials cannot overtake this one's
// reduction; two streams never share a buffer.  (Allocation happens on a stream's first large product -- never inside a captured
// graph if the capture was preceded by a warm-up of the same step.)
struct WsBuf { float* p = nullptr; size_t bytes = 0; };
static WsBuf* ws_for(hipStream_t s, size_t bytes) {
  static WsBuf pool[64];
  static hipStream_t keys[64];
  static int n = 0;
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  int i = 0;
  for (; i < n; i++) if (keys[i] == s) break;
  if (i == n) { if (n == 64) return nullptr; keys[n++] = s; pool[i] = WsBuf{}; }
  WsBuf& b = pool[i];
  if (b.bytes < bytes) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) == hipSuccess && st != hipStreamCaptureStatusNone) return nullptr;   // cannot grow inside a capture
    // (the outgrown buffer is NOT freed: a captured hipGraph may hold its address -- graph_step.py replays launches recorded on this
    // stream -- and it is tens of megabytes at most)
    size_t want = bytes + bytes / 4;
    if (hipMalloc(reinterpret_cast<void**>(&b.p), want) != hipSuccess) { b = WsBuf{}; return nullptr; }
    b.bytes = want;
  }
  return &b;
}
}  // namespace ptv

namespace ptv { extern int g_splitk_ordered; }
extern "C" int ptv_wgrad_dma(int enable) { ptv::g_wgrad_dma = enable ? 1 : 0; return PTV_OK; }
extern "C" int ptv_wgrad_mode(int ordered) { ptv::g_wgrad_mode = ptv::g_splitk_ordered = ordered ? 1 : 0; return PTV_OK; }

static int wgrad_impl(int M, int N, int K, const void* A, long lda, const void* A2, long lda2, int split, const void* B, long ldb, float* C, long ldc,
                      float alpha, int accumulate, int dtypes, int slabs, float* colsum_a, const int* k_top, long k_unit, int k_rev, void* stream) {
  if (M < 0 || N < 0 || K < 0 || !A || !B || !C) return PTV_ERR_ARG;
  if (k_top && (k_unit <= 0 || k_unit % WBK)) return PTV_ERR_ARG;
  if (M == 0 || N == 0) return PTV_OK;
  {                                                      // timing experiment only (results invalid): PTV_WGRAD_DRY=1 skips every product
    static const int dry = getenv("PTV_WGRAD_DRY") ? atoi(getenv("PTV_WGRAD_DRY")) : 0;
    if (dry) return PTV_OK;
  }
  hipStream_t s = (hipStream_t)stream;
  if (K == 0) {
    if (!accumulate) {
      const long total = (long)M * N;
      int nb = (int)((total + 255) / 256); if (nb > 2048) nb = 2048;
      hipLaunchKernelGGL(wgrad_zero_kernel, dim3(nb), dim3(256), 0, s, C, ldc, M, N);
    }
    PTV_CHECK_LAUNCH();
    return PTV_OK;
  }
  const bool af = !(dtypes & 1), bf = !(dtypes & 2);
  const bool vec = (lda % (af ? 4 : 8) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0) &&
                   (!A2 || ((lda2 % (af ? 4 : 8) == 0) && ((reinterpret_cast<uintptr_t>(A2) & 15) == 0))) &&
                   (ldb % (bf ? 4 : 8) == 0) && ((reinterpret_cast<uintptr_t>(B) & 15) == 0);
  // the unguarded kernel reads whole 8-column chunks: where a chunk straddles M or N it spills into the next row, so it stops
  // short of the operands' last row; the guarded kernel takes the remaining <= 32 rows (and everything when rows are unaligned)
  const bool odd = (M % 8) || (N % 8) || (A2 && (split % 8));
  const int kfast = !vec ? 0 : (odd ? ((K - 1) / WBK) * WBK : (K / WBK) * WBK);
  const int nset = 2;                                              // (4 register sets of prefetch, bf16 sources only: 2028 vs 2011 us over the step's shapes)
  WgArgs sent[2]; int nsent[2] = {0, 0};
  float* ws = nullptr; float* ws_csum = nullptr; int ws_slabs = 0;
  const bool has_a = kfast > 0, has_b = kfast < K;
  bool zeroed = false;
  auto launch = [&](bool guard, int k0, int kn, int want_slabs, int pass) -> int {
    WgArgs g{static_cast<const char*>(A) + (long)k0 * lda * (af ? 4 : 2), lda, static_cast<const char*>(B) + (long)k0 * ldb * (bf ? 4 : 2), ldb,
             C, ldc, M, N, kn, 0, cdiv(M, WBM), cdiv(N, WBN), 1, 0, alpha, colsum_a, k_top, k_unit, k_rev, g_gemm_prio,
             nullptr, nullptr, 0,
             A2 ? static_cast<const char*>(A2) + (long)k0 * lda2 * (af ? 4 : 2) : nullptr, lda2, split, k0};
    // (the guarded tail launch takes the limits too, shifted by its first row: the rows a limit declares zero may never have been WRITTEN by
    // whoever produced the other operand -- a forward that stopped at the batch's last live note step)
    const int tiles = g.tiles_m * g.tiles_n;     // BLOCKS per slab
    // slab count (measured optima of scripts/bench_wgrad.py sweep on MI355X).  Every slab pays M*N atomics, and a grid that is
    // just over one block per CU leaves a tail, so: about one block per CU (never more) for the skinny, HBM-bound products;
    // about three per CU for the MFMA-heavy ones (many tiles), where co-resident blocks hide each other's stalls; a slab is at least
    // 32 (short K) or 64 stages deep
    int ns = want_slabs;
    if (ns <= 0) {
      const bool shortk = kn <= 8192;
      ns = shortk ? (256 + tiles - 1) / tiles : (tiles >= 32 ? 768 : 256) / tiles;
      const int deep = shortk ? (kn >= 2048 ? kn / 1024 : kn / 256) : kn / 2048;
      if (ns > deep) ns = deep;
      if (ns >= 8) ns = ns / 8 * 8;
      if (ns < 1) ns = 1;
    }
    const int maxs = kn / (4 * WBK) > 0 ? kn / (4 * WBK) : 1;
    if (ns > maxs) ns = maxs;
    // block -> (slab, tile) so that the blocks of one XCD (dispatch is round-robin over the 8) share operand columns in ITS L2:
    // map 2 gives an XCD a range of tiles for all slabs (B columns are then fetched by every XCD: fine when K is short), map 1 gives it
    // whole slabs -- every K row is fetched by one XCD only (PMC on 1536 x 512 x 245760: L2 hit 29 % and 3.3x the algorithmic bytes from
    // the fabric with map 2)
    const bool can1 = ns >= 8 && ns % 8 == 0, can2 = tiles % 8 == 0;
    const bool deepk = kn >= 16384;
    if (can1 && (deepk || !can2)) g.map = 1;
    else if (can2) g.map = 2;
    g.kper = cdiv(cdiv(kn, ns), WBK) * WBK;
    if (g.map != 1) ns = cdiv(kn, g.kper);
    g.nslab = ns;
    if (pass == 0) {                                             // planning pass: slab counts only
      sent[guard ? 1 : 0] = g; nsent[guard ? 1 : 0] = ns;
      return PTV_OK;
    }
    if (ws) {
      g.ws = ws; g.ws_csum = colsum_a ? ws_csum : nullptr; g.slab0 = guard ? nsent[0] : 0;
    }
    else if (!accumulate && !zeroed) {
      const long total = (long)M * N;
      int nb = (int)((total + 255) / 256); if (nb > 2048) nb = 2048;
      hipLaunchKernelGGL(wgrad_zero_kernel, dim3(nb), dim3(256), 0, s, C, ldc, M, N);
      zeroed = true;
    }
    sent[guard ? 1 : 0] = g;
    const dim3 grid((unsigned)(tiles * ns));
    constexpr int lds_pad = 0;       // (unused dynamic LDS per block to keep product blocks off the CUs of the latency chains: no change, round 4)
#define WG_LAUNCH(AF, BF)                                                                                      \
    do {                                                                                                       \
      if (guard) hipLaunchKernelGGL((wgrad_kernel<AF, BF, true, 2>), grid, dim3(256), lds_pad, s, g);                \
      else if (nset == 2) hipLaunchKernelGGL((wgrad_kernel<AF, BF, false, 2>), grid, dim3(256), lds_pad, s, g);      \
      else if constexpr (!(AF) && !(BF)) hipLaunchKernelGGL((wgrad_kernel<false, false, false, 4>), grid, dim3(256), lds_pad, s, g); \
    } while (0)
    // (default OFF: standalone the LDS-DMA kernel is 5-12 % faster on the deep products -- 1536 x 512 x 245760 571 -> 544 us -- but in the
    // step it measured 7.82-7.98 ms against 7.72-7.75: its 48 KB of LDS per block co-reside worse with the persistent recurrences' 96-KB
    // workgroups than the register-staged kernel's 36 KB.  PTV_WGRAD_DMA=1 enables it; tests/test_gpu_switches.py runs the step on it.)
    if (!af && !bf && !guard && g_wgrad_dma && (kn % WBK) == 0) { hipLaunchKernelGGL(wgrad_dma_kernel, grid, dim3(256), 0, s, g); return PTV_OK; }
    if (af && bf) WG_LAUNCH(true, true);
    else if (af) WG_LAUNCH(true, false);
    else if (bf) WG_LAUNCH(false, true);
    else WG_LAUNCH(false, false);
#undef WG_LAUNCH
    return PTV_OK;
  };
  if (has_a) launch(false, 0, kfast, slabs, 0);
  if (has_b) launch(true, kfast, K - kfast, has_a ? 1 : slabs, 0);
  const int total_slabs = nsent[0] + nsent[1];
  // one slab in all: the product kernel is the only writer of every element (plain read-modify-write, already reproducible)
  if (g_wgrad_mode == 1 && total_slabs > 1) {
    const int tiles = cdiv(M, WBM) * cdiv(N, WBN);
    const size_t tile_bytes = (size_t)total_slabs * tiles * WBM * WBN * sizeof(float);
    const size_t sum_bytes = colsum_a ? (size_t)total_slabs * M * sizeof(float) : 0;
    WsBuf* wb = ws_for(s, tile_bytes + sum_bytes);
    if (!wb) g_ord_fallbacks++;
    if (wb) { ws = wb->p; ws_csum = colsum_a ? wb->p + tile_bytes / sizeof(float) : nullptr; ws_slabs = total_slabs; }
  }
  const int pi = prof::want(5, M, N) ? prof::begin(s) : -1;        // bench.py's roofline block: the family's launches, product + reduction
  if (has_a) launch(false, 0, kfast, slabs, 1);
  if (has_b) launch(true, kfast, K - kfast, has_a ? 1 : slabs, 1);
  if (ws) {
    const bool vec4 = (N & 3) == 0 && (ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0;
    const long total = vec4 ? ((M * (long)N) >> 2) : M * (long)N;
    int nb = (int)((total + 255) / 256); if (nb > 4096) nb = 4096; if (nb < 1) nb = 1;
    // (the guarded launch alone -- unaligned operands -- may have many slabs: it then plays the fast launch's part in the reduction)
    if (!has_a) hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(nb), dim3(256), 0, s, sent[1], sent[1], 1, 0, accumulate);
    else hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(nb), dim3(256), 0, s, sent[0], sent[1], 1, has_b ? 1 : 0, accumulate);
  }
  (void)ws_slabs;
  if (pi >= 0) prof::end(pi, s, 2.0 * M * N * K);                  // (full K: a k_top limit is a device value)
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

extern "C" int ptv_wgrad(int M, int N, int K, const void* A, long lda, const void* B, long ldb, float* C, long ldc, float alpha,
                         int accumulate, int dtypes, int slabs, float* colsum_a, const int* k_top, long k_unit, int k_rev, void* stream) {
  return wgrad_impl(M, N, K, A, lda, nullptr, 0, 0, B, ldb, C, ldc, alpha, accumulate, dtypes, slabs, colsum_a, k_top, k_unit, k_rev, stream);
}

// C[M1 + M2, N] (+)= alpha * [A1 | A2]^T . B: the column blocks of two row-per-sample matrices of the same dtype against ONE pass over B
// (M1 a multiple of 128).  The notes GRU's weight_hh gradient is [dgi[:, :1024] | dgh_n]^T . h (the r / z thirds of the gate gradients are
// shared with the input side, ptv_notes_gru_persist_bwd): two products read the 252-MB state matrix twice.
extern "C" int ptv_wgrad_cat(int M1, const void* A1, long lda1, int M2, const void* A2, long lda2, int N, int K, const void* B, long ldb,
                             float* C, long ldc, float alpha, int accumulate, int dtypes, int slabs, float* colsum_a, const int* k_top,
                             long k_unit, int k_rev, void* stream) {
  if (M1 <= 0 || M2 <= 0 || (M1 % WBM) || !A1 || !A2) return PTV_ERR_ARG;
  return wgrad_impl(M1 + M2, N, K, A1, lda1, A2, lda2, M1, B, ldb, C, ldc, alpha, accumulate, dtypes, slabs, colsum_a, k_top, k_unit, k_rev, stream);
}

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
  const int* seg_n; int seg_unit, seg_period;   // K segments (ptv_wgrad_job): of unit q = row / seg_unit only the first seg_n[q % seg_period] rows are live
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
  if (g.seg_n) {                                                  // (kper divides seg_unit: a slab lies inside one unit)
    const long row = (long)k_begin + g.k_base;
    const int q = (int)(row / g.seg_unit), r0 = (int)(row - (long)q * g.seg_unit);
    // (a negative period: the units run in REVERSED order -- the reversed direction of a GRU indexes its rows by processing step)
    const int qi = g.seg_period > 0 ? q % g.seg_period : -g.seg_period - 1 - q % (-g.seg_period);
    const int live = g.seg_n[qi] - r0;                            // live rows from the slab's first row on
    if (live <= 0) return false;
    if (k_end - k_begin > live) k_end = k_begin + live;
  }
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

// one block's share of a product: block number b of the launch that `g` describes (a slab of K for one 128 x 128 output tile).
// GUARD = false: rows 16-byte aligned, K a multiple of 32 and not reaching the operands' last row (see ptv_wgrad).
// Shared by the one-product kernel and the batched one (wgrad_batch_kernel): the same plan gives the same bits either way.
template <bool AF32, bool BF32, bool GUARD>
__device__ __forceinline__ void wgrad_block(const WgArgs& g, int b, __bf16* As, __bf16* Bs) {
  constexpr int NSET = 2;            // register sets of prefetch (4, bf16 sources only, measured 2028 vs 2011 us over the step's shapes: removed)
  if (g.prio) __builtin_amdgcn_s_setprio(3);
  const int tiles = g.tiles_m * g.tiles_n;
  int slab, tile;
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
  }
  if (t0 < nst) WG_BODY(0, t0);
#undef WG_BODY
#undef WG_FETCH
  wgrad_store(g, acc, accs, do_sum, slab, tile, tiles, m_blk, n_blk, wm, wn, lane);
}

template <bool AF32, bool BF32, bool GUARD>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(WgArgs g) {
  __shared__ __attribute__((aligned(16))) __bf16 As[2 * WSTAGE];
  __shared__ __attribute__((aligned(16))) __bf16 Bs[2 * WSTAGE];
  wgrad_block<AF32, BF32, GUARD>(g, blockIdx.x, As, Bs);
}

// ---------------------------------------------------------------------------------------------
// Several products in ONE launch (ptv_wgrad_batch): the parameter-gradient products that become ready at the same point of a backward
// pass -- a GRU's W_ih / W_hh / bias gradients, a decoder's z projections -- are each a few dozen to a few hundred blocks; launched one
// by one every small product leaves most of the chip idle for its ramp-up and tail, and each is followed by its own reduction launch.
// The table holds one entry per (product, fast | guarded part); a block finds its entry by its number and runs wgrad_block on it.
// ---------------------------------------------------------------------------------------------
constexpr int WG_MAX_ENT = 16;          // entries per launch (8 products with a guarded tail each)
struct WgBatch {
  WgArgs g[WG_MAX_ENT];
  int first[WG_MAX_ENT + 1];            // entry e owns blocks first[e] .. first[e + 1] - 1
  unsigned char kind[WG_MAX_ENT];       // bit 0: A fp32, bit 1: B fp32, bit 2: guarded
  int n;
};

template <bool HAS_F32, bool HAS_GUARD>
__global__ __launch_bounds__(256, 2) void wgrad_batch_kernel(WgBatch bt) {
  __shared__ __attribute__((aligned(16))) __bf16 As[2 * WSTAGE];
  __shared__ __attribute__((aligned(16))) __bf16 Bs[2 * WSTAGE];
  int e = 0;
  while (e + 1 < bt.n && (int)blockIdx.x >= bt.first[e + 1]) e++;
  e = __builtin_amdgcn_readfirstlane(e);
  const WgArgs& g = bt.g[e];
  const int b = (int)blockIdx.x - bt.first[e];
  const int kind = bt.kind[e];
  if (!HAS_F32 && !HAS_GUARD) { wgrad_block<false, false, false>(g, b, As, Bs); return; }
  switch (kind) {
    case 0: wgrad_block<false, false, false>(g, b, As, Bs); break;
    case 1: if constexpr (HAS_F32) wgrad_block<true, false, false>(g, b, As, Bs); break;
    case 2: if constexpr (HAS_F32) wgrad_block<false, true, false>(g, b, As, Bs); break;
    case 3: if constexpr (HAS_F32) wgrad_block<true, true, false>(g, b, As, Bs); break;
    case 4: if constexpr (HAS_GUARD) wgrad_block<false, false, true>(g, b, As, Bs); break;
    case 5: if constexpr (HAS_F32 && HAS_GUARD) wgrad_block<true, false, true>(g, b, As, Bs); break;
    case 6: if constexpr (HAS_F32 && HAS_GUARD) wgrad_block<false, true, true>(g, b, As, Bs); break;
    default: if constexpr (HAS_F32 && HAS_GUARD) wgrad_block<true, true, true>(g, b, As, Bs); break;
  }
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
// bi / nb: this block's number among the nb blocks that work on this product (grid-stride over its elements).
__device__ __forceinline__ void wgrad_reduce_block(const WgArgs& ga, const WgArgs& gb, int has_a, int has_b, int accumulate, int bi, int nb) {
  const int tiles = ga.tiles_m * ga.tiles_n;
  // the live slabs of the fast launch are one contiguous range (the k_top limits cut K at one end): [s0, s1) -- or, with K segments, a list
  // (one thread per slab finds out, a ballot keeps them in slab order: at most 256 slabs, wgrad_run)
  __shared__ unsigned short lst[256];
  __shared__ int wtot[4];
  const bool listed = ga.seg_n != nullptr;
  int s0 = 0, s1 = 0;
  if (has_a && listed) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int kb, ke;
    const bool lv = tid < ga.nslab && slab_range(ga, tid, kb, ke);
    const unsigned long long m = __ballot(lv);
    if (lane == 0) wtot[wv] = __popcll(m);
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wv; w++) base += wtot[w];
    if (lv) lst[base + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)tid;
    s1 = wtot[0] + wtot[1] + wtot[2] + wtot[3];
    __syncthreads();
  } else if (has_a) {
    bool any = false;
    for (int sl = 0; sl < ga.nslab; sl++) {
      int kb, ke;
      if (slab_range(ga, sl, kb, ke)) { if (!any) s0 = sl; s1 = sl + 1; any = true; }
    }
  }
  auto SL = [&](int i) { return listed ? (long)lst[i] : (long)i; };
  bool live_b = false;
  if (has_b) { int kb, ke; live_b = slab_range(gb, 0, kb, ke); }
  const long tstride = (long)tiles * (WBM * WBN);
  const float* wb = has_b ? gb.ws + (long)gb.slab0 * tstride : nullptr;
  const bool vec = (ga.N & 3) == 0 && (ga.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(ga.C) & 15) == 0;
  const long gstride = (long)nb * blockDim.x, gstart = (long)bi * blockDim.x + threadIdx.x;
  if (vec) {
    const int N4 = ga.N >> 2;
    const long total4 = (long)ga.M * N4;
    for (long i = gstart; i < total4; i += gstride) {
      const int m = (int)(i / N4), n = (int)(i % N4) * 4;
      const long off = (long)((m / WBM) * ga.tiles_n + n / WBN) * (WBM * WBN) + (long)(m % WBM) * WBN + (n % WBN);
      const float* p = ga.ws + off;
      float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
      int sl = s0;
      for (; sl + 8 <= s1; sl += 8) {                               // eight partials in flight, added in slab order
        float4 v[8];
#pragma unroll
        for (int q = 0; q < 8; q++) v[q] = *reinterpret_cast<const float4*>(p + SL(sl + q) * tstride);
#pragma unroll
        for (int q = 0; q < 8; q++) { sum.x += v[q].x; sum.y += v[q].y; sum.z += v[q].z; sum.w += v[q].w; }
      }
      for (; sl < s1; sl++) {
        const float4 v = *reinterpret_cast<const float4*>(p + SL(sl) * tstride);
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
    for (long i = gstart; i < total; i += gstride) {
      const int m = (int)(i / ga.N), n = (int)(i % ga.N);
      const long off = (long)((m / WBM) * ga.tiles_n + n / WBN) * (WBM * WBN) + (long)(m % WBM) * WBN + (n % WBN);
      float sum = 0.f;
      const float* p = ga.ws + off;
      int sl = s0;
      for (; sl + 8 <= s1; sl += 8) {
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; q++) v[q] = p[SL(sl + q) * tstride];
#pragma unroll
        for (int q = 0; q < 8; q++) sum += v[q];
      }
      for (; sl < s1; sl++) sum += p[SL(sl) * tstride];
      if (live_b) sum += wb[off];
      float* cp = ga.C + (long)m * ga.ldc + n;
      *cp = (accumulate ? *cp : 0.f) + ga.alpha * sum;
    }
  }
  if (ga.ws_csum) {                                                 // the bias gradient (always accumulates)
    for (long m = gstart; m < ga.M; m += gstride) {
      float sum = 0.f;
      for (int sl = s0; sl < s1; sl++) sum += ga.ws_csum[SL(sl) * ga.M + m];
      if (live_b) sum += gb.ws_csum[(long)gb.slab0 * ga.M + m];
      ga.csum[m] += sum;
    }
  }
}

__global__ void wgrad_reduce_kernel(WgArgs ga, WgArgs gb, int has_a, int has_b, int accumulate) {
  wgrad_reduce_block(ga, gb, has_a, has_b, accumulate, blockIdx.x, gridDim.x);
}

// the reductions of a batch in one launch: product j owns blocks first[j] .. first[j + 1] - 1
constexpr int WG_MAX_JOBS = 8;
struct WgRedBatch {
  WgArgs ga[WG_MAX_JOBS], gb[WG_MAX_JOBS];
  int first[WG_MAX_JOBS + 1];
  unsigned char has_a[WG_MAX_JOBS], has_b[WG_MAX_JOBS], acc[WG_MAX_JOBS];
  int n;
};
__global__ void wgrad_reduce_batch_kernel(WgRedBatch rb) {
  int j = 0;
  while (j + 1 < rb.n && (int)blockIdx.x >= rb.first[j + 1]) j++;
  j = __builtin_amdgcn_readfirstlane(j);
  wgrad_reduce_block(rb.ga[j], rb.gb[j], rb.has_a[j], rb.has_b[j], rb.acc[j], (int)blockIdx.x - rb.first[j], rb.first[j + 1] - rb.first[j]);
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
// how ptv_wgrad_batch issues its products (ptv_wgrad_batch_mode / PTV_WGRAD_BATCH; same bits in every mode).  Measured in the B = 512 step,
// same process, alternating (profiles/r06_ab_runs.txt): 0 = one call each 6.98-7.01 ms, 1 = ONE launch for everything 7.26-7.33 (the mixed
// launch runs every product at the register budget of its fp32-source variant and five operand streams thrash each XCD's L2), 2 = single
// product launches + one reduction launch 6.98-7.09, 3 = small products batched, deep ones alone, one reduction launch 6.93-7.03: the default
static int g_wgrad_batch = [] { const char* e = getenv("PTV_WGRAD_BATCH"); return e ? atoi(e) : 3; }();
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
extern "C" int ptv_wgrad_batch_mode(int mode) { if (mode < 0 || mode > 3) return PTV_ERR_ARG; ptv::g_wgrad_batch = mode; return PTV_OK; }
extern "C" int ptv_wgrad_dma(int enable) { ptv::g_wgrad_dma = enable ? 1 : 0; return PTV_OK; }
extern "C" int ptv_wgrad_mode(int ordered) { ptv::g_wgrad_mode = ptv::g_splitk_ordered = ordered ? 1 : 0; return PTV_OK; }

extern "C" int ptv_wgrad_seg_supported(long K, long seg_unit);
namespace {
// one product of a call: what the caller asked for (ptv_wgrad_job + the two-source form of ptv_wgrad_cat) ...
struct Job {
  int M, N, K; const void* A; long lda; const void* A2; long lda2; int split; const void* B; long ldb; float* C; long ldc;
  float alpha; int accumulate, dtypes, slabs; float* colsum_a; const int* k_top; long k_unit; int k_rev;
  const int* seg_n = nullptr; long seg_unit = 0; int seg_period = 0;
};
// ... and how it runs: the fast (unguarded) launch over rows [0, kfast) and the guarded one over the rest, each cut into K slabs.
// A pure function of the job (shapes, alignment) and of the process-wide mode: one product planned alone or inside a batch runs the
// same blocks on the same slabs and reduces them in the same order -- the same bits.
struct Plan {
  WgArgs part[2]; bool has[2]; int nslab[2]; int tiles; bool af, bf; int total_slabs;
  size_t tile_floats, sum_floats;           // workspace need when the reduction is ordered (total_slabs > 1)
};

int plan_job(const Job& j, Plan& p) {
  const bool af = !(j.dtypes & 1), bf = !(j.dtypes & 2);
  p.af = af; p.bf = bf;
  const bool vec = (j.lda % (af ? 4 : 8) == 0) && ((reinterpret_cast<uintptr_t>(j.A) & 15) == 0) &&
                   (!j.A2 || ((j.lda2 % (af ? 4 : 8) == 0) && ((reinterpret_cast<uintptr_t>(j.A2) & 15) == 0))) &&
                   (j.ldb % (bf ? 4 : 8) == 0) && ((reinterpret_cast<uintptr_t>(j.B) & 15) == 0);
  // the unguarded kernel reads whole 8-column chunks: where a chunk straddles M or N it spills into the next row, so it stops
  // short of the operands' last row; the guarded kernel takes the remaining <= 32 rows (and everything when rows are unaligned)
  const bool odd = (j.M % 8) || (j.N % 8) || (j.A2 && (j.split % 8));
  const int kfast = !vec ? 0 : (odd ? ((j.K - 1) / WBK) * WBK : (j.K / WBK) * WBK);
  p.has[0] = kfast > 0; p.has[1] = kfast < j.K;
  p.tiles = cdiv(j.M, WBM) * cdiv(j.N, WBN);
  auto part = [&](bool guard, int k0, int kn, int want_slabs) {
    WgArgs g{static_cast<const char*>(j.A) + (long)k0 * j.lda * (af ? 4 : 2), j.lda, static_cast<const char*>(j.B) + (long)k0 * j.ldb * (bf ? 4 : 2), j.ldb,
             j.C, j.ldc, j.M, j.N, kn, 0, cdiv(j.M, WBM), cdiv(j.N, WBN), 1, 0, j.alpha, j.colsum_a, j.k_top, j.k_unit, j.k_rev, g_gemm_prio,
             nullptr, nullptr, 0,
             j.A2 ? static_cast<const char*>(j.A2) + (long)k0 * j.lda2 * (af ? 4 : 2) : nullptr, j.lda2, j.split, k0,
             j.seg_n, (int)j.seg_unit, j.seg_period};
    // (the guarded tail launch takes the limits too, shifted by its first row: the rows a limit declares zero may never have been WRITTEN by
    // whoever produced the other operand -- a forward that stopped at the batch's last live note step)
    const int tiles = p.tiles;     // BLOCKS per slab
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
    if (j.seg_n) {
      // K segments: a slab must lie inside one unit -- kper = the largest power of two <= the wanted depth that divides seg_unit (>= 128 rows);
      // at most 256 slabs (the reduction lists the live ones with one thread per slab)
      int kper = 128;
      int want = cdiv(kn, ns);
      if (want_slabs <= 0 && want > 4096) want = 4096;           // (measured, scripts/bench_wgrad_seg.py: the clipped slabs of a unit balance better when a unit holds >= 4 of them)
      while (kper * 2 <= want && (j.seg_unit % (kper * 2)) == 0) kper *= 2;
      while (cdiv(kn, kper) > 248 && (j.seg_unit % (kper * 2)) == 0) kper *= 2;
      g.kper = kper; ns = cdiv(kn, kper);
      // whole slabs per XCD (map 1: every K row is fetched into ONE L2) wants a multiple of 8 slabs: pad with slabs beyond K (they find
      // no rows and leave).  With tile ranges per XCD (map 2) every XCD fetched every row of both operands: PMC, +0.1 GB per step
      // instead of -0.6 for the clipped products
      const bool can2 = tiles % 8 == 0;
      if (ns >= 8 && (kn >= 16384 || !can2) && (ns + 7) / 8 * 8 <= 256) { ns = (ns + 7) / 8 * 8; g.map = 1; }
      else if (can2) g.map = 2;
      g.nslab = ns;
      p.part[guard ? 1 : 0] = g; p.nslab[guard ? 1 : 0] = ns;
      return;
    }
    const bool can1 = ns >= 8 && ns % 8 == 0, can2 = tiles % 8 == 0;
    const bool deepk = kn >= 16384;
    if (can1 && (deepk || !can2)) g.map = 1;
    else if (can2) g.map = 2;
    g.kper = cdiv(cdiv(kn, ns), WBK) * WBK;
    if (g.map != 1) ns = cdiv(kn, g.kper);
    g.nslab = ns;
    p.part[guard ? 1 : 0] = g; p.nslab[guard ? 1 : 0] = ns;
  };
  p.nslab[0] = p.nslab[1] = 0;
  if (p.has[0]) part(false, 0, kfast, j.slabs);
  if (p.has[1]) part(true, kfast, j.K - kfast, p.has[0] ? 1 : j.slabs);
  p.total_slabs = p.nslab[0] + p.nslab[1];
  p.tile_floats = (size_t)p.total_slabs * p.tiles * WBM * WBN;
  p.sum_floats = j.colsum_a ? (size_t)p.total_slabs * j.M : 0;
  return PTV_OK;
}

void zero_c(const Job& j, hipStream_t s) {
  const long total = (long)j.M * j.N;
  int nb = (int)((total + 255) / 256); if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(wgrad_zero_kernel, dim3(nb), dim3(256), 0, s, j.C, j.ldc, j.M, j.N);
}

int reduce_blocks(const Job& j) {
  const bool vec4 = (j.N & 3) == 0 && (j.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(j.C) & 15) == 0;
  const long total = vec4 ? ((j.M * (long)j.N) >> 2) : j.M * (long)j.N;
  int nb = (int)((total + 255) / 256); if (nb > 4096) nb = 4096; if (nb < 1) nb = 1;
  return nb;
}

// the products of `jobs` (disjoint outputs), as one product launch + one reduction launch when there are several
int wgrad_run(const Job* jobs, int njobs, hipStream_t s) {
  if (njobs <= 0) return PTV_OK;
  if (!g_wgrad_batch && njobs > 1) {
    for (int i = 0; i < njobs; i++) PTV_TRY(wgrad_run(jobs + i, 1, s));
    return PTV_OK;
  }
  if (njobs > WG_MAX_JOBS) {                                       // more than a table holds: in chunks
    for (int i = 0; i < njobs; i += WG_MAX_JOBS) PTV_TRY(wgrad_run(jobs + i, njobs - i < WG_MAX_JOBS ? njobs - i : WG_MAX_JOBS, s));
    return PTV_OK;
  }
  {                                                      // timing experiment only (results invalid): PTV_WGRAD_DRY=1 skips every product
    static const int dry = getenv("PTV_WGRAD_DRY") ? atoi(getenv("PTV_WGRAD_DRY")) : 0;
    if (dry) return PTV_OK;
  }
  Job live[WG_MAX_JOBS]; Plan plan[WG_MAX_JOBS]; int n = 0;
  double flops = 0.0, lim15 = 0.0, lim16 = 0.0, seg = 0.0; int pm = 0, pn = 0;
  for (int i = 0; i < njobs; i++) {
    const Job& j = jobs[i];
    if (j.M < 0 || j.N < 0 || j.K < 0 || !j.A || !j.B || !j.C) return PTV_ERR_ARG;
    if (j.k_top && (j.k_unit <= 0 || j.k_unit % WBK)) return PTV_ERR_ARG;
    if (j.seg_n && (j.seg_period == 0 || j.A2 || !ptv_wgrad_seg_supported(j.K, j.seg_unit))) return PTV_ERR_ARG;
    if (j.M == 0 || j.N == 0) continue;
    if (j.K == 0) { if (!j.accumulate) zero_c(j, s); continue; }
    live[n] = j; PTV_TRY(plan_job(j, plan[n]));
    flops += 2.0 * j.M * j.N * j.K; pm = j.M; pn = j.N;
    if (j.k_top && j.k_unit > 0) {
      const long units = j.k_rev > 0 ? j.k_rev : j.K / j.k_unit;
      if (units == 15) lim15 += 2.0 * j.M * j.N * j.K; else if (units == 16) lim16 += 2.0 * j.M * j.N * j.K;
    }
    if (j.seg_n) seg += 2.0 * j.M * j.N * j.K;
    n++;
  }
  if (n == 0) { PTV_CHECK_LAUNCH(); return PTV_OK; }
  // workspace of the ordered reductions: one region per product that has more than one slab in all (a single slab is the only writer of
  // every element: plain read-modify-write, already reproducible)
  float* wsp[WG_MAX_JOBS]; float* wsc[WG_MAX_JOBS];
  for (int i = 0; i < n; i++) wsp[i] = wsc[i] = nullptr;
  if (g_wgrad_mode == 1) {
    size_t need = 0;
    for (int i = 0; i < n; i++) if (plan[i].total_slabs > 1) need += plan[i].tile_floats + ((plan[i].sum_floats + 3) & ~(size_t)3);
    if (need) {
      WsBuf* wb = ws_for(s, need * sizeof(float));
      if (!wb) g_ord_fallbacks++;
      else {
        float* q = wb->p;
        for (int i = 0; i < n; i++) if (plan[i].total_slabs > 1) {
          wsp[i] = q; q += plan[i].tile_floats;
          if (live[i].colsum_a) { wsc[i] = q; q += (plan[i].sum_floats + 3) & ~(size_t)3; }
        }
      }
    }
  }
  const int pi = prof::want(5, pm, pn) ? prof::begin(s) : -1;       // bench.py's roofline block: the family's launches, product + reduction
  for (int i = 0; i < n; i++) {
    for (int k = 0; k < 2; k++) if (plan[i].has[k]) {
      WgArgs& g = plan[i].part[k];
      if (wsp[i]) { g.ws = wsp[i]; g.ws_csum = wsc[i]; g.slab0 = k ? plan[i].nslab[0] : 0; }
    }
    if (!wsp[i] && !live[i].accumulate) zero_c(live[i], s);          // (atomics / single slab: the product adds into C)
  }
  // how a call with several products runs (g_wgrad_batch; same bits in every mode):
  //   1  ONE product launch for all of them (wgrad_batch_kernel) + ONE reduction launch
  //   2  one product launch each (the kernel specialised for its operand dtypes), ONE reduction launch for all
  //   3  the small products (<= SMALL_BLOCKS blocks) in one launch, the others one launch each, ONE reduction launch
  // (0: every product its own call -- handled above)
  constexpr int SMALL_BLOCKS = 160;
  auto launch_single = [&](const Plan& p) {
    for (int k = 0; k < 2; k++) if (p.has[k]) {
      const WgArgs& g = p.part[k];
      const dim3 grid((unsigned)(p.tiles * p.nslab[k]));
      const bool guard = k == 1;
#define WG_LAUNCH(AF, BF)                                                                                      \
      do {                                                                                                     \
        if (guard) hipLaunchKernelGGL((wgrad_kernel<AF, BF, true>), grid, dim3(256), 0, s, g);                 \
        else hipLaunchKernelGGL((wgrad_kernel<AF, BF, false>), grid, dim3(256), 0, s, g);                      \
      } while (0)
      // (default OFF: standalone the LDS-DMA kernel is 5-12 % faster on the deep products -- 1536 x 512 x 245760 571 -> 544 us -- but in the
      // step it measured 7.82-7.98 ms against 7.72-7.75: its 48 KB of LDS per block co-reside worse with the persistent recurrences' 96-KB
      // workgroups than the register-staged kernel's 36 KB.  PTV_WGRAD_DMA=1 enables it; tests/test_gpu_switches.py runs the step on it.)
      if (!p.af && !p.bf && !guard && g_wgrad_dma && (g.K % WBK) == 0) { hipLaunchKernelGGL(wgrad_dma_kernel, grid, dim3(256), 0, s, g); continue; }
      if (p.af && p.bf) WG_LAUNCH(true, true);
      else if (p.af) WG_LAUNCH(true, false);
      else if (p.bf) WG_LAUNCH(false, true);
      else WG_LAUNCH(false, false);
#undef WG_LAUNCH
    }
  };
  if (n == 1) {
    const Plan& p = plan[0];
    launch_single(p);
    if (wsp[0]) {
      // (the guarded launch alone -- unaligned operands -- may have many slabs: it then plays the fast launch's part in the reduction)
      const int nb = reduce_blocks(live[0]);
      if (!p.has[0]) hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(nb), dim3(256), 0, s, p.part[1], p.part[1], 1, 0, live[0].accumulate);
      else hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(nb), dim3(256), 0, s, p.part[0], p.part[1], 1, p.has[1] ? 1 : 0, live[0].accumulate);
    }
  } else {
    // ---- product launches.  Batched entries go in order of decreasing work (the long blocks start first), every entry's first block a
    // multiple of 8 so that the block -> XCD relation the slab / tile maps count on (block b runs on XCD b % 8) holds inside the entry
    bool batched[WG_MAX_JOBS]; int nbat = 0;
    for (int i = 0; i < n; i++) {
      const int blocks = plan[i].tiles * (plan[i].nslab[0] + plan[i].nslab[1]);
      batched[i] = g_wgrad_batch == 1 || (g_wgrad_batch == 3 && blocks <= SMALL_BLOCKS);
      nbat += batched[i] ? 1 : 0;
    }
    if (nbat == 1) { for (int i = 0; i < n; i++) batched[i] = false; nbat = 0; }
    for (int i = 0; i < n; i++) if (!batched[i]) launch_single(plan[i]);
    if (nbat) {
      WgBatch bt; bt.n = 0;
      int order[2 * WG_MAX_JOBS]; double work[2 * WG_MAX_JOBS]; int ne = 0;
      for (int i = 0; i < n; i++) for (int k = 0; k < 2; k++) if (batched[i] && plan[i].has[k]) {
        order[ne] = i * 2 + k; work[ne] = (double)plan[i].part[k].kper * (k ? 4.0 : 1.0); ne++;       // (a block's time ~ its slab depth; guarded rows cost more)
      }
      for (int a = 1; a < ne; a++) {                                   // insertion sort by work, stable
        const int o = order[a]; const double w = work[a]; int b = a - 1;
        while (b >= 0 && work[b] < w) { order[b + 1] = order[b]; work[b + 1] = work[b]; b--; }
        order[b + 1] = o; work[b + 1] = w;
      }
      bool any_f32 = false, any_guard = false;
      int nblk = 0;
      for (int a = 0; a < ne; a++) {
        const int i = order[a] >> 1, k = order[a] & 1;
        const Plan& p = plan[i];
        bt.g[a] = p.part[k];
        bt.kind[a] = (unsigned char)((p.af ? 1 : 0) | (p.bf ? 2 : 0) | (k ? 4 : 0));
        any_f32 |= p.af || p.bf; any_guard |= k == 1;
        bt.first[a] = nblk;
        nblk += (p.tiles * p.nslab[k] + 7) / 8 * 8;                  // (surplus blocks find their slab beyond K and return)
      }
      bt.first[ne] = nblk; bt.n = ne;
      for (int a = ne + 1; a <= WG_MAX_ENT; a++) bt.first[a] = nblk;
      const dim3 grid((unsigned)nblk);
      if (any_f32 && any_guard) hipLaunchKernelGGL((wgrad_batch_kernel<true, true>), grid, dim3(256), 0, s, bt);
      else if (any_f32) hipLaunchKernelGGL((wgrad_batch_kernel<true, false>), grid, dim3(256), 0, s, bt);
      else if (any_guard) hipLaunchKernelGGL((wgrad_batch_kernel<false, true>), grid, dim3(256), 0, s, bt);
      else hipLaunchKernelGGL((wgrad_batch_kernel<false, false>), grid, dim3(256), 0, s, bt);
    }
    // ---- one reduction launch for the products that went through the workspace
    WgRedBatch rb; rb.n = 0; int rblk = 0;
    for (int i = 0; i < n; i++) if (wsp[i]) {
      const Plan& p = plan[i];
      const int q = rb.n++;
      if (!p.has[0]) { rb.ga[q] = p.part[1]; rb.gb[q] = p.part[1]; rb.has_a[q] = 1; rb.has_b[q] = 0; }
      else { rb.ga[q] = p.part[0]; rb.gb[q] = p.has[1] ? p.part[1] : p.part[0]; rb.has_a[q] = 1; rb.has_b[q] = p.has[1] ? 1 : 0; }
      rb.acc[q] = (unsigned char)(live[i].accumulate ? 1 : 0);
      rb.first[q] = rblk; rblk += reduce_blocks(live[i]);
    }
    for (int q = rb.n; q <= WG_MAX_JOBS; q++) rb.first[q] = rblk;
    if (rb.n) hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3(rblk), dim3(256), 0, s, rb);
  }
  if (pi >= 0) { prof::end(pi, s, flops); prof::aux(pi, lim15, lim16, seg); }    // (full K: a k_top limit is a device value; the limited part separately)
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}
}  // namespace

// K segments need slabs that never straddle a unit -- a power of two (>= 128 rows) that divides seg_unit -- and at most 256 of them (the
// reduction lists the live ones with one thread per slab; 8 are kept for the padding to whole slabs per XCD and the guarded tail)
extern "C" int ptv_wgrad_seg_supported(long K, long seg_unit) {
  if (K <= 0 || seg_unit < 128 || (seg_unit & 127) || (K % seg_unit)) return 0;
  long kper = 128;
  while ((seg_unit % (kper * 2)) == 0) kper *= 2;
  return (K + kper - 1) / kper <= 248 ? 1 : 0;
}

extern "C" int ptv_wgrad(int M, int N, int K, const void* A, long lda, const void* B, long ldb, float* C, long ldc, float alpha,
                         int accumulate, int dtypes, int slabs, float* colsum_a, const int* k_top, long k_unit, int k_rev, void* stream) {
  const Job j{M, N, K, A, lda, nullptr, 0, 0, B, ldb, C, ldc, alpha, accumulate, dtypes, slabs, colsum_a, k_top, k_unit, k_rev};
  return wgrad_run(&j, 1, (hipStream_t)stream);
}

// C[M1 + M2, N] (+)= alpha * [A1 | A2]^T . B: the column blocks of two row-per-sample matrices of the same dtype against ONE pass over B
// (M1 a multiple of 128).  The notes GRU's weight_hh gradient is [dgi[:, :1024] | dgh_n]^T . h (the r / z thirds of the gate gradients are
// shared with the input side, ptv_notes_gru_persist_bwd): two products read the 252-MB state matrix twice.
extern "C" int ptv_wgrad_cat(int M1, const void* A1, long lda1, int M2, const void* A2, long lda2, int N, int K, const void* B, long ldb,
                             float* C, long ldc, float alpha, int accumulate, int dtypes, int slabs, float* colsum_a, const int* k_top,
                             long k_unit, int k_rev, void* stream) {
  if (M1 <= 0 || M2 <= 0 || (M1 % WBM) || !A1 || !A2) return PTV_ERR_ARG;
  const Job j{M1 + M2, N, K, A1, lda1, A2, lda2, M1, B, ldb, C, ldc, alpha, accumulate, dtypes, slabs, colsum_a, k_top, k_unit, k_rev};
  return wgrad_run(&j, 1, (hipStream_t)stream);
}

// several products, one launch (+ one reduction launch): include/ptvae_hip.h
extern "C" int ptv_wgrad_batch(const ptv_wgrad_job* jobs, int njobs, void* stream) {
  if (njobs < 0 || (njobs > 0 && !jobs)) return PTV_ERR_ARG;
  if (njobs > 64) return PTV_ERR_ARG;
  Job js[64];
  for (int i = 0; i < njobs; i++) {
    const ptv_wgrad_job& q = jobs[i];
    js[i] = Job{q.M, q.N, q.K, q.A, q.lda, nullptr, 0, 0, q.B, q.ldb, q.C, q.ldc, q.alpha, q.accumulate, q.dtypes, q.slabs, q.colsum_a, q.k_top,
                q.k_unit, q.k_rev, q.seg_n, q.seg_unit, q.seg_period};
  }
  return wgrad_run(js, njobs, (hipStream_t)stream);
}

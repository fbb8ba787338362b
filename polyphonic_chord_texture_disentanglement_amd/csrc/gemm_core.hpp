// gemm_core.hpp -- LDS-tiled MFMA GEMM core for gfx950 (MI355X), shared by every dense contraction
// on the polyphonic-VAE train-step path (GRU input/hidden GEMMs, Linear layers, their dX and dW).
//
// C[m, n] (+epilogue) = sum_k A(m, k) * B(n, k)        A: M x K, B: N x K (logical)
//
// * operands live in HBM as fp32; a tile is staged global -> registers -> LDS and converted to
//   the compute type on the way (bf16: v_cvt_pk_bf16_f32, RNE; f32: unchanged)
// * compute type BF16 -> v_mfma_f32_16x16x32_bf16, F32 -> v_mfma_f32_16x16x4_f32 (exact fp32
//   FMA chain; the parity path).  fp32 accumulation in both.
// * either operand may be "K-major" in memory (element (r,k) at p[k*ld + r]): the loader
//   transposes 4x4 register blocks so the LDS image is always [row][k]  -> NT / NN / TN products
// * 256 threads = 4 wave64 in a WGM x WGN grid, each wave owns FM x (NG*FN) 16x16 accumulators
// * NG = 3 makes the B tile gather the r/z/n gate rows of a GRU weight for the same hidden units
//   (rows g*gate_stride + j), so a fused GRU-cell epilogue sees all three gates of (m, j) in one lane
// * MFMA is issued with swapped operands (B fragment as srcA) so a lane ends up with 4
//   CONSECUTIVE n of one m: epilogues use 16-byte loads / stores
// * split-K over blockIdx.z for the weight-gradient products (K = rows x steps is huge, M x N small)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ptv {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int NTHREADS = 256;

struct BF16 {
  using T = __bf16;
  static constexpr int BK = 64;    // k per LDS tile (128 B per row)
  static constexpr int LDS_LD = 64;  // unpadded; bank conflicts are removed by the XOR swizzle below
  static constexpr int CH = 8;     // elements per 16-byte chunk
};
struct F32 {
  using T = float;
  static constexpr int BK = 32;
  static constexpr int LDS_LD = 32;
  static constexpr int CH = 4;
};

struct GemmArgs {
  const float* A; long lda;   // KMAJOR_A ? A[k*lda + m] : A[m*lda + k]
  const float* B; long ldb;   // KMAJOR_B ? B[k*ldb + n] : B[n*ldb + k]
  int M, N, K;
  int k_per_split;            // K range handled by one blockIdx.z (multiple of BK); == K when no split
  long gate_stride;           // NG==3: B row of gate g, unit j is g*gate_stride + j   (N = #units)
};

// LDS image of a tile: [row][128 bytes], the eight 16-byte chunks of row r XOR-permuted by
// ((r >> 2) ^ r) & 7.  With this permutation the transposing stores of the K-major loader
// (ds_write_b64/b128 from 8 row-groups x k-quads), the row stores of the K-contiguous loader and
// the ds_read_b128 fragment reads (16 rows x 4 chunks per wave) are all bank-conflict free under
// the gfx950 lane-group rules (brute-forced in scripts/lds_swizzle_check.py).
template <class CT> __device__ __forceinline__ int swz(int row, int k) {
  return row * CT::LDS_LD + ((((k / CT::CH) ^ ((row >> 2) ^ row)) & 7) * CT::CH) + (k % CT::CH);
}

// ---------------------------------------------------------------------------------------------
// tile staging
// ---------------------------------------------------------------------------------------------
template <class CT> __device__ __forceinline__ void lds_store4(typename CT::T* dst, float a, float b, float c, float d);
template <> __device__ __forceinline__ void lds_store4<BF16>(__bf16* dst, float a, float b, float c, float d) {
  bf16x4 v; v[0] = (__bf16)a; v[1] = (__bf16)b; v[2] = (__bf16)c; v[3] = (__bf16)d;
  *reinterpret_cast<bf16x4*>(dst) = v;
}
template <> __device__ __forceinline__ void lds_store4<F32>(float* dst, float a, float b, float c, float d) {
  *reinterpret_cast<float4*>(dst) = make_float4(a, b, c, d);
}

// K-contiguous source: tile ROWS x BK.  `rowbase(r)` gives the global row for tile row r or -1.
template <class CT, int ROWS>
struct StageKC {
  static constexpr int VPR = CT::BK / 4;            // float4 per row
  static constexpr int RPP = NTHREADS / VPR;        // rows per pass
  static constexpr int NP = (ROWS + RPP - 1) / RPP;
  float4 v[NP];

  template <class RowMap>
  __device__ __forceinline__ void load(const float* __restrict__ p, long ld, int k0, int kend, bool vec_ok, RowMap rowmap) {
    const int tid = threadIdx.x;
    const int vr = tid % VPR, r0 = tid / VPR;
    const int k = k0 + vr * 4;
#pragma unroll
    for (int i = 0; i < NP; i++) {
      const int r = r0 + i * RPP;
      float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
      long g = (ROWS % RPP == 0 || r < ROWS) ? rowmap(r) : -1;
      if (g >= 0) {
        const float* q = p + g * ld + k;
        if (vec_ok && k + 3 < kend) {
          x = *reinterpret_cast<const float4*>(q);
        } else {
          if (k + 0 < kend) x.x = q[0];
          if (k + 1 < kend) x.y = q[1];
          if (k + 2 < kend) x.z = q[2];
          if (k + 3 < kend) x.w = q[3];
        }
      }
      v[i] = x;
    }
  }
  // whole tile in range and 16-byte aligned: straight-line loads, no per-element predicates (a
  // branchy loader makes hipcc wait for every load before the next branch -- nothing pipelines)
  template <class RowMap>
  __device__ __forceinline__ void load_fast(const float* __restrict__ p, long ld, int k0, RowMap rowmap) {
    const int tid = threadIdx.x;
    const int vr = tid % VPR, r0 = tid / VPR;
    const float* base = p + k0 + vr * 4;
#pragma unroll
    for (int i = 0; i < NP; i++) {
      int r = r0 + i * RPP;
      if (ROWS % RPP != 0 && r >= ROWS) r = ROWS - 1;       // tail threads re-read the last row (never stored)
      v[i] = *reinterpret_cast<const float4*>(base + rowmap(r) * ld);
    }
  }
  __device__ __forceinline__ void store(typename CT::T* s) const {
    const int tid = threadIdx.x;
    const int vr = tid % VPR, r0 = tid / VPR;
#pragma unroll
    for (int i = 0; i < NP; i++) {
      const int r = r0 + i * RPP;
      if (ROWS % RPP == 0 || r < ROWS) lds_store4<CT>(s + swz<CT>(r, vr * 4), v[i].x, v[i].y, v[i].z, v[i].w);
    }
  }
};

// K-major source: element (r, k) at p[k*ld + r].  Work item = 4 k x 4 r register block.
template <class CT, int ROWS>
struct StageKM {
  static constexpr int RG = ROWS / 4;                  // row groups
  static constexpr int KQ = CT::BK / 4;                // k quads
  static constexpr int ITEMS = RG * KQ;
  static constexpr int NP = (ITEMS + NTHREADS - 1) / NTHREADS;
  float4 v[NP][4];
  // work item -> (row group, k quad): 8 adjacent lanes cover 8 row groups (one 128-byte line per k
  // row), the next lanes walk the k quads
  static __device__ __forceinline__ int wi_rg(int w) { return (w & 7) + 8 * (w / (8 * KQ)); }
  static __device__ __forceinline__ int wi_kq(int w) { return (w >> 3) % KQ; }

  __device__ __forceinline__ void load(const float* __restrict__ p, long ld, int k0, int kend, bool vec_ok, long row0, long nrows) {
#pragma unroll
    for (int i = 0; i < NP; i++) {
      const int w = threadIdx.x + i * NTHREADS;
      const int rg = wi_rg(w), kq = wi_kq(w);
      const long r = row0 + rg * 4;
#pragma unroll
      for (int kk = 0; kk < 4; kk++) {
        const int k = k0 + kq * 4 + kk;
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((ITEMS % NTHREADS == 0 || w < ITEMS) && k < kend) {
          const float* q = p + (long)k * ld + r;
          if (vec_ok && r + 3 < nrows) {
            x = *reinterpret_cast<const float4*>(q);
          } else {
            if (r + 0 < nrows) x.x = q[0];
            if (r + 1 < nrows) x.y = q[1];
            if (r + 2 < nrows) x.z = q[2];
            if (r + 3 < nrows) x.w = q[3];
          }
        }
        v[i][kk] = x;
      }
    }
  }
  __device__ __forceinline__ void load_fast(const float* __restrict__ p, long ld, int k0, long row0) {
#pragma unroll
    for (int i = 0; i < NP; i++) {
      int w = threadIdx.x + i * NTHREADS;
      if (ITEMS % NTHREADS != 0 && w >= ITEMS) w = ITEMS - 1;
      const float* q = p + (long)(k0 + wi_kq(w) * 4) * ld + row0 + wi_rg(w) * 4;
#pragma unroll
      for (int kk = 0; kk < 4; kk++) v[i][kk] = *reinterpret_cast<const float4*>(q + kk * ld);
    }
  }
  __device__ __forceinline__ void store(typename CT::T* s) const {
#pragma unroll
    for (int i = 0; i < NP; i++) {
      const int w = threadIdx.x + i * NTHREADS;
      if (ITEMS % NTHREADS == 0 || w < ITEMS) {
        const int rg = wi_rg(w), kq = wi_kq(w);
        lds_store4<CT>(s + swz<CT>(rg * 4 + 0, kq * 4), v[i][0].x, v[i][1].x, v[i][2].x, v[i][3].x);
        lds_store4<CT>(s + swz<CT>(rg * 4 + 1, kq * 4), v[i][0].y, v[i][1].y, v[i][2].y, v[i][3].y);
        lds_store4<CT>(s + swz<CT>(rg * 4 + 2, kq * 4), v[i][0].z, v[i][1].z, v[i][2].z, v[i][3].z);
        lds_store4<CT>(s + swz<CT>(rg * 4 + 3, kq * 4), v[i][0].w, v[i][1].w, v[i][2].w, v[i][3].w);
      }
    }
  }
};

// ---------------------------------------------------------------------------------------------
// MFMA over one LDS tile.  acc[fm][fn] : lane holds m = fm*16 + (lane&15), n = fn*16 + (lane>>4)*4 + reg
// ---------------------------------------------------------------------------------------------
template <class CT, int FM, int FNT> struct TileMma;

template <int FM, int FNT>
struct TileMma<BF16, FM, FNT> {
  // a_rows / b_rows: LDS pointers to this wave's first A / B row for each fragment
  template <class BRow>
  static __device__ __forceinline__ void run(const __bf16* As, int a_row0, const __bf16* Bs, BRow b_row, f32x4 (&acc)[FM][FNT]) {
    const int lane = threadIdx.x & 63;
    const int r = lane & 15, kq = (lane >> 4) * 8;
#pragma unroll
    for (int ks = 0; ks < BF16::BK; ks += 32) {
      bf16x8 a[FM], b[FNT];
#pragma unroll
      for (int i = 0; i < FM; i++) a[i] = *reinterpret_cast<const bf16x8*>(As + swz<BF16>(a_row0 + i * 16 + r, ks + kq));
#pragma unroll
      for (int j = 0; j < FNT; j++) b[j] = *reinterpret_cast<const bf16x8*>(Bs + swz<BF16>(b_row(j) + r, ks + kq));
#pragma unroll
      for (int i = 0; i < FM; i++)
#pragma unroll
        for (int j = 0; j < FNT; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
    }
  }
};

template <int FM, int FNT>
struct TileMma<F32, FM, FNT> {
  template <class BRow>
  static __device__ __forceinline__ void run(const float* As, int a_row0, const float* Bs, BRow b_row, f32x4 (&acc)[FM][FNT]) {
    const int lane = threadIdx.x & 63;
    const int r = lane & 15, kq = (lane >> 4) * 4;
#pragma unroll
    for (int ks = 0; ks < F32::BK; ks += 16) {
      float4 a[FM], b[FNT];
#pragma unroll
      for (int i = 0; i < FM; i++) a[i] = *reinterpret_cast<const float4*>(As + swz<F32>(a_row0 + i * 16 + r, ks + kq));
#pragma unroll
      for (int j = 0; j < FNT; j++) b[j] = *reinterpret_cast<const float4*>(Bs + swz<F32>(b_row(j) + r, ks + kq));
      // lane group g=(lane>>4) feeds k = ks + 4g + e on MFMA e: a consistent k permutation of A and B
#pragma unroll
      for (int i = 0; i < FM; i++)
#pragma unroll
        for (int j = 0; j < FNT; j++) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j].x, a[i].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j].y, a[i].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j].z, a[i].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j].w, a[i].w, acc[i][j], 0, 0, 0);
        }
    }
  }
};

// ---------------------------------------------------------------------------------------------
// the kernel body.  Epi::apply(ep, acc, m0, j0, M, N) is called once per wave with
//   m0 = first m of the wave's tile, j0 = first unit (n) of the wave's tile;
//   acc[fm][g*FN + fn] is gate g, fragment (fm, fn).
// ---------------------------------------------------------------------------------------------
template <class CT, int BM, int BN, int WGM, int WGN, int NG, bool KMAJOR_A, bool KMAJOR_B, class Epi>
__device__ __forceinline__ void gemm_body(const GemmArgs& g, const typename Epi::Params& ep) {
  static_assert(WGM * WGN == 4, "4 waves");
  static_assert(!(NG == 3 && KMAJOR_B), "gate gather needs K-contiguous weights");
  using T = typename CT::T;
  constexpr int WTM = BM / WGM, WTN = BN / WGN;
  constexpr int FM = WTM / 16, FN = WTN / 16;
  constexpr int BROWS = NG * BN;
  __shared__ __attribute__((aligned(16))) T As[BM * CT::LDS_LD];
  __shared__ __attribute__((aligned(16))) T Bs[BROWS * CT::LDS_LD];

  const int m_blk = blockIdx.y * BM, n_blk = blockIdx.x * BN;
  const int kbeg = blockIdx.z * g.k_per_split;
  const int kend = min(g.K, kbeg + g.k_per_split);
  const int wave = threadIdx.x >> 6;
  const int wm = wave / WGN, wn = wave % WGN;

  const bool vecA = ((g.lda & 3) == 0) && ((reinterpret_cast<uintptr_t>(g.A) & 15) == 0);
  const bool vecB = ((g.ldb & 3) == 0) && ((reinterpret_cast<uintptr_t>(g.B) & 15) == 0);

  auto rowmapA = [&](int r) -> long { long m = m_blk + r; return m < g.M ? m : -1; };
  auto rowmapB = [&](int r) -> long {
    const int gate = r / BN, jj = r % BN;
    long n = n_blk + jj;
    return n < g.N ? (long)gate * g.gate_stride + n : -1;
  };

  typename std::conditional<KMAJOR_A, StageKM<CT, BM>, StageKC<CT, BM>>::type sa;
  typename std::conditional<KMAJOR_B, StageKM<CT, BROWS>, StageKC<CT, BROWS>>::type sb;

  f32x4 acc[FM][NG * FN];
#pragma unroll
  for (int i = 0; i < FM; i++)
#pragma unroll
    for (int j = 0; j < NG * FN; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const bool fullA = vecA && (m_blk + BM <= g.M);          // block-uniform: rows of the tile all valid
  const bool fullB = vecB && (n_blk + BN <= g.N);
  auto rowfastA = [&](int r) -> long { return m_blk + r; };
  auto rowfastB = [&](int r) -> long { return (long)(r / BN) * g.gate_stride + n_blk + (r % BN); };
  auto load_tiles = [&](int k0) {
    const bool kfull = k0 + CT::BK <= kend;
    if (fullA && kfull) {
      if constexpr (KMAJOR_A) sa.load_fast(g.A, g.lda, k0, m_blk);
      else sa.load_fast(g.A, g.lda, k0, rowfastA);
    } else {
      if constexpr (KMAJOR_A) sa.load(g.A, g.lda, k0, kend, vecA, m_blk, g.M);
      else sa.load(g.A, g.lda, k0, kend, vecA, rowmapA);
    }
    if (fullB && kfull) {
      if constexpr (KMAJOR_B) sb.load_fast(g.B, g.ldb, k0, n_blk);
      else sb.load_fast(g.B, g.ldb, k0, rowfastB);
    } else {
      if constexpr (KMAJOR_B) sb.load(g.B, g.ldb, k0, kend, vecB, n_blk, g.N);
      else sb.load(g.B, g.ldb, k0, kend, vecB, rowmapB);
    }
  };

  if (kbeg < kend) load_tiles(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += CT::BK) {
    __syncthreads();              // previous tile fully consumed
    sa.store(As);
    sb.store(Bs);
    __syncthreads();
    if (k0 + CT::BK < kend) load_tiles(k0 + CT::BK);     // prefetch next tile under the MFMAs
    TileMma<CT, FM, NG * FN>::run(As, wm * WTM, Bs,
                                  [&](int j) { return (j / FN) * BN + wn * WTN + (j % FN) * 16; }, acc);
  }
  Epi::template apply<FM, FN, NG>(ep, acc, m_blk + wm * WTM, n_blk + wn * WTN, g.M, g.N);
}

}  // namespace ptv

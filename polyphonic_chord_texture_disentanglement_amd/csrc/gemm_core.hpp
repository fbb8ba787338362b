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
#include <type_traits>

namespace ptv {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int NTHREADS = 256;

// 4 consecutive elements of an fp32 or bf16 array (runtime dtype flag; epilogue traffic in bf16 mode)
__device__ __forceinline__ float4 ld4f(const void* p, long i, bool bf) {
  if (bf) {
    const bf16x4 v = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(p) + i);
    return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
  }
  return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p) + i);
}
__device__ __forceinline__ void st4f(void* p, long i, bool bf, float a, float b, float c, float d) {
  if (bf) {
    bf16x4 v; v[0] = (__bf16)a; v[1] = (__bf16)b; v[2] = (__bf16)c; v[3] = (__bf16)d;
    *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(p) + i) = v;
  } else {
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(p) + i) = make_float4(a, b, c, d);
  }
}
__device__ __forceinline__ float ld1f(const void* p, long i, bool bf) {
  return bf ? (float)reinterpret_cast<const __bf16*>(p)[i] : reinterpret_cast<const float*>(p)[i];
}
__device__ __forceinline__ void st1f(void* p, long i, bool bf, float v) {
  if (bf) reinterpret_cast<__bf16*>(p)[i] = (__bf16)v; else reinterpret_cast<float*>(p)[i] = v;
}

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
  const void* A; long lda;    // KMAJOR_A ? A[k*lda + m] : A[m*lda + k]   (fp32, or bf16 when the kernel's SA is set)
  const void* B; long ldb;    // KMAJOR_B ? B[k*ldb + n] : B[n*ldb + k]
  int M, N, K;
  int k_per_split;            // K range handled by one blockIdx.z (multiple of BK); == K when no split
  long gate_stride;           // NG==3: B row of gate g, unit j is g*gate_stride + j   (N = #units)
  const int* m_top; long m_unit;   // rows of A from (*m_top + 1) * m_unit on are known to be zero (or null): those tiles skip the K loop
  int prio;                   // plain products: raise the wave priority (a launch of the latency chain next to sibling-stream products)
  // row segments of A (or null): the rows are units of seg_unit rows (a note step's decoder rows in length order) of which only the first
  // seg_n[unit % seg_period] (device ints, multiples of 128) hold anything -- the other row tiles are dead like the ones beyond m_top
  const int* seg_n = nullptr; int seg_unit = 0, seg_period = 0;
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
// one 16-byte LDS chunk = CT::CH consecutive k of one row.  SB = the HBM source already holds bf16
// (activations / gradients that only ever feed MFMA operands are kept as bf16 copies in bf16 mode:
// half the operand traffic, no conversion while staging); otherwise fp32 converted on the way.
template <bool SB> struct SrcT { using T = float; };
template <> struct SrcT<true> { using T = __bf16; };

template <class CT, bool SB> struct Chunk;
template <> struct Chunk<BF16, false> {
  float4 lo, hi;                                   // k .. k+7
  __device__ __forceinline__ void zero() { lo = hi = make_float4(0.f, 0.f, 0.f, 0.f); }
  __device__ __forceinline__ void set(int e, float x) { if (e < 4) (&lo.x)[e] = x; else (&hi.x)[e - 4] = x; }
  __device__ __forceinline__ void load16(const float* q) { lo = *reinterpret_cast<const float4*>(q); hi = *reinterpret_cast<const float4*>(q + 4); }
  __device__ __forceinline__ void store(__bf16* d) const {
    bf16x8 v;
    v[0] = (__bf16)lo.x; v[1] = (__bf16)lo.y; v[2] = (__bf16)lo.z; v[3] = (__bf16)lo.w;
    v[4] = (__bf16)hi.x; v[5] = (__bf16)hi.y; v[6] = (__bf16)hi.z; v[7] = (__bf16)hi.w;
    *reinterpret_cast<bf16x8*>(d) = v;
  }
};
template <> struct Chunk<BF16, true> {
  bf16x8 v;
  __device__ __forceinline__ void zero() { for (int e = 0; e < 8; e++) v[e] = (__bf16)0.f; }
  __device__ __forceinline__ void set(int e, __bf16 x) { v[e] = x; }
  __device__ __forceinline__ void load16(const __bf16* q) { v = *reinterpret_cast<const bf16x8*>(q); }
  __device__ __forceinline__ void store(__bf16* d) const { *reinterpret_cast<bf16x8*>(d) = v; }
};
template <> struct Chunk<F32, false> {
  float4 lo;
  __device__ __forceinline__ void zero() { lo = make_float4(0.f, 0.f, 0.f, 0.f); }
  __device__ __forceinline__ void set(int e, float x) { (&lo.x)[e] = x; }
  __device__ __forceinline__ void load16(const float* q) { lo = *reinterpret_cast<const float4*>(q); }
  __device__ __forceinline__ void store(float* d) const { *reinterpret_cast<float4*>(d) = lo; }
};

// K-contiguous source: tile ROWS x BK, one 16-byte LDS chunk per thread per pass.
//   8 threads cover a row (BK = 8 chunks), RPP = 32 rows per pass.
template <class CT, int ROWS, bool SB>
struct StageKC {
  using S = typename SrcT<SB>::T;
  static constexpr int CPR = 8;                     // chunks per row
  static constexpr int RPP = NTHREADS / CPR;        // 32 rows per pass
  static constexpr int NP = ROWS / RPP;
  static_assert(ROWS % RPP == 0, "tile rows must be a multiple of 32");
  Chunk<CT, SB> v[NP];

  // general path: bounds-checked, any alignment
  template <class RowMap>
  __device__ __forceinline__ void load(const S* __restrict__ p, long ld, int k0, int kend, bool vec_ok, RowMap rowmap) {
    const int tid = threadIdx.x;
    const int c = tid % CPR, r0 = tid / CPR;
    const int k = k0 + c * CT::CH;
#pragma unroll
    for (int i = 0; i < NP; i++) {
      v[i].zero();
      const long g = rowmap(r0 + i * RPP);
      if (g >= 0) {
        const S* q = p + g * ld + k;
        if (vec_ok && k + CT::CH <= kend) v[i].load16(q);
        else {
#pragma unroll
          for (int e = 0; e < CT::CH; e++) if (k + e < kend) v[i].set(e, q[e]);
        }
      }
    }
  }
  // fast path: whole tile in range and 16-byte aligned -> straight-line loads, block-uniform row bases
  // (SGPR) + one 32-bit per-thread offset.  (A branchy loader makes hipcc wait for every load before
  // the next branch: nothing pipelines.)
  template <class RowU>
  __device__ __forceinline__ void load_fast(const S* __restrict__ p, long ld, int k0, RowU rowu) {
    const int tid = threadIdx.x;
    const unsigned toff = (unsigned)((tid / CPR) * ld + (tid % CPR) * CT::CH);
#pragma unroll
    for (int i = 0; i < NP; i++) v[i].load16(p + rowu(i * RPP) * ld + k0 + toff);
  }
  __device__ __forceinline__ void store(typename CT::T* s) const {
    const int tid = threadIdx.x;
    const int c = tid % CPR, r0 = tid / CPR;
#pragma unroll
    for (int i = 0; i < NP; i++) v[i].store(s + swz<CT>(r0 + i * RPP, c * CT::CH));
  }
};

// K-major fp32 source: element (r, k) at p[k*ld + r].  Work item = CH k x 4 r register block -> four
// 16-byte LDS chunks (rows r..r+3).  8 adjacent lanes cover 8 row groups (one 128-byte line per k row),
// the next lanes walk the 8 k-chunks.
template <class CT, int ROWS, bool SB>
struct StageKM {
  static constexpr int RG = ROWS / 4;                  // row groups
  static constexpr int ITEMS = RG * 8;                 // x 8 k-chunks
  static constexpr int NP = (ITEMS + NTHREADS - 1) / NTHREADS;
  static_assert(ITEMS % NTHREADS == 0 || NTHREADS % ITEMS == 0, "item count");
  float4 v[NP][CT::CH];
  // item w -> row group (w & 7) + 8 * (w / 64), k-chunk (w >> 3) & 7
  static __device__ __forceinline__ int item() { return ITEMS < NTHREADS ? (int)(threadIdx.x % ITEMS) : (int)threadIdx.x; }

  __device__ __forceinline__ void load(const float* __restrict__ p, long ld, int k0, int kend, bool vec_ok, long row0, long nrows) {
#pragma unroll
    for (int i = 0; i < NP; i++) {
      const int w = item() + i * NTHREADS;
      const int rg = (w & 7) + 8 * (w >> 6), kc = (w >> 3) & 7;
      const long r = row0 + rg * 4;
#pragma unroll
      for (int kk = 0; kk < CT::CH; kk++) {
        const int k = k0 + kc * CT::CH + kk;
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < kend) {
          const float* q = p + (long)k * ld + r;
          if (vec_ok && r + 3 < nrows) x = *reinterpret_cast<const float4*>(q);
          else {
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
    const int w0 = item();
    const unsigned toff = (unsigned)((((w0 >> 3) & 7) * CT::CH) * ld + ((w0 & 7) + 8 * (w0 >> 6)) * 4);
#pragma unroll
    for (int i = 0; i < NP; i++) {
      const float* ub = p + (long)k0 * ld + row0 + i * (NTHREADS / 64) * 8 * 4;      // block-uniform
#pragma unroll
      for (int kk = 0; kk < CT::CH; kk++) v[i][kk] = *reinterpret_cast<const float4*>(ub + kk * ld + toff);
    }
  }
  __device__ __forceinline__ void store(typename CT::T* s) const {
    if (ITEMS < NTHREADS && threadIdx.x >= ITEMS) return;
#pragma unroll
    for (int i = 0; i < NP; i++) {
      const int w = item() + i * NTHREADS;
      const int rg = (w & 7) + 8 * (w >> 6), kc = (w >> 3) & 7;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        Chunk<CT, false> c;
#pragma unroll
        for (int kk = 0; kk < CT::CH; kk++) c.set(kk, (&v[i][kk].x)[j]);
        c.store(s + swz<CT>(rg * 4 + j, kc * CT::CH));
      }
    }
  }
};

// K-major bf16 source: work item = 4 k x 8 r block of bf16 (four 16-byte loads, one per k row) -> eight
// 8-byte LDS half-chunks (rows r..r+7, 4 consecutive k).  LW adjacent lanes cover LW row groups (up to one
// 128-byte line per k row), the next lanes walk the 16 k-quads; ROWS = 128 keeps all 256 threads loading.
template <int ROWS>
struct StageKM<BF16, ROWS, true> {
  static constexpr int RG = ROWS / 8;
  static constexpr int LW = RG < 8 ? RG : 8;
  static constexpr int ITEMS = RG * 16;                // x 16 k-quads
  static_assert(ITEMS <= NTHREADS, "one item per thread");
  bf16x8 v[4];
  static __device__ __forceinline__ int item() { return (int)(threadIdx.x % ITEMS); }
  static __device__ __forceinline__ int wi_rg(int w) { return (w % LW) + LW * (w / (LW * 16)); }
  static __device__ __forceinline__ int wi_kq(int w) { return (w / LW) % 16; }

  __device__ __forceinline__ void load(const __bf16* __restrict__ p, long ld, int k0, int kend, bool vec_ok, long row0, long nrows) {
    const int w = item();
    const long r = row0 + wi_rg(w) * 8;
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
      const int k = k0 + wi_kq(w) * 4 + kk;
      bf16x8 x;
#pragma unroll
      for (int e = 0; e < 8; e++) x[e] = (__bf16)0.f;
      if (k < kend) {
        const __bf16* q = p + (long)k * ld + r;
        if (vec_ok && r + 7 < nrows) x = *reinterpret_cast<const bf16x8*>(q);
        else {
#pragma unroll
          for (int e = 0; e < 8; e++) if (r + e < nrows) x[e] = q[e];
        }
      }
      v[kk] = x;
    }
  }
  __device__ __forceinline__ void load_fast(const __bf16* __restrict__ p, long ld, int k0, long row0) {
    const int w = item();
    const unsigned toff = (unsigned)((wi_kq(w) * 4) * ld + wi_rg(w) * 8);
    const __bf16* ub = p + (long)k0 * ld + row0;
#pragma unroll
    for (int kk = 0; kk < 4; kk++) v[kk] = *reinterpret_cast<const bf16x8*>(ub + kk * ld + toff);
  }
  __device__ __forceinline__ void store(__bf16* s) const {
    if (ITEMS < NTHREADS && threadIdx.x >= ITEMS) return;
    const int w = item();
    const int rg = wi_rg(w), kq = wi_kq(w);
#pragma unroll
    for (int j = 0; j < 8; j++) {
      bf16x4 c;
#pragma unroll
      for (int kk = 0; kk < 4; kk++) c[kk] = v[kk][j];
      *reinterpret_cast<bf16x4*>(s + swz<BF16>(rg * 8 + j, kq * 4)) = c;
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
// RowStage: accumulator fragments -> wave-private LDS -> row-contiguous lanes.
// In the MFMA C layout a lane owns 4 units of ONE row and the 16 lanes of a column group sit on 16
// different rows, so every global access of an epilogue touches 16 rows x 32-64 bytes; measured on the
// GRU cell's operand mix that pattern streams at 3.3 TB/s where row-contiguous lanes reach 4.2 (8 lanes
// per row) to 4.8 TB/s (16 lanes per row) -- scripts/micro/epi_pattern.hip.  Epilogues therefore pass each
// 16-row fragment row through LDS: afterwards W/4 consecutive lanes cover one row's W units (W = units of
// the wave tile, 32 or 64) and a wave instruction touches 8 or 4 rows in whole 64-256 byte runs.
//   * C-layout ds_write_b128 (8 contiguous lanes = 8 rows): row stride S = NG*W + 4 floats, S/4 odd -> the
//     8 rows land on 8 disjoint 4-bank groups: conflict-free.
//   * read-back ds_read_b128: each hardware lane group ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32 --
//     MI355X_MICROARCH.md LDS table) is mapped onto one whole row (W = 64), or onto two rows 8 apart
//     (W = 32: 8*S = 32 mod 64 banks, the halves tile the 64 banks): conflict-free.
// ---------------------------------------------------------------------------------------------
template <int NG, int W>
struct RowStage {
  static_assert(W == 32 || W == 64, "wave tiles of 32 or 64 units");
  static constexpr int NF = NG * W / 16;                  // fragments per staged row
  static constexpr int S = NG * W + 4;                    // floats per staged row
  static constexpr int WAVE_BYTES = 16 * S * 4;
  static constexpr int PASSES = W == 64 ? 4 : 2;          // wave instructions per 16-row fragment row
  static __device__ __forceinline__ float* base(char* lds) { return reinterpret_cast<float*>(lds) + (threadIdx.x >> 6) * (16 * S); }
  // acc row i of the wave (fragments g*FN + f) -> LDS; st[row][g*W + unit]
  static __device__ __forceinline__ void put(float* st, const f32x4 (&a)[NF]) {
    const int lane = threadIdx.x & 63;
    float* w = st + (lane & 15) * S + (lane >> 4) * 4;
#pragma unroll
    for (int f = 0; f < NF; f++) *reinterpret_cast<f32x4*>(w + f * 16) = a[f];
    __builtin_amdgcn_wave_barrier();
  }
  // lane -> row (0..15) of pass ps and first unit (0..W-4) of its 4-unit chunk
  static __device__ __forceinline__ int row(int ps) {
    const int lane = threadIdx.x & 63, q = (lane >> 2) & 7;
    const int grp = (lane >> 5) * 2 + ((0x96 >> q) & 1);          // hardware b128 lane group 0..3
    return W == 64 ? ps * 4 + grp : ps * 4 + grp + 8 * (q >> 2);
  }
  static __device__ __forceinline__ int unit() {
    const int lane = threadIdx.x & 63, q = (lane >> 2) & 7;
    return W == 64 ? ((q >> 1) * 4 + (lane & 3)) * 4 : (((q >> 1) & 1) * 4 + (lane & 3)) * 4;
  }
  static __device__ __forceinline__ f32x4 get(const float* st, int r, int col) {
    return *reinterpret_cast<const f32x4*>(st + r * S + col);
  }
};

// ---------------------------------------------------------------------------------------------
// the kernel body.  Epi::apply(ep, acc, m0, j0, M, N) is called once per wave with
//   m0 = first m of the wave's tile, j0 = first unit (n) of the wave's tile;
//   acc[fm][g*FN + fn] is gate g, fragment (fm, fn).
// ---------------------------------------------------------------------------------------------
template <class CT, int BM, int BN, int WGM, int WGN, int NG, bool KMAJOR_A, bool KMAJOR_B, class Epi, bool SA = false, bool SB = false, int PF = 1>
__device__ __forceinline__ void gemm_body(const GemmArgs& g, const typename Epi::Params& ep) {
  static_assert(WGM * WGN == 4, "4 waves");
  static_assert(!(NG == 3 && KMAJOR_B), "gate gather needs K-contiguous weights");
  using T = typename CT::T;
  constexpr int WTM = BM / WGM, WTN = BN / WGN;
  constexpr int FM = WTM / 16, FN = WTN / 16;
  constexpr int BROWS = NG * BN;
  // one LDS arena: the operand tiles of the main loop, reused by epilogues that stage accumulators (RowStage)
  constexpr int MAIN_BYTES = (BM + BROWS) * CT::LDS_LD * (int)sizeof(T);
  constexpr int EPI_BYTES = Epi::template lds_bytes<FM, FN, NG>();
  __shared__ __attribute__((aligned(16))) char smem[MAIN_BYTES > EPI_BYTES ? MAIN_BYTES : EPI_BYTES];
  T* As = reinterpret_cast<T*>(smem);
  T* Bs = As + BM * CT::LDS_LD;

  // Block -> (m tile, n tile, K split), XCD-aware.  Blocks are dispatched round-robin over the 8 XCDs
  // (block id % 8; speed only, never correctness -- MI355X_MICROARCH.md), each with a private 4 MB L2:
  //  * no K split: XCD x owns the m tiles = x (mod 8) and walks them n-fastest, so the n tiles that
  //    re-read one A row panel run back-to-back on ONE XCD (the big activation operand crosses the
  //    fabric once and is still in that L2 when its next n tile starts)
  //  * K splits (weight gradients): a whole split -- every tile that re-reads the same K slab of both
  //    operands -- is pinned to one XCD
  const int ntn = gridDim.x, ntm = gridDim.y, nsp = gridDim.z;
  const int bid = blockIdx.x + ntn * (blockIdx.y + ntm * blockIdx.z);
  const int per = ntm * ntn;
  int mt, nt, split;
  if (nsp > 1 && (nsp & 7) == 0) {
    const int slot = bid >> 3;
    split = (bid & 7) + 8 * (slot / per);
    const int tile = slot % per;
    mt = tile / ntn; nt = tile % ntn;
  } else if (nsp == 1 && g.M >= g.N && (ntm & 7) == 0) {       // A is the big operand: pin its row panels
    const int slot = bid >> 3;
    split = 0;
    mt = (slot / ntn) * 8 + (bid & 7); nt = slot % ntn;
  } else if (nsp == 1 && g.M < g.N && (ntn & 7) == 0) {        // B (weights) is the big operand: pin its panels
    const int slot = bid >> 3;
    split = 0;
    nt = (slot / ntm) * 8 + (bid & 7); mt = slot % ntm;
  } else {
    split = bid / per;
    const int tile = bid % per;
    mt = tile / ntn; nt = tile % ntn;
  }
  const int m_blk = mt * BM, n_blk = nt * BN;
  const int kbeg = split * g.k_per_split;
  // a row tile that lies in the part of A its producer declared zero contributes nothing: empty K range, the epilogue still runs
  // (C = bias / unchanged)
  const bool dead = (g.m_top != nullptr && m_blk >= ((long)*g.m_top + 1) * g.m_unit) ||
                    (g.seg_n != nullptr && (m_blk % g.seg_unit) >= g.seg_n[(m_blk / g.seg_unit) % g.seg_period]);
  if (dead && Epi::dead_is_noop(ep)) return;                    // C += 0: nothing to read or write (block-uniform, before any barrier)
  const int kend = dead ? kbeg : min(g.K, kbeg + g.k_per_split);
  const int wave = threadIdx.x >> 6;
  const int wm = wave / WGN, wn = wave % WGN;

  static_assert(!(SA || SB) || std::is_same<CT, BF16>::value, "bf16 sources feed the bf16 MFMA path only");
  using TA = typename SrcT<SA>::T;
  using TB = typename SrcT<SB>::T;
  const TA* Ap = reinterpret_cast<const TA*>(g.A);
  const TB* Bp = reinterpret_cast<const TB*>(g.B);
  const bool vecA = ((g.lda & (SA ? 7 : 3)) == 0) && ((reinterpret_cast<uintptr_t>(g.A) & 15) == 0);
  const bool vecB = ((g.ldb & (SB ? 7 : 3)) == 0) && ((reinterpret_cast<uintptr_t>(g.B) & 15) == 0);

  auto rowmapA = [&](int r) -> long { long m = m_blk + r; return m < g.M ? m : -1; };
  auto rowmapB = [&](int r) -> long {
    const int gate = r / BN, jj = r % BN;
    long n = n_blk + jj;
    return n < g.N ? (long)gate * g.gate_stride + n : -1;
  };

  // PF register stages: global loads for PF tiles are in flight ahead of the MFMAs (the small-M
  // recurrent steps run one block per CU, so load latency is only hidden by depth, not by occupancy)
  typename std::conditional<KMAJOR_A, StageKM<CT, BM, SA>, StageKC<CT, BM, SA>>::type sa[PF];
  typename std::conditional<KMAJOR_B, StageKM<CT, BROWS, SB>, StageKC<CT, BROWS, SB>>::type sb[PF];

  // epilogue operands an Epi wants in flight under the main loop (empty for most)
  typename Epi::template Pre<FM, FN, NG> pre;
  Epi::template prefetch<FM, FN, NG>(ep, pre, m_blk + wm * WTM, n_blk + wn * WTN, g.M, g.N);

  f32x4 acc[FM][NG * FN];
#pragma unroll
  for (int i = 0; i < FM; i++)
#pragma unroll
    for (int j = 0; j < NG * FN; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const bool fullA = vecA && (m_blk + BM <= g.M);          // block-uniform: rows of the tile all valid
  const bool fullB = vecB && (n_blk + BN <= g.N);
  auto rowfastA = [&](int r) -> long { return m_blk + r; };                  // r = i*RPP: block-uniform
  auto rowfastB = [&](int r) -> long { return (long)(r / BN) * g.gate_stride + n_blk + (r % BN); };
  static_assert(BN % 32 == 0 && BM % 32 == 0, "row passes must not straddle a gate block");
  auto load_tiles = [&](auto& sa_, auto& sb_, int k0) {
    const bool kfull = k0 + CT::BK <= kend;
    if (fullA && kfull) {
      if constexpr (KMAJOR_A) sa_.load_fast(Ap, g.lda, k0, m_blk);
      else sa_.load_fast(Ap, g.lda, k0, rowfastA);
    } else {
      if constexpr (KMAJOR_A) sa_.load(Ap, g.lda, k0, kend, vecA, m_blk, g.M);
      else sa_.load(Ap, g.lda, k0, kend, vecA, rowmapA);
    }
    if (fullB && kfull) {
      if constexpr (KMAJOR_B) sb_.load_fast(Bp, g.ldb, k0, n_blk);
      else sb_.load_fast(Bp, g.ldb, k0, rowfastB);
    } else {
      if constexpr (KMAJOR_B) sb_.load(Bp, g.ldb, k0, kend, vecB, n_blk, g.N);
      else sb_.load(Bp, g.ldb, k0, kend, vecB, rowmapB);
    }
  };
  auto fast_tiles = [&](auto& sa_, auto& sb_, int k0) {          // no predicates at all (steady state)
    if constexpr (KMAJOR_A) sa_.load_fast(Ap, g.lda, k0, m_blk);
    else sa_.load_fast(Ap, g.lda, k0, rowfastA);
    if constexpr (KMAJOR_B) sb_.load_fast(Bp, g.ldb, k0, n_blk);
    else sb_.load_fast(Bp, g.ldb, k0, rowfastB);
  };
  auto consume = [&](auto& sa_, auto& sb_) {
    __syncthreads();              // previous tile fully consumed
    sa_.store(As);
    sb_.store(Bs);
    __syncthreads();
  };
  auto mma = [&]() {
    TileMma<CT, FM, NG * FN>::run(As, wm * WTM, Bs,
                                  [&](int j) { return (j / FN) * BN + wn * WTN + (j % FN) * 16; }, acc);
  };

  int k0 = kbeg;
  {
    // pipeline with a branch-free steady state: every stage's loads are PF tiles old when consumed, and the
    // straight-line body lets hipcc wait with a counted vmcnt instead of draining the queue
    if (fullA && fullB && kbeg + 2 * PF * CT::BK <= kend) {
#pragma unroll
      for (int p = 0; p < PF; p++) fast_tiles(sa[p], sb[p], k0 + p * CT::BK);
      for (; k0 + 2 * PF * CT::BK <= kend; k0 += PF * CT::BK) {
#pragma unroll
        for (int p = 0; p < PF; p++) {
          consume(sa[p], sb[p]);
          fast_tiles(sa[p], sb[p], k0 + (PF + p) * CT::BK);
          mma();
        }
      }
#pragma unroll
      for (int p = 0; p < PF; p++) {                       // drain the PF tiles still in registers
        consume(sa[p], sb[p]);
        mma();
      }
      k0 += PF * CT::BK;
    }
  }
  if (k0 < kend) load_tiles(sa[0], sb[0], k0);
  for (; k0 < kend; k0 += CT::BK) {
    consume(sa[0], sb[0]);
    if (k0 + CT::BK < kend) load_tiles(sa[0], sb[0], k0 + CT::BK);     // prefetch next tile under the MFMAs
    mma();
  }
  if constexpr (EPI_BYTES > 0) __syncthreads();            // every wave is done with the operand tiles
  Epi::template apply<FM, FN, NG>(ep, acc, pre, m_blk + wm * WTM, n_blk + wn * WTN, g.M, g.N, split, smem);
}

}  // namespace ptv

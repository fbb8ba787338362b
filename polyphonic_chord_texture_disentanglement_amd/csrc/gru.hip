// gru.hip -- fused GRU cell kernels (forward step and BPTT step) on the MFMA core, and the
// sequence drivers that launch one step kernel per recurrent step on a caller-supplied stream.
//
// Follows torch.nn.GRU cell semantics used by every recurrent layer of the reference
// (ptvae.py:14,39,103,262-286):
//   r = s(gi_r + W_hr h + b_hr)  z = s(gi_z + W_hz h + b_hz)  n = tanh(gi_n + r*(W_hn h + b_hn))
//   h' = (1-z)*n + z*h            (gi = W_i x + b_i is produced by a batched input-side GEMM)
#include <type_traits>
#include "common.hpp"
#include "gemm_core.hpp"
#include "prof.hpp"
#include "../../include/ptvae_hip.h"
#include "../../include/ptvae_hip_debug.h"

namespace ptv {

// ---------------------------------------------------------------------------------------------
// forward step epilogue: acc[.][g*FN+fn] = (h_prev . W_h{g}^T) for unit j of gate g
// ---------------------------------------------------------------------------------------------
struct GruFwdParams {
  const float* hprev; long ld_hprev;
  const void* gi; long ld_gi;         // [M, 3H] input-side pre-activations (b_ih included); fp32 or bf16
  const void* gi2; long ld_gi2;       // optional second addend (e.g. per-step token part), may be null
  const float* bhh;                   // [3H]
  float* hout; long ld_hout;
  __bf16* hout16;                     // optional bf16 copy of the new state [M,H] dense (MFMA operand for later products)
  void* gates; long plane;            // saved (r, z, n, hn) as 4 planes [M,H]; null = do not save; fp32 or bf16
  const int* lengths; int t;          // packed-sequence mask: row m is updated iff t < lengths[m]
  const int* gi_idx;                  // optional: row m reads gi row gi_idx[m] (token-indexed gate table)
  int H;
  int flags;                          // PTV_GRU_*_BF16
};

// FAST: the teacher-forced bf16-storage case (gi and gates bf16, no gi2 / gi_idx), dtype branches resolved at
// compile time and the per-element operands (gi, h_prev) PREFETCHED into registers before the main loop:
// the cell is a streaming kernel with a short K loop, and its HBM rate is set by how many bytes each CU
// keeps in flight -- operands requested at kernel start arrive under the MFMAs instead of being waited for
// one cell at a time in the epilogue.
template <int MODE>   // 0 generic, 1 FAST, 2 FAST with a bf16 gi2 addend
struct EpiGruFwdT {
  template <class P> static __device__ __forceinline__ bool dead_is_noop(const P&) { return false; }   // (GemmArgs::m_top is never set here)
  using Params = GruFwdParams;
  static constexpr bool FAST = MODE > 0, G2 = MODE == 2;
  // wave tiles 32 or 64 units wide go through RowStage (row-contiguous lanes); narrower ones stay in the C layout.
  // A lane then owns NC cells (row, 4 units) per 16-row fragment row i.
  template <int FN> static constexpr bool staged() { return FN == 2 || FN == 4; }
  template <int FN> static constexpr int ncell() { return staged<FN>() ? RowStage<3, (staged<FN>() ? FN * 16 : 64)>::PASSES : FN; }
  template <int FM, int FN, int NG> static constexpr int lds_bytes() { return staged<FN>() ? 4 * RowStage<3, (staged<FN>() ? FN * 16 : 64)>::WAVE_BYTES : 0; }
  template <int FN> static __device__ __forceinline__ void coord(int i, int c, int& row, int& u) {
    if constexpr (staged<FN>()) {
      using RS = RowStage<3, FN * 16>;
      row = i * 16 + RS::row(c); u = RS::unit();
    } else {
      const int lane = threadIdx.x & 63;
      row = i * 16 + (lane & 15); u = c * 16 + (lane >> 4) * 4;
    }
  }
  template <int FM, int FN, int NG> struct Pre {
    bf16x4 g[FAST ? FM : 1][FAST ? ncell<FN>() : 1][3];
    bf16x4 g2[G2 ? FM : 1][G2 ? ncell<FN>() : 1][3];
    float4 hp[FAST ? FM : 1][FAST ? ncell<FN>() : 1];
  };
  template <int FM, int FN, int NG>
  static __device__ __forceinline__ void prefetch(const Params& p, Pre<FM, FN, NG>& pre, int m0, int j0, int M, int H) {
    if constexpr (FAST) {
      const __bf16* gi = reinterpret_cast<const __bf16*>(p.gi);
#pragma unroll
      for (int i = 0; i < FM; i++)
#pragma unroll
        for (int c = 0; c < ncell<FN>(); c++) {
          int row, u; coord<FN>(i, c, row, u);
          const long m = min(m0 + row, M - 1);        // clamped: always a valid address, no branch
          const int j = min(j0 + u, H - 4);
#pragma unroll
          for (int gt = 0; gt < 3; gt++) pre.g[i][c][gt] = *reinterpret_cast<const bf16x4*>(gi + m * p.ld_gi + gt * H + j);
          if constexpr (G2) {
            const __bf16* gi2 = reinterpret_cast<const __bf16*>(p.gi2);
#pragma unroll
            for (int gt = 0; gt < 3; gt++) pre.g2[i][c][gt] = *reinterpret_cast<const bf16x4*>(gi2 + m * p.ld_gi2 + gt * H + j);
          }
          pre.hp[i][c] = *reinterpret_cast<const float4*>(p.hprev + m * p.ld_hprev + j);
        }
    }
  }
  // one cell: row m, units j..j+3; a_r/a_z/a_n = h_prev . W_h{r,z,n}^T
  static __device__ __forceinline__ void cell(const Params& p, int m, int j, int H, const f32x4& a_r, const f32x4& a_z, const f32x4& a_n,
                                              const bf16x4* pg, const bf16x4* pg2, const float4* php) {
    const bool gbf = FAST || (p.flags & PTV_GRU_GATES_BF16), ibf = p.flags & PTV_GRU_GI_BF16, i2bf = p.flags & PTV_GRU_GI2_BF16;
    const bool live = p.lengths == nullptr || p.t < p.lengths[m];
    // H is a multiple of 4 for every GRU on the path (host checks) -> vector accesses
    float gir[4], giz[4], gin[4], hP[4];
    if constexpr (FAST) {
#pragma unroll
      for (int e = 0; e < 4; e++) { gir[e] = (float)pg[0][e]; giz[e] = (float)pg[1][e]; gin[e] = (float)pg[2][e]; }
      if constexpr (G2) {
#pragma unroll
        for (int e = 0; e < 4; e++) { gir[e] += (float)pg2[0][e]; giz[e] += (float)pg2[1][e]; gin[e] += (float)pg2[2][e]; }
      }
      hP[0] = php->x; hP[1] = php->y; hP[2] = php->z; hP[3] = php->w;
    } else {
      const long gm = p.gi_idx ? p.gi_idx[m] : m;
      const float4 gr = ld4f(p.gi, gm * p.ld_gi + j, ibf);
      const float4 gz = ld4f(p.gi, gm * p.ld_gi + H + j, ibf);
      const float4 gn = ld4f(p.gi, gm * p.ld_gi + 2 * H + j, ibf);
      float4 hr = make_float4(0, 0, 0, 0), hz = hr, hn2 = hr;
      if (p.gi2) {
        hr = ld4f(p.gi2, (long)m * p.ld_gi2 + j, i2bf);
        hz = ld4f(p.gi2, (long)m * p.ld_gi2 + H + j, i2bf);
        hn2 = ld4f(p.gi2, (long)m * p.ld_gi2 + 2 * H + j, i2bf);
      }
      const float4 hp = *reinterpret_cast<const float4*>(p.hprev + (long)m * p.ld_hprev + j);
      gir[0] = gr.x + hr.x; gir[1] = gr.y + hr.y; gir[2] = gr.z + hr.z; gir[3] = gr.w + hr.w;
      giz[0] = gz.x + hz.x; giz[1] = gz.y + hz.y; giz[2] = gz.z + hz.z; giz[3] = gz.w + hz.w;
      gin[0] = gn.x + hn2.x; gin[1] = gn.y + hn2.y; gin[2] = gn.z + hn2.z; gin[3] = gn.w + hn2.w;
      hP[0] = hp.x; hP[1] = hp.y; hP[2] = hp.z; hP[3] = hp.w;
    }
    const float4 br = *reinterpret_cast<const float4*>(p.bhh + j);
    const float4 bz = *reinterpret_cast<const float4*>(p.bhh + H + j);
    const float4 bn = *reinterpret_cast<const float4*>(p.bhh + 2 * H + j);
    const float bR[4] = {br.x, br.y, br.z, br.w}, bZ[4] = {bz.x, bz.y, bz.z, bz.w}, bN[4] = {bn.x, bn.y, bn.z, bn.w};
    float r[4], z[4], n[4], hn[4], h[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
      r[e] = sigmoidf_(gir[e] + a_r[e] + bR[e]);
      z[e] = sigmoidf_(giz[e] + a_z[e] + bZ[e]);
      hn[e] = a_n[e] + bN[e];
      n[e] = tanhf_(gin[e] + r[e] * hn[e]);
      if (!live) { r[e] = 0.f; z[e] = 1.f; n[e] = 0.f; }     // masked row: h' = h, zero gate grads
      h[e] = (1.0f - z[e]) * n[e] + z[e] * hP[e];
    }
    *reinterpret_cast<float4*>(p.hout + (long)m * p.ld_hout + j) = make_float4(h[0], h[1], h[2], h[3]);
    if (FAST || p.hout16) st4f(p.hout16, (long)m * H + j, true, h[0], h[1], h[2], h[3]);
    if (p.gates) {
      const long gs = (long)m * H + j;
      st4f(p.gates, gs + 0 * p.plane, gbf, r[0], r[1], r[2], r[3]);
      st4f(p.gates, gs + 1 * p.plane, gbf, z[0], z[1], z[2], z[3]);
      st4f(p.gates, gs + 2 * p.plane, gbf, n[0], n[1], n[2], n[3]);
      st4f(p.gates, gs + 3 * p.plane, gbf, hn[0], hn[1], hn[2], hn[3]);
    }
  }
  template <int FM, int FN, int NG>
  static __device__ __forceinline__ void apply(const Params& p, f32x4 (&acc)[FM][NG * FN], const Pre<FM, FN, NG>& pre,
                                               int m0, int j0, int M, int H, int split, char* lds) {
    static_assert(NG == 3, "GRU epilogue needs the three gates");
    constexpr int PI = FAST ? 1 : 0, P2 = G2 ? 1 : 0;      // Pre arrays are [1][1] when unused
    if constexpr (staged<FN>()) {
      constexpr int W = FN * 16;
      using RS = RowStage<3, W>;
      float* st = RS::base(lds);
#pragma unroll
      for (int i = 0; i < FM; i++) {
        RS::put(st, acc[i]);
#pragma unroll
        for (int c = 0; c < RS::PASSES; c++) {
          int row, u; coord<FN>(i, c, row, u);
          const int m = m0 + row, j = j0 + u, rl = row - i * 16;
          if (m < M && j < H)
            cell(p, m, j, H, RS::get(st, rl, u), RS::get(st, rl, W + u), RS::get(st, rl, 2 * W + u), pre.g[i * PI][c * PI], pre.g2[i * P2][c * P2], &pre.hp[i * PI][c * PI]);
        }
        __builtin_amdgcn_wave_barrier();
      }
    } else {
#pragma unroll
      for (int i = 0; i < FM; i++)
#pragma unroll
        for (int f = 0; f < FN; f++) {
          int row, u; coord<FN>(i, f, row, u);
          const int m = m0 + row, j = j0 + u;
          if (m < M && j < H)
            cell(p, m, j, H, acc[i][0 * FN + f], acc[i][1 * FN + f], acc[i][2 * FN + f], pre.g[i * PI][f * PI], pre.g2[i * P2][f * P2], &pre.hp[i * PI][f * PI]);
        }
    }
  }
};
using EpiGruFwd = EpiGruFwdT<0>;

// ---------------------------------------------------------------------------------------------
// BPTT step epilogue: acc = dgh_{s+1} . W_hh  (grad reaching h_{s+1}... see gru_seq_bwd)
// ---------------------------------------------------------------------------------------------
struct GruBwdParams {
  const float* dhz_next;              // [M,H] dh (x) z carried from the later step, null at the last step
  const void* dh_ext; long ld_ext;    // [M,H] gradient arriving at this step's output from outside (fp32, or bf16 with PTV_GRU_EXT_BF16), may be null
  const float* dh_ext2; long ld_ext2; // second external addend (e.g. final-state grad), may be null
  const float* lr_a; long lr_lda; int lr_k; const float* lr_b;   // optional low-rank addend: dh += lr_a[m, 0:k] . lr_b[k, H]
  const void* gates; long plane;      // saved r,z,n,hn of this step (fp32 or bf16)
  const float* hprev; long ld_hprev;
  void* dgi; void* dgh;               // [M,3H] each (fp32 or bf16)
  float* dhz;                         // [M,H] out: dh (x) z
  int H;
  int flags;
};

// FAST: bf16 gates and gate gradients, low-rank addend of rank <= 2 (the teacher-forced bf16 path): dtype branches
// resolved at compile time, and the operands of cell c+1 are requested before cell c is computed, so a
// lane always has one cell's loads in flight behind the arithmetic and stores of the previous one.
template <bool FAST>
struct EpiGruBwdT {
  template <class P> static __device__ __forceinline__ bool dead_is_noop(const P&) { return false; }   // (GemmArgs::m_top is never set here)
  using Params = GruBwdParams;
  struct Ops { bf16x4 g[4]; float4 hp, dz, e1, e2; float la[2]; };   // la: low-rank coefficients of the row (lr_k <= 2)
  static __device__ __forceinline__ Ops load_ops(const Params& p, int m_, int j_, int M, int H) {
    Ops o;
    const long m = min(m_, M - 1); const int j = min(j_, H - 4);            // clamped: valid address, no branch
    const __bf16* gt = reinterpret_cast<const __bf16*>(p.gates) + m * H + j;
#pragma unroll
    for (int q = 0; q < 4; q++) o.g[q] = *reinterpret_cast<const bf16x4*>(gt + q * p.plane);
    o.hp = *reinterpret_cast<const float4*>(p.hprev + m * p.ld_hprev + j);
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    o.dz = p.dhz_next ? *reinterpret_cast<const float4*>(p.dhz_next + m * H + j) : zero;
    o.e1 = p.dh_ext ? ld4f(p.dh_ext, m * p.ld_ext + j, p.flags & PTV_GRU_EXT_BF16) : zero;
    o.e2 = p.dh_ext2 ? *reinterpret_cast<const float4*>(p.dh_ext2 + m * p.ld_ext2 + j) : zero;
    o.la[0] = p.lr_a ? p.lr_a[m * p.lr_lda] : 0.f;
    o.la[1] = (p.lr_a && p.lr_k > 1) ? p.lr_a[m * p.lr_lda + 1] : 0.f;
    return o;
  }
  static __device__ __forceinline__ void cell_fast(const Params& p, int m, int j, int H, const f32x4& av, const Ops& o) {
    const float dzn[4] = {o.dz.x, o.dz.y, o.dz.z, o.dz.w}, e1[4] = {o.e1.x, o.e1.y, o.e1.z, o.e1.w}, e2[4] = {o.e2.x, o.e2.y, o.e2.z, o.e2.w};
    const float hp[4] = {o.hp.x, o.hp.y, o.hp.z, o.hp.w};
    float lr[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.lr_a) {                                                           // dh += lr_a[m, 0:k] . lr_b[k, H], k <= 2
      const float4 b0 = *reinterpret_cast<const float4*>(p.lr_b + j);
      lr[0] = o.la[0] * b0.x; lr[1] = o.la[0] * b0.y; lr[2] = o.la[0] * b0.z; lr[3] = o.la[0] * b0.w;
      if (p.lr_k > 1) {
        const float4 b1 = *reinterpret_cast<const float4*>(p.lr_b + H + j);
        lr[0] += o.la[1] * b1.x; lr[1] += o.la[1] * b1.y; lr[2] += o.la[1] * b1.z; lr[3] += o.la[1] * b1.w;
      }
    }
    float dr[4], dz[4], dn[4], dnr[4], dhz[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const float dh = av[e] + dzn[e] + e1[e] + e2[e] + lr[e];
      const float r = (float)o.g[0][e], z = (float)o.g[1][e], n = (float)o.g[2][e], hn = (float)o.g[3][e];
      dn[e] = dh * (1.0f - z) * (1.0f - n * n);
      dz[e] = dh * (hp[e] - n) * z * (1.0f - z);
      dr[e] = dn[e] * hn * r * (1.0f - r);
      dnr[e] = dn[e] * r;
      dhz[e] = dh * z;
    }
    const long go = (long)m * 3 * H + j;
    st4f(p.dgi, go, true, dr[0], dr[1], dr[2], dr[3]);
    st4f(p.dgi, go + H, true, dz[0], dz[1], dz[2], dz[3]);
    st4f(p.dgi, go + 2 * H, true, dn[0], dn[1], dn[2], dn[3]);
    st4f(p.dgh, go, true, dr[0], dr[1], dr[2], dr[3]);
    st4f(p.dgh, go + H, true, dz[0], dz[1], dz[2], dz[3]);
    st4f(p.dgh, go + 2 * H, true, dnr[0], dnr[1], dnr[2], dnr[3]);
    *reinterpret_cast<float4*>(p.dhz + (long)m * H + j) = make_float4(dhz[0], dhz[1], dhz[2], dhz[3]);
  }
  template <int FN> static __device__ __forceinline__ void coord(int c, int& row, int& u) {   // cell c of a 16-row fragment row
    if constexpr (staged<FN>()) {
      using RS = RowStage<1, FN * 16>;
      row = RS::row(c); u = RS::unit();
    } else {
      const int lane = threadIdx.x & 63;
      row = lane & 15; u = c * 16 + (lane >> 4) * 4;
    }
  }
  template <int FM, int FN, int NG> struct Pre {};
  template <int FM, int FN, int NG>
  static __device__ __forceinline__ void prefetch(const Params&, Pre<FM, FN, NG>&, int, int, int, int) {}
  template <int FN> static constexpr bool staged() { return FN == 2 || FN == 4; }
  template <int FM, int FN, int NG> static constexpr int lds_bytes() { return staged<FN>() ? 4 * RowStage<1, (staged<FN>() ? FN * 16 : 64)>::WAVE_BYTES : 0; }
  template <int FM, int FN, int NG>
  static __device__ __forceinline__ void apply(const Params& p, f32x4 (&acc)[FM][NG * FN], const Pre<FM, FN, NG>&,
                                               int m0, int j0, int M, int H, int split, char* lds) {
    if constexpr (FAST) {
      constexpr int NC = staged<FN>() ? RowStage<1, (staged<FN>() ? FN * 16 : 64)>::PASSES : FN;
      Ops o[2];
      { int row, u; coord<FN>(0, row, u); o[0] = load_ops(p, m0 + row, j0 + u, M, H); }
      [[maybe_unused]] float* st = nullptr;
      if constexpr (staged<FN>()) st = RowStage<1, (staged<FN>() ? FN * 16 : 64)>::base(lds);
#pragma unroll
      for (int i = 0; i < FM; i++) {
        if constexpr (staged<FN>()) RowStage<1, (staged<FN>() ? FN * 16 : 64)>::put(st, acc[i]);
#pragma unroll
        for (int c = 0; c < NC; c++) {
          constexpr int dummy = 0; (void)dummy;
          const int idx = i * NC + c;
          if (idx + 1 < FM * NC) {
            int row, u; coord<FN>((idx + 1) % NC, row, u);
            o[(idx + 1) & 1] = load_ops(p, m0 + ((idx + 1) / NC) * 16 + row, j0 + u, M, H);
          }
          int row, u; coord<FN>(c, row, u);
          const int m = m0 + i * 16 + row, j = j0 + u;
          f32x4 av;
          if constexpr (staged<FN>()) av = RowStage<1, (staged<FN>() ? FN * 16 : 64)>::get(st, row, u); else av = acc[i][c];
          if (m < M && j < H) cell_fast(p, m, j, H, av, o[idx & 1]);
        }
        if constexpr (staged<FN>()) __builtin_amdgcn_wave_barrier();
      }
      return;
    }
    if constexpr (staged<FN>()) {
      using RS = RowStage<1, FN * 16>;
      float* st = RS::base(lds);
      const int u = RS::unit();
#pragma unroll
      for (int i = 0; i < FM; i++) {
        RS::put(st, acc[i]);
#pragma unroll
        for (int c = 0; c < RS::PASSES; c++) {
          const int row = RS::row(c), m = m0 + i * 16 + row, j = j0 + u;
          if (m < M && j < H) cell(p, m, j, H, RS::get(st, row, u));
        }
        __builtin_amdgcn_wave_barrier();
      }
    } else {
      const int lane = threadIdx.x & 63;
#pragma unroll
      for (int i = 0; i < FM; i++)
#pragma unroll
        for (int f = 0; f < FN; f++) {
          const int m = m0 + i * 16 + (lane & 15), j = j0 + f * 16 + (lane >> 4) * 4;
          if (m < M && j < H) cell(p, m, j, H, acc[i][f]);
        }
    }
  }
  // one cell: row m, units j..j+3; a = dgh_{s+1} . W_hh
  static __device__ __forceinline__ void cell(const Params& p, int m, int j, int H, const f32x4& av) {
    const bool gbf = p.flags & PTV_GRU_GATES_BF16, dbf = p.flags & PTV_GRU_DG_BF16;
    {
      {
        float dh[4] = {av[0], av[1], av[2], av[3]};
        if (p.dhz_next) { const float4 q = *reinterpret_cast<const float4*>(p.dhz_next + (long)m * H + j); dh[0] += q.x; dh[1] += q.y; dh[2] += q.z; dh[3] += q.w; }
        if (p.dh_ext) { const float4 q = ld4f(p.dh_ext, (long)m * p.ld_ext + j, p.flags & PTV_GRU_EXT_BF16); dh[0] += q.x; dh[1] += q.y; dh[2] += q.z; dh[3] += q.w; }
        if (p.dh_ext2) { const float4 q = *reinterpret_cast<const float4*>(p.dh_ext2 + (long)m * p.ld_ext2 + j); dh[0] += q.x; dh[1] += q.y; dh[2] += q.z; dh[3] += q.w; }
        if (p.lr_a) {
          for (int k = 0; k < p.lr_k; k++) {
            const float a = p.lr_a[(long)m * p.lr_lda + k];
            const float4 q = *reinterpret_cast<const float4*>(p.lr_b + (long)k * H + j);
            dh[0] += a * q.x; dh[1] += a * q.y; dh[2] += a * q.z; dh[3] += a * q.w;
          }
        }
        const long gs = (long)m * H + j;
        const float4 r4 = ld4f(p.gates, gs + 0 * p.plane, gbf);
        const float4 z4 = ld4f(p.gates, gs + 1 * p.plane, gbf);
        const float4 n4 = ld4f(p.gates, gs + 2 * p.plane, gbf);
        const float4 q4 = ld4f(p.gates, gs + 3 * p.plane, gbf);
        const float4 hp4 = *reinterpret_cast<const float4*>(p.hprev + (long)m * p.ld_hprev + j);
        const float r[4] = {r4.x, r4.y, r4.z, r4.w}, z[4] = {z4.x, z4.y, z4.z, z4.w}, n[4] = {n4.x, n4.y, n4.z, n4.w};
        const float hn[4] = {q4.x, q4.y, q4.z, q4.w}, hp[4] = {hp4.x, hp4.y, hp4.z, hp4.w};
        float dr[4], dz[4], dn[4], dnr[4], dhz[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
          dn[e] = dh[e] * (1.0f - z[e]) * (1.0f - n[e] * n[e]);
          dz[e] = dh[e] * (hp[e] - n[e]) * z[e] * (1.0f - z[e]);
          dr[e] = dn[e] * hn[e] * r[e] * (1.0f - r[e]);
          dnr[e] = dn[e] * r[e];
          dhz[e] = dh[e] * z[e];
        }
        const long go = (long)m * 3 * H + j;
        st4f(p.dgi, go, dbf, dr[0], dr[1], dr[2], dr[3]);
        st4f(p.dgi, go + H, dbf, dz[0], dz[1], dz[2], dz[3]);
        st4f(p.dgi, go + 2 * H, dbf, dn[0], dn[1], dn[2], dn[3]);
        st4f(p.dgh, go, dbf, dr[0], dr[1], dr[2], dr[3]);
        st4f(p.dgh, go + H, dbf, dz[0], dz[1], dz[2], dz[3]);
        st4f(p.dgh, go + 2 * H, dbf, dnr[0], dnr[1], dnr[2], dnr[3]);
        *reinterpret_cast<float4*>(p.dhz + (long)m * H + j) = make_float4(dhz[0], dhz[1], dhz[2], dhz[3]);
      }
    }
  }
};

template <class CT, int BM, int BJ, bool SA, bool SB, int MODE>
__global__ __launch_bounds__(NTHREADS, 2) void gru_fwd_step_kernel(GemmArgs g, GruFwdParams ep) {
  // the recurrent steps are the serial chain of the step: their waves outrank the weight-gradient products that
  // share the CUs from sibling streams (wave priority only orders instruction issue inside a CU)
  __builtin_amdgcn_s_setprio(3);
  gemm_body<CT, BM, BJ, 2, 2, 3, false, false, EpiGruFwdT<MODE>, SA, SB, (BM * BJ <= 64 * 32 ? 3 : 1)>(g, ep);
}

__global__ void cast_bf16_kernel(const float* __restrict__ src, long lds, __bf16* __restrict__ dst, long rows, int cols) {
  const long total = rows * cols;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x)
    dst[i] = (__bf16)src[(i / cols) * lds + (i % cols)];
}
// SB: the weight operand is the bf16 TRANSPOSED shadow W_hh^T [H, 3H] (K-contiguous rows: the fast loader);
// otherwise the fp32 parameter W_hh [3H, H] read K-major
template <class CT, int BM, int BN, bool SA, bool SB, bool FAST>
__global__ __launch_bounds__(NTHREADS) void gru_bwd_step_kernel(GemmArgs g, GruBwdParams ep) {
  __builtin_amdgcn_s_setprio(3);
  gemm_body<CT, BM, BN, 2, 2, 1, false, !SB, EpiGruBwdT<FAST>, SA, SB, (BM * BN <= 64 * 32 ? 4 : (BM * BN <= 64 * 64 ? 3 : 2))>(g, ep);
}


// ---------------------------------------------------------------------------------------------
// optional per-launch timing of one kernel family with HIP events on the launch stream
// (bench.py's roofline: average duration of the dominant kernel over the timed region)
// ---------------------------------------------------------------------------------------------
namespace prof {
// launch timing for bench.py's roofline block (declared in prof.hpp; csrc/notes_persist.hip records through the same table)
constexpr int MAXEV = 8192;
static int enabled = 0, filt_M = 0, filt_H = 0;          // enabled: bit (tag-1) set = record launches of that family
static hipEvent_t ev0[MAXEV], ev1[MAXEV];
static int slot_tag[MAXEV];
static double slot_flops[MAXEV], slot_aux[MAXEV][3];
static int created = 0, used = 0, cur_tag = 0;
bool want(int tag, int M, int H) {
  // (tags 5 = weight-gradient products, 6 = BPTT of the persistent small-M recurrences: whole families, no shape filter)
  const bool ok = (enabled & (1 << (tag - 1))) && (tag >= 5 || ((filt_M == 0 || filt_M == M) && (filt_H == 0 || filt_H == H))) && used < MAXEV;
  if (ok) cur_tag = tag;
  return ok;
}
int begin(hipStream_t s) {
  if (used >= created) {
    if (hipEventCreate(&ev0[created]) != hipSuccess || hipEventCreate(&ev1[created]) != hipSuccess) return -1;
    created++;
  }
  (void)hipEventRecord(ev0[used], s);
  slot_tag[used] = cur_tag;
  return used;
}
void end(int i, hipStream_t s, double fl) {
  (void)hipEventRecord(ev1[i], s);
  slot_flops[i] = fl;
  slot_aux[i][0] = slot_aux[i][1] = slot_aux[i][2] = 0.0;
  used = i + 1;
}
void aux(int i, double a, double b, double c) { slot_aux[i][0] = a; slot_aux[i][1] = b; slot_aux[i][2] = c; }
}  // namespace prof

template <class CT, bool SA, bool SB, int FAST>
static void launch_fwd_step(const GemmArgs& g, const GruFwdParams& ep, hipStream_t s) {
  // 64 rows x 64 units (x3 gates) per block when that fills the chip, else 64 x 32 (the small-M recurrent steps)
  const long blocks_big = (long)cdiv(g.M, 64) * cdiv(g.N, 64);
  if (blocks_big >= 384) {
    hipLaunchKernelGGL((gru_fwd_step_kernel<CT, 64, 64, SA, SB, FAST>), dim3(cdiv(g.N, 64), cdiv(g.M, 64)), dim3(NTHREADS), 0, s, g, ep);
  } else {
    hipLaunchKernelGGL((gru_fwd_step_kernel<CT, 64, 32, SA, SB, FAST>), dim3(cdiv(g.N, 32), cdiv(g.M, 64)), dim3(NTHREADS), 0, s, g, ep);
  }
}
static void launch_fwd_any(int prec, bool a16, bool w16, const GemmArgs& g, const GruFwdParams& ep, hipStream_t s) {
  // FAST epilogue: all-bf16 storage, plain row indexing (the teacher-forced path)
  const bool fast = a16 && w16 && ep.hout16 && (ep.flags & PTV_GRU_GI_BF16) && (!ep.gates || (ep.flags & PTV_GRU_GATES_BF16)) &&
                    (!ep.gi2 || (ep.flags & PTV_GRU_GI2_BF16)) && !ep.gi_idx && g.M >= 1 && ep.H >= 4;
  if (prec == PTV_PREC_BF16) {
    if (fast && ep.gi2) launch_fwd_step<BF16, true, true, 2>(g, ep, s);
    else if (fast) launch_fwd_step<BF16, true, true, 1>(g, ep, s);
    else if (a16 && w16) launch_fwd_step<BF16, true, true, 0>(g, ep, s);
    else if (a16) launch_fwd_step<BF16, true, false, 0>(g, ep, s);
    else launch_fwd_step<BF16, false, false, 0>(g, ep, s);
  } else launch_fwd_step<F32, false, false, 0>(g, ep, s);
}
static void cast_rows_bf16(const float* src, long lds, void* dst, long rows, int cols, hipStream_t s) {
  long nb = (rows * cols + 255) / 256; if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(cast_bf16_kernel, dim3((int)nb), dim3(256), 0, s, src, lds, (__bf16*)dst, rows, cols);
}

template <class CT, bool SA, bool SB, bool FAST>
static void launch_bwd_step(const GemmArgs& g, const GruBwdParams& ep, hipStream_t s) {
  const long blocks_big = (long)cdiv(g.M, 128) * cdiv(g.N, 128);
  const long blocks_mid = (long)cdiv(g.M, 64) * cdiv(g.N, 64);
  if (blocks_big >= 192 && g.N > 64) {
    hipLaunchKernelGGL((gru_bwd_step_kernel<CT, 128, 128, SA, SB, FAST>), dim3(cdiv(g.N, 128), cdiv(g.M, 128)), dim3(NTHREADS), 0, s, g, ep);
  } else if (blocks_mid >= 192) {
    hipLaunchKernelGGL((gru_bwd_step_kernel<CT, 64, 64, SA, SB, FAST>), dim3(cdiv(g.N, 64), cdiv(g.M, 64)), dim3(NTHREADS), 0, s, g, ep);
  } else {
    hipLaunchKernelGGL((gru_bwd_step_kernel<CT, 64, 32, SA, SB, FAST>), dim3(cdiv(g.N, 32), cdiv(g.M, 64)), dim3(NTHREADS), 0, s, g, ep);
  }
}

}  // namespace ptv

using namespace ptv;

extern "C" int ptv_gru_seq_fwd(int prec, int M, int H, int T,
                               const void* gi, long gi_step_stride, long gi_ld,
                               const void* gi2, long gi2_step_stride, long gi2_ld,
                               const void* w_hh, const float* b_hh,
                               float* hall, void* hall16, void* gates,
                               const int* lengths, int reverse, const int* gi_idx, int flags, void* stream) {
  if (M <= 0 || H <= 0 || T <= 0 || (H & 3) || !gi || !w_hh || !b_hh || !hall) return PTV_ERR_ARG;
  if ((gi_ld & 3) || (gi_step_stride & 3) || (gi2 && ((gi2_ld & 3) || (gi2_step_stride & 3)))) return PTV_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const long MH = (long)M * H;
  const bool a16 = hall16 != nullptr;
  const bool w16 = flags & PTV_GRU_W_BF16;
  if ((a16 || w16) && (prec != PTV_PREC_BF16 || (H & 7))) return PTV_ERR_ARG;
  if (w16 && !a16) return PTV_ERR_ARG;                           // bf16 weights ride with the bf16 state shadow
  if (a16 && !(flags & PTV_GRU_SKIP_CAST0)) cast_rows_bf16(hall, H, hall16, M, H, s);   // slot 0 (the caller wrote it in fp32)
  for (int step = 0; step < T; step++) {
    const int t = reverse ? T - 1 - step : step;
    GemmArgs g{a16 ? (const void*)((const __bf16*)hall16 + step * MH) : (const void*)(hall + step * MH), H, w_hh, H, M, H, H, H, (long)H};
    const long esz_gi = (flags & PTV_GRU_GI_BF16) ? 2 : 4, esz_gi2 = (flags & PTV_GRU_GI2_BF16) ? 2 : 4;
    const long esz_g = (flags & PTV_GRU_GATES_BF16) ? 2 : 4;
    EpiGruFwd::Params ep{hall + step * MH, H,
                         (const char*)gi + t * gi_step_stride * esz_gi, gi_ld,
                         gi2 ? (const char*)gi2 + t * gi2_step_stride * esz_gi2 : nullptr, gi2_ld,
                         b_hh, hall + (step + 1) * MH, H, a16 ? (__bf16*)hall16 + (step + 1) * MH : nullptr,
                         gates ? (char*)gates + (long)step * 4 * MH * esz_g : nullptr, MH,
                         lengths, t, gi_idx, H, flags};
    const int pi = prof::want(1, M, H) ? prof::begin(s) : -1;
    launch_fwd_any(prec, a16, w16, g, ep, s);
    if (pi >= 0) prof::end(pi, s, 2.0 * M * 3.0 * H * H);
  }
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

extern "C" int ptv_gru_seq_bwd(int prec, int M, int H, int T,
                               const float* hall, const void* gates, const void* w_hh,
                               const void* dh_ext, long ext_step_stride, long ext_ld,
                               const float* dh_last, long last_ld,
                               const float* lr_a, long lr_step_stride, long lr_lda, int lr_k, const float* lr_b,
                               void* dgi, void* dgh, float* dhz, float* dh0,
                               int reverse, int flags, void* stream) {
  if (M <= 0 || H <= 0 || T <= 0 || (H & 3) || !hall || !gates || !w_hh || !dgi || !dgh || !dhz) return PTV_ERR_ARG;
  if ((flags & PTV_GRU_DG_BF16) && (prec != PTV_PREC_BF16 || (H & 7))) return PTV_ERR_ARG;
  const bool w16 = flags & PTV_GRU_W_BF16;
  if (w16 && !(flags & PTV_GRU_DG_BF16)) return PTV_ERR_ARG;
  if (dh_ext && ((ext_ld & 3) || (ext_step_stride & 3))) return PTV_ERR_ARG;
  if ((flags & PTV_GRU_EXT_BF16) && prec != PTV_PREC_BF16) return PTV_ERR_ARG;
  if (dh_last && (last_ld & 3)) return PTV_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const long MH = (long)M * H, M3H = 3 * MH;
  for (int step = T - 1; step >= 0; step--) {
    const int t = reverse ? T - 1 - step : step;
    const bool last = step == T - 1;
    // dh_{step+1} = dgh_{step+1} . W_hh (K = 3H; K = 0 at the last step) + dhz_{step+1} + external grads
    const bool dbf = flags & PTV_GRU_DG_BF16;
    const long esz_d = dbf ? 2 : 4, esz_g = (flags & PTV_GRU_GATES_BF16) ? 2 : 4;
    GemmArgs g{last ? dgh : (const char*)dgh + (long)(step + 1) * M3H * esz_d, 3L * H, w_hh, w16 ? 3L * H : (long)H, M, H, last ? 0 : 3 * H, last ? 0 : 3 * H, 0};
    GruBwdParams ep{last ? nullptr : dhz + ((step + 1) & 1) * MH,
                         dh_ext ? (const char*)dh_ext + (long)step * ext_step_stride * ((flags & PTV_GRU_EXT_BF16) ? 2 : 4) : nullptr, ext_ld,
                         last ? dh_last : nullptr, last_ld,
                         lr_a ? lr_a + (long)step * lr_step_stride : nullptr, lr_lda, lr_k, lr_b,
                         (const char*)gates + (long)step * 4 * MH * esz_g, MH,
                         hall + (long)step * MH, H,
                         (char*)dgi + (long)t * M3H * esz_d, (char*)dgh + (long)step * M3H * esz_d,
                         dhz + (step & 1) * MH, H, flags};
    const int pi = prof::want(2, M, H) ? prof::begin(s) : -1;
    const bool fast = dbf && w16 && (flags & PTV_GRU_GATES_BF16) && (!lr_a || lr_k <= 2);   // all-bf16 storage
    if (prec == PTV_PREC_BF16) {
      if (fast) launch_bwd_step<BF16, true, true, true>(g, ep, s);
      else if (dbf && w16) launch_bwd_step<BF16, true, true, false>(g, ep, s);
      else if (dbf) launch_bwd_step<BF16, true, false, false>(g, ep, s);
      else launch_bwd_step<BF16, false, false, false>(g, ep, s);
    } else launch_bwd_step<F32, false, false, false>(g, ep, s);
    if (pi >= 0) prof::end(pi, s, last ? 0.0 : 2.0 * M * 3.0 * H * H);
  }
  PTV_CHECK_LAUNCH();
  if (dh0) {
    // dh0 = dhz_0 + dgh_0 . W_hh
    if (hipMemcpyAsync(dh0, dhz, MH * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess) return PTV_ERR_LAUNCH;
    PTV_TRY(ptv_gemm(prec, 0, w16 ? 0 : 1, M, H, 3 * H, dgh, 3L * H, w_hh, w16 ? 3L * H : (long)H, dh0, H, nullptr, 1.0f, 1, 0, -1,
                     ((flags & PTV_GRU_DG_BF16) ? 1 : 0) | (w16 ? 2 : 0), stream));
  }
  return PTV_OK;
}

// single GRU cell step with fully explicit strides (the free-running decoder walks row slices of the
// step-major buffers: its per-time-step batch is a [B]-row window of the [32*B]-row matrices)
extern "C" int ptv_gru_step_fwd(int prec, int M, int H,
                                const float* hprev, long ld_hprev, const void* hprev16, void* hout16,
                                const void* gi, long gi_ld, const void* gi2, long gi2_ld,
                                const void* w_hh, const float* b_hh,
                                float* hout, long ld_hout,
                                void* gates, long gates_plane,
                                const int* lengths, int t, const int* gi_idx, int flags, void* stream) {
  if (M <= 0 || H <= 0 || (H & 3) || !hprev || !gi || !w_hh || !b_hh || !hout) return PTV_ERR_ARG;
  if ((gi_ld & 3) || (ld_hprev & 3) || (ld_hout & 3) || (gi2 && (gi2_ld & 3)) || (gates && (gates_plane & 3))) return PTV_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const bool a16 = hprev16 != nullptr;
  const bool w16 = flags & PTV_GRU_W_BF16;
  if ((a16 || hout16 || w16) && (prec != PTV_PREC_BF16 || (H & 7))) return PTV_ERR_ARG;
  if (w16 && !a16) return PTV_ERR_ARG;
  GemmArgs g{a16 ? hprev16 : (const void*)hprev, a16 ? (long)H : ld_hprev, w_hh, H, M, H, H, H, (long)H};
  EpiGruFwd::Params ep{hprev, ld_hprev, gi, gi_ld, gi2, gi2_ld, b_hh, hout, ld_hout, (__bf16*)hout16, gates, gates_plane, lengths, t, gi_idx, H, flags};
  launch_fwd_any(prec, a16, w16, g, ep, s);
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

extern "C" int ptv_prof_enable(int mask) { ptv::prof::enabled = mask; return PTV_OK; }
extern "C" int ptv_prof_config(int M, int H) { ptv::prof::filt_M = M; ptv::prof::filt_H = H; return PTV_OK; }
extern "C" int ptv_prof_reset(void) { ptv::prof::used = 0; return PTV_OK; }
extern "C" int ptv_prof_read_tag(int tag, long* count, double* total_ms, double* flops) {
  double tot = 0.0, fl = 0.0;
  long n = 0;
  for (int i = 0; i < ptv::prof::used; i++) {
    if (tag != 0 && ptv::prof::slot_tag[i] != tag) continue;
    float ms = 0.f;
    if (hipEventSynchronize(ptv::prof::ev1[i]) != hipSuccess) return PTV_ERR_LAUNCH;
    if (hipEventElapsedTime(&ms, ptv::prof::ev0[i], ptv::prof::ev1[i]) != hipSuccess) return PTV_ERR_LAUNCH;
    tot += ms; fl += ptv::prof::slot_flops[i]; n++;
  }
  if (count) *count = n;
  if (total_ms) *total_ms = tot;
  if (flops) *flops = fl;
  return PTV_OK;
}
// the part of a tag's FLOPs that is subject to a device-side row limit (ptv_wgrad's k_top), by the number of units the limit counts in:
// lim15 = products over the 15 note steps of the decoder, lim16 = over the 16 note positions of the note-summary GRU.  bench.py scales
// them by the live fraction of the benchmark batch to report EXECUTED next to nominal FLOPs.
extern "C" int ptv_prof_read_limited(int tag, double* lim15, double* lim16) {
  double a = 0.0, b = 0.0;
  for (int i = 0; i < ptv::prof::used; i++) {
    if (tag != 0 && ptv::prof::slot_tag[i] != tag) continue;
    a += ptv::prof::slot_aux[i][0]; b += ptv::prof::slot_aux[i][1];
  }
  if (lim15) *lim15 = a;
  if (lim16) *lim16 = b;
  return PTV_OK;
}
// ... and the part of lim15 whose products also clip K segments (ptv_wgrad_job.seg_n: the dead blocks of every note step)
extern "C" int ptv_prof_read_segmented(int tag, double* seg) {
  double c = 0.0;
  for (int i = 0; i < ptv::prof::used; i++) {
    if (tag != 0 && ptv::prof::slot_tag[i] != tag) continue;
    c += ptv::prof::slot_aux[i][2];
  }
  if (seg) *seg = c;
  return PTV_OK;
}
extern "C" int ptv_prof_read(long* count, double* total_ms, double* flops) { return ptv_prof_read_tag(0, count, total_ms, flops); }

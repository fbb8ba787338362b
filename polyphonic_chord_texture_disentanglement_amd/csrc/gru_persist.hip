// gru_persist.hip -- persistent, weight-stationary GRU recurrences for the small-M chains of the train step
// (time GRU ptvae.py:461-462, encoder bi-GRUs ptvae.py:23,116, chord decoder ptvae.py:64-65): ONE launch per
// sequence (or per group of up to 4 independent sequences, e.g. the two directions of a bi-GRU) instead of one
// launch per recurrent step.
//
// Why: at M = B <= 1024 rows a per-step launch is pure latency -- 256 one-per-CU blocks walk K = H (forward) or
// K = 3H (BPTT) with nothing resident and re-fetch the 6.3 MB W_hh from L2/HBM every step (16.5 / 44 us per
// step at M = 512, H = 1024).  Here a workgroup owns (row group, 16 hidden units) for ALL T steps:
//   * its W_hh slice (forward: 3 gates x 16 units x H; BPTT: 16 units x 3H of W_hh^T; 96 KB bf16 at H = 1024)
//     is loaded into LDS once and stays there: weights cross the fabric once per launch, not once per step
//   * the recurrent state of its own cells (h_prev / dh (x) z) stays in registers across the steps
//   * per step only the bf16 MFMA operand is exchanged: every workgroup publishes its [rows x 16 units] slice
//     of h_{s+1} (BPTT: its 3 gate slices of dgh_s) with write-through (sc1) stores, drains vmcnt, and one lane
//     bumps the row group's arrival counter; consumers poll that ONE word relaxed and then read the row group's
//     rows with sc1 (L1-bypassing) loads straight into MFMA A fragments (MI355X_MICROARCH.md: visibility rules,
//     recipe R1).  Only the UG = H/16 workgroups of one row group synchronise with each other.
//   * the exchanged operand lives in its own scratch tensor `xch`, K-BLOCKED and row-interleaved: [step][k/8][row][8]
//     bf16.  An MFMA A fragment wants lane (r, q) to hold 8 k of row r, so ADJACENT lanes are adjacent ROWS: in a
//     row-major [row][k] tensor every lane of a wave-load hits a different cache line (64 tag lookups per 1-KB
//     load: the step was bound by the texture-address path at ~16 B/clk/CU, time linear in M); in the blocked
//     layout 16 adjacent lanes read 256 contiguous bytes.  One slot per step: no address is rewritten in a launch.
//   * what the backward needs (fp32 states, bf16 states, saved gates) is streamed out as before, same layouts as
//     ptv_gru_seq_fwd / ptv_gru_seq_bwd, so every consumer (dW products, heads) is unchanged.
// Residency: the grid is at most one 256-thread workgroup per CU (LDS 96 KB + registers admit exactly one), the
// host sizes RG from the CU count and refuses (PTV_ERR_UNSUPPORTED) what does not fit; every spin is bounded and
// reports through the error word of `sync`.  Callers must not run two persistent launches concurrently (two
// half-resident grids could wait on each other): the Python host chains them with events (functional.py).
// bf16 precision with bf16 storage only; other shapes/precisions use the per-step kernels of gru.hip.
#include "common.hpp"
#include "prof.hpp"
#include "gemm_core.hpp"
#include "../../include/ptvae_hip.h"

namespace ptv {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((address_space(1))) unsigned gu32;
typedef __attribute__((address_space(1))) unsigned long long gu64;

constexpr int PU = 16;                      // hidden units per workgroup
constexpr int PMAXC = 4;                    // chains per launch
constexpr int PLDS = 96 * 1024;             // W slice: 48 x H (fwd) or 16 x 3H (bwd) bf16, H <= 1024
constexpr unsigned SPIN_LIMIT = 4000000u;   // ~ seconds: a stranded grid gives up instead of hanging the GPU

struct PChainF {
  const __bf16* w_hh; const float* b_hh;                 // [3H,H] bf16, [3H]
  const __bf16* gi; long gi_step, gi_ld;                 // [T][M][3H] by TIME index
  const __bf16* gi2; long gi2_step, gi2_ld;              // optional second addend
  float* hall; __bf16* hall16; __bf16* gates;            // [T+1][M][H], [T+1][M][H], [T][4][M][H] or null
  __bf16* xch;                                           // exchange scratch [T+1][H/8][M][8]
  const int* lengths; int reverse;
};
struct PGruFwdArgs {
  PChainF c[PMAXC];
  int NC, M, H, T, RG, UG, rows_wg, dbg;
  unsigned* sync;                                        // word 0: error flag; word 16*(1+g): arrival counter of group g (one 64-B line each)
};

struct PChainB {
  const __bf16* w_t;                                     // W_hh^T [H][3H] bf16
  const float* hall; const __bf16* gates;
  const void* dh_ext; long ext_step, ext_ld; int ext_bf16;
  const float* dh_last; long last_ld;
  __bf16* dgi; __bf16* dgh; float* dh0;
  __bf16* xch;                                           // exchange scratch [T][3H/8][M][8]
  int reverse;
  float* part;                                           // split-K teams only: fp32 partial tiles [2][RG][teams][dest][src][rows_wg][16]
};
struct PGruBwdArgs {
  PChainB c[PMAXC];
  int NC, M, H, T, RG, UG, rows_wg, dbg;
  unsigned* sync;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, long bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)(bytes > 0x7fffffffL ? 0x7fffffffL : bytes), 0x00020000);
}
__device__ __forceinline__ bf16x8 as_bf16x8(const u32x4& v) {
  union { u32x4 u; bf16x8 b; } x; x.u = v; return x.b;
}
__device__ __forceinline__ void store_bf16x4_sc1(__bf16* p, float a, float b, float c, float d) {
  union { bf16x4 v; unsigned long long u; } x;
  x.v[0] = (__bf16)a; x.v[1] = (__bf16)b; x.v[2] = (__bf16)c; x.v[3] = (__bf16)d;
  __hip_atomic_store((gu64*)(reinterpret_cast<unsigned long long*>(p)), x.u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float4 ld_bf16x4(const __bf16* p) {
  const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
  return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
__device__ __forceinline__ void st_bf16x4(__bf16* p, float a, float b, float c, float d) {
  bf16x4 v; v[0] = (__bf16)a; v[1] = (__bf16)b; v[2] = (__bf16)c; v[3] = (__bf16)d;
  *reinterpret_cast<bf16x4*>(p) = v;
}

// block -> (group = chain*RG + row group, index inside the group).  Blocks are dispatched round-robin over the
// 8 XCDs (speed only): a row group's workgroups are kept on as few XCDs as possible, so the rows they exchange
// cross the fabric into 8/G L2s instead of all 8.
__device__ __forceinline__ void block_map(int G, int UG, int& grp, int& idx) {
  const int bid = blockIdx.x, nW = G * UG;
  if (G <= 8 && (8 % G) == 0 && (nW & 7) == 0) {
    const int per = 8 / G, x = bid & 7;
    grp = x / per; idx = (x % per) * (nW >> 3) + (bid >> 3);
  } else { grp = bid / UG; idx = bid % UG; }
}

// W slice -> LDS as K-tiles of [NBR rows][64 k] in the XOR-swizzled layout of gemm_core.hpp (conflict-free b128
// fragment reads).  src row r starts at src + rowoff(r).
template <int NBR, class RowOff>
__device__ __forceinline__ void load_w_slice(__bf16* Ws, const __bf16* src, int K, RowOff rowoff) {
  const int cpr = K >> 3;                               // 16-byte chunks per row
  for (int c = threadIdx.x; c < NBR * cpr; c += NTHREADS) {
    const int r = c / cpr, kc = c - r * cpr;
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(src + rowoff(r) + kc * 8);
    *reinterpret_cast<bf16x8*>(Ws + (kc >> 3) * (NBR * 64) + swz<BF16>(r, (kc & 7) * 8)) = v;
  }
}

// one lane polls the group's counter (relaxed, L2-served), the workgroup then proceeds to sc1 loads
template <bool ACQ>
__device__ __forceinline__ void wait_arrivals(gu32* cnt, unsigned target, gu32* err, bool& dead) {
  if (threadIdx.x == 0) {
    if (!dead) {
      unsigned spins = 0;
      while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > SPIN_LIMIT) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); dead = true; break; }
      }
    }
    if (ACQ) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // ONE buffer_inv sc1 per CU per step, after the match
  }
  __syncthreads();
}
// every storing wave drains its write-through stores, then ONE lane bumps the counter
__device__ __forceinline__ void publish(gu32* cnt) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// acc[i][g] += A[rows of frag i][K] . Wslice[g*16 + unit][K]^T with A fragments read straight from global memory
// (sc1: another CU wrote them in this launch) and B fragments from the LDS-resident slice.  Register double buffer
// of KU k-blocks (32 k each): the loads of block j+1 are in flight under the MFMAs of block j.
// LP = how the exchanged operand is read: 0 = sc1 loads (agent scope: served from the fabric, every read crosses it),
// 1 = nt loads (bypass L1, L2-served), 2 = plain loads behind ONE agent-scope acquire per step (L1 invalidated, L2-served)
template <int LP> struct LoadAux { static constexpr int v = LP == 0 ? 16 : (LP == 1 ? 2 : 0); };
template <int FM, int KU, int NB, int LP>
struct PMma {
  u32x4 buf[2][KU][FM];
  unsigned kstride;                                        // bytes between consecutive k-blocks (32 k) of one lane
  __device__ __forceinline__ void load(int b, __amdgpu_buffer_rsrc_t rs, const unsigned (&rowoff)[FM], int kb0) {
#pragma unroll
    for (int j = 0; j < KU; j++)
#pragma unroll
      for (int i = 0; i < FM; i++) buf[b][j][i] = __builtin_amdgcn_raw_buffer_load_b128(rs, rowoff[i] + (unsigned)(kb0 + j) * kstride, 0, LoadAux<LP>::v);
  }
  __device__ __forceinline__ void mma(int b, const __bf16* Ws, int kb0, f32x4 (&acc)[FM][NB]) {
    const int lane = threadIdx.x & 63, rl = lane & 15, kq = (lane >> 4) * 8;
#pragma unroll
    for (int j = 0; j < KU; j++) {
      const int kb = kb0 + j;
      const __bf16* tile = Ws + (kb >> 1) * (NB * 16 * 64);
      bf16x8 w[NB];
#pragma unroll
      for (int g = 0; g < NB; g++) w[g] = *reinterpret_cast<const bf16x8*>(tile + swz<BF16>(g * 16 + rl, (kb & 1) * 32 + kq));
#pragma unroll
      for (int i = 0; i < FM; i++)
#pragma unroll
        for (int g = 0; g < NB; g++) acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[g], as_bf16x8(buf[b][j][i]), acc[i][g], 0, 0, 0);
    }
  }
  // K = nkb k-blocks (a multiple of 2*KU)
  __device__ __forceinline__ void run(__amdgpu_buffer_rsrc_t rs, const unsigned (&rowoff)[FM], const __bf16* Ws, int nkb, f32x4 (&acc)[FM][NB]) {
    load(0, rs, rowoff, 0);
    for (int kb = 0; kb < nkb; kb += 2 * KU) {
      load(1, rs, rowoff, kb + KU);
      mma(0, Ws, kb, acc);
      if (kb + 2 * KU < nkb) load(0, rs, rowoff, kb + 2 * KU);
      mma(1, Ws, kb + KU, acc);
    }
  }
};

// =============================================================================================
// forward
// =============================================================================================
template <int FM, int KU, int LP>
__global__ __launch_bounds__(NTHREADS, 2) void pgru_fwd_kernel(PGruFwdArgs a) {
  __builtin_amdgcn_s_setprio(3);                                         // a launch of the latency chain: wins instruction issue against sibling-stream products
  __shared__ __attribute__((aligned(16))) char smem[PLDS];
  __bf16* Ws = reinterpret_cast<__bf16*>(smem);
  int grp, ug;
  block_map(a.NC * a.RG, a.UG, grp, ug);
  const int ch = grp / a.RG, rg = grp - ch * a.RG;
  const PChainF& c = a.c[ch];
  const int H = a.H, M = a.M, T = a.T;
  const long MH = (long)M * H;
  const int u0 = ug * PU;
  gu32* cnt = (gu32*)(a.sync + 16 * (1 + grp));
  gu32* err = (gu32*)(a.sync);

  load_w_slice<48>(Ws, c.w_hh, H, [&](int r) { return ((long)(r >> 4) * H + u0 + (r & 15)) * H; });
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rl = lane & 15, kq = lane >> 4;
  const int u = u0 + kq * 4;                                  // this lane's 4 units
  const int row0 = rg * a.rows_wg + wave * (FM * 16);
  int row[FM]; bool ok[FM]; int len[FM]; unsigned rowoff[FM]; float4 hp[FM];
#pragma unroll
  for (int i = 0; i < FM; i++) {
    const int r = row0 + i * 16 + rl;
    ok[i] = r < M; row[i] = r < M ? r : M - 1;
    len[i] = c.lengths ? c.lengths[row[i]] : 0x7fffffff;
    rowoff[i] = (unsigned)(((long)kq * M + row[i]) * 16);         // blocked layout: chunk kq of k-block 0, this row
    hp[i] = *reinterpret_cast<const float4*>(c.hall + (long)row[i] * H + u);
  }
  const float4 br = *reinterpret_cast<const float4*>(c.b_hh + u);
  const float4 bz = *reinterpret_cast<const float4*>(c.b_hh + H + u);
  const float4 bn = *reinterpret_cast<const float4*>(c.b_hh + 2 * H + u);
  bool dead = false;
  PMma<FM, KU, 3, LP> mm;
  mm.kstride = (unsigned)M * 64u;
  const int xc = (u >> 3), xe = (kq & 1) * 4;                  // this lane's half chunk of the exchange layout

  for (int s = 0; s < T; s++) {
    const int t = c.reverse ? T - 1 - s : s;
    // input-side pre-activations of this lane's cells: produced by earlier launches, requested before the wait
    float4 g1[FM][3], g2[FM][3];
#pragma unroll
    for (int i = 0; i < FM; i++)
#pragma unroll
      for (int g = 0; g < 3; g++) {
        g1[i][g] = ld_bf16x4(c.gi + (long)t * c.gi_step + (long)row[i] * c.gi_ld + g * H + u);
        g2[i][g] = c.gi2 ? ld_bf16x4(c.gi2 + (long)t * c.gi2_step + (long)row[i] * c.gi2_ld + g * H + u) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    if (s > 0 && !(a.dbg & 1)) wait_arrivals<LP == 2>(cnt, (unsigned)(a.UG * s), err, dead);      // h_s of the whole row group is published
    f32x4 acc[FM][3];
#pragma unroll
    for (int i = 0; i < FM; i++)
#pragma unroll
      for (int g = 0; g < 3; g++) acc[i][g] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!(a.dbg & 4)) mm.run(make_rsrc(c.xch + (long)((a.dbg & 1) ? 0 : s) * MH, MH * 2), rowoff, Ws, H >> 5, acc);

    float4 gR[FM], gZ[FM], gN[FM], gHN[FM];
#pragma unroll
    for (int i = 0; i < FM; i++) {
      const bool live = t < len[i];
      const float bR[4] = {br.x, br.y, br.z, br.w}, bZ[4] = {bz.x, bz.y, bz.z, bz.w}, bN[4] = {bn.x, bn.y, bn.z, bn.w};
      const float ir[4] = {g1[i][0].x + g2[i][0].x, g1[i][0].y + g2[i][0].y, g1[i][0].z + g2[i][0].z, g1[i][0].w + g2[i][0].w};
      const float iz[4] = {g1[i][1].x + g2[i][1].x, g1[i][1].y + g2[i][1].y, g1[i][1].z + g2[i][1].z, g1[i][1].w + g2[i][1].w};
      const float in_[4] = {g1[i][2].x + g2[i][2].x, g1[i][2].y + g2[i][2].y, g1[i][2].z + g2[i][2].z, g1[i][2].w + g2[i][2].w};
      const float hP[4] = {hp[i].x, hp[i].y, hp[i].z, hp[i].w};
      float r[4], z[4], n[4], hn[4], h[4];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        r[e] = sigmoid_fast(ir[e] + acc[i][0][e] + bR[e]);
        z[e] = sigmoid_fast(iz[e] + acc[i][1][e] + bZ[e]);
        hn[e] = acc[i][2][e] + bN[e];
        n[e] = tanh_fast(in_[e] + r[e] * hn[e]);
        if (!live) { r[e] = 0.f; z[e] = 1.f; n[e] = 0.f; }            // masked row: h' = h, zero gate grads
        h[e] = (1.0f - z[e]) * n[e] + z[e] * hP[e];
      }
      hp[i] = make_float4(h[0], h[1], h[2], h[3]);
      gR[i] = make_float4(r[0], r[1], r[2], r[3]); gZ[i] = make_float4(z[0], z[1], z[2], z[3]);
      gN[i] = make_float4(n[0], n[1], n[2], n[3]); gHN[i] = make_float4(hn[0], hn[1], hn[2], hn[3]);
      // the exchanged operand goes out first and alone: the publish below drains exactly these stores
      if (ok[i] && !(a.dbg & 8)) store_bf16x4_sc1(c.xch + (long)(s + 1) * MH + ((long)xc * M + row[i]) * 8 + xe, h[0], h[1], h[2], h[3]);
    }
    if (s + 1 < T && !(a.dbg & 2)) publish(cnt);
    // what only later launches read (fp32 state, saved gates) streams out behind the publish, under the next step's wait
#pragma unroll
    for (int i = 0; i < FM; i++) {
      if (ok[i] && !(a.dbg & 8)) {
        const long o = (long)row[i] * H + u;
        *reinterpret_cast<float4*>(c.hall + (long)(s + 1) * MH + o) = hp[i];
        st_bf16x4(c.hall16 + (long)(s + 1) * MH + o, hp[i].x, hp[i].y, hp[i].z, hp[i].w);
        if (c.gates) {
          __bf16* gp = c.gates + (long)s * 4 * MH + o;
          st_bf16x4(gp, gR[i].x, gR[i].y, gR[i].z, gR[i].w);
          st_bf16x4(gp + MH, gZ[i].x, gZ[i].y, gZ[i].z, gZ[i].w);
          st_bf16x4(gp + 2 * MH, gN[i].x, gN[i].y, gN[i].z, gN[i].w);
          st_bf16x4(gp + 3 * MH, gHN[i].x, gHN[i].y, gHN[i].z, gHN[i].w);
        }
      }
    }
  }
}

// =============================================================================================
// BPTT
// =============================================================================================
template <int FM, int KU, int LP>
__global__ __launch_bounds__(NTHREADS, 2) void pgru_bwd_kernel(PGruBwdArgs a) {
  __builtin_amdgcn_s_setprio(3);                                         // a launch of the latency chain: wins instruction issue against sibling-stream products
  __shared__ __attribute__((aligned(16))) char smem[PLDS];
  __bf16* Ws = reinterpret_cast<__bf16*>(smem);
  int grp, ug;
  block_map(a.NC * a.RG, a.UG, grp, ug);
  const int ch = grp / a.RG, rg = grp - ch * a.RG;
  const PChainB& c = a.c[ch];
  const int H = a.H, M = a.M, T = a.T;
  const long MH = (long)M * H, M3H = 3 * MH;
  const int u0 = ug * PU;
  gu32* cnt = (gu32*)(a.sync + 16 * (1 + grp));
  gu32* err = (gu32*)(a.sync);

  load_w_slice<16>(Ws, c.w_t, 3 * H, [&](int r) { return (long)(u0 + r) * 3 * H; });
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rl = lane & 15, kq = lane >> 4;
  const int u = u0 + kq * 4;
  const int row0 = rg * a.rows_wg + wave * (FM * 16);
  int row[FM]; bool ok[FM]; unsigned rowoff[FM]; float4 dhz[FM];
#pragma unroll
  for (int i = 0; i < FM; i++) {
    const int r = row0 + i * 16 + rl;
    ok[i] = r < M; row[i] = r < M ? r : M - 1;
    rowoff[i] = (unsigned)(((long)kq * M + row[i]) * 16);
    dhz[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  bool dead = false;
  PMma<FM, KU, 1, LP> mm;
  mm.kstride = (unsigned)M * 64u;
  const int xe = (kq & 1) * 4;
  const int nkb = (3 * H) >> 5;

  for (int step = T - 1; step >= (c.dh0 ? -1 : 0); step--) {
    const int t = step < 0 ? 0 : (c.reverse ? T - 1 - step : step);
    const bool last = step == T - 1;
    // this lane's cells: saved gates, previous state, external gradients (all from earlier launches)
    float4 gr[FM], gz[FM], gn[FM], gh[FM], hpv[FM], ex[FM];
    if (step >= 0) {
#pragma unroll
      for (int i = 0; i < FM; i++) {
        const long o = (long)row[i] * H + u;
        const __bf16* gp = c.gates + (long)step * 4 * MH + o;
        gr[i] = ld_bf16x4(gp); gz[i] = ld_bf16x4(gp + MH); gn[i] = ld_bf16x4(gp + 2 * MH); gh[i] = ld_bf16x4(gp + 3 * MH);
        hpv[i] = *reinterpret_cast<const float4*>(c.hall + (long)step * MH + o);
        float4 e = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c.dh_ext) {
          const long eo = (long)step * c.ext_step + (long)row[i] * c.ext_ld + u;
          e = c.ext_bf16 ? ld_bf16x4(reinterpret_cast<const __bf16*>(c.dh_ext) + eo) : *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(c.dh_ext) + eo);
        }
        if (last && c.dh_last) {
          const float4 q = *reinterpret_cast<const float4*>(c.dh_last + (long)row[i] * c.last_ld + u);
          e.x += q.x; e.y += q.y; e.z += q.z; e.w += q.w;
        }
        ex[i] = e;
      }
    }
    f32x4 acc[FM][1];
#pragma unroll
    for (int i = 0; i < FM; i++) acc[i][0] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!last) {
      wait_arrivals<LP == 2>(cnt, (unsigned)(a.UG * (T - 1 - step)), err, dead);        // dgh_{step+1} of the row group is published
      mm.run(make_rsrc(c.xch + (long)(step + 1) * M3H, M3H * 2), rowoff, Ws, nkb, acc);
    }
    if (step < 0) {                                                            // dh0 = dhz_0 + dgh_0 . W_hh
#pragma unroll
      for (int i = 0; i < FM; i++)
        if (ok[i]) *reinterpret_cast<float4*>(c.dh0 + (long)row[i] * H + u) =
            make_float4(acc[i][0][0] + dhz[i].x, acc[i][0][1] + dhz[i].y, acc[i][0][2] + dhz[i].z, acc[i][0][3] + dhz[i].w);
      break;
    }
    float4 dR[FM], dZ[FM], dN[FM], dNR[FM];
#pragma unroll
    for (int i = 0; i < FM; i++) {
      const float dzn[4] = {dhz[i].x, dhz[i].y, dhz[i].z, dhz[i].w}, e1[4] = {ex[i].x, ex[i].y, ex[i].z, ex[i].w};
      const float R[4] = {gr[i].x, gr[i].y, gr[i].z, gr[i].w}, Z[4] = {gz[i].x, gz[i].y, gz[i].z, gz[i].w};
      const float N[4] = {gn[i].x, gn[i].y, gn[i].z, gn[i].w}, HN[4] = {gh[i].x, gh[i].y, gh[i].z, gh[i].w};
      const float hP[4] = {hpv[i].x, hpv[i].y, hpv[i].z, hpv[i].w};
      float dr[4], dz[4], dn[4], dnr[4], dq[4];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const float dh = acc[i][0][e] + dzn[e] + e1[e];
        dn[e] = dh * (1.0f - Z[e]) * (1.0f - N[e] * N[e]);
        dz[e] = dh * (hP[e] - N[e]) * Z[e] * (1.0f - Z[e]);
        dr[e] = dn[e] * HN[e] * R[e] * (1.0f - R[e]);
        dnr[e] = dn[e] * R[e];
        dq[e] = dh * Z[e];
      }
      dhz[i] = make_float4(dq[0], dq[1], dq[2], dq[3]);
      dR[i] = make_float4(dr[0], dr[1], dr[2], dr[3]); dZ[i] = make_float4(dz[0], dz[1], dz[2], dz[3]);
      dN[i] = make_float4(dn[0], dn[1], dn[2], dn[3]);
      dNR[i] = make_float4(dnr[0], dnr[1], dnr[2], dnr[3]);
      if (ok[i]) {                                                             // exchanged operand: write-through, first and alone
        __bf16* px = c.xch + (long)step * M3H + (long)row[i] * 8 + xe;
        store_bf16x4_sc1(px + (long)((u) >> 3) * M * 8, dr[0], dr[1], dr[2], dr[3]);
        store_bf16x4_sc1(px + (long)((H + u) >> 3) * M * 8, dz[0], dz[1], dz[2], dz[3]);
        store_bf16x4_sc1(px + (long)((2 * H + u) >> 3) * M * 8, dnr[0], dnr[1], dnr[2], dnr[3]);
      }
    }
    if (step > 0 || c.dh0) publish(cnt);
#pragma unroll
    for (int i = 0; i < FM; i++) {
      if (ok[i]) {                                                             // dgi / dgh: read by later launches only
        __bf16* ph = c.dgh + (long)step * M3H + (long)row[i] * 3 * H + u;
        st_bf16x4(ph, dR[i].x, dR[i].y, dR[i].z, dR[i].w);
        st_bf16x4(ph + H, dZ[i].x, dZ[i].y, dZ[i].z, dZ[i].w);
        st_bf16x4(ph + 2 * H, dNR[i].x, dNR[i].y, dNR[i].z, dNR[i].w);
        __bf16* pi = c.dgi + (long)t * M3H + (long)row[i] * 3 * H + u;
        st_bf16x4(pi, dR[i].x, dR[i].y, dR[i].z, dR[i].w);
        st_bf16x4(pi + H, dZ[i].x, dZ[i].y, dZ[i].z, dZ[i].w);
        st_bf16x4(pi + 2 * H, dN[i].x, dN[i].y, dN[i].z, dN[i].w);
      }
    }
  }
}

// =============================================================================================
// BPTT, split-K teams (round 4).  The kernel above makes every workgroup read ALL K = 3H of its row group's exchanged operand per
// step -- 64 unit groups x M x 3H bf16 = 201 MB per step and chain at M = 512, H = 1024, at the ~65 GB/s a CU gets out of a
// handed-off tile (MI355X_MICROARCH.md, handoff-payload): 12-16 us of a 21-us step.  Here S consecutive workgroups (a TEAM, on one
// XCD under the XCD-aware block map) share 16*S units and split K: a workgroup keeps W_hh^T[16*S units][3H/S] in LDS (the same
// 96 KB), reads only its K range of the operand (1/S of the bytes), and the team exchanges fp32 partial tiles: workgroup ks
// FINALISES units [16*ks, 16*ks+16) of the team -- it parks the other S-1 tiles [rows x 16] in a scratch ring (16-byte sc1
// stores), bumps the team's counter, and adds the S-1 tiles it receives to its own IN SOURCE ORDER (fixed association: the result
// does not depend on arrival order).  Per step and workgroup at rows = 128: 192 KB + 24 KB read instead of 768 KB, one more
// (4-party, same-XCD) hand-off.  The epilogue operands (gates, previous state, external gradient) are requested after the
// partials are published, while the accumulators are dead, so the kernel stays inside 256 registers at 512 rows per workgroup:
// all four chains of the two encoder bi-GRUs fit ONE launch.  Epilogue, exchange layout, dgi / dgh outputs: as above.
// =============================================================================================
__device__ __forceinline__ u32x4 as_u32x4(const f32x4& v) { union { f32x4 f; u32x4 u; } x; x.f = v; return x.u; }
__device__ __forceinline__ f32x4 as_f32x4(const u32x4& v) { union { u32x4 u; f32x4 f; } x; x.u = v; return x.f; }

template <int FM, int KU, int S>
__global__ __launch_bounds__(NTHREADS, 2) void pgru_bwd_sk_kernel(PGruBwdArgs a) {
  __builtin_amdgcn_s_setprio(3);
  __shared__ __attribute__((aligned(16))) char smem[PLDS];
  __bf16* Ws = reinterpret_cast<__bf16*>(smem);
  int grp, idx;
  block_map(a.NC * a.RG, a.UG, grp, idx);
  const int ch = grp / a.RG, rg = grp - ch * a.RG;
  const int team = idx / S, ks = idx - team * S;
  const PChainB& c = a.c[ch];
  const int H = a.H, M = a.M, T = a.T;
  const long MH = (long)M * H, M3H = 3 * MH;
  const int KS = 3 * H / S;                                   // this workgroup's K range [ks*KS, (ks+1)*KS)
  const int uq0 = team * (PU * S), u0 = uq0 + ks * PU;       // the team's units / the 16 this workgroup finalises
  const int TPG = a.UG / S;                                   // teams per row group
  gu32* cnt = (gu32*)(a.sync + 16 * (1 + grp));
  gu32* tcnt = (gu32*)(a.sync + 16 * (33 + grp * TPG + team));
  gu32* err = (gu32*)(a.sync);

  load_w_slice<PU * S>(Ws, c.w_t + (long)ks * KS, KS, [&](int r) { return (long)(uq0 + r) * 3 * H; });
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rl = lane & 15, kq = lane >> 4;
  const int u = u0 + kq * 4;
  const int rw0 = wave * (FM * 16);
  const int row0 = rg * a.rows_wg + rw0;
  int row[FM]; bool ok[FM]; unsigned rowoff[FM]; float4 dhz[FM];
#pragma unroll
  for (int i = 0; i < FM; i++) {
    const int r = row0 + i * 16 + rl;
    ok[i] = r < M; row[i] = r < M ? r : M - 1;
    rowoff[i] = (unsigned)(((long)kq * M + row[i]) * 16);
    dhz[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  bool dead = false;
  PMma<FM, KU, S, 0> mm;
  mm.kstride = (unsigned)M * 64u;
  const int xe = (kq & 1) * 4;
  const int nkb = KS >> 5;
  const long xks = (long)ks * KS * M;                         // this K range inside a slot of [3H/8][M][8]
  const long ptile = (long)a.rows_wg * 16;                    // floats of one [rows x 16] partial tile
  const long pteam = (long)S * S * ptile, pslot = (long)a.RG * TPG * pteam;
  float* pbase = c.part + ((long)rg * TPG + team) * pteam;
  const unsigned plane = (unsigned)(((rw0 + rl) * 16 + kq * 4) * 4);       // byte offset of this lane inside a tile (+ i*1024)

  for (int step = T - 1; step >= (c.dh0 ? -1 : 0); step--) {
    const int t = step < 0 ? 0 : (c.reverse ? T - 1 - step : step);
    const bool last = step == T - 1;
    f32x4 own[FM];
#pragma unroll
    for (int i = 0; i < FM; i++) own[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    __amdgpu_buffer_rsrc_t prs = make_rsrc(pbase + ((step + 1) & 1) * pslot, pteam * 4);
    if (!last) {
      wait_arrivals<false>(cnt, (unsigned)(a.UG * (T - 1 - step)), err, dead);          // dgh_{step+1} of the row group is published
      f32x4 acc[FM][S];
#pragma unroll
      for (int i = 0; i < FM; i++)
#pragma unroll
        for (int j = 0; j < S; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      mm.run(make_rsrc(c.xch + (long)(step + 1) * M3H + xks, (long)KS * M * 2), rowoff, Ws, nkb, acc);
#pragma unroll
      for (int j = 0; j < S; j++) {
        if (j == ks) {
#pragma unroll
          for (int i = 0; i < FM; i++) own[i] = acc[i][j];
        } else {
#pragma unroll
          for (int i = 0; i < FM; i++)
            __builtin_amdgcn_raw_buffer_store_b128(as_u32x4(acc[i][j]), prs, plane + (unsigned)i * 1024u, (unsigned)((j * S + ks) * ptile * 4), 16);
        }
      }
      publish(tcnt);
    }
    // this lane's cells: saved gates, previous state, external gradients (all from earlier launches)
    float4 gr[FM], gz[FM], gn[FM], gh[FM], hpv[FM], ex[FM];
    if (step >= 0) {
#pragma unroll
      for (int i = 0; i < FM; i++) {
        const long o = (long)row[i] * H + u;
        const __bf16* gp = c.gates + (long)step * 4 * MH + o;
        gr[i] = ld_bf16x4(gp); gz[i] = ld_bf16x4(gp + MH); gn[i] = ld_bf16x4(gp + 2 * MH); gh[i] = ld_bf16x4(gp + 3 * MH);
        hpv[i] = *reinterpret_cast<const float4*>(c.hall + (long)step * MH + o);
        float4 e = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c.dh_ext) {
          const long eo = (long)step * c.ext_step + (long)row[i] * c.ext_ld + u;
          e = c.ext_bf16 ? ld_bf16x4(reinterpret_cast<const __bf16*>(c.dh_ext) + eo) : *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(c.dh_ext) + eo);
        }
        if (last && c.dh_last) {
          const float4 q = *reinterpret_cast<const float4*>(c.dh_last + (long)row[i] * c.last_ld + u);
          e.x += q.x; e.y += q.y; e.z += q.z; e.w += q.w;
        }
        ex[i] = e;
      }
    }
    f32x4 sum[FM];
    if (!last) {
      wait_arrivals<false>(tcnt, (unsigned)(S * (T - 1 - step)), err, dead);            // the team's partial tiles are parked
      f32x4 p[S][FM];
#pragma unroll
      for (int q = 0; q < S; q++) {
        if (q == ks) {
#pragma unroll
          for (int i = 0; i < FM; i++) p[q][i] = own[i];
        } else {
#pragma unroll
          for (int i = 0; i < FM; i++)
            p[q][i] = as_f32x4(__builtin_amdgcn_raw_buffer_load_b128(prs, plane + (unsigned)i * 1024u, (unsigned)((ks * S + q) * ptile * 4), 16));
        }
      }
#pragma unroll
      for (int i = 0; i < FM; i++) {
        f32x4 s_ = p[0][i];
#pragma unroll
        for (int q = 1; q < S; q++) s_ += p[q][i];
        sum[i] = s_;
      }
    } else {
#pragma unroll
      for (int i = 0; i < FM; i++) sum[i] = own[i];
    }
    if (step < 0) {                                                            // dh0 = dhz_0 + dgh_0 . W_hh
#pragma unroll
      for (int i = 0; i < FM; i++)
        if (ok[i]) *reinterpret_cast<float4*>(c.dh0 + (long)row[i] * H + u) =
            make_float4(sum[i][0] + dhz[i].x, sum[i][1] + dhz[i].y, sum[i][2] + dhz[i].z, sum[i][3] + dhz[i].w);
      break;
    }
    float4 dR[FM], dZ[FM], dN[FM], dNR[FM];
#pragma unroll
    for (int i = 0; i < FM; i++) {
      const float dzn[4] = {dhz[i].x, dhz[i].y, dhz[i].z, dhz[i].w}, e1[4] = {ex[i].x, ex[i].y, ex[i].z, ex[i].w};
      const float R[4] = {gr[i].x, gr[i].y, gr[i].z, gr[i].w}, Z[4] = {gz[i].x, gz[i].y, gz[i].z, gz[i].w};
      const float N[4] = {gn[i].x, gn[i].y, gn[i].z, gn[i].w}, HN[4] = {gh[i].x, gh[i].y, gh[i].z, gh[i].w};
      const float hP[4] = {hpv[i].x, hpv[i].y, hpv[i].z, hpv[i].w};
      float dr[4], dz[4], dn[4], dnr[4], dq[4];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const float dh = sum[i][e] + dzn[e] + e1[e];
        dn[e] = dh * (1.0f - Z[e]) * (1.0f - N[e] * N[e]);
        dz[e] = dh * (hP[e] - N[e]) * Z[e] * (1.0f - Z[e]);
        dr[e] = dn[e] * HN[e] * R[e] * (1.0f - R[e]);
        dnr[e] = dn[e] * R[e];
        dq[e] = dh * Z[e];
      }
      dhz[i] = make_float4(dq[0], dq[1], dq[2], dq[3]);
      dR[i] = make_float4(dr[0], dr[1], dr[2], dr[3]); dZ[i] = make_float4(dz[0], dz[1], dz[2], dz[3]);
      dN[i] = make_float4(dn[0], dn[1], dn[2], dn[3]);
      dNR[i] = make_float4(dnr[0], dnr[1], dnr[2], dnr[3]);
      if (ok[i]) {                                                             // exchanged operand: write-through, first and alone
        __bf16* px = c.xch + (long)step * M3H + (long)row[i] * 8 + xe;
        store_bf16x4_sc1(px + (long)((u) >> 3) * M * 8, dr[0], dr[1], dr[2], dr[3]);
        store_bf16x4_sc1(px + (long)((H + u) >> 3) * M * 8, dz[0], dz[1], dz[2], dz[3]);
        store_bf16x4_sc1(px + (long)((2 * H + u) >> 3) * M * 8, dnr[0], dnr[1], dnr[2], dnr[3]);
      }
    }
    if (step > 0 || c.dh0) publish(cnt);
#pragma unroll
    for (int i = 0; i < FM; i++) {
      if (ok[i]) {                                                             // dgi / dgh: read by later launches only
        __bf16* ph = c.dgh + (long)step * M3H + (long)row[i] * 3 * H + u;
        st_bf16x4(ph, dR[i].x, dR[i].y, dR[i].z, dR[i].w);
        st_bf16x4(ph + H, dZ[i].x, dZ[i].y, dZ[i].z, dZ[i].w);
        st_bf16x4(ph + 2 * H, dNR[i].x, dNR[i].y, dNR[i].z, dNR[i].w);
        __bf16* pi = c.dgi + (long)t * M3H + (long)row[i] * 3 * H + u;
        st_bf16x4(pi, dR[i].x, dR[i].y, dR[i].z, dR[i].w);
        st_bf16x4(pi + H, dZ[i].x, dZ[i].y, dZ[i].z, dZ[i].w);
        st_bf16x4(pi + 2 * H, dN[i].x, dN[i].y, dN[i].z, dN[i].w);
      }
    }
  }
}

// x[(k/8)*M*8 + row*8 + k%8] = bf16(h[row*H + k]): slot 0 of the forward exchange tensor from the caller's fp32 state
__global__ void pack_blocked_kernel(const float* __restrict__ h, __bf16* __restrict__ x, int M, int H) {
  const long n = (long)M * (H >> 3);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int row = (int)(i % M), kc = (int)(i / M);
    const float4 a = *reinterpret_cast<const float4*>(h + (long)row * H + kc * 8);
    const float4 b = *reinterpret_cast<const float4*>(h + (long)row * H + kc * 8 + 4);
    bf16x8 v;
    v[0] = (__bf16)a.x; v[1] = (__bf16)a.y; v[2] = (__bf16)a.z; v[3] = (__bf16)a.w;
    v[4] = (__bf16)b.x; v[5] = (__bf16)b.y; v[6] = (__bf16)b.z; v[7] = (__bf16)b.w;
    *reinterpret_cast<bf16x8*>(x + i * 8) = v;
  }
}

static int g_load_policy = 0, g_dbg = 0;
static int g_num_cu = 0, g_cu_reserve = 0;
static int num_cu() {
  if (g_num_cu == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = -1;
    g_num_cu = n;
  }
  return g_num_cu;
}

// row groups / rows per workgroup for NC chains of M rows with UG = H/16 unit groups: one workgroup per CU at most
static int plan(int NC, int M, int H, int& RG, int& rows_wg, int& FM, int fm_max = 4) {
  if (NC < 1 || NC > PMAXC || M <= 0 || H < 256 || H > 1024 || (H & 255)) return PTV_ERR_UNSUPPORTED;
  const int ncu = num_cu() - g_cu_reserve;
  const int UG = H / PU;
  int rg = ncu / (NC * UG);
  if (rg < 1) return PTV_ERR_UNSUPPORTED;
  int p2 = 1; while (p2 * 2 <= rg) p2 *= 2;
  rg = p2;
  while (rg > 1 && (M + rg - 1) / rg < 64 && (M + rg / 2 - 1) / (rg / 2) <= 256) rg /= 2;     // no emptier than one 64-row panel
  int fm = ((M + rg - 1) / rg + 63) / 64;
  if (fm == 3) fm = 4;
  if (fm > 4 && fm <= 8) fm = 8;
  if (fm > fm_max) return PTV_ERR_UNSUPPORTED;
  RG = rg; rows_wg = fm * 64; FM = fm;
  return PTV_OK;
}
static bool splitk_ok(int H, int S) { return (S == 2 || S == 4) && (H / PU) % S == 0 && (3 * H / S) % 192 == 0; }

}  // namespace ptv

using namespace ptv;

extern "C" int ptv_gru_persist_load_policy(int lp) {
  if (lp >= 100) { ptv::g_dbg = lp - 100; return PTV_OK; }      // timing experiments only (results invalid)
  if (lp < 0 || lp > 2) return PTV_ERR_ARG;
  ptv::g_load_policy = lp;
  return PTV_OK;
}

extern "C" int ptv_gru_persist_cu_reserve(int cus) {
  if (cus < 0 || cus > 255) return PTV_ERR_ARG;
  ptv::g_cu_reserve = cus;
  return PTV_OK;
}

extern "C" int ptv_gru_persist_supported(int NC, int M, int H) {
  int RG, rows, FM;
  return plan(NC, M, H, RG, rows, FM) == PTV_OK ? 1 : 0;
}

extern "C" int ptv_gru_persist_fwd(int NC, int M, int H, int T,
                                   const void* const* gi, const long* gi_step, const long* gi_ld,
                                   const void* const* gi2, const long* gi2_step, const long* gi2_ld,
                                   const void* const* w_hh16, const float* const* b_hh,
                                   float* const* hall, void* const* hall16, void* const* gates,
                                   const int* const* lengths, const int* reverse, void* const* xch, unsigned* sync, void* stream) {
  if (T <= 0 || !gi || !gi_step || !gi_ld || !w_hh16 || !b_hh || !hall || !hall16 || !reverse || !xch || !sync) return PTV_ERR_ARG;
  int RG, rows, FM;
  PTV_TRY(plan(NC, M, H, RG, rows, FM));
  PGruFwdArgs a{};
  for (int i = 0; i < NC; i++) {
    if (!gi[i] || !w_hh16[i] || !b_hh[i] || !hall[i] || !hall16[i] || !xch[i]) return PTV_ERR_ARG;
    if ((gi_ld[i] & 3) || (gi_step[i] & 3)) return PTV_ERR_ARG;
    const bool has2 = gi2 && gi2[i];
    if (has2 && ((gi2_ld[i] & 3) || (gi2_step[i] & 3))) return PTV_ERR_ARG;
    a.c[i] = PChainF{(const __bf16*)w_hh16[i], b_hh[i], (const __bf16*)gi[i], gi_step[i], gi_ld[i],
                     has2 ? (const __bf16*)gi2[i] : nullptr, has2 ? gi2_step[i] : 0, has2 ? gi2_ld[i] : 0,
                     hall[i], (__bf16*)hall16[i], gates ? (__bf16*)gates[i] : nullptr, (__bf16*)xch[i],
                     lengths ? lengths[i] : nullptr, reverse[i]};
  }
  a.NC = NC; a.M = M; a.H = H; a.T = T; a.RG = RG; a.UG = H / PU; a.rows_wg = rows; a.sync = sync; a.dbg = g_dbg;
  for (int i = 0; i < NC; i++) {                               // slot 0 of the bf16 state and of the exchange tensor (caller wrote fp32)
    PTV_TRY(ptv_cast_bf16(hall[i], hall16[i], (long)M * H, stream));
    long nb = ((long)M * (H >> 3) + 255) / 256; if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(pack_blocked_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, hall[i], (__bf16*)xch[i], M, H);
  }
  const dim3 grid(NC * RG * a.UG), block(NTHREADS);
  hipStream_t s = (hipStream_t)stream;
  // KU = k-blocks (32 k) per register buffer half (2*KU*32 must divide K), sized so that a wave stays within 256 registers:
  // the weight-gradient products on sibling streams then co-reside on the CU (one persistent wave + two GEMM waves per SIMD)
  // and run in the cycles the latency-bound recurrence leaves idle.
#define PTV_PG_LAUNCH(K, LP_, KK)                                                     \
  do {                                                                                \
    const bool k8 = ((KK) % 512) == 0;                                                \
    if (FM == 1) { if (k8) hipLaunchKernelGGL((K<1, 8, LP_>), grid, block, 0, s, a); else hipLaunchKernelGGL((K<1, 4, LP_>), grid, block, 0, s, a); } \
    else if (FM == 2) hipLaunchKernelGGL((K<2, 4, LP_>), grid, block, 0, s, a);       \
    else hipLaunchKernelGGL((K<4, 2, LP_>), grid, block, 0, s, a);                    \
  } while (0)
  if (g_load_policy == 0) PTV_PG_LAUNCH(pgru_fwd_kernel, 0, H);
  else if (g_load_policy == 1) PTV_PG_LAUNCH(pgru_fwd_kernel, 1, H);
  else PTV_PG_LAUNCH(pgru_fwd_kernel, 2, H);
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

extern "C" int ptv_gru_persist_bwd(int NC, int M, int H, int T,
                                   const float* const* hall, const void* const* gates, const void* const* w_t16,
                                   const void* const* dh_ext, const long* ext_step, const long* ext_ld, const int* ext_bf16,
                                   const float* const* dh_last, const long* last_ld,
                                   void* const* dgi, void* const* dgh, float* const* dh0,
                                   const int* reverse, void* const* xch, unsigned* sync, void* stream) {
  if (T <= 0 || !hall || !gates || !w_t16 || !dgi || !dgh || !reverse || !xch || !sync) return PTV_ERR_ARG;
  int RG, rows, FM;
  PTV_TRY(plan(NC, M, H, RG, rows, FM));
  PGruBwdArgs a{};
  for (int i = 0; i < NC; i++) {
    if (!hall[i] || !gates[i] || !w_t16[i] || !dgi[i] || !dgh[i] || !xch[i]) return PTV_ERR_ARG;
    const bool hext = dh_ext && dh_ext[i];
    if (hext && ((ext_ld[i] & 3) || (ext_step[i] & 3))) return PTV_ERR_ARG;
    const bool hl = dh_last && dh_last[i];
    if (hl && (last_ld[i] & 3)) return PTV_ERR_ARG;
    a.c[i] = PChainB{(const __bf16*)w_t16[i], hall[i], (const __bf16*)gates[i],
                     hext ? dh_ext[i] : nullptr, hext ? ext_step[i] : 0, hext ? ext_ld[i] : 0, hext && ext_bf16 ? ext_bf16[i] : 0,
                     hl ? dh_last[i] : nullptr, hl ? last_ld[i] : 0,
                     (__bf16*)dgi[i], (__bf16*)dgh[i], dh0 ? dh0[i] : nullptr, (__bf16*)xch[i], reverse[i]};
  }
  // every chain of one launch takes the dh0 tail or none does (the step loop bound is per chain, the counters are not shared)
  a.NC = NC; a.M = M; a.H = H; a.T = T; a.RG = RG; a.UG = H / PU; a.rows_wg = rows; a.sync = sync;
  const dim3 grid(NC * RG * a.UG), block(NTHREADS);
  hipStream_t s = (hipStream_t)stream;
  const int pi = prof::want(6, M, H) ? prof::begin(s) : -1;
  if (g_load_policy == 0) PTV_PG_LAUNCH(pgru_bwd_kernel, 0, 3 * H);
  else if (g_load_policy == 1) PTV_PG_LAUNCH(pgru_bwd_kernel, 1, 3 * H);
  else PTV_PG_LAUNCH(pgru_bwd_kernel, 2, 3 * H);
  if (pi >= 0) prof::end(pi, s, 2.0 * NC * M * 3.0 * H * H * T);
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

// ---- split-K teams (pgru_bwd_sk_kernel): S = 2 or 4 workgroups share 16*S units and split K = 3H
extern "C" int ptv_gru_persist_splitk_supported(int NC, int M, int H, int S) {
  int RG, rows, FM;
  return splitk_ok(H, S) && plan(NC, M, H, RG, rows, FM, 8) == PTV_OK ? 1 : 0;
}

extern "C" long ptv_gru_persist_part_elems(int NC, int M, int H, int S) {
  int RG, rows, FM;
  if (!splitk_ok(H, S) || plan(NC, M, H, RG, rows, FM, 8) != PTV_OK) return 0;
  return 2L * RG * rows * H * S;                               // [2][RG][H/(16 S) teams][S][S][rows][16]
}

extern "C" int ptv_gru_persist_bwd_splitk(int S, int NC, int M, int H, int T,
                                          const float* const* hall, const void* const* gates, const void* const* w_t16,
                                          const void* const* dh_ext, const long* ext_step, const long* ext_ld, const int* ext_bf16,
                                          const float* const* dh_last, const long* last_ld,
                                          void* const* dgi, void* const* dgh, float* const* dh0,
                                          const int* reverse, void* const* xch, float* const* part, unsigned* sync, void* stream) {
  if (T <= 0 || !hall || !gates || !w_t16 || !dgi || !dgh || !reverse || !xch || !part || !sync) return PTV_ERR_ARG;
  if (!splitk_ok(H, S)) return PTV_ERR_UNSUPPORTED;
  int RG, rows, FM;
  PTV_TRY(plan(NC, M, H, RG, rows, FM, 8));
  PGruBwdArgs a{};
  for (int i = 0; i < NC; i++) {
    if (!hall[i] || !gates[i] || !w_t16[i] || !dgi[i] || !dgh[i] || !xch[i] || !part[i]) return PTV_ERR_ARG;
    const bool hext = dh_ext && dh_ext[i];
    if (hext && ((ext_ld[i] & 3) || (ext_step[i] & 3))) return PTV_ERR_ARG;
    const bool hl = dh_last && dh_last[i];
    if (hl && (last_ld[i] & 3)) return PTV_ERR_ARG;
    a.c[i] = PChainB{(const __bf16*)w_t16[i], hall[i], (const __bf16*)gates[i],
                     hext ? dh_ext[i] : nullptr, hext ? ext_step[i] : 0, hext ? ext_ld[i] : 0, hext && ext_bf16 ? ext_bf16[i] : 0,
                     hl ? dh_last[i] : nullptr, hl ? last_ld[i] : 0,
                     (__bf16*)dgi[i], (__bf16*)dgh[i], dh0 ? dh0[i] : nullptr, (__bf16*)xch[i], reverse[i], part[i]};
  }
  a.NC = NC; a.M = M; a.H = H; a.T = T; a.RG = RG; a.UG = H / PU; a.rows_wg = rows; a.sync = sync;
  if (NC * RG > 32 || NC * RG * (a.UG / S) > 128) return PTV_ERR_UNSUPPORTED;            // counters: 16 * (33 + 128) words
  const dim3 grid(NC * RG * a.UG), block(NTHREADS);
  hipStream_t s = (hipStream_t)stream;
#define PTV_SK_LAUNCH(S_)                                                                                  \
  do {                                                                                                     \
    if (FM == 1) hipLaunchKernelGGL((pgru_bwd_sk_kernel<1, 3, S_>), grid, block, 0, s, a);                 \
    else if (FM == 2) hipLaunchKernelGGL((pgru_bwd_sk_kernel<2, 3, S_>), grid, block, 0, s, a);            \
    else if (FM == 4) hipLaunchKernelGGL((pgru_bwd_sk_kernel<4, 3, S_>), grid, block, 0, s, a);            \
    else hipLaunchKernelGGL((pgru_bwd_sk_kernel<8, 1, S_>), grid, block, 0, s, a);                         \
  } while (0)
  const int pi = prof::want(6, M, H) ? prof::begin(s) : -1;
  if (S == 4) PTV_SK_LAUNCH(4); else PTV_SK_LAUNCH(2);
  if (pi >= 0) prof::end(pi, s, 2.0 * NC * M * 3.0 * H * H * T);
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

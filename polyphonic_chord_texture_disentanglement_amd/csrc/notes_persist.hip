// notes_persist.hip -- the teacher-forced notes GRU (dec_notes_gru, ptvae.py:395-398 restructured to ONE 15-step sequence over
// 32*B rows) as row-partitioned persistent kernels: forward and BPTT, one launch each.
//
// 71 % of the forward FLOPs of the model live in this recurrence.  As one launch per step it was the largest exclusive item
// of the train step (15 x 97 us forward at 0.34 of the HBM roofline, 15 x 129 us backward): every step re-read the state from
// HBM (fp32 + bf16), the input-side pre-activations of a separately launched product (0.75 GB written and read back), and
// started cold.  Rows are independent, so a workgroup can own 64 rows (four MFMA M tiles) for ALL 15 steps:
//   * the MFMA operand copy of the state (bf16) never leaves the CU (LDS, double buffered); the fp32 state is written once
//     per step (the backward needs it anyway) and read back by the lane that wrote it; BPTT: dh (x) z in LDS (fp32, 128 KB), dgh -- the next step's A operand, 64 x 1536, too large for
//     LDS -- in a private, L2-resident scratch tile held K-blocked so fragment loads are contiguous
//   * W_hh is streamed from L2 in MFMA-fragment-major packing (ptv_pack_mfma_b), one fragment load serving four M tiles:
//     the same 1.5 MB per CU per step the per-step kernels pulled through L2, now the ONLY large read besides GC
//   * the token product (emb . W_ih[:, Ht:]^T) is fused: K = 128 more per step instead of a [15*R, 1536] tensor
//   * the weight tiles are packed pair-interleaved (ptv_pack_mfma_b, pairs = 1): a lane's accumulators of two adjacent tiles are 8
//     consecutive units of one row, so the epilogue runs in the MFMA's own lane layout with 16- / 32-byte accesses and no cross-lane
//     movement (a first version transposed lanes with 64 ds_bpermute per tile pair: the LDS crossbar became the bottleneck)
// No workgroup waits for another: no flags, no residency requirement, any grid size.
// bf16 MFMA operands, fp32 state and accumulation (the bf16 precision policy); Hn = 512, E = 128 (init_model() geometry).
#include "common.hpp"
#include "gemm_core.hpp"
#include "prof.hpp"
#include "../../include/ptvae_hip.h"

namespace ptv {

extern int g_gemm_prio;            // gemm.hip: set by ptv_gemm_priority (the caller is on a latency chain)

// The saved gate planes are private to this file's forward / BPTT pair and live UNIT-BLOCKED: plane[(u / 32)][row][u % 32].  A lane of the
// epilogue owns 8 consecutive units of a row and the 4 lane groups of a wave 32 of them, so in a row-major plane a wave instruction
// touches 16 rows x 64 bytes -- half a cache line per row; blocked it is ONE contiguous kilobyte.  Measured at R = 16384, T = 15:
// forward 999 -> 973 us, dense BPTT 1245 -> 1090 us (scripts/bench_notes.py).
__device__ __forceinline__ long gate_off(long row, int u, long R) { return ((long)(u >> 5) * R + row) * 32 + (u & 31); }
// the gradient arriving at the states, the [T*R][H] matrix of the heads' input-gradient products, column-blocked by 32 the same way
// (ptv_gemm dtypes bit 3 over all T*R rows): element (step s, row, u)
__device__ __forceinline__ long ext_off(int s, long row, int u, long R, int T) { return ((long)(u >> 5) * ((long)T * R) + (long)s * R + row) * 32 + (u & 31); }

constexpr int NE = 128, NRP = 64;                    // input (token) width, rows per workgroup
constexpr int NT16LD = NE + 16;                      // bf16 LDS row strides (+16): conflict-free b128 fragment reads

__device__ __forceinline__ float nsig(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float ntanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

// the two accumulator fragments of a PAIR of unit tiles packed with ptv_pack_mfma_b(pairs = 1): lane (row = lane % 16,
// quad = lane / 16) already holds the 8 consecutive units 8*quad .. 8*quad+7 of the pair's 32 -- no cross-lane movement
__device__ __forceinline__ void pair_to_rows(const f32x4& f0, const f32x4& f1, float (&o)[8]) {
#pragma unroll
  for (int e = 0; e < 4; e++) { o[e] = f0[e]; o[4 + e] = f1[e]; }
}

// streaming traffic (read once / written once per launch) carries the non-temporal hint: with the default policy the ~26 MB per
// XCD per step of activations evicted the W_hh fragments every workgroup re-reads each step (L2 hit rate 60 %)
typedef __attribute__((ext_vector_type(4))) float f4v;
__device__ __forceinline__ bf16x8 ldnt_bf16x8(const __bf16* p) { return __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(p)); }
__device__ __forceinline__ float4 ldnt_f4(const float* p) {
  const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p));
  return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void stnt_bf16x8(__bf16* p, const float (&v)[8]) {
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; e++) o[e] = (__bf16)v[e];
  __builtin_nontemporal_store(o, reinterpret_cast<bf16x8*>(p));
}
__device__ __forceinline__ void stnt_f32x8(float* p, const float (&v)[8]) {
  __builtin_nontemporal_store(f4v{v[0], v[1], v[2], v[3]}, reinterpret_cast<f4v*>(p));
  __builtin_nontemporal_store(f4v{v[4], v[5], v[6], v[7]}, reinterpret_cast<f4v*>(p + 4));
}

__device__ __forceinline__ void ld_bf16x8(const __bf16* p, float (&o)[8]) {
  const bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int e = 0; e < 8; e++) o[e] = (float)v[e];
}
__device__ __forceinline__ void st_bf16x8(__bf16* p, const float (&v)[8]) {
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; e++) o[e] = (__bf16)v[e];
  *reinterpret_cast<bf16x8*>(p) = o;
}
__device__ __forceinline__ void ld_f32x8(const float* p, float (&o)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
__device__ __forceinline__ void st_f32x8(float* p, const float (&v)[8]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}

// =============================================================================================
// forward.  H = 512: the notes GRU (hoisted input part GC, no mask).  H = 128: one direction of dec_notes_emb_gru, the note
// summary bi-GRU over the 16 notes of each of the 32*B steps (ptvae.py:446-453,480-486): b_ih, packed-sequence mask by length,
// reversed time order for the *_reverse direction, final state written to its half of the summary.
// =============================================================================================
struct RowGruFwdArgs {
  const bf16x8 *w_hh, *w_x;        // pair-interleaved packing: W_hh [3H/16 tiles][H/32 kb][64], W_x [3H/16][4][64]
  const float* b_hh; const float* b_ih;   // [3H]; b_ih may be null (folded into gc)
  const __bf16* gc;                // the [R][3H] hoisted input part (b_ih included), COLUMN-BLOCKED by 32: [3H/32][R][32] (ptv_gemm dtypes bit 3), or null
  const float* x; long x_step;     // fed tokens fp32: x + t*x_step + row*128
  const int* lengths;              // [R] or null: row m is updated at time t iff t < lengths[m]
  float* HN; __bf16* HN16;         // [T+1][R][H]; slot 0 of HN written by the caller
  __bf16* gates;                   // [T][4][H/32][R][32] (unit-blocked planes, see gate_off) or null
  float* out; long out_ld;         // final state -> out[row*out_ld + unit], or null
  int R, T, reverse, dbg;
  int skip;                        // pass over the steps beyond the longest row of the panel (EMB with lengths)
  int prio;                        // EMB: raised wave priority (the launch is part of a latency chain)
  const int* perm;                 // EMB, or null: panel position p works on row perm[p] of x / lengths / out (rows sorted by length: a panel's
                                   // longest row is then close to all of its rows); HN / HN16 / gates are indexed by POSITION
};

// EMB = false: the notes GRU (hoisted input part gc, b_ih folded in, no mask, no final-state output); EMB = true: a direction of
// dec_notes_emb_gru (b_ih, optional length mask, optional reversed time, final state)
template <int H, bool EMB>
__device__ __forceinline__ void row_gru_fwd_body(const RowGruFwdArgs& a, const long panel, char* nsm) {
  constexpr int KBH = H / 32, NUT = H / 16, NPASS = H / 128, UTW = NUT / 4, HLD = H + 16, KT = KBH + 4;
  __bf16* h16 = reinterpret_cast<__bf16*>(nsm);                          // [2][64][HLD]
  __bf16* tok16 = h16 + 2 * NRP * HLD;                                   // [64][NT16LD]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rl = lane & 15, kq = (lane >> 4) * 8;                        // fragment coordinates
  const int erow = lane & 15, eq = lane >> 4;                            // epilogue coordinates = the MFMA C layout (pair-interleaved tiles)
  const long R = a.R;
  const long r0 = panel * NRP;
  const long RH = R * H;

  // ---- initial state: bf16 operand copy -> LDS and HN16 slot 0
  for (int i = tid; i < NRP * (H / 8); i += 256) {
    const int row = i / (H / 8), c8 = (i % (H / 8)) * 8;
    const long gr = min(r0 + row, R - 1);
    float v[8];
    ld_f32x8(a.HN + gr * H + c8, v);
    st_bf16x8(h16 + row * HLD + c8, v);
    if (r0 + row < R) st_bf16x8(a.HN16 + gr * H + c8, v);
  }
  long grow[4]; bool ok[4]; int len[4];
  long tnat[4];                                                          // the rows whose tokens this thread stages each step
#pragma unroll
  for (int i = 0; i < 4; i++) {
    ok[i] = r0 + i * 16 + erow < R; grow[i] = min(r0 + i * 16 + erow, R - 1);
    len[i] = 0x7fffffff;
    tnat[i] = min(r0 + (tid >> 4) + i * 16, R - 1);
    if constexpr (EMB) {
      if (a.perm) tnat[i] = a.perm[tnat[i]];
      if (a.lengths) len[i] = a.lengths[a.perm ? (long)a.perm[grow[i]] : grow[i]];
    }
  }
  static_assert(NE == 128, "token staging: 16 chunks per row, 4 rows per thread");
  // EMB with lengths: a (panel, time) pair beyond the longest row of the panel is the identity for all 64 rows (the reference packs
  // the sequences, ptvae.py:446-453: on this data the mean length is 3.7 of 16 notes) -- such steps only pass the state on
  int pmax = a.T, gmax = a.T;
  if constexpr (EMB) {
    if (a.lengths && a.skip) {
      int* misc = reinterpret_cast<int*>(tok16 + NRP * NT16LD);
      if (tid == 0) misc[0] = 0;
      __syncthreads();
      if (tid < NRP && r0 + tid < R) atomicMax(misc, a.lengths[a.perm ? (long)a.perm[r0 + tid] : r0 + tid]);
      __syncthreads();
      pmax = min(misc[0], a.T);
      // ... and a time index beyond the longest row of the WHOLE launch is dead for every panel: nothing downstream reads its state slots
      // (the BPTT stops at its own panel limit; the weight-gradient products and the dX product stop at the launch-wide limit the BPTT
      // reports, top_step = gmax - 1, which exists when R % 32 == 0).  Every workgroup finds gmax itself: R ints from L2, ~1 us --
      // the per-step copies it saves were a quarter of this launch's HBM traffic (round 6; PMC in profiles/r06_pmc_by_kernel.json)
      if ((R & 31) == 0) {
        int g = 0;
        for (long i = tid; i < R; i += 256) g = max(g, a.lengths[i]);
        if (tid == 0) misc[0] = 0;
        __syncthreads();
        atomicMax(misc, g);
        __syncthreads();
        gmax = min(misc[0], a.T);
        __syncthreads();
      }
    }
  }
  int slot = 0, cur = 0;                                                   // HN slot holding the current fp32 state; current bf16 LDS buffer
  // workgroups of one XCD run in near lockstep and would all ask the L2 for the same few fragment lines at the same moment (a
  // handful of its 16 channels busy, the rest idle): each walks the passes and the k-blocks from its own starting point
  const int prot = (a.dbg & 8) ? 0 : ((int)panel >> 3) & (NPASS - 1), krot = (a.dbg & 8) ? 0 : ((((int)panel >> 5) & 7) * 2) & (KBH - 1);

  // inside the step loop the waves exchange through LDS only (the fp32 state a lane re-reads from HN is its own store): lds_barrier()
  // lets a step's 117 MB of state / gate stores drain under the next step's products instead of at the step boundary
  for (int n = 0; n < a.T; n++) {
    const int tt = (EMB && a.reverse) ? a.T - 1 - n : n;
    if constexpr (EMB) {
      if (tt >= gmax) continue;                                            // dead for every row of the launch: the state stays in `slot`, nobody reads slot n + 1
      if (slot != n) {
        // the first step after a launch-wide dead PREFIX (the reversed direction): the state before this step is read as slot n by the
        // BPTT and by the weight_hh gradient product (whose k_rev limit starts exactly here) -- materialise it once instead of once per dead step
        __syncthreads();
        for (int i = tid; i < NRP * (H / 8); i += 256) {
          const int row = i / (H / 8), c8 = (i % (H / 8)) * 8;
          if (r0 + row < R) {
            const long gr = r0 + row;
            *reinterpret_cast<bf16x8*>(a.HN16 + (long)n * RH + gr * H + c8) = *reinterpret_cast<const bf16x8*>(h16 + cur * NRP * HLD + row * HLD + c8);
            float v[8];
            ld_f32x8(a.HN + (long)slot * RH + gr * H + c8, v);
            st_f32x8(a.HN + (long)n * RH + gr * H + c8, v);
          }
        }
        __syncthreads();
        slot = n;
      }
      if (tt >= pmax) {                                                    // whole panel masked: h' = h
        __syncthreads();                                                   // the previous step's state stores (other lanes' mapping) are complete
        for (int i = tid; i < NRP * (H / 8); i += 256) {
          const int row = i / (H / 8), c8 = (i % (H / 8)) * 8;
          if (r0 + row < R) {
            const long gr = r0 + row;
            *reinterpret_cast<bf16x8*>(a.HN16 + (long)(n + 1) * RH + gr * H + c8) = *reinterpret_cast<const bf16x8*>(h16 + cur * NRP * HLD + row * HLD + c8);
            float v[8];
            ld_f32x8(a.HN + (long)slot * RH + gr * H + c8, v);
            st_f32x8(a.HN + (long)(n + 1) * RH + gr * H + c8, v);
          }
        }
        __syncthreads();                                                   // the copy is read back through other lanes' mapping: global fence
        slot = n + 1;
        continue;
      }
    }
    const int nxt = cur ^ 1;
    const __bf16* hc = h16 + cur * NRP * HLD;
    __bf16* hn_ = h16 + nxt * NRP * HLD;
    // ---- this step's fed tokens -> LDS (bf16 MFMA operand)
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int row = (tid >> 4) + k * 16, c8 = (tid & 15) * 8;
      float v[8];
      ld_f32x8(a.x + (long)tt * a.x_step + tnat[k] * NE + c8, v);
      st_bf16x8(tok16 + row * NT16LD + c8, v);
    }
    lds_barrier();
#pragma unroll 1
    for (int p0 = 0; p0 < NPASS; p0++) {
      const int p = (p0 + prot) & (NPASS - 1);
      const int ut0 = wave * UTW + p * 2;
      const int u = ut0 * 16 + eq * 8;                                   // this lane's first unit of the pass
      // acc[i][0..3] = r0 r1 z0 z1 (h and token parts summed), [4,5] = W_hn h, [6,7] = W_in token
      // H = 128: the 4 M tiles of the panel in two halves (the fragments stream twice: 48 KB per wave and half) so that two
      // workgroups fit the register file of a CU
      constexpr int MH = H == 128 ? 2 : 4;
#pragma unroll 1
      for (int mh = 0; mh < 4; mh += MH) {
      // the accumulators START at the biases (a lane's registers of a tile pair are 8 consecutive units of one row: the bias depends
        // on the unit only), so the epilogue carries no bias registers
        f32x4 acc[MH][8];
        {
          float bR[8], bZ[8], bN[8], bI[8];
          ld_f32x8(a.b_hh + u, bR); ld_f32x8(a.b_hh + H + u, bZ); ld_f32x8(a.b_hh + 2 * H + u, bN);
  #pragma unroll
          for (int e = 0; e < 8; e++) bI[e] = 0.f;
          if constexpr (EMB) {
            float t8[8];
            ld_f32x8(a.b_ih + u, t8);
  #pragma unroll
            for (int e = 0; e < 8; e++) bR[e] += t8[e];
            ld_f32x8(a.b_ih + H + u, t8);
  #pragma unroll
            for (int e = 0; e < 8; e++) bZ[e] += t8[e];
            ld_f32x8(a.b_ih + 2 * H + u, bI);
          }
  #pragma unroll
          for (int i = 0; i < MH; i++) {
            acc[i][0] = f32x4{bR[0], bR[1], bR[2], bR[3]}; acc[i][1] = f32x4{bR[4], bR[5], bR[6], bR[7]};
            acc[i][2] = f32x4{bZ[0], bZ[1], bZ[2], bZ[3]}; acc[i][3] = f32x4{bZ[4], bZ[5], bZ[6], bZ[7]};
            acc[i][4] = f32x4{bN[0], bN[1], bN[2], bN[3]}; acc[i][5] = f32x4{bN[4], bN[5], bN[6], bN[7]};
            acc[i][6] = f32x4{bI[0], bI[1], bI[2], bI[3]}; acc[i][7] = f32x4{bI[4], bI[5], bI[6], bI[7]};
          }
        }
        const int tl[6] = {ut0, ut0 + 1, NUT + ut0, NUT + 1 + ut0, 2 * NUT + ut0, 2 * NUT + 1 + ut0};
        // epilogue operands of the pass (GC and the fp32 state of this lane's cells) are requested BEFORE the products: they come
        // from HBM, and waiting for them per M tile in the epilogue exposed that latency 16 times per step
        bf16x8 gq[MH][3]; float4 hq[MH][2];
  #pragma unroll
        for (int i = 0; i < MH; i++) {
          if constexpr (!EMB) {
            const __bf16* g = a.gc + gate_off(grow[mh + i], u, R);            // column-blocked by 32: [3H/32][R][32]
            const long gstep = R * H;
  #pragma unroll
            for (int gt = 0; gt < 3; gt++) gq[i][gt] = ldnt_bf16x8(g + gt * gstep);
          }
          if constexpr (H != 128) {                                        // (H = 128: loaded per tile in the epilogue -- registers)
            const float* hp = a.HN + (long)n * RH + grow[mh + i] * H + u;
            hq[i][0] = ldnt_f4(hp); hq[i][1] = ldnt_f4(hp + 4);
          }
        }
        // software-pipelined stream of KBH + 4 k-blocks through a ring of 4 fragment buffers: the loads of k-block k+3 are issued
        // before the MFMAs of k-block k (24 MFMAs = ~400 cycles per k-block against ~900 cycles of L2 latency)
        // (H = 128: two workgroups share a CU -- both directions of the note-summary GRU run side by side -- so half the registers:
        // a ring of 2)
        constexpr int RD = H == 128 ? 2 : 4;
        bf16x8 b[RD][6];
        auto ldw = [&](bf16x8 (&d)[6], int k) {                            // k < KBH: W_hh block (k + krot) % KBH; else W_x block k - KBH
  #pragma unroll
          for (int j = 0; j < 6; j++)
            d[j] = k < KBH ? a.w_hh[((long)tl[j] * KBH + ((k + krot) & (KBH - 1))) * 64 + lane] : a.w_x[((long)tl[j] * 4 + (k - KBH)) * 64 + lane];
        };
  #pragma unroll
        for (int k = 0; k < RD - 1; k++) ldw(b[k], k);
  #pragma unroll
        for (int k = 0; k < KT; k++) {
          if (k + RD - 1 < KT) ldw(b[(k + RD - 1) % RD], k + RD - 1);
          const bool tokpart = k >= KBH;
          const __bf16* A = tokpart ? tok16 : hc;
          const int lda = tokpart ? NT16LD : HLD, kb = tokpart ? k - KBH : ((k + krot) & (KBH - 1));
  #pragma unroll
          for (int i = 0; i < MH; i++) {
            const bf16x8 av = *reinterpret_cast<const bf16x8*>(A + ((mh + i) * 16 + rl) * lda + kb * 32 + kq);
  #pragma unroll
            for (int j = 0; j < 6; j++) {
              const int slot = j < 4 ? j : (tokpart ? j + 2 : j);
              acc[i][slot] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[k % RD][j], av, acc[i][slot], 0, 0, 0);
            }
          }
        }
        // ---- epilogue: GRU cell on this lane's cells of the pass
  #pragma unroll
        for (int i = 0; i < MH; i++) {
          float gR[8], gZ[8], gN[8], aR[8], aZ[8], aH[8], aI[8], hp[8];
  #pragma unroll
          for (int e = 0; e < 8; e++) {
            gR[e] = EMB ? 0.f : (float)gq[i][0][e]; gZ[e] = EMB ? 0.f : (float)gq[i][1][e]; gN[e] = EMB ? 0.f : (float)gq[i][2][e];
          }
          // fp32 state of these cells: written by this very lane one step ago (the whole fp32 state, 128 KB per workgroup at H = 512,
          // fits neither LDS next to the bf16 operand copies nor the register file next to the accumulators)
          if constexpr (H == 128) {
            const float* hpp = a.HN + (long)slot * RH + grow[mh + i] * H + u;
            hq[i][0] = ldnt_f4(hpp); hq[i][1] = ldnt_f4(hpp + 4);
          }
          hp[0] = hq[i][0].x; hp[1] = hq[i][0].y; hp[2] = hq[i][0].z; hp[3] = hq[i][0].w;
          hp[4] = hq[i][1].x; hp[5] = hq[i][1].y; hp[6] = hq[i][1].z; hp[7] = hq[i][1].w;
          pair_to_rows(acc[i][0], acc[i][1], aR); pair_to_rows(acc[i][2], acc[i][3], aZ);
          pair_to_rows(acc[i][4], acc[i][5], aH); pair_to_rows(acc[i][6], acc[i][7], aI);
          const bool live = !EMB || tt < len[mh + i];
          float r[8], z[8], nn[8], hn[8], h[8];
  #pragma unroll
          for (int e = 0; e < 8; e++) {
            r[e] = nsig(aR[e] + gR[e]);
            z[e] = nsig(aZ[e] + gZ[e]);
            hn[e] = aH[e];
            nn[e] = ntanh(aI[e] + gN[e] + r[e] * hn[e]);
            if (!live) { r[e] = 0.f; z[e] = 1.f; nn[e] = 0.f; }             // masked row: h' = h, zero gate gradients
            h[e] = (1.0f - z[e]) * nn[e] + z[e] * hp[e];
          }
          st_bf16x8(hn_ + ((mh + i) * 16 + erow) * HLD + u, h);
          if (ok[mh + i]) {
            const long o = (long)(n + 1) * RH + grow[mh + i] * H + u;
            st_f32x8(a.HN + o, h);                                       // read back next step: default policy
            // (HN16 is NOT written here: 16 bytes per lane are 64-byte row segments, half a cache line.  The whole bf16 state of the
            // panel is in LDS once the step's barrier has passed -- it goes out from there in full rows, under the next step's products)
            if (a.gates) {
              __bf16* gp = a.gates + (long)n * 4 * RH + gate_off(grow[mh + i], u, R);
              stnt_bf16x8(gp, r); stnt_bf16x8(gp + RH, z); stnt_bf16x8(gp + 2 * RH, nn); stnt_bf16x8(gp + 3 * RH, hn);
            }
            }
          __builtin_amdgcn_sched_barrier(0);                               // keep the M tiles' epilogues (and the passes) apart: register pressure
        }
      }
    }
    lds_barrier();
    slot = n + 1; cur = nxt;
    // bf16 state after this step: LDS (complete, and stable until the step after next overwrites this buffer) -> HN16 slot n + 1, one
    // kilobyte (H = 512) of contiguous row per 64 lanes
    {
      const __bf16* hs = h16 + cur * NRP * HLD;
      for (int i = tid; i < NRP * (H / 8); i += 256) {
        const int row = i / (H / 8), c8 = (i % (H / 8)) * 8;
        if (r0 + row < R)
          __builtin_nontemporal_store(*reinterpret_cast<const bf16x8*>(hs + row * HLD + c8),
                                      reinterpret_cast<bf16x8*>(a.HN16 + (long)(n + 1) * RH + (r0 + row) * H + c8));
      }
    }
  }
  if constexpr (EMB) {
    if (a.out) {                                                           // final state = the last written slot
      __syncthreads();
      for (int i = tid; i < NRP * (H / 8); i += 256) {
        const int row = i / (H / 8), c8 = (i % (H / 8)) * 8;
        if (r0 + row < R) {
          float v[8];
          ld_f32x8(a.HN + (long)slot * RH + (r0 + row) * H + c8, v);
          st_f32x8(a.out + (a.perm ? (long)a.perm[r0 + row] : r0 + row) * a.out_ld + c8, v);
        }
      }
    }
  }
}

template <int H, bool EMB>
__global__ __launch_bounds__(256, H == 128 ? 2 : 1) void row_gru_fwd_kernel(RowGruFwdArgs a) {
  // a launch of the latency chain wins instruction issue against sibling-stream products.  The ground-truth note summaries (EMB) are
  // NOT on the chain when they run beside the encoders at the head of the step (their consumer is the time GRU, after the encoders):
  // there the caller's priority state decides (ptv_gemm_priority, 0 inside a side-stream call)
  if (!EMB || a.prio) __builtin_amdgcn_s_setprio(3);
  extern __shared__ __attribute__((aligned(16))) char nsm[];
  // (Measured and removed, round 5: with the rows sorted by length, HALF the panels as the grid and workgroup b running panel b, then panel
  // G-1-b -- a long one and a short one, equal work per workgroup.  The launch is latency-bound per step, not throughput-bound: twice the
  // sequential steps per workgroup made the step 0.3 ms slower, 8.08-8.10 against 7.75-7.88 ms; profiles/r05_ab_runs.txt)
  row_gru_fwd_body<H, EMB>(a, blockIdx.x, nsm);
}

// =============================================================================================
// BPTT
// =============================================================================================
struct RowGruBwdArgs {
  const bf16x8* wt;                // pair-interleaved packing of W_hh^T: [H/16 tiles of output units][3H/32 kb][64]
  const float* HN;                 // EMB: fp32 states [T+1][R][H]
  const __bf16* HN16;              // !EMB (the notes GRU, whose forward keeps the fp32 state in registers): the bf16 states [T+1][R][H]
  const __bf16* gates;             // EMB: planes unit-blocked by 32 (gate_off); !EMB: by 16 (the wave-role forward, notes_roles.hip)
  const __bf16* ext;               // bf16 gradient arriving at the state after step s: the [T*R][H] matrix COLUMN-BLOCKED by 32 ([H/32][T*R][32]), or null
  const float* dh_last; long last_ld;   // gradient arriving at the final state only (rows of stride last_ld), or null
  const int* lengths;              // EMB: the lengths the forward ran with (it skipped the panel's fully masked steps), or null
  int* top_step;                   // atomicMax'ed with the last step at which a gradient arrived for any panel (!EMB) / the last time
                                   // index any row reaches (EMB with lengths), or null
  __bf16* dgi; __bf16* dgh;        // dgi [T][R][3H] by TIME index; dgh by processing step: [T][R][3H] (EMB) or its n third only [T][R][H]
  float* dh0;                      // [R][H] or null
  __bf16* scratch;                 // [grid][2][3H/8 chunks][64 rows][8]: dgh of the workgroup's rows, K-blocked (A operand of the next step)
  int R, T, reverse;
  int skip;                        // pass over steps whose result is exactly zero
  const int* perm;                 // EMB, or null: the forward's row permutation -- dh_last, lengths and dgi (operand of products against the
                                   // forward's INPUT rows) by row perm[p], HN / gates / dgh (against the forward's states) by position p
  const int* bound;                // !EMB, or null: no gradient arrives after step *bound and nobody reads dgi / dgh beyond top_step <= *bound
  const int* row_len;              // !EMB, or null (needs bound): rows sorted by descending length -- the panel's own last live step is row_len[first row] - 1
};

// EMB = false: the notes GRU (gradient arrives at every state: ext; forward time order; dh0 wanted); EMB = true: a direction of
// dec_notes_emb_gru (gradient arrives at the final state only: dh_last; optional reversed time)
template <int H, bool EMB>
__device__ __forceinline__ void row_gru_bwd_body(const RowGruBwdArgs& a, const long panel, char* nsm) {
  constexpr int KT = 3 * H / 32, NTW = H / 64, NCH = 3 * H / 8;           // k-blocks, output tiles per wave, scratch chunks
  float* dhz = reinterpret_cast<float*>(nsm);                            // [64][H] fp32: dh (x) z carried to the earlier step
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rl = lane & 15, kqi = lane >> 4;
  const int erow = lane & 15, eq = lane >> 4;
  const long R = a.R;
  const long r0 = panel * NRP;
  const long RH = R * H, R3H = 3 * RH;
  __bf16* sc = a.scratch + (long)blockIdx.x * 2 * (NCH * NRP * 8);
  long grow[4]; bool ok[4];
  long gnat[4];                                                          // the same rows in the order of the forward's INPUT (EMB with perm)
#pragma unroll
  for (int i = 0; i < 4; i++) {
    ok[i] = r0 + i * 16 + erow < R; grow[i] = min(r0 + i * 16 + erow, R - 1);
    gnat[i] = grow[i];
    if constexpr (EMB) { if (a.perm) gnat[i] = a.perm[grow[i]]; }
  }
  int pmax = a.T, gmax = a.T;
  if constexpr (EMB) {
    if (a.lengths) {                                                     // given iff the forward skipped: must match it
      // (the scratch word is the first word of dhz, zeroed afterwards: the kernel's LDS is exactly 64*H*4 bytes -- at H = 128 two of
      // these workgroups plus one 96-KB persistent-GRU workgroup fill the CU's 160 KB to the byte, 16 bytes more and they exclude
      // each other: the note-summary BPTT could not run under the encoders' BPTT chains)
      int* misc = reinterpret_cast<int*>(dhz);
      if (tid == 0) misc[0] = 0;
      __syncthreads();
      if (tid < NRP && r0 + tid < R) atomicMax(misc, a.lengths[a.perm ? (long)a.perm[r0 + tid] : r0 + tid]);
      __syncthreads();
      pmax = min(misc[0], a.T);
      if (a.top_step && tid == 0 && pmax > 0) atomicMax(a.top_step, pmax - 1);   // last TIME index with a live row in any panel
      __syncthreads();
      // the launch-wide limit the consumers of dgi / dgh will be given (top_step = gmax - 1 once every panel has reported): the zero rows
      // of time indices at or beyond gmax are never read -- not written (200 of this launch's 570 MB of stores, round 6)
      if (a.top_step && (R & 31) == 0) {
        int g = 0;
        for (long i = tid; i < R; i += 256) g = max(g, a.lengths[i]);
        if (tid == 0) misc[0] = 0;
        __syncthreads();
        atomicMax(misc, g);
        __syncthreads();
        gmax = min(misc[0], a.T);
        __syncthreads();
      }
    }
  }
  for (int i = tid; i < NRP * H; i += 256) dhz[i] = 0.f;
  __syncthreads();
  bool first_active = true;                                               // no later step has handed a dgh over yet
  int s_top = a.T - 1;
  if constexpr (!EMB) {
    // A step whose arriving gradient is zero for all 64 rows, with nothing arriving from later steps either, produces exactly zero
    // gate gradients: the loss ignores the padded note slots (CrossEntropyLoss(ignore_index), ptvae.py:498-511), which are the LATE
    // steps of every row -- on this data 8 of the 15.  Tested on the arriving gradient itself (one 64-KB read per step), so it holds
    // for whatever loss produced it.  The BPTT proper starts at the last step that has something.
    if (a.skip && a.bound) s_top = min(a.T - 1, max(*a.bound, -1));         // (steps beyond a caller-given bound are not touched at all)
    const int s_panel = (a.skip && a.bound && a.row_len) ? min(s_top, a.row_len[r0 & ~127L] - 1) : s_top;     // (deadness per 128-row block, as the heads)
    for (; a.skip && s_top >= 0; s_top--) {
      unsigned nz = 0;
      if (s_top <= s_panel)
      for (int i = tid; i < NRP * (H / 8); i += 256) {
        const int row = i / (H / 8), c8 = (i % (H / 8)) * 8;
        if (r0 + row < R) {
          typedef __attribute__((ext_vector_type(4))) unsigned u4v;
          const u4v w = *reinterpret_cast<const u4v*>(a.ext + ext_off(s_top, r0 + row, c8, R, a.T));
          nz |= (w[0] | w[1] | w[2] | w[3]) & 0x7fff7fffu;                // -0.0 is zero too
        }
      }
      if (__syncthreads_or(nz != 0)) break;
      for (int i = tid; i < NRP * (H / 8); i += 256) {
        const int row = i / (H / 8), c8 = (i % (H / 8)) * 8;
        if (r0 + row < R) {
          const bf16x8 zz = {};
          __bf16* pi = a.dgi + (long)s_top * R3H + (r0 + row) * (3 * H) + c8;
          *reinterpret_cast<bf16x8*>(pi) = zz; *reinterpret_cast<bf16x8*>(pi + H) = zz; *reinterpret_cast<bf16x8*>(pi + 2 * H) = zz;
          *reinterpret_cast<bf16x8*>(a.dgh + (long)s_top * RH + (r0 + row) * H + c8) = zz;
        }
      }
    }
    if (a.top_step && tid == 0 && s_top >= 0) atomicMax(a.top_step, s_top);   // consumers of dgi / dgh may stop after this step's rows
  }

  for (int s = s_top; s >= (a.dh0 ? -1 : 0); s--) {
    const int tt = s < 0 ? 0 : ((EMB && a.reverse) ? a.T - 1 - s : s);
    if constexpr (EMB) {
      if (s >= 0 && tt >= gmax) continue;                                  // beyond the launch-wide limit: rows nobody reads
      if (s >= 0 && tt >= pmax) {                                          // the forward passed the state through: zero gate gradients
        for (int i = tid; i < NRP * (3 * H / 8); i += 256) {
          const int row = i / (3 * H / 8), c8 = (i % (3 * H / 8)) * 8;
          if (r0 + row < R) {
            const bf16x8 zz = {};
            *reinterpret_cast<bf16x8*>(a.dgh + (long)s * R3H + (r0 + row) * (3 * H) + c8) = zz;
            *reinterpret_cast<bf16x8*>(a.dgi + (long)tt * R3H + (a.perm ? (long)a.perm[r0 + row] : r0 + row) * (3 * H) + c8) = zz;
          }
        }
        continue;
      }
    }
    const bool last = first_active;
    first_active = false;
    const __bf16* scr = sc + ((s + 1) & 1) * (NCH * NRP * 8);             // dgh_{s+1}, written by the previous iteration
    __bf16* scw = sc + (s & 1) * (NCH * NRP * 8);
    // HBM operands of the epilogue items (tile pairs x 4 M tiles; saved gates, previous state, external gradient) run 2 items
    // ahead of the arithmetic through a ring of 3 register sets; the first two are requested before the products
    struct Ops { bf16x8 g[4]; bf16x8 ex; float4 hp[2]; bf16x8 hp16; };
    Ops ops[3];
    constexpr int NIT = (NTW / 2) * 4;
    auto ldops = [&](Ops& o, int it) {
      const int pr = it >> 2, i = it & 3;
      const int u = (wave * NTW + pr * 2) * 16 + eq * 8;
      const long base = (long)s * RH + grow[i] * H + u;
      const __bf16* gp = a.gates + (long)s * 4 * RH + (EMB ? gate_off(grow[i], u, R) : ((long)(u >> 4) * R + grow[i]) * 16 + (u & 15));
#pragma unroll
      for (int q = 0; q < 4; q++) o.g[q] = ldnt_bf16x8(gp + q * RH);
      if constexpr (!EMB) {
        o.ex = ldnt_bf16x8(a.ext + ext_off(s, grow[i], u, R, a.T));
        o.hp16 = *reinterpret_cast<const bf16x8*>(a.HN16 + base);
      } else {
        o.hp[0] = ldnt_f4(a.HN + base); o.hp[1] = ldnt_f4(a.HN + base + 4);
      }
    };
    if (s >= 0) { ldops(ops[0], 0); ldops(ops[1], 1); }
    // acc[i][j]: M tile i, unit tile wave*NTW + j
    f32x4 acc[4][NTW];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int j = 0; j < NTW; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!last) {
      // dh = dgh_{s+1} . W_hh: K = 3H; A fragments from the K-blocked scratch (chunk = 8 k of 64 rows: the 16 lanes of a quad read
      // 256 contiguous bytes), B fragments of W_hh^T from L2; ring of 3: the loads of k-block k+2 precede the MFMAs of k-block k
      bf16x8 bw[3][NTW], aw[3][4];
      auto ldg = [&](int buf, int k) {
#pragma unroll
        for (int j = 0; j < NTW; j++) bw[buf][j] = a.wt[((long)(wave * NTW + j) * KT + k) * 64 + lane];
#pragma unroll
        for (int i = 0; i < 4; i++) aw[buf][i] = *reinterpret_cast<const bf16x8*>(scr + (((long)(k * 4 + kqi)) * NRP + i * 16 + rl) * 8);
      };
      auto mm = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
          for (int j = 0; j < NTW; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[buf][j], aw[buf][i], acc[i][j], 0, 0, 0);
      };
      static_assert(KT % 3 == 0, "ring of 3");
      ldg(0, 0); ldg(1, 1);
#pragma unroll 1
      for (int k = 0; k < KT; k += 3) {
        ldg(2, k + 2);
        mm(0);
        if (k + 3 < KT) ldg(0, k + 3);
        mm(1);
        if (k + 4 < KT) ldg(1, k + 4);
        mm(2);
      }
    }
    // ---- epilogue items in the MFMA lane layout (pair-interleaved tiles: 8 consecutive units per lane)
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int pr = it >> 2, i = it & 3;
      const int u = (wave * NTW + pr * 2) * 16 + eq * 8;
      if (s >= 0 && it + 2 < NIT) ldops(ops[(it + 2) % 3], it + 2);
      float dh[8];
      pair_to_rows(acc[i][2 * pr], acc[i][2 * pr + 1], dh);
      float* dzp = dhz + (i * 16 + erow) * H + u;
      float cz[8];
      ld_f32x8(dzp, cz);
      if (s < 0) {                                                       // dh0 = dhz_0 + dgh_0 . W_hh
#pragma unroll
        for (int e = 0; e < 8; e++) dh[e] += cz[e];
        if (ok[i]) st_f32x8(a.dh0 + grow[i] * H + u, dh);
        continue;
      }
      const Ops& o = ops[it % 3];
      float hp[8];
      if constexpr (EMB) {
        hp[0] = o.hp[0].x; hp[1] = o.hp[0].y; hp[2] = o.hp[0].z; hp[3] = o.hp[0].w; hp[4] = o.hp[1].x; hp[5] = o.hp[1].y; hp[6] = o.hp[1].z; hp[7] = o.hp[1].w;
      } else {
#pragma unroll
        for (int e = 0; e < 8; e++) hp[e] = (float)o.hp16[e];
      }
      float lastg[8];
#pragma unroll
      for (int e = 0; e < 8; e++) lastg[e] = 0.f;
      if constexpr (EMB) { if (last && a.dh_last) ld_f32x8(a.dh_last + gnat[i] * a.last_ld + u, lastg); }
      float dr[8], dz[8], dn[8], dnr[8], dq[8];
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const float gr = (float)o.g[0][e], gz = (float)o.g[1][e], gn = (float)o.g[2][e], gh = (float)o.g[3][e];
        const float d = dh[e] + cz[e] + (EMB ? lastg[e] : (float)o.ex[e]);
        dn[e] = d * (1.0f - gz) * (1.0f - gn * gn);
        dz[e] = d * (hp[e] - gn) * gz * (1.0f - gz);
        dr[e] = dn[e] * gh * gr * (1.0f - gr);
        dnr[e] = dn[e] * gr;
        dq[e] = d * gz;
      }
      st_f32x8(dzp, dq);
      // the next step's A operand: chunk (gate*H + u)/8 of the K-blocked scratch, this row
      __bf16* sp = scw + ((long)(u >> 3) * NRP + i * 16 + erow) * 8;
      st_bf16x8(sp, dr); st_bf16x8(sp + (long)(H / 8) * NRP * 8, dz); st_bf16x8(sp + (long)(2 * H / 8) * NRP * 8, dnr);
      if (ok[i]) {
        if constexpr (EMB) {
          __bf16* ph = a.dgh + (long)s * R3H + grow[i] * (3 * H) + u;
          stnt_bf16x8(ph, dr); stnt_bf16x8(ph + H, dz); stnt_bf16x8(ph + 2 * H, dnr);
        } else {
          // forward time order: the r and z thirds of dgh ARE dgi's (same rows, same step) -- only the n third (dn * r) goes out
          stnt_bf16x8(a.dgh + (long)s * RH + grow[i] * H + u, dnr);
        }
        __bf16* pi = a.dgi + (long)tt * R3H + gnat[i] * (3 * H) + u;
        stnt_bf16x8(pi, dr); stnt_bf16x8(pi + H, dz); stnt_bf16x8(pi + 2 * H, dn);
      }
    }
    __syncthreads();                                                     // scratch + dhz of this step complete before the next products
  }
}

template <int H, bool EMB>
__global__ __launch_bounds__(256, H == 128 ? 2 : 1) void row_gru_bwd_kernel(RowGruBwdArgs a) {
  __builtin_amdgcn_s_setprio(3);                                         // a launch of the latency chain: wins instruction issue against sibling-stream products
  extern __shared__ __attribute__((aligned(16))) char nsm[];
  row_gru_bwd_body<H, EMB>(a, blockIdx.x, nsm);
}

template <int H, bool EMB>
static int launch_fwd(const RowGruFwdArgs& a, hipStream_t s) {
  const size_t lds = (size_t)(2 * NRP * (H + 16) + NRP * NT16LD) * sizeof(__bf16) + 16;
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(row_gru_fwd_kernel<H, EMB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return PTV_ERR_LAUNCH;
    attr = true;
  }
  hipLaunchKernelGGL((row_gru_fwd_kernel<H, EMB>), dim3((unsigned)((a.R + NRP - 1) / NRP)), dim3(256), lds, s, a);
  return PTV_OK;
}
template <int H, bool EMB>
static int launch_bwd(const RowGruBwdArgs& a, hipStream_t s) {
  const size_t lds = (size_t)NRP * H * sizeof(float);
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(row_gru_bwd_kernel<H, EMB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return PTV_ERR_LAUNCH;
    attr = true;
  }
  hipLaunchKernelGGL((row_gru_bwd_kernel<H, EMB>), dim3((unsigned)((a.R + NRP - 1) / NRP)), dim3(256), lds, s, a);
  return PTV_OK;
}

// perm = the rows ordered by DESCENDING length, ties in row order (a stable counting sort: one workgroup, each thread owns a contiguous
// chunk of rows; cnt[bin][thread] -> exclusive scan in (bin descending, thread ascending) order -> positions).  lengths in [0, nb - 1].
__global__ __launch_bounds__(1024) void rows_by_length_kernel(const int* __restrict__ lengths, int* __restrict__ perm, long R, int nb) {
  extern __shared__ int cnt[];                                           // [nb][1024] + [nb][16] wave totals + [nb] bin bases
  int* wtot = cnt + nb * 1024;
  int* base = wtot + nb * 16;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long chunk = (R + 1023) / 1024, r0 = tid * chunk, r1 = min(R, r0 + chunk);
  for (int b = 0; b < nb; b++) cnt[b * 1024 + tid] = 0;
  for (long r = r0; r < r1; r++) cnt[min(max(lengths[r], 0), nb - 1) * 1024 + tid]++;
  // per bin: exclusive scan over the 1024 threads (wave scan by shuffles, wave totals through LDS)
  for (int b = 0; b < nb; b++) {
    const int v = cnt[b * 1024 + tid];
    int incl = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(incl, d, 64);
      if (lane >= d) incl += o;
    }
    if (lane == 63) wtot[b * 16 + wave] = incl;
    cnt[b * 1024 + tid] = incl - v;                                      // exclusive within the wave
  }
  __syncthreads();
  if (tid < nb) {                                                        // bin tid: its waves' offsets, its total
    int run = 0;
    for (int w = 0; w < 16; w++) { const int t = wtot[tid * 16 + w]; wtot[tid * 16 + w] = run; run += t; }
    base[tid] = run;
  }
  __syncthreads();
  if (tid == 0) {                                                        // bins in descending order of length
    int run = 0;
    for (int b = nb - 1; b >= 0; b--) { const int t = base[b]; base[b] = run; run += t; }
  }
  __syncthreads();
  for (long r = r0; r < r1; r++) {
    const int b = min(max(lengths[r], 0), nb - 1);
    const int pos = base[b] + wtot[b * 16 + wave] + cnt[b * 1024 + tid]++;
    perm[pos] = (int)r;
  }
}

}  // namespace ptv

using namespace ptv;

extern "C" int ptv_rows_by_length(const int* lengths, int* perm, long R, int max_len, void* stream) {
  if (!lengths || !perm || R <= 0 || R > 0x7fffffffL || max_len < 0) return PTV_ERR_ARG;
  const int nb = max_len + 1;
  const size_t lds = (size_t)(nb * 1024 + nb * 16 + nb) * sizeof(int);
  if (lds > 160 * 1024) return PTV_ERR_UNSUPPORTED;                       // lengths up to 38
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(rows_by_length_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return PTV_ERR_LAUNCH;
    attr = true;
  }
  hipLaunchKernelGGL(rows_by_length_kernel, dim3(1), dim3(1024), lds, (hipStream_t)stream, lengths, perm, R, nb);
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

extern "C" int ptv_notes_bwd8(const void* wt, const void* HN16, const void* gates, const void* ext, void* dgi, void* dgh, float* dh0, void* scratch,
                              long R, int T, const int* bound, const int* row_len, int* top_step, void* stream);
// the notes GRU's BPTT: 1 = the 8-wave kernel of notes_roles.hip (LDS-resident operand, carry in registers), 0 = the 4-wave kernel of this file
static int g_notes_bwd8 = 1;
extern "C" int ptv_notes_bwd_variant(int eight_waves) { g_notes_bwd8 = eight_waves ? 1 : 0; return PTV_OK; }

extern "C" int ptv_row_gru_persist_fwd_perm(int H, const void* w_hh, const void* w_x, const float* b_hh, const float* b_ih, const void* gc,
                                            const float* x, long x_step, const int* lengths, const int* perm, float* HN, void* HN16,
                                            void* gates, float* out, long out_ld, long R, int T, int reverse, void* stream);
extern "C" int ptv_row_gru_persist_fwd(int H, const void* w_hh, const void* w_x, const float* b_hh, const float* b_ih, const void* gc,
                                       const float* x, long x_step, const int* lengths, float* HN, void* HN16, void* gates,
                                       float* out, long out_ld, long R, int T, int reverse, void* stream) {
  return ptv_row_gru_persist_fwd_perm(H, w_hh, w_x, b_hh, b_ih, gc, x, x_step, lengths, nullptr, HN, HN16, gates, out, out_ld, R, T, reverse, stream);
}

extern "C" int ptv_row_gru_persist_fwd_perm(int H, const void* w_hh, const void* w_x, const float* b_hh, const float* b_ih, const void* gc,
                                            const float* x, long x_step, const int* lengths, const int* perm, float* HN, void* HN16,
                                            void* gates, float* out, long out_ld, long R, int T, int reverse, void* stream) {
  if (!w_hh || !w_x || !b_hh || !x || !HN || !HN16 || R <= 0 || T <= 0 || (H != 512 && H != 128)) return PTV_ERR_ARG;
  if (out && (out_ld & 3)) return PTV_ERR_ARG;
  // H = 512 is the notes GRU (gc given, bias folded, dense): the wave-role kernel of notes_roles.hip; H = 128 the note-summary GRU (b_ih
  // given, mask / reverse / final state)
  if (H == 512) {
    if (!gc || b_ih || lengths || perm || reverse || out || x_step != R * NE) return PTV_ERR_UNSUPPORTED;
    return ptv_notes_gru_persist_fwd(w_hh, w_x, b_hh, gc, x, HN, HN16, gates, R, T, stream);
  }
  RowGruFwdArgs a{(const bf16x8*)w_hh, (const bf16x8*)w_x, b_hh, b_ih, (const __bf16*)gc, x, x_step, lengths, HN, (__bf16*)HN16,
                  (__bf16*)gates, out, out_ld, (int)R, T & 0xff, reverse, T >> 8, g_zero_skip, g_gemm_prio, perm};
  const int pi = prof::want(3, (int)R, H) ? prof::begin((hipStream_t)stream) : -1;
  if (gc || !b_ih) return PTV_ERR_UNSUPPORTED;
  PTV_TRY((launch_fwd<128, true>(a, (hipStream_t)stream)));
  if (pi >= 0) prof::end(pi, (hipStream_t)stream, 2.0 * R * 3.0 * H * (H + NE) * (T & 0xff));
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

extern "C" long ptv_row_gru_persist_scratch_elems(int H, long R) { return ((R + NRP - 1) / NRP) * 2 * ((3L * H / 8) * NRP * 8); }

extern "C" int ptv_row_gru_persist_bwd_perm(int H, const void* wt, const void* HN, const void* gates, const void* ext,
                                            const float* dh_last, long last_ld, const int* lengths, const int* perm, void* dgi, void* dgh,
                                            float* dh0, void* scratch, long R, int T, int reverse, int* top_step, void* stream);
static int row_gru_bwd_any(int H, const void* wt, const void* HN, const void* gates, const void* ext, const float* dh_last, long last_ld,
                           const int* lengths, const int* perm, void* dgi, void* dgh, float* dh0, void* scratch, long R, int T, int reverse,
                           const int* bound, const int* row_len, int* top_step, void* stream);
extern "C" int ptv_row_gru_persist_bwd(int H, const void* wt, const void* HN, const void* gates, const void* ext,
                                       const float* dh_last, long last_ld, const int* lengths, void* dgi, void* dgh, float* dh0,
                                       void* scratch, long R, int T, int reverse, int* top_step, void* stream) {
  return ptv_row_gru_persist_bwd_perm(H, wt, HN, gates, ext, dh_last, last_ld, lengths, nullptr, dgi, dgh, dh0, scratch, R, T, reverse, top_step, stream);
}

extern "C" int ptv_row_gru_persist_bwd_perm(int H, const void* wt, const void* HN, const void* gates, const void* ext,
                                            const float* dh_last, long last_ld, const int* lengths, const int* perm, void* dgi, void* dgh,
                                            float* dh0, void* scratch, long R, int T, int reverse, int* top_step, void* stream) {
  return row_gru_bwd_any(H, wt, HN, gates, ext, dh_last, last_ld, lengths, perm, dgi, dgh, dh0, scratch, R, T, reverse, nullptr, nullptr, top_step, stream);
}

static int row_gru_bwd_any(int H, const void* wt, const void* HN, const void* gates, const void* ext, const float* dh_last, long last_ld,
                           const int* lengths, const int* perm, void* dgi, void* dgh, float* dh0, void* scratch, long R, int T, int reverse,
                           const int* bound, const int* row_len, int* top_step, void* stream) {
  if (!wt || !HN || !gates || !dgi || !dgh || !scratch || R <= 0 || T <= 0 || (H != 512 && H != 128)) return PTV_ERR_ARG;
  if (dh_last && (last_ld & 3)) return PTV_ERR_ARG;
  if (H == 512 && (lengths || perm)) return PTV_ERR_UNSUPPORTED;
  if (perm && dh0) return PTV_ERR_UNSUPPORTED;                            // (dh0 would be indexed by position)
  RowGruBwdArgs a{(const bf16x8*)wt, H == 512 ? nullptr : (const float*)HN, H == 512 ? (const __bf16*)HN : nullptr, (const __bf16*)gates, (const __bf16*)ext, dh_last, last_ld, lengths, top_step, (__bf16*)dgi, (__bf16*)dgh, dh0,
                  (__bf16*)scratch, (int)R, T, reverse, g_zero_skip, perm, bound, row_len};
  const int pi = prof::want(4, (int)R, H) ? prof::begin((hipStream_t)stream) : -1;
  if (H == 512 && (!ext || dh_last || reverse)) return PTV_ERR_UNSUPPORTED;
  if (H == 128 && ext) return PTV_ERR_UNSUPPORTED;
  PTV_TRY(H == 512 ? (launch_bwd<512, false>(a, (hipStream_t)stream)) : (launch_bwd<128, true>(a, (hipStream_t)stream)));
  if (pi >= 0) prof::end(pi, (hipStream_t)stream, 2.0 * R * 3.0 * H * H * (T - 1 + (dh0 ? 1 : 0)));
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

extern "C" long ptv_notes_gru_persist_scratch_elems(long R) { return ptv_row_gru_persist_scratch_elems(512, R); }

extern "C" int ptv_notes_gru_persist_bwd_rows(const void* wt, const void* HN16, const void* gates, const void* ext, void* dgi, void* dgh,
                                              float* dh0, void* scratch, long R, int T, const int* bound, const int* row_len, int* top_step,
                                              void* stream) {
  if (!ext || (bound && !top_step) || (row_len && !bound)) return PTV_ERR_ARG;
  if (g_notes_bwd8) return ptv_notes_bwd8(wt, HN16, gates, ext, dgi, dgh, dh0, scratch, R, T, bound, row_len, top_step, stream);
  return row_gru_bwd_any(512, wt, HN16, gates, ext, nullptr, 0, nullptr, nullptr, dgi, dgh, dh0, scratch, R, T & 0xffff, 0, bound, row_len, top_step,
                         stream);                                 // (this kernel always writes the zero rows: T bit 16 is a permission, not an order)
}
extern "C" int ptv_notes_gru_persist_bwd_top(const void* wt, const void* HN16, const void* gates, const void* ext, void* dgi, void* dgh,
                                             float* dh0, void* scratch, long R, int T, const int* bound, int* top_step, void* stream) {
  return ptv_notes_gru_persist_bwd_rows(wt, HN16, gates, ext, dgi, dgh, dh0, scratch, R, T, bound, nullptr, top_step, stream);
}
extern "C" int ptv_notes_gru_persist_bwd(const void* wt, const void* HN16, const void* gates, const void* ext, void* dgi, void* dgh,
                                         float* dh0, void* scratch, long R, int T, int* top_step, void* stream) {
  return ptv_notes_gru_persist_bwd_top(wt, HN16, gates, ext, dgi, dgh, dh0, scratch, R, T, nullptr, top_step, stream);
}

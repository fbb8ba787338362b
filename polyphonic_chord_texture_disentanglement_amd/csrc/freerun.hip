// freerun.hip -- the free-running / scheduled-sampling PianoTree decoder as row-partitioned persistent kernels.
//
// Reference: PtvaeDecoder.decode_notes / decode_note (ptvae.py:336-428) and the re-summarisation of the predicted notes
// (ptvae.py:476-486) -- what the reference's train.py schedule runs from its third batch on and what
// inference_decode runs always (model.py:124-131).  Every next token is an argmax of the previous step, so the forward is a
// 32 x 15 x (1 + 5) deep dependency chain; as per-step launches on [B]-row windows it was ~9,000 launches per forward, each
// using a sliver of the chip (6.1k samples/s at B = 512 against 30k teacher-forced).
//
// Samples are independent, so the chain is partitioned by ROWS: a workgroup owns a panel of 16 samples (one MFMA M tile)
// and walks, for one time step t, ALL 15 note steps by itself:
//     notes-GRU cell  ->  pitch head  ->  argmax  ->  dur_hid  ->  5-step duration GRU + argmax feedback  ->  token embed
// with the panel's state in LDS / registers (h fp32 + bf16, logits, duration state) and the weights streamed from L2 in an
// MFMA-fragment-major packing (ptv_pack_mfma_b: a B fragment of 16 units x 32 k is ONE contiguous 1-KB wave load; reading
// fragments out of the row-major weight costs 64 cache-line lookups per load and binds on the address path).  Nothing is
// exchanged between workgroups: no flags, no residency requirement.  What the batched backward needs (states, gates,
// logits, tokens) is streamed out in the step-major layouts of functional_free.DecoderStepFn, which is unchanged.
// A second kernel runs the bi-GRU over a panel's 16 predicted notes (packed-sequence masking by the predicted length) that
// produces the next time-step token.  The time GRU step itself (M = B rows, H = 1024: 7.8 MB of weights per step) stays on
// the chip-wide step kernels of gru.hip.
// Specialised to the init_model() geometry (E = 128, Hn = 512, Hd = 64, He = 128, 130 pitches), bf16 MFMA operands, fp32
// state / logits -- the bf16 precision policy of the teacher-forced path.  Other configurations use the step loop.
#include <type_traits>
#include "common.hpp"
#include "prof.hpp"
#include "gemm_core.hpp"
#include "../../include/ptvae_hip.h"

namespace ptv {

constexpr int FE = 128, FHN = 512, FHD = 64, FNP = 130, FHE = 128;
constexpr int FP = 16;                       // rows per panel
constexpr int H16LD = FHN + 16;              // bf16 row strides: stride/2 words = 8 (mod 64) -> conflict-free b128 fragment reads
constexpr int T16LD = FE + 16;
constexpr int P16LD = 160 + 16;              // pitch logits as an MFMA operand, K padded 130 -> 160
constexpr int D16LD = FHD + 16;
constexpr int PITLD = 144;                   // fp32 logits row (9 tiles)

__device__ __forceinline__ float fsig(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float ftanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

// accumulator fragment (C layout: lane = row + 16*quad) -> epilogue layout (lane = 4*row + quad): afterwards 4 ADJACENT lanes
// hold the 16 units of one row, so global traffic of the epilogue goes out in 32-64 byte runs instead of 64 scattered pieces
__device__ __forceinline__ f32x4 to_rowmajor_lanes(const f32x4& v) {
  const int lane = threadIdx.x & 63, src = (lane >> 2) + 16 * (lane & 3);
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; e++) o[e] = __shfl(v[e], src, 64);
  return o;
}

// acc[j] += A(LDS bf16 [16][lda], K = KB*32) . Wp(tile j)^T ; packed tiles: frag (tile, kb) at wp[(tile*KB + kb)*64 + lane].
// The k loop is unrolled by UN only: UN*NT fragment loads (16 B per lane each) are in flight per iteration -- a full unroll
// lets the compiler hoist every load of the product and spill.
template <int NT, int KB, int UN>
__device__ __forceinline__ void panel_mma(const bf16x8* __restrict__ wp, const int (&tile)[NT], const __bf16* A, int lda, f32x4 (&acc)[NT]) {
  const int lane = threadIdx.x & 63, rl = lane & 15, kq = (lane >> 4) * 8;
  static_assert(KB % UN == 0 || KB < UN, "unroll factor");
#pragma unroll 1
  for (int k0 = 0; k0 < KB; k0 += UN) {
    bf16x8 b[UN][NT];
#pragma unroll
    for (int q = 0; q < UN; q++)
#pragma unroll
      for (int j = 0; j < NT; j++)
        if (k0 + q < KB) b[q][j] = wp[((long)tile[j] * KB + k0 + q) * 64 + lane];
#pragma unroll
    for (int q = 0; q < UN; q++) {
      if (k0 + q < KB) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(A + rl * lda + (k0 + q) * 32 + kq);
#pragma unroll
        for (int j = 0; j < NT; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[q][j], a, acc[j], 0, 0, 0);
      }
    }
  }
}

// the gate products of one pass (6 tiles: r0 r1 z0 z1 n0 n1): accH += h . W_hh^T (K = 512), accT += token . W_ih[:, Ht:]^T
// (K = 128) as ONE explicitly software-pipelined stream of 5 groups of 4 k-blocks (24 fragment loads = 24 KB per wave each):
// the loads of group g+1 are issued before the MFMAs of group g, with static register double buffers.  Left to the compiler the
// same loop came out either pipelined (14 us per note step) or with every group's latency exposed (36 us) from build to build.
template <int KBH, int KBT>
__device__ __forceinline__ void gate_products(const bf16x8* __restrict__ wh, const bf16x8* __restrict__ wt, const int (&tile)[6],
                                              const __bf16* Ah, int ldh, const __bf16* At, int ldt, f32x4 (&accH)[6], f32x4 (&accT)[6]) {
  static_assert(KBH % 4 == 0 && KBT == 4, "groups of 4 k-blocks");
  const int lane = threadIdx.x & 63, rl = lane & 15, kq = (lane >> 4) * 8;
  constexpr int NG = KBH / 4;
  bf16x8 b[2][4][6];
  auto ld = [&](bf16x8 (&d)[4][6], const bf16x8* w, int KB, int k0) {
#pragma unroll
    for (int q = 0; q < 4; q++)
#pragma unroll
      for (int j = 0; j < 6; j++) d[q][j] = w[((long)tile[j] * KB + k0 + q) * 64 + lane];
  };
  auto mm = [&](const bf16x8 (&d)[4][6], const __bf16* A, int lda, int k0, f32x4 (&acc)[6]) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(A + rl * lda + (k0 + q) * 32 + kq);
#pragma unroll
      for (int j = 0; j < 6; j++) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(d[q][j], a, acc[j], 0, 0, 0);
    }
  };
  ld(b[0], wh, KBH, 0);
#pragma unroll
  for (int g = 0; g < NG; g++) {
    if (g + 1 < NG) ld(b[(g + 1) & 1], wh, KBH, (g + 1) * 4);
    else ld(b[(g + 1) & 1], wt, KBT, 0);
    mm(b[g & 1], Ah, ldh, g * 4, accH);
  }
  mm(b[NG & 1], At, ldt, 0, accT);
}

// the same stream for NT tiles in groups of G k-blocks (the 16 k-blocks of h . W_hh^T and the 4 of token . W_ih^T as one sequence of 20):
// <3, 4, 3> is the half pass of an 8-member cluster -- one unit tile x 3 gates, 24 fragment loads in flight per wave like <6, 4>'s, in
// a ring of three groups (two groups of 8 k-blocks spilled).  Every
// accumulator still adds its k-blocks in ascending order: bit-identical to gate_products.
template <int NT, int G, int D>
__device__ __forceinline__ void gate_products_g(const bf16x8* __restrict__ wh, const bf16x8* __restrict__ wt, const int (&tile)[NT],
                                                const __bf16* Ah, int ldh, const __bf16* At, int ldt, f32x4 (&accH)[NT], f32x4 (&accT)[NT]) {
  constexpr int KBH = 16, KBT = 4, NV = KBH + KBT, NGR = (NV + G - 1) / G;
  const int lane = threadIdx.x & 63, rl = lane & 15, kq = (lane >> 4) * 8;
  bf16x8 b[D][G][NT];                                             // ring of D groups: D - 1 groups of loads in flight under the MFMAs of one
  auto ld = [&](bf16x8 (&d)[G][NT], int g) {
#pragma unroll
    for (int q = 0; q < G; q++) {
      const int vk = g * G + q;
      if (vk < NV) {
#pragma unroll
        for (int j = 0; j < NT; j++) d[q][j] = vk < KBH ? wh[((long)tile[j] * KBH + vk) * 64 + lane] : wt[((long)tile[j] * KBT + vk - KBH) * 64 + lane];
      }
    }
  };
  auto mm = [&](const bf16x8 (&d)[G][NT], int g) {
#pragma unroll
    for (int q = 0; q < G; q++) {
      const int vk = g * G + q;
      if (vk < NV) {
        const bf16x8 av = vk < KBH ? *reinterpret_cast<const bf16x8*>(Ah + rl * ldh + vk * 32 + kq) : *reinterpret_cast<const bf16x8*>(At + rl * ldt + (vk - KBH) * 32 + kq);
#pragma unroll
        for (int j = 0; j < NT; j++) {
          if (vk < KBH) accH[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(d[q][j], av, accH[j], 0, 0, 0);
          else accT[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(d[q][j], av, accT[j], 0, 0, 0);
        }
      }
    }
  };
#pragma unroll
  for (int g = 0; g < D - 1 && g < NGR; g++) ld(b[g % D], g);
#pragma unroll
  for (int g = 0; g < NGR; g++) {
    if (g + D - 1 < NGR) ld(b[(g + D - 1) % D], g + D - 1);
    mm(b[g % D], g);
  }
}

__device__ __forceinline__ void st_bf16x4_lds(__bf16* p, float a, float b, float c, float d) {
  bf16x4 v; v[0] = (__bf16)a; v[1] = (__bf16)b; v[2] = (__bf16)c; v[3] = (__bf16)d;
  *reinterpret_cast<bf16x4*>(p) = v;
}

// =============================================================================================
// weight packing: W fp32 [N][ld] (columns 0..K-1 of the given base) -> MFMA B-fragment-major bf16
//   out[((nt*KB + kb)*64 + lane)*8 + e] = W[nt*16 + (lane & 15)][kb*32 + (lane >> 4)*8 + e]   (0 beyond N / K)
// =============================================================================================
// pairs = 1: rows are interleaved inside every PAIR of tiles so that a lane's accumulator registers of the two tiles are 8
// CONSECUTIVE output columns: fragment row c of tile 2t   <- source row 32t + (c / 4) * 8 + c % 4,
//                                              tile 2t+1 <- source row 32t + (c / 4) * 8 + 4 + c % 4
// (MFMA C layout: lane = row + 16 * quad holds columns 4 * quad .. 4 * quad + 3 of a tile) -> lane (row, quad) owns columns
// 32t + 8 * quad .. + 7 with no cross-lane movement: 16- / 32-byte epilogue accesses straight from the accumulators.
__global__ void pack_b_kernel(const float* __restrict__ W, long ld, int N, int K, __bf16* __restrict__ out, int NT, int KB, int pairs) {
  const long total = (long)NT * KB * 64;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63);
    const long f = i >> 6;
    const int kb = (int)(f % KB), nt = (int)(f / KB);
    const int c = lane & 15;
    const int n = pairs ? (nt >> 1) * 32 + (c >> 2) * 8 + (nt & 1) * 4 + (c & 3) : nt * 16 + c;
    const int k0 = kb * 32 + (lane >> 4) * 8;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = (__bf16)((n < N && k0 + e < K) ? W[(long)n * ld + k0 + e] : 0.f);
    *reinterpret_cast<bf16x8*>(out + i * 8) = v;
  }
}

// =============================================================================================
// note loop of one time step for one 16-row panel
// =============================================================================================
typedef __attribute__((address_space(1))) unsigned gu32;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

struct NoteLoopArgs {
  const bf16x8 *wg_h, *wg_t, *wp, *wd_h, *wd_p, *wdur;
  const float *b_hh_n, *b_p, *b_dh, *b_hh_d, *tab0, *tab, *w_out, *b_out, *w_embT, *b_emb;
  const float* gc; long ld_gc;     // hoisted input part of the notes GRU for this t (b_ih included): row b at gc + b*ld_gc, 1536 wide
  const float* h0; long ld_h0;     // initial notes-GRU state of this t: row b at h0 + b*ld_h0 (null: slot 0 of HN holds it already)
  const float* emb;                // ground-truth embedding, step-major [16][R][128]; null in inference
  float* HN; __bf16* gates_n; float* pitch; long ld_pitch; float* HD; __bf16* gates_d; float* dur; int* idx;
  float* TOK; float* PRED; long* xhat; int* plen;
  const int* force_pitch; const int* force_dur;
  long* dbg_out;                  // timing experiments: per workgroup {XCC id, CU/SE id word, cycles}
  __bf16* HN16; __bf16* HD16;     // optional bf16 copies of the states (operands of the backward's products); HD16 replaces HD[1..5]
  int B, t, R, M;
  unsigned coin_mask;              // bit n: the token fed to note step n+1 is the ground truth (teacher forcing coin, ptvae.py:420)
  int train;                       // save what the backward needs (states, gates)
  int tok_store;                   // save the fed tokens (train, or the light mode whose caller recomputes states and gates batched)
  int dbg;                         // timing experiments: skip phases (results invalid)
  // cluster mode (note_loop_kernel only): S workgroups share a panel -- each streams 1/S of the gate weights, the new bf16 state is
  // all-gathered through `xch` once per note step, everything after the cell is computed by all S redundantly
  int S;
  __bf16* xch;                     // [panels][2][16][512] bf16 exchange buffer
  unsigned* cnt;                   // [panels] arrival counters (zeroed by the caller before t = 0) + [1] error word
};

// lane-exchange helpers of the RES = 1 kernel (VALU, no LDS round trip; same operands and the same order of additions as the __shfl_xor forms)
// (inline asm: hipcc 7.2 miscompiles __builtin_amdgcn_permlane16_swap / 32_swap -- it reads the FIRST result for both elements of the
// returned pair, scripts/micro/permlane_swap.hip; the s_nop covers the VALU-write -> permlane-read hazard the compiler cannot see)
__device__ __forceinline__ float xor16_sum(float x) {           // x[l] + x[l ^ 16]: swap the odd 16-lane rows of a with the even rows of b
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}
__device__ __forceinline__ float xor32_sum(float x) {           // x[l] + x[l ^ 32]: swap the upper half of a with the lower half of b
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}
template <int N> __device__ __forceinline__ int row_ror(int x) { return __builtin_amdgcn_update_dpp(0, x, 0x120 + N, 0xf, 0xf, false); }
// (value, index) maximum over the 16 lanes of a DPP row, first maximal index on ties; every lane ends with the result
template <int N> __device__ __forceinline__ void argmax_ror(float& best, int& bi) {
  const float ov = __builtin_bit_cast(float, row_ror<N>(__builtin_bit_cast(int, best)));
  const int oi = row_ror<N>(bi);
  const bool take = ov > best || (ov == best && oi < bi);
  best = take ? ov : best; bi = take ? oi : bi;
}

// RES = 1 (round 6): the weights of the HEAD phases never change over the 15 x 32 note steps, and a wave is alone on its SIMD (512
// registers): the two pitch-head tiles of a wave stay in registers for the whole launch, the third tile of wave 0 (columns 128 / 129
// only: 2 KB) and the logits part of dur_hid_linear (20 KB) in LDS, the state part of dur_hid_linear is requested when the head
// phase begins and consumed two phases later, the predicted token's embedding row right after the argmax.  The head phases then
// hold no exposed L2 round trip (RES = 0 had 8 + 1 + 1 of them per note step on wave 0).  Same products, same k order: bit-identical.
template <int RES, int NUK = 2>                                  // NUK = 1: the eight-member cluster's kernel (one unit tile per wave and note step)
__global__ __launch_bounds__(256, 1) void note_loop_kernel(NoteLoopArgs a) {
  __shared__ __attribute__((aligned(16))) bf16x8 wp8[RES ? 16 * 4 * 2 : 1];   // pitch-head tile 8, rows 128 / 129 only: [kb][quad][row]
  __shared__ __attribute__((aligned(16))) bf16x8 wdp[RES ? 4 * 5 * 64 : 1];   // dur_hid_linear, logits part (4 tiles x 5 k-blocks)
  __shared__ __attribute__((aligned(16))) float embc[RES ? 6 : 1][FE];        // note_embedding: bias + the 5 duration columns
  __shared__ __attribute__((aligned(16))) float hf[FP][FHN];                   // notes-GRU state, fp32
  __shared__ __attribute__((aligned(16))) __bf16 h16[2][FP][H16LD];            // its bf16 MFMA-operand copy (double buffered)
  __shared__ __attribute__((aligned(16))) __bf16 tok16[FP][T16LD];             // current input token
  __shared__ __attribute__((aligned(16))) float pit[FP][PITLD];                // pitch logits
  __shared__ __attribute__((aligned(16))) __bf16 pit16[FP][P16LD];
  __shared__ __attribute__((aligned(16))) float hdf[FP][FHD];                  // duration-GRU state
  __shared__ __attribute__((aligned(16))) __bf16 hd16[2][FP][D16LD];
  __shared__ __attribute__((aligned(16))) bf16x8 wdl[12 * 2 * 64];             // duration W_hh, fragment-major (24 KB)
  __shared__ float tabs[3][3 * FHD];
  __shared__ float bhd[3 * FHD];
  __shared__ float wo[2 * FHD + 2];
  __shared__ __attribute__((aligned(8))) float part[2][4][FP][2];              // double buffered: one barrier per duration step
  __shared__ int pidx[FP];
  __shared__ int bits[FP][5];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (scalar: branches on it are not divergent)
  const long t_begin = a.dbg_out ? (long)__builtin_amdgcn_s_memtime() : 0;
  const int crow = lane & 15, ckq = lane >> 4;                // accumulator (C) layout
  const int erow = lane >> 2, eq = lane & 3;                  // epilogue layout
  const int B = a.B, R = a.R, t = a.t;
  const long M = a.M;
  // cluster mode: the S members of a panel are S consecutive workgroups OF ONE XCD (workgroups go round-robin over the 8 XCDs:
  // speed only -- the exchange is agent-scope either way), member 0 is the one that writes the panel's outputs
  const int S = a.S;
  int panel = blockIdx.x, mem = 0;
  if (S > 1) {
    const int x = blockIdx.x & 7, q = blockIdx.x >> 3;
    panel = (q / S) * 8 + x; mem = q % S;
    if (panel * FP >= B) return;                              // (whole panels only: nobody waits for these)
  }
  const bool lead = mem == 0;
  const int p_lo = mem * (4 / S), p_hi = p_lo + 4 / S;        // the P1 passes (32 units per wave each) this member computes
  gu32* const xcnt = S > 1 ? (gu32*)(a.cnt + panel) : nullptr;
  gu32* const xerr = S > 1 ? (gu32*)(a.cnt + (B + FP - 1) / FP) : nullptr;
  bool dead = false;
  const int r0 = panel * FP;                                  // first sample of the panel
  const int rE = min(r0 + erow, B - 1), rC = min(r0 + crow, B - 1);
  const bool okE = r0 + erow < B, okC = r0 + crow < B && lead;   // (okC guards output stores only)
  const long wrowE = (long)t * B + rE, wrowC = (long)t * B + rC;   // row in the [R]-row step-major matrices

  // ---- one-time loads: duration GRU weights / tables -> LDS, initial state and first token -> LDS
  for (int i = tid; i < 12 * 2 * 64; i += 256) wdl[i] = a.wdur[i];
  for (int i = tid; i < 3 * FHD; i += 256) { tabs[0][i] = a.tab0[i]; tabs[1][i] = a.tab[i]; tabs[2][i] = a.tab[3 * FHD + i]; bhd[i] = a.b_hh_d[i]; }
  for (int i = tid; i < 2 * FHD; i += 256) wo[i] = a.w_out[i];
  if (tid < 2) wo[2 * FHD + tid] = a.b_out[tid];
  for (int i = tid; i < FP * (FHN / 4); i += 256) {
    const int row = i / (FHN / 4), c4 = (i % (FHN / 4)) * 4;
    const int rb = min(r0 + row, B - 1);
    const float4 v = a.h0 ? *reinterpret_cast<const float4*>(a.h0 + (long)rb * a.ld_h0 + c4)
                          : *reinterpret_cast<const float4*>(a.HN + ((long)t * B + rb) * FHN + c4);
    if (a.h0 && r0 + row < B && lead) *reinterpret_cast<float4*>(a.HN + ((long)t * B + rb) * FHN + c4) = v;   // slot 0 of HN for the backward
    *reinterpret_cast<float4*>(&hf[row][c4]) = v;
    st_bf16x4_lds(&h16[0][row][c4], v.x, v.y, v.z, v.w);
    if (a.train && a.HN16 && r0 + row < B && lead) st_bf16x4_lds(a.HN16 + ((long)t * B + r0 + row) * FHN + c4, v.x, v.y, v.z, v.w);
  }
  for (int i = tid; i < FP * (FE / 4); i += 256) {
    const int row = i / (FE / 4), c4 = (i % (FE / 4)) * 4;
    const float4 v = *reinterpret_cast<const float4*>(a.TOK + ((long)t * B + min(r0 + row, B - 1)) * FE + c4);
    st_bf16x4_lds(&tok16[row][c4], v.x, v.y, v.z, v.w);
  }
  for (int i = tid; i < FP * (P16LD - FNP); i += 256) pit16[i / (P16LD - FNP)][FNP + i % (P16LD - FNP)] = (__bf16)0.f;   // K padding of the logits operand
  if (tid < FP) pidx[tid] = 0;
  bf16x8 rp[2][16];                                             // RES: pitch-head tiles `wave` and `wave + 4`, resident
  if constexpr (RES) {
    for (int i = tid; i < 16 * 4 * 2; i += 256) {
      const int r = i & 1, quad = (i >> 1) & 3, kb = i >> 3;
      wp8[i] = a.wp[((long)8 * 16 + kb) * 64 + quad * 16 + r];
    }
    for (int i = tid; i < 4 * 5 * 64; i += 256) wdp[i] = a.wd_p[i];
    for (int i = tid; i < 6 * FE; i += 256) embc[i / FE][i % FE] = i < FE ? a.b_emb[i] : a.w_embT[(long)(FNP + i / FE - 1) * FE + i % FE];
#pragma unroll
    for (int kb = 0; kb < 16; kb++) {
      rp[0][kb] = a.wp[((long)wave * 16 + kb) * 64 + lane];
      rp[1][kb] = a.wp[((long)(wave + 4) * 16 + kb) * 64 + lane];
    }
  }
  float bp_r[3][4] = {};                                        // RES: the biases of this lane's head columns (a global load between the head's MFMAs
  float4 bdh_r = make_float4(0.f, 0.f, 0.f, 0.f);               //      and its LDS stores is a whole L2 round trip on the chain, every note step)
  if constexpr (RES) {
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const int c = (j == 2 ? 8 : wave + 4 * j) * 16 + ckq * 4 + e;
        bp_r[j][e] = c < FNP ? a.b_p[c] : 0.f;
      }
    bdh_r = *reinterpret_cast<const float4*>(a.b_dh + wave * 16 + ckq * 4);
  }
  float4 ew0 = make_float4(0.f, 0.f, 0.f, 0.f), ew1 = ew0;      // RES: embedding row of the decision, requested in P3
  // timing experiments (dbg_out): 100-MHz ticks wave 0 spends per phase, summed over the 15 note steps -> dbg_out[3 * grid + 8 * block + i],
  // i = 0 cell (products + epilogue), 1 barrier + state exchange, 2 pitch head, 3 argmax + dur_hid, 4 duration GRU, 5 token embedding
  long tph[6] = {0, 0, 0, 0, 0, 0}, tlast = 0;
  auto PH = [&](int i) {
    if (a.dbg_out) { const long now = (long)__builtin_amdgcn_s_memrealtime(); if (i >= 0) tph[i] += now - tlast; tlast = now; }
  };

  __syncthreads();

  // waves exchange through LDS only inside the loop (every global store is an output nobody here reads back from another
  // wave): lds_barrier() keeps the training-mode stores of states and gates in flight across the 14 barriers of a note step
  PH(-1);
  for (int n = 0; n < 15; n++) {
    const int cur = n & 1, nxt = cur ^ 1;
    // cluster mode: exchange buffer of this note step (double buffered) as 16 x 256 8-byte words {2 bf16 units, step tag}
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(S > 1 ? (void*)(a.xch + ((long)panel * 2 + (n & 1)) * (FP * FHN * 2)) : nullptr, 0,
                                                                        S > 1 ? FP * FHN * 4 : 0, 0x00020000);
    const unsigned xseq = (unsigned)(t * 15 + n + 1);
    // ================= P1: notes-GRU cell.  wave w owns units [w*128, w*128+128) = 8 tiles of 16, two per pass =================
    // one pass = NU unit tiles (of 16 units) x 3 gates of this wave: two for 1 / 2 / 4 members per panel, ONE for eight (member m: tile m of the wave's 8)
    auto cell_pass = [&](auto nu_c, const int ut0) {
      constexpr int NU = decltype(nu_c)::value;
      // per-lane constants in the epilogue layout (4 adjacent lanes = 16 units of one row): the hoisted input part GC (b_ih
      // included) and b_hh, requested before the products
      float4 gR[NU], gZ[NU], gN[NU], bR[NU], bZ[NU], bN[NU];
#pragma unroll
      for (int j = 0; j < NU; j++) {
        const int u = (ut0 + j) * 16 + eq * 4;
        const float* g = a.gc + (long)rE * a.ld_gc;
        gR[j] = *reinterpret_cast<const float4*>(g + u); gZ[j] = *reinterpret_cast<const float4*>(g + FHN + u); gN[j] = *reinterpret_cast<const float4*>(g + 2 * FHN + u);
        bR[j] = *reinterpret_cast<const float4*>(a.b_hh_n + u); bZ[j] = *reinterpret_cast<const float4*>(a.b_hh_n + FHN + u);
        bN[j] = *reinterpret_cast<const float4*>(a.b_hh_n + 2 * FHN + u);
      }
      f32x4 accH[3 * NU], accT[3 * NU];                         // (r.., z.., n..): h . W_hh^T and token . W_ih[:, Ht:]^T
#pragma unroll
      for (int j = 0; j < 3 * NU; j++) accH[j] = accT[j] = f32x4{0.f, 0.f, 0.f, 0.f};
      int tl[3 * NU];
#pragma unroll
      for (int j = 0; j < NU; j++) { tl[j] = ut0 + j; tl[NU + j] = 32 + ut0 + j; tl[2 * NU + j] = 64 + ut0 + j; }
      if (!(a.dbg & 1)) {
        if constexpr (NU == 2) gate_products<16, 4>(a.wg_h, a.wg_t, tl, &h16[cur][0][0], H16LD, &tok16[0][0], T16LD, accH, accT);
        else gate_products_g<3, 4, 3>(a.wg_h, a.wg_t, tl, &h16[cur][0][0], H16LD, &tok16[0][0], T16LD, accH, accT);
      }
#pragma unroll
      for (int j = 0; j < NU; j++) {
        const int u = (ut0 + j) * 16 + eq * 4;
        f32x4 sR, sZ;
#pragma unroll
        for (int e = 0; e < 4; e++) { sR[e] = accH[j][e] + accT[j][e]; sZ[e] = accH[NU + j][e] + accT[NU + j][e]; }
        const f32x4 aR = to_rowmajor_lanes(sR), aZ = to_rowmajor_lanes(sZ), aI = to_rowmajor_lanes(accT[2 * NU + j]), aH = to_rowmajor_lanes(accH[2 * NU + j]);
        const float4 hp4 = *reinterpret_cast<const float4*>(&hf[erow][u]);
        const float hp[4] = {hp4.x, hp4.y, hp4.z, hp4.w};
        const float kR[4] = {gR[j].x + bR[j].x, gR[j].y + bR[j].y, gR[j].z + bR[j].z, gR[j].w + bR[j].w};
        const float kZ[4] = {gZ[j].x + bZ[j].x, gZ[j].y + bZ[j].y, gZ[j].z + bZ[j].z, gZ[j].w + bZ[j].w};
        const float kN[4] = {gN[j].x, gN[j].y, gN[j].z, gN[j].w}, kB[4] = {bN[j].x, bN[j].y, bN[j].z, bN[j].w};
        float r[4], z[4], nn[4], hn[4], h[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
          r[e] = fsig(aR[e] + kR[e]);
          z[e] = fsig(aZ[e] + kZ[e]);
          hn[e] = aH[e] + kB[e];
          nn[e] = ftanh(aI[e] + kN[e] + r[e] * hn[e]);
          h[e] = (1.0f - z[e]) * nn[e] + z[e] * hp[e];
        }
        *reinterpret_cast<float4*>(&hf[erow][u]) = make_float4(h[0], h[1], h[2], h[3]);
        st_bf16x4_lds(&h16[nxt][erow][u], h[0], h[1], h[2], h[3]);
        if (S > 1) {                                              // cluster mode: the new state leaves for the other members right here
          bf16x4 hv; hv[0] = (__bf16)h[0]; hv[1] = (__bf16)h[1]; hv[2] = (__bf16)h[2]; hv[3] = (__bf16)h[3];
          const u32x2 d2 = __builtin_bit_cast(u32x2, hv);
          const int wofs = (erow * FHN + u) * 4;                  // byte offset of the 8-byte word {2 units, step tag} of units u, u + 1
          __builtin_amdgcn_raw_buffer_store_b64(u32x2{d2[0], xseq}, xr, wofs, 0, 16);        // sc1
          __builtin_amdgcn_raw_buffer_store_b64(u32x2{d2[1], xseq}, xr, wofs + 8, 0, 16);
        }
        if (a.train && okE) {
          *reinterpret_cast<float4*>(a.HN + ((long)(n + 1) * R + wrowE) * FHN + u) = make_float4(h[0], h[1], h[2], h[3]);
          if (a.HN16) st_bf16x4_lds(a.HN16 + ((long)(n + 1) * R + wrowE) * FHN + u, h[0], h[1], h[2], h[3]);
          __bf16* gp = a.gates_n + (((long)n * 4) * R + wrowE) * FHN + u;
          const long pl = (long)R * FHN;
          st_bf16x4_lds(gp, r[0], r[1], r[2], r[3]);
          st_bf16x4_lds(gp + pl, z[0], z[1], z[2], z[3]);
          st_bf16x4_lds(gp + 2 * pl, nn[0], nn[1], nn[2], nn[3]);
          st_bf16x4_lds(gp + 3 * pl, hn[0], hn[1], hn[2], hn[3]);
        }
      }
    };
    if (!(a.dbg & 2)) {
      if constexpr (NUK == 1) {
        // (opaque per note step: the 60 fragment addresses of the pass are loop-invariant otherwise, and hipcc hoists them out of the note loop --
        // 120 registers, 220 bytes of spills; the two-tile kernels are safe behind their runtime pass loop)
        int ut = wave * 8 + mem;
        asm volatile("" : "+s"(ut));
        cell_pass(std::integral_constant<int, 1>{}, ut);
      }
      else {
#pragma unroll 1
        for (int p = p_lo; p < p_hi; p++) cell_pass(std::integral_constant<int, 2>{}, wave * 8 + p * 2);
      }
    }
    bf16x8 wdh[16];                                              // RES: this wave's tile of dur_hid_linear's state part -- requested here, in
    if constexpr (RES) {                                         // flight across the state exchange, consumed in the head phase
#pragma unroll
      for (int kb = 0; kb < 16; kb++) wdh[kb] = a.wd_h[((long)wave * 16 + kb) * 64 + lane];
    }
    PH(0);
    if (S == 1) lds_barrier();
    if (S > 1) {
      // ---- all-gather of the new bf16 state (round 6: flag-in-data).  Every member has written its units as 8-byte words {2 units, step
      // tag} straight from the cell epilogue (single-copy atomic, sc1 = agent scope); a reader polls the WORDS it needs until they carry
      // this step's tag -- no arrival counter to bump and poll, no store acknowledgement to wait for: one L2 round trip after the last
      // member's stores land instead of four.  The caller zeroes xch before t = 0 (tag 0 never matches); a buffer is rewritten two
      // note steps later, which its writer can only reach after every reader has sent the step in between.
      // a wave's 128 units are 4 groups of 32 (a pass; 1, 2 or 4 members: NUK = 2) or 8 groups of 16 (eight members: NUK = 1); a thread fetches
      // NQ pieces of PW words: piece = a quarter of a group's units of one row
      constexpr int PW = NUK == 2 ? 4 : 2, NQ = NUK == 2 ? 3 : 7, GU = NUK == 2 ? 32 : 16;
      const int nfg = NUK == 2 ? 4 - 4 / S : 7;                  // foreign groups per wave
      const int own0 = NUK == 2 ? p_lo : mem, nown = NUK == 2 ? 4 / S : 1;
      unsigned pend = 0;
      int wofs[NQ];
#pragma unroll
      for (int q = 0; q < NQ; q++) {
        const int i2 = tid + 256 * q, piece = i2 & 3, row = (i2 >> 2) & 15, wg = i2 >> 6;   // wg: (wave, foreign group)
        const int fg = wg % nfg, g = fg < own0 ? fg : fg + nown;
        const int u = (wg / nfg) * 128 + g * GU + piece * (2 * PW);
        wofs[q] = (row * FHN + u) * 4;
        if (q < nfg) pend |= ((1u << PW) - 1u) << (PW * q);
      }
      if (tid == 0) __hip_atomic_fetch_add(xcnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (diagnostics only: nobody waits on it)
      unsigned spins = 0;
      // throttle: wave 0 first watches ONE word per foreign group (a late one: row 15, last units of wave 3's tile) with a sleep between
      // looks -- 256 lanes x 12 words of polling per workgroup slowed the weight streams of the members still in their cell (B = 1024:
      // +1 us per note step); the words themselves are still checked one by one below
      // (eight members: their cells are short and in step -- every lane polls its own words from the start, 13.8 -> 13.2 us; with four
      // members at B = 1024 the same costs 20.2 -> 22.0: dbg 16 switches the watch phase off there for timing)
      while (!dead && wave == 0 && NUK == 2 && !(a.dbg & 16)) {
        bool all = true;
#pragma unroll
        for (int fg = 0; fg < 7; fg++)
          if (fg < nfg) {
            const int g = fg < own0 ? fg : fg + nown;
            const u32x2 sv = __builtin_amdgcn_raw_buffer_load_b64(xr, (15 * FHN + 3 * 128 + g * GU + GU - 2) * 4, 0, 16);
            all = all && sv[1] == xseq;
          }
        if (all) break;
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1u << 22)) { __hip_atomic_store(xerr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); dead = true; }
      }
      lds_barrier();                                              // (wave 0 watched for the other three)
      while (pend && !dead) {
        u32x2 v[NQ * PW];
#pragma unroll
        for (int w = 0; w < NQ * PW; w++)
          if ((pend >> w) & 1u) v[w] = __builtin_amdgcn_raw_buffer_load_b64(xr, wofs[w / PW] + 8 * (w % PW), 0, 16);                    // sc1
#pragma unroll
        for (int w = 0; w < NQ * PW; w++)
          if (((pend >> w) & 1u) && v[w][1] == xseq) {
            const int o = wofs[w / PW] >> 2;                     // = row * 512 + u
            *reinterpret_cast<unsigned*>(&h16[nxt][o >> 9][(o & 511) + 2 * (w % PW)]) = v[w][0];
            pend &= ~(1u << w);
          }
        if (pend && ++spins > (1u << 22)) { __hip_atomic_store(xerr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); dead = true; }
      }
      lds_barrier();
    }
    PH(1);
    // ================= P2: pitch head (9 tiles over 4 waves) + the state part of dur_hid_linear (one tile per wave) =================
    f32x4 accD[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
    if constexpr (RES) {
      if (!(a.dbg & 4)) {
        // operands first, all at once (left to the scheduler -- at this register pressure -- every LDS read is issued right before the MFMA
        // that consumes it and its latency is paid 16 times), then the MFMAs back to back
        const int rl = lane & 15, kq = (lane >> 4) * 8;
        f32x4 acc[3] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int k0 = 0; k0 < 16; k0 += 8) {                       // (two halves: 64 registers of operands instead of 128)
          bf16x8 av[8], b8[8];
#pragma unroll
          for (int q = 0; q < 8; q++) av[q] = *reinterpret_cast<const bf16x8*>(&h16[nxt][rl][(k0 + q) * 32 + kq]);
#pragma unroll
          for (int q = 0; q < 8; q++) b8[q] = wp8[((k0 + q) * 4 + (lane >> 4)) * 2 + (rl & 1)];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int q = 0; q < 8; q++) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rp[0][k0 + q], av[q], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rp[1][k0 + q], av[q], acc[1], 0, 0, 0);
            // tile 8 is wave 0's, but every wave runs it: wave 0 is the longest chain either way, and a branch between the MFMAs makes the
            // compiler copy the accumulators around it
            if (rl >= 2) b8[q] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b8[q], av[q], acc[2], 0, 0, 0);
            accD[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wdh[k0 + q], av[q], accD[0], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < 3; j++) {
          if (j == 2 && wave != 0) break;
          const int c0 = (j == 2 ? 8 : wave + 4 * j) * 16 + ckq * 4;
#pragma unroll
          for (int e = 0; e < 4; e++) {
            const int c = c0 + e;
            const float v = c < FNP ? acc[j][e] + bp_r[j][e] : 0.f;
            pit[crow][c] = v;
            pit16[crow][c] = (__bf16)v;
          }
        }
      }
    } else {
      const int tl[1] = {wave};
      panel_mma<1, 16, 8>(a.wd_h, tl, &h16[nxt][0][0], H16LD, accD);
    }
    for (int nt = wave; nt < ((a.dbg & 4) || RES ? 0 : 9); nt += 4) {
      f32x4 acc[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
      const int tl[1] = {nt};
      panel_mma<1, 16, 8>(a.wp, tl, &h16[nxt][0][0], H16LD, acc);
      const int c0 = nt * 16 + ckq * 4;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const int c = c0 + e;
        const float v = c < FNP ? acc[0][e] + a.b_p[c] : 0.f;
        pit[crow][c] = v;
        pit16[crow][c] = (__bf16)v;
      }
    }
    lds_barrier();
    PH(2);
    // ================= P3: argmax over the 130 logits (16 lanes per row, first maximal index) + logits out =================
    {
      const int row = tid >> 4, j = tid & 15;
      float best = -INFINITY; int bi = 0x7fffffff;
      const long pr = (long)n * R + (long)t * B + min(r0 + row, B - 1);
      const bool ok = r0 + row < B && lead;
      if constexpr (RES) {
        float pv[9];
#pragma unroll
        for (int k = 0; k < 9; k++) pv[k] = pit[row][j + 16 * k];              // (columns 130..143 of the row exist and hold zeros)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 9; k++) {
          const int c = j + 16 * k;
          const bool take = c < FNP && pv[k] > best;
          best = take ? pv[k] : best; bi = take ? c : bi;
        }
        argmax_ror<8>(best, bi); argmax_ror<4>(best, bi); argmax_ror<2>(best, bi); argmax_ror<1>(best, bi);
        if (ok) {
#pragma unroll
          for (int k = 0; k < 9; k++) if (j + 16 * k < FNP) a.pitch[pr * a.ld_pitch + j + 16 * k] = pv[k];
        }
      } else {
#pragma unroll
      for (int k = 0; k < 9; k++) {
        const int c = j + 16 * k;
        if (c < FNP) {
          const float v = pit[row][c];
          if (ok) a.pitch[pr * a.ld_pitch + c] = v;
          if (v > best) { best = v; bi = c; }
        }
      }
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
      }
      }
      if (a.force_pitch) bi = a.force_pitch[(long)n * R + (long)t * B + min(r0 + row, B - 1)];
      if (j == 0) pidx[row] = bi;
      if constexpr (RES) {                                        // (every lane of the row holds the decision; P6 maps threads the same way)
        const float* wr_ = a.w_embT + (long)bi * FE + j * 8;
        ew0 = *reinterpret_cast<const float4*>(wr_); ew1 = *reinterpret_cast<const float4*>(wr_ + 4);
      }
    }
    // ================= P4: dur_hid_linear([h | logits]) -> initial duration state (wave w = units w*16..) =================
    {
      if constexpr (RES) {
        const int rl = lane & 15, kq = (lane >> 4) * 8;
        bf16x8 pv16[5], wv[5];
#pragma unroll
        for (int kb = 0; kb < 5; kb++) {
          pv16[kb] = *reinterpret_cast<const bf16x8*>(&pit16[rl][kb * 32 + kq]);
          wv[kb] = wdp[(wave * 5 + kb) * 64 + lane];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kb = 0; kb < 5; kb++) accD[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv[kb], pv16[kb], accD[0], 0, 0, 0);
      } else {
        const int tl[1] = {wave};
        panel_mma<1, 5, 8>(a.wd_p, tl, &pit16[0][0], P16LD, accD);             // + the logits part (K = 130 padded to 160)
      }
      const int u = wave * 16 + ckq * 4;
      const float4 b4 = RES ? bdh_r : *reinterpret_cast<const float4*>(a.b_dh + u);
      const float h[4] = {accD[0][0] + b4.x, accD[0][1] + b4.y, accD[0][2] + b4.z, accD[0][3] + b4.w};
      *reinterpret_cast<float4*>(&hdf[crow][u]) = make_float4(h[0], h[1], h[2], h[3]);
      st_bf16x4_lds(&hd16[0][crow][u], h[0], h[1], h[2], h[3]);
      if (a.train && okC) {
        *reinterpret_cast<float4*>(a.HD + ((long)n * R + wrowC) * FHD + u) = make_float4(h[0], h[1], h[2], h[3]);
        if (a.HD16) st_bf16x4_lds(a.HD16 + ((long)n * R + wrowC) * FHD + u, h[0], h[1], h[2], h[3]);
      }
    }
    lds_barrier();
    PH(3);
    // ================= P5: 5-step duration GRU, argmax feedback (wave w = units w*16..w*16+15) =================
    if constexpr (RES) {
      // The five steps unrolled with every invariant of the phase in registers (weight fragments, the three possible input rows, biases,
      // this lane's fp32 state) and the LDS reads that follow a barrier issued together: the chain of a step is barrier -> {partial
      // logits, state operand} in ONE LDS round trip -> decision (VALU) beside the MFMAs -> cell.  Before: the input row was looked up in
      // LDS AFTER the decision and the four partial sums were read one round trip each (0.85 us per step).  Same arithmetic, same order.
      const long prC = (long)n * R + wrowC;
      const int u = wave * 16 + ckq * 4;
      if (!(a.dbg & 8)) {
        float4 tb[3][3], bh[3];
#pragma unroll
        for (int g = 0; g < 3; g++) {
#pragma unroll
          for (int tk = 0; tk < 3; tk++) tb[tk][g] = *reinterpret_cast<const float4*>(&tabs[tk][g * FHD + u]);
          bh[g] = *reinterpret_cast<const float4*>(&bhd[g * FHD + u]);
        }
        const float4 wo0 = *reinterpret_cast<const float4*>(&wo[u]), wo1 = *reinterpret_cast<const float4*>(&wo[FHD + u]);
        const float wb0 = wo[2 * FHD], wb1 = wo[2 * FHD + 1];
        bf16x8 wd[3][2];
#pragma unroll
        for (int g = 0; g < 3; g++)
#pragma unroll
          for (int kb = 0; kb < 2; kb++) wd[g][kb] = wdl[((g * 4 + wave) * 2 + kb) * 64 + lane];
        const float4 hp0 = *reinterpret_cast<const float4*>(&hdf[crow][u]);     // (this lane wrote it in P4)
        float hp[4] = {hp0.x, hp0.y, hp0.z, hp0.w};
        bf16x8 av[2];
#pragma unroll
        for (int kb = 0; kb < 2; kb++) av[kb] = *reinterpret_cast<const bf16x8*>(&hd16[0][crow][kb * 32 + ckq * 8]);
        float2 q[4] = {};
        int dtk = 0;                                                           // this lane's row: 0 = <sos>, 1 + previous decision
        auto decide = [&](int d) {                                             // logits and decision of duration step d from the four waves' partial sums
          const float e0 = q[0].x + q[1].x + q[2].x + q[3].x + wb0;
          const float e1 = q[0].y + q[1].y + q[2].y + q[3].y + wb1;
          int id = e1 > e0 ? 1 : 0;                                             // first max wins ties (torch.max)
          if (a.force_dur) id = a.force_dur[(long)d * M + prC];
          dtk = 1 + id;
          if (tid < FP) {                                                      // wave 0, lanes 0..15: crow == tid
            if (okC) {
              a.dur[prC * 10 + 2 * d] = e0; a.dur[prC * 10 + 2 * d + 1] = e1;
              a.idx[(long)d * M + prC] = id;
            }
            bits[tid][d] = id;
          }
        };
#pragma unroll
        for (int d = 0; d < 5; d++) {
          const int dc = d & 1, dn = dc ^ 1;
          __builtin_amdgcn_sched_barrier(0);
          if (d > 0) decide(d - 1);
          f32x4 acc[3];
#pragma unroll
          for (int g = 0; g < 3; g++) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kb = 0; kb < 2; kb++)
#pragma unroll
            for (int g = 0; g < 3; g++) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wd[g][kb], av[kb], acc[g], 0, 0, 0);
          float gi[3][4];
#pragma unroll
          for (int g = 0; g < 3; g++) {
            const float4 t0 = d == 0 ? tb[0][g] : tb[1][g], t1 = d == 0 ? tb[0][g] : tb[2][g];
            const bool two = dtk == 2;
            gi[g][0] = two ? t1.x : t0.x; gi[g][1] = two ? t1.y : t0.y; gi[g][2] = two ? t1.z : t0.z; gi[g][3] = two ? t1.w : t0.w;
          }
          const float bR[4] = {bh[0].x, bh[0].y, bh[0].z, bh[0].w}, bZ[4] = {bh[1].x, bh[1].y, bh[1].z, bh[1].w}, bN[4] = {bh[2].x, bh[2].y, bh[2].z, bh[2].w};
          const float w0v[4] = {wo0.x, wo0.y, wo0.z, wo0.w}, w1v[4] = {wo1.x, wo1.y, wo1.z, wo1.w};
          float r[4], z[4], nn[4], hn[4], h[4], o0 = 0.f, o1 = 0.f;
#pragma unroll
          for (int e = 0; e < 4; e++) {
            r[e] = fsig(gi[0][e] + acc[0][e] + bR[e]);
            z[e] = fsig(gi[1][e] + acc[1][e] + bZ[e]);
            hn[e] = acc[2][e] + bN[e];
            nn[e] = ftanh(gi[2][e] + r[e] * hn[e]);
            h[e] = (1.0f - z[e]) * nn[e] + z[e] * hp[e];
            o0 += w0v[e] * h[e]; o1 += w1v[e] * h[e];
          }
#pragma unroll
          for (int e = 0; e < 4; e++) hp[e] = h[e];
          st_bf16x4_lds(&hd16[dn][crow][u], h[0], h[1], h[2], h[3]);
          if (a.train && okC) {
            if (a.HD16) st_bf16x4_lds(a.HD16 + ((long)(d + 1) * M + prC) * FHD + u, h[0], h[1], h[2], h[3]);
            else *reinterpret_cast<float4*>(a.HD + ((long)(d + 1) * M + prC) * FHD + u) = make_float4(h[0], h[1], h[2], h[3]);
            __bf16* gp = a.gates_d + (((long)d * 4) * M + prC) * FHD + u;
            const long pl = M * FHD;
            st_bf16x4_lds(gp, r[0], r[1], r[2], r[3]);
            st_bf16x4_lds(gp + pl, z[0], z[1], z[2], z[3]);
            st_bf16x4_lds(gp + 2 * pl, nn[0], nn[1], nn[2], nn[3]);
            st_bf16x4_lds(gp + 3 * pl, hn[0], hn[1], hn[2], hn[3]);
          }
          o0 = xor16_sum(o0); o1 = xor16_sum(o1); o0 = xor32_sum(o0); o1 = xor32_sum(o1);
          if (lane < 16) *reinterpret_cast<float2*>(&part[dc][wave][lane][0]) = make_float2(o0, o1);
          lds_barrier();
#pragma unroll
          for (int w = 0; w < 4; w++) q[w] = *reinterpret_cast<const float2*>(&part[dc][w][crow][0]);
          if (d < 4) {
#pragma unroll
            for (int kb = 0; kb < 2; kb++) av[kb] = *reinterpret_cast<const bf16x8*>(&hd16[dn][crow][kb * 32 + ckq * 8]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        decide(4);
      }
    } else {
      const long prC = (long)n * R + wrowC;
      const int u = wave * 16 + ckq * 4;
      int dtk = 0;                                                             // this lane's row: 0 = <sos>, 1 + previous decision
#pragma unroll 1
      for (int d = 0; d < ((a.dbg & 8) ? 0 : 5); d++) {
        const int dc = d & 1, dn = dc ^ 1;
        f32x4 acc[3];
#pragma unroll
        for (int g = 0; g < 3; g++) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 2; kb++) {
          const bf16x8 av = *reinterpret_cast<const bf16x8*>(&hd16[dc][crow][kb * 32 + ckq * 8]);
#pragma unroll
          for (int g = 0; g < 3; g++) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wdl[((g * 4 + wave) * 2 + kb) * 64 + lane], av, acc[g], 0, 0, 0);
        }
        const float* gi = tabs[dtk];
        const float4 hp4 = *reinterpret_cast<const float4*>(&hdf[crow][u]);
        const float hp[4] = {hp4.x, hp4.y, hp4.z, hp4.w};
        float r[4], z[4], nn[4], hn[4], h[4], o0 = 0.f, o1 = 0.f;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int jj = u + e;
          r[e] = fsig(gi[jj] + acc[0][e] + bhd[jj]);
          z[e] = fsig(gi[FHD + jj] + acc[1][e] + bhd[FHD + jj]);
          hn[e] = acc[2][e] + bhd[2 * FHD + jj];
          nn[e] = ftanh(gi[2 * FHD + jj] + r[e] * hn[e]);
          h[e] = (1.0f - z[e]) * nn[e] + z[e] * hp[e];
          o0 += wo[jj] * h[e]; o1 += wo[FHD + jj] * h[e];
        }
        *reinterpret_cast<float4*>(&hdf[crow][u]) = make_float4(h[0], h[1], h[2], h[3]);
        st_bf16x4_lds(&hd16[dn][crow][u], h[0], h[1], h[2], h[3]);
        if (a.train && okC) {
          if (a.HD16) st_bf16x4_lds(a.HD16 + ((long)(d + 1) * M + prC) * FHD + u, h[0], h[1], h[2], h[3]);
          else *reinterpret_cast<float4*>(a.HD + ((long)(d + 1) * M + prC) * FHD + u) = make_float4(h[0], h[1], h[2], h[3]);
          __bf16* gp = a.gates_d + (((long)d * 4) * M + prC) * FHD + u;
          const long pl = M * FHD;
          st_bf16x4_lds(gp, r[0], r[1], r[2], r[3]);
          st_bf16x4_lds(gp + pl, z[0], z[1], z[2], z[3]);
          st_bf16x4_lds(gp + 2 * pl, nn[0], nn[1], nn[2], nn[3]);
          st_bf16x4_lds(gp + 3 * pl, hn[0], hn[1], hn[2], hn[3]);
        }
        if constexpr (RES) { o0 = xor16_sum(o0); o1 = xor16_sum(o1); o0 = xor32_sum(o0); o1 = xor32_sum(o1); }
        else {
          o0 += __shfl_xor(o0, 16, 64); o1 += __shfl_xor(o1, 16, 64);
          o0 += __shfl_xor(o0, 32, 64); o1 += __shfl_xor(o1, 32, 64);
        }
        if (lane < 16) { part[dc][wave][lane][0] = o0; part[dc][wave][lane][1] = o1; }
        lds_barrier();
        // every lane forms the two logits of ITS row from the four waves' partial sums and takes the decision itself (no second
        // barrier to pass the token around); the panel's first 16 threads also publish them
        {
          float e0, e1;
          if constexpr (RES) {                                                  // (the four reads in flight together; same order of additions)
            const float2 q0 = *reinterpret_cast<const float2*>(&part[dc][0][crow][0]), q1 = *reinterpret_cast<const float2*>(&part[dc][1][crow][0]);
            const float2 q2 = *reinterpret_cast<const float2*>(&part[dc][2][crow][0]), q3 = *reinterpret_cast<const float2*>(&part[dc][3][crow][0]);
            const float w0_ = wo[2 * FHD], w1_ = wo[2 * FHD + 1];
            __builtin_amdgcn_sched_barrier(0);
            e0 = q0.x + q1.x + q2.x + q3.x + w0_;
            e1 = q0.y + q1.y + q2.y + q3.y + w1_;
          } else {
            e0 = part[dc][0][crow][0] + part[dc][1][crow][0] + part[dc][2][crow][0] + part[dc][3][crow][0] + wo[2 * FHD];
            e1 = part[dc][0][crow][1] + part[dc][1][crow][1] + part[dc][2][crow][1] + part[dc][3][crow][1] + wo[2 * FHD + 1];
          }
          int id = e1 > e0 ? 1 : 0;                                             // first max wins ties (torch.max)
          if (a.force_dur) id = a.force_dur[(long)d * M + prC];
          dtk = 1 + id;
          if (tid < FP) {                                                      // wave 0, lanes 0..15: crow == tid
            if (okC) {
              a.dur[prC * 10 + 2 * d] = e0; a.dur[prC * 10 + 2 * d + 1] = e1;
              a.idx[(long)d * M + prC] = id;
            }
            bits[tid][d] = id;
          }
        }
      }
    }
    lds_barrier();                                                             // bits[] of the last duration step
    PH(4);
    // ================= P6: predicted token = note_embedding(onehot(pitch) | 5 duration bits); next input token =================
    {
      const int row = tid >> 4, e0 = (tid & 15) * 8;
      const bool ok = r0 + row < B && lead;
      const long wr = (long)t * B + min(r0 + row, B - 1);
      const int pch = pidx[row];
      float v[8];
      float4 b0, b1, w0, w1;
      if constexpr (RES) {
        b0 = *reinterpret_cast<const float4*>(&embc[0][e0]); b1 = *reinterpret_cast<const float4*>(&embc[0][e0 + 4]);
        w0 = ew0; w1 = ew1;
      } else {
        b0 = *reinterpret_cast<const float4*>(a.b_emb + e0); b1 = *reinterpret_cast<const float4*>(a.b_emb + e0 + 4);
        w0 = *reinterpret_cast<const float4*>(a.w_embT + (long)pch * FE + e0); w1 = *reinterpret_cast<const float4*>(a.w_embT + (long)pch * FE + e0 + 4);
      }
      v[0] = b0.x + w0.x; v[1] = b0.y + w0.y; v[2] = b0.z + w0.z; v[3] = b0.w + w0.w;
      v[4] = b1.x + w1.x; v[5] = b1.y + w1.y; v[6] = b1.z + w1.z; v[7] = b1.w + w1.w;
#pragma unroll
      for (int d = 0; d < 5; d++) {
        float4 q0, q1;
        if constexpr (RES) { q0 = *reinterpret_cast<const float4*>(&embc[1 + d][e0]); q1 = *reinterpret_cast<const float4*>(&embc[1 + d][e0 + 4]); }
        else { q0 = *reinterpret_cast<const float4*>(a.w_embT + (long)(FNP + d) * FE + e0); q1 = *reinterpret_cast<const float4*>(a.w_embT + (long)(FNP + d) * FE + e0 + 4); }
        const float f = (float)bits[row][d];
        v[0] += f * q0.x; v[1] += f * q0.y; v[2] += f * q0.z; v[3] += f * q0.w;
        v[4] += f * q1.x; v[5] += f * q1.y; v[6] += f * q1.z; v[7] += f * q1.w;
      }
      if (ok) {
        float* pp = a.PRED + ((long)(n + 1) * R + wr) * FE + e0;
        *reinterpret_cast<float4*>(pp) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(pp + 4) = make_float4(v[4], v[5], v[6], v[7]);
      }
      if (n < 14) {
        if ((a.coin_mask >> n) & 1u) {                                          // teacher forcing: the ground-truth note n+1
          const float* gp = a.emb + ((long)(n + 1) * R + wr) * FE + e0;
          const float4 g0 = *reinterpret_cast<const float4*>(gp), g1 = *reinterpret_cast<const float4*>(gp + 4);
          v[0] = g0.x; v[1] = g0.y; v[2] = g0.z; v[3] = g0.w; v[4] = g1.x; v[5] = g1.y; v[6] = g1.z; v[7] = g1.w;
        }
        if (a.tok_store && ok) {
          float* tp = a.TOK + ((long)(n + 1) * R + wr) * FE + e0;
          *reinterpret_cast<float4*>(tp) = make_float4(v[0], v[1], v[2], v[3]);
          *reinterpret_cast<float4*>(tp + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
        st_bf16x4_lds(&tok16[row][e0], v[0], v[1], v[2], v[3]);
        st_bf16x4_lds(&tok16[row][e0 + 4], v[4], v[5], v[6], v[7]);
      }
      if (tid < FP && r0 + tid < B && lead) {
        const int rw = r0 + tid;
        long* xr = a.xhat + (((long)rw * 32 + t) * 16 + n + 1) * 6;
        const int pb = pidx[tid];
        xr[0] = pb;
#pragma unroll
        for (int d = 0; d < 5; d++) xr[1 + d] = bits[tid][d];
        int L = a.plen[(long)t * B + rw];
        if (L == 0 && pb == 129) L = n + 1;                                     // first <eos>            (ptvae.py:415-416)
        if (n == 14 && L == 0) L = n + 1;                                       // no <eos> by the end     (ptvae.py:425)
        a.plen[(long)t * B + rw] = L;
      }
    }
    lds_barrier();
    PH(5);
  }
  if (a.dbg_out && tid == 0) {
    for (int i = 0; i < 6; i++) a.dbg_out[3L * gridDim.x + 8L * blockIdx.x + i] = tph[i];
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    a.dbg_out[blockIdx.x * 3 + 0] = xcc; a.dbg_out[blockIdx.x * 3 + 1] = hwid;
    a.dbg_out[blockIdx.x * 3 + 2] = (long)__builtin_amdgcn_s_memtime() - t_begin;
  }
}

// =============================================================================================
// The same note loop with the state products taken off the critical path: 8 waves.  The notes-GRU products W_hh . h are bound by
// the CU's L2 port (1.6 MB of fragments per note step) and need only the state, not the next token -- so waves 0-3 ("producers")
// stream them for step n+1 WHILE waves 4-7 ("heads") run pitch head, argmax, dur_hid, the duration GRU and the token embedding of
// step n.  When the token arrives, the producers add the short token part (K = 128), run the cell epilogue and hand the new state
// over.  Both groups execute the same nine workgroup barriers per note step (gfx950 has no named barriers), so each of the eight
// product segments is paired with one head phase:
//   producers: token part + epilogue | B0 | segment 1 | B1 | ... | segment 8 | B8
//   heads:                      wait | B0 | pitch head | B1 | argmax, dur_hid | B2 | dur step 0..4 | B3..B7 | embedding | B8
// =============================================================================================
__global__ __launch_bounds__(512, 1) void note_loop2_kernel(NoteLoopArgs a) {
  __shared__ __attribute__((aligned(16))) float hf[FP][FHN];                   // notes-GRU state, fp32
  __shared__ __attribute__((aligned(16))) __bf16 h16[2][FP][H16LD];            // its bf16 MFMA-operand copy (double buffered)
  __shared__ __attribute__((aligned(16))) __bf16 tok16[FP][T16LD];             // current input token
  __shared__ __attribute__((aligned(16))) float pit[FP][PITLD];                // pitch logits
  __shared__ __attribute__((aligned(16))) __bf16 pit16[FP][P16LD];
  __shared__ __attribute__((aligned(16))) float hdf[FP][FHD];                  // duration-GRU state
  __shared__ __attribute__((aligned(16))) __bf16 hd16[2][FP][D16LD];
  __shared__ __attribute__((aligned(16))) bf16x8 wdl[12 * 2 * 64];             // duration W_hh, fragment-major (24 KB)
  __shared__ float tabs[3][3 * FHD];
  __shared__ float bhd[3 * FHD];
  __shared__ float wo[2 * FHD + 2];
  __shared__ float part[2][4][FP][2];                                          // double buffered: one barrier per duration step
  __shared__ int pidx[FP];
  __shared__ int bits[FP][5];


  const int tid0 = threadIdx.x, lane = tid0 & 63;
  const bool producer = tid0 < 256;
  const int tid = tid0 & 255, wave = __builtin_amdgcn_readfirstlane((tid0 >> 6) & 3);   // thread / wave index inside the group (wave: SGPR)
  const int crow = lane & 15, ckq = lane >> 4;                // accumulator (C) layout
  const int erow = lane >> 2, eq = lane & 3;                  // epilogue layout
  const int B = a.B, R = a.R, t = a.t;
  const long M = a.M;
  const int r0 = blockIdx.x * FP;                             // first sample of the panel
  const int rE = min(r0 + erow, B - 1), rC = min(r0 + crow, B - 1);
  const bool okE = r0 + erow < B, okC = r0 + crow < B;
  const long wrowE = (long)t * B + rE, wrowC = (long)t * B + rC;   // row in the [R]-row step-major matrices

  // ---- one-time loads: duration GRU weights / tables -> LDS, initial state and first token -> LDS
  for (int i = tid0; i < 12 * 2 * 64; i += 512) wdl[i] = a.wdur[i];
  for (int i = tid0; i < 3 * FHD; i += 512) { tabs[0][i] = a.tab0[i]; tabs[1][i] = a.tab[i]; tabs[2][i] = a.tab[3 * FHD + i]; bhd[i] = a.b_hh_d[i]; }
  for (int i = tid0; i < 2 * FHD; i += 512) wo[i] = a.w_out[i];
  if (tid0 < 2) wo[2 * FHD + tid0] = a.b_out[tid0];
  for (int i = tid0; i < FP * (FHN / 4); i += 512) {
    const int row = i / (FHN / 4), c4 = (i % (FHN / 4)) * 4;
    const int rb = min(r0 + row, B - 1);
    const float4 v = a.h0 ? *reinterpret_cast<const float4*>(a.h0 + (long)rb * a.ld_h0 + c4)
                          : *reinterpret_cast<const float4*>(a.HN + ((long)t * B + rb) * FHN + c4);
    if (a.h0 && r0 + row < B) *reinterpret_cast<float4*>(a.HN + ((long)t * B + rb) * FHN + c4) = v;   // slot 0 of HN for the backward
    *reinterpret_cast<float4*>(&hf[row][c4]) = v;
    st_bf16x4_lds(&h16[0][row][c4], v.x, v.y, v.z, v.w);
    if (a.train && a.HN16 && r0 + row < B) st_bf16x4_lds(a.HN16 + ((long)t * B + r0 + row) * FHN + c4, v.x, v.y, v.z, v.w);
  }
  for (int i = tid0; i < FP * (FE / 4); i += 512) {
    const int row = i / (FE / 4), c4 = (i % (FE / 4)) * 4;
    const float4 v = *reinterpret_cast<const float4*>(a.TOK + ((long)t * B + min(r0 + row, B - 1)) * FE + c4);
    st_bf16x4_lds(&tok16[row][c4], v.x, v.y, v.z, v.w);
  }
  for (int i = tid0; i < FP * (P16LD - FNP); i += 512) pit16[i / (P16LD - FNP)][FNP + i % (P16LD - FNP)] = (__bf16)0.f;   // K padding of the logits operand
  if (tid0 < FP) pidx[tid0] = 0;

  __syncthreads();

  if (producer) {
    const int rl = lane & 15, kq = (lane >> 4) * 8;             // fragment coordinates
    f32x4 accH[4][6];                                           // W_hh . h of the step being prepared, all four passes
    bf16x8 tf[4][6];                                            // token-part fragments of the pass about to run
    auto tiles = [&](int p, int (&tl)[6]) {
      const int ut0 = wave * 8 + p * 2;
      tl[0] = ut0; tl[1] = ut0 + 1; tl[2] = 32 + ut0; tl[3] = 33 + ut0; tl[4] = 64 + ut0; tl[5] = 65 + ut0;
    };
    auto load_tok_frags = [&](const bf16x8* wt, int p) {
      int tl[6]; tiles(p, tl);
#pragma unroll
      for (int q = 0; q < 4; q++)
#pragma unroll
        for (int j = 0; j < 6; j++) tf[q][j] = (wt + (tl[j] * 4 + q) * 64)[lane];          // uniform base (SGPR) + lane offset: no 64-bit VGPR address per load
    };
    // the state products as ONE software pipeline over 64 units (pass, k-block) with a ring of 5 fragment sets; `sync` puts a
    // workgroup barrier after every 8th unit (the pairing with the head phases), loads keep flying across it
    auto h_products = [&](const bf16x8* wh, const __bf16* hsrc, bool sync) {
      bf16x8 b[5][6];
      auto ldu = [&](bf16x8 (&d)[6], int un) {
        int tl[6]; tiles(un >> 4, tl);
#pragma unroll
        for (int j = 0; j < 6; j++) d[j] = (wh + (tl[j] * 16 + (un & 15)) * 64)[lane];
      };
#pragma unroll
      for (int p = 0; p < 4; p++)
#pragma unroll
        for (int j = 0; j < 6; j++) accH[p][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      ldu(b[0], 0); ldu(b[1], 1); ldu(b[2], 2); ldu(b[3], 3);
#pragma unroll
      for (int un = 0; un < 64; un++) {
        if (un + 4 < 64) ldu(b[(un + 4) % 5], un + 4);
        const bf16x8 av = *reinterpret_cast<const bf16x8*>(hsrc + rl * H16LD + (un & 15) * 32 + kq);
#pragma unroll
        for (int j = 0; j < 6; j++) accH[un >> 4][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[un % 5][j], av, accH[un >> 4][j], 0, 0, 0);
        if (sync && (un & 7) == 7) lds_barrier();
        __builtin_amdgcn_sched_barrier(0);                        // or the scheduler hoists all 384 fragment loads and spills
      }
    };
    h_products(a.wg_h, &h16[0][0][0], false);                    // step 0: the heads wait at B0 meanwhile
    load_tok_frags(a.wg_t, 0);
    for (int n = 0; n < 15; n++) {
      const int cur = n & 1, nxt = cur ^ 1;
      (void)cur;
      // the weight pointers pass through an opaque register copy per note step: otherwise the compiler hoists the address arithmetic
      // of all 480 fragment loads out of this loop and spills it
      const bf16x8 *wh = a.wg_h, *wt = a.wg_t;
      asm volatile("" : "+s"(wh), "+s"(wt));
      // ---- token part + cell epilogue, pass by pass
#pragma unroll
      for (int p = 0; p < 4; p++) {
        const int ut0 = wave * 8 + p * 2;
        float4 gR[2], gZ[2], gN[2], bR[2], bZ[2], bN[2];
#pragma unroll
        for (int j = 0; j < 2; j++) {
          const int u = (ut0 + j) * 16 + eq * 4;
          const float* g = a.gc + (long)rE * a.ld_gc;
          gR[j] = *reinterpret_cast<const float4*>(g + u); gZ[j] = *reinterpret_cast<const float4*>(g + FHN + u); gN[j] = *reinterpret_cast<const float4*>(g + 2 * FHN + u);
          bR[j] = *reinterpret_cast<const float4*>(a.b_hh_n + u); bZ[j] = *reinterpret_cast<const float4*>(a.b_hh_n + FHN + u);
          bN[j] = *reinterpret_cast<const float4*>(a.b_hh_n + 2 * FHN + u);
        }
        f32x4 accT[6];
#pragma unroll
        for (int j = 0; j < 6; j++) accT[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const bf16x8 av = *reinterpret_cast<const bf16x8*>(&tok16[0][0] + rl * T16LD + q * 32 + kq);
#pragma unroll
          for (int j = 0; j < 6; j++) accT[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tf[q][j], av, accT[j], 0, 0, 0);
        }
#pragma unroll
      for (int j = 0; j < 2; j++) {
        const int u = (ut0 + j) * 16 + eq * 4;
        f32x4 sR, sZ;
#pragma unroll
        for (int e = 0; e < 4; e++) { sR[e] = accH[p][j][e] + accT[j][e]; sZ[e] = accH[p][2 + j][e] + accT[2 + j][e]; }
        const f32x4 aR = to_rowmajor_lanes(sR), aZ = to_rowmajor_lanes(sZ), aI = to_rowmajor_lanes(accT[4 + j]), aH = to_rowmajor_lanes(accH[p][4 + j]);
        const float4 hp4 = *reinterpret_cast<const float4*>(&hf[erow][u]);
        const float hp[4] = {hp4.x, hp4.y, hp4.z, hp4.w};
        const float kR[4] = {gR[j].x + bR[j].x, gR[j].y + bR[j].y, gR[j].z + bR[j].z, gR[j].w + bR[j].w};
        const float kZ[4] = {gZ[j].x + bZ[j].x, gZ[j].y + bZ[j].y, gZ[j].z + bZ[j].z, gZ[j].w + bZ[j].w};
        const float kN[4] = {gN[j].x, gN[j].y, gN[j].z, gN[j].w}, kB[4] = {bN[j].x, bN[j].y, bN[j].z, bN[j].w};
        float r[4], z[4], nn[4], hn[4], h[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
          r[e] = fsig(aR[e] + kR[e]);
          z[e] = fsig(aZ[e] + kZ[e]);
          hn[e] = aH[e] + kB[e];
          nn[e] = ftanh(aI[e] + kN[e] + r[e] * hn[e]);
          h[e] = (1.0f - z[e]) * nn[e] + z[e] * hp[e];
        }
        *reinterpret_cast<float4*>(&hf[erow][u]) = make_float4(h[0], h[1], h[2], h[3]);
        st_bf16x4_lds(&h16[nxt][erow][u], h[0], h[1], h[2], h[3]);
        if (a.train && okE) {
          *reinterpret_cast<float4*>(a.HN + ((long)(n + 1) * R + wrowE) * FHN + u) = make_float4(h[0], h[1], h[2], h[3]);
          if (a.HN16) st_bf16x4_lds(a.HN16 + ((long)(n + 1) * R + wrowE) * FHN + u, h[0], h[1], h[2], h[3]);
          __bf16* gp = a.gates_n + (((long)n * 4) * R + wrowE) * FHN + u;
          const long pl = (long)R * FHN;
          st_bf16x4_lds(gp, r[0], r[1], r[2], r[3]);
          st_bf16x4_lds(gp + pl, z[0], z[1], z[2], z[3]);
          st_bf16x4_lds(gp + 2 * pl, nn[0], nn[1], nn[2], nn[3]);
          st_bf16x4_lds(gp + 3 * pl, hn[0], hn[1], hn[2], hn[3]);
        }
      }
        if (p < 3) load_tok_frags(wt, p + 1);
      }
      lds_barrier();                                             // B0: the new state is in LDS
      if (n < 14) {
        h_products(wh, &h16[nxt][0][0], true);                   // B1..B8 inside
        load_tok_frags(wt, 0);
      } else {
#pragma unroll 1
        for (int s = 0; s < 8; s++) lds_barrier();
      }
    }
  } else {
    // the decision of duration step d for `row` from the four waves' partial sums (complete after that step's barrier)
    auto dur_decision = [&](int d, int row, long pr) {
      const int dc = d & 1;
      const float e0 = part[dc][0][row][0] + part[dc][1][row][0] + part[dc][2][row][0] + part[dc][3][row][0] + wo[2 * FHD];
      const float e1 = part[dc][0][row][1] + part[dc][1][row][1] + part[dc][2][row][1] + part[dc][3][row][1] + wo[2 * FHD + 1];
      int id = e1 > e0 ? 1 : 0;
      if (a.force_dur) id = a.force_dur[(long)d * M + pr];
      return id;
    };
    for (int n = 0; n < 15; n++) {
      const int cur = n & 1, nxt = cur ^ 1;
      (void)cur;
      lds_barrier();                                             // B0
    // ================= P2: pitch head (9 tiles over 4 waves) + the state part of dur_hid_linear (one tile per wave) =================
    f32x4 accD[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
    {
      const int tl[1] = {wave};
      panel_mma<1, 16, 8>(a.wd_h, tl, &h16[nxt][0][0], H16LD, accD);
    }
    for (int nt = wave; nt < 9; nt += 4) {
      f32x4 acc[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
      const int tl[1] = {nt};
      panel_mma<1, 16, 8>(a.wp, tl, &h16[nxt][0][0], H16LD, acc);
      const int c0 = nt * 16 + ckq * 4;
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const int c = c0 + e;
        const float v = c < FNP ? acc[0][e] + a.b_p[c] : 0.f;
        pit[crow][c] = v;
        pit16[crow][c] = (__bf16)v;
      }
    }
    lds_barrier();                                               // B1
    // ================= P3: argmax over the 130 logits (16 lanes per row, first maximal index) + logits out =================
    {
      const int row = tid >> 4, j = tid & 15;
      float best = -INFINITY; int bi = 0x7fffffff;
      const long pr = (long)n * R + (long)t * B + min(r0 + row, B - 1);
      const bool ok = r0 + row < B;
#pragma unroll
      for (int k = 0; k < 9; k++) {
        const int c = j + 16 * k;
        if (c < FNP) {
          const float v = pit[row][c];
          if (ok) a.pitch[pr * a.ld_pitch + c] = v;
          if (v > best) { best = v; bi = c; }
        }
      }
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
      }
      if (a.force_pitch) bi = a.force_pitch[(long)n * R + (long)t * B + min(r0 + row, B - 1)];
      if (j == 0) pidx[row] = bi;
    }
    // ================= P4: dur_hid_linear([h | logits]) -> initial duration state (wave w = units w*16..) =================
    {
      const int tl[1] = {wave};
      panel_mma<1, 5, 8>(a.wd_p, tl, &pit16[0][0], P16LD, accD);               // + the logits part (K = 130 padded to 160)
      const int u = wave * 16 + ckq * 4;
      const float4 b4 = *reinterpret_cast<const float4*>(a.b_dh + u);
      const float h[4] = {accD[0][0] + b4.x, accD[0][1] + b4.y, accD[0][2] + b4.z, accD[0][3] + b4.w};
      *reinterpret_cast<float4*>(&hdf[crow][u]) = make_float4(h[0], h[1], h[2], h[3]);
      st_bf16x4_lds(&hd16[0][crow][u], h[0], h[1], h[2], h[3]);
      if (a.train && okC) {
        *reinterpret_cast<float4*>(a.HD + ((long)n * R + wrowC) * FHD + u) = make_float4(h[0], h[1], h[2], h[3]);
        if (a.HD16) st_bf16x4_lds(a.HD16 + ((long)n * R + wrowC) * FHD + u, h[0], h[1], h[2], h[3]);
      }
    }
    lds_barrier();                                               // B2
    // ================= P5: 5-step duration GRU, argmax feedback (wave w = units w*16..w*16+15) =================
    {
      const long prC = (long)n * R + wrowC;
      const int u = wave * 16 + ckq * 4;
      int dtk = 0;                                                             // this lane's row: 0 = <sos>, 1 + previous decision
#pragma unroll 1
      for (int d = 0; d < 5; d++) {
        const int dc = d & 1, dn = dc ^ 1;
        f32x4 acc[3];
#pragma unroll
        for (int g = 0; g < 3; g++) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 2; kb++) {
          const bf16x8 av = *reinterpret_cast<const bf16x8*>(&hd16[dc][crow][kb * 32 + ckq * 8]);
#pragma unroll
          for (int g = 0; g < 3; g++) acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wdl[((g * 4 + wave) * 2 + kb) * 64 + lane], av, acc[g], 0, 0, 0);
        }
        const float* gi = tabs[dtk];
        const float4 hp4 = *reinterpret_cast<const float4*>(&hdf[crow][u]);
        const float hp[4] = {hp4.x, hp4.y, hp4.z, hp4.w};
        float r[4], z[4], nn[4], hn[4], h[4], o0 = 0.f, o1 = 0.f;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int jj = u + e;
          r[e] = fsig(gi[jj] + acc[0][e] + bhd[jj]);
          z[e] = fsig(gi[FHD + jj] + acc[1][e] + bhd[FHD + jj]);
          hn[e] = acc[2][e] + bhd[2 * FHD + jj];
          nn[e] = ftanh(gi[2 * FHD + jj] + r[e] * hn[e]);
          h[e] = (1.0f - z[e]) * nn[e] + z[e] * hp[e];
          o0 += wo[jj] * h[e]; o1 += wo[FHD + jj] * h[e];
        }
        *reinterpret_cast<float4*>(&hdf[crow][u]) = make_float4(h[0], h[1], h[2], h[3]);
        st_bf16x4_lds(&hd16[dn][crow][u], h[0], h[1], h[2], h[3]);
        if (a.train && okC) {
          if (a.HD16) st_bf16x4_lds(a.HD16 + ((long)(d + 1) * M + prC) * FHD + u, h[0], h[1], h[2], h[3]);
          else *reinterpret_cast<float4*>(a.HD + ((long)(d + 1) * M + prC) * FHD + u) = make_float4(h[0], h[1], h[2], h[3]);
          __bf16* gp = a.gates_d + (((long)d * 4) * M + prC) * FHD + u;
          const long pl = M * FHD;
          st_bf16x4_lds(gp, r[0], r[1], r[2], r[3]);
          st_bf16x4_lds(gp + pl, z[0], z[1], z[2], z[3]);
          st_bf16x4_lds(gp + 2 * pl, nn[0], nn[1], nn[2], nn[3]);
          st_bf16x4_lds(gp + 3 * pl, hn[0], hn[1], hn[2], hn[3]);
        }
        o0 += __shfl_xor(o0, 16, 64); o1 += __shfl_xor(o1, 16, 64);
        o0 += __shfl_xor(o0, 32, 64); o1 += __shfl_xor(o1, 32, 64);
        if (lane < 16) { part[dc][wave][lane][0] = o0; part[dc][wave][lane][1] = o1; }
        lds_barrier();
        // every lane forms the two logits of ITS row from the four waves' partial sums and takes the decision itself (no second
        // barrier to pass the token around); the panel's first 16 threads also publish them
        {
          const float e0 = part[dc][0][crow][0] + part[dc][1][crow][0] + part[dc][2][crow][0] + part[dc][3][crow][0] + wo[2 * FHD];
          const float e1 = part[dc][0][crow][1] + part[dc][1][crow][1] + part[dc][2][crow][1] + part[dc][3][crow][1] + wo[2 * FHD + 1];
          int id = e1 > e0 ? 1 : 0;                                             // first max wins ties (torch.max)
          if (a.force_dur) id = a.force_dur[(long)d * M + prC];
          dtk = 1 + id;
          if (tid < FP) {                                                      // wave 0, lanes 0..15: crow == tid
            if (okC) {
              a.dur[prC * 10 + 2 * d] = e0; a.dur[prC * 10 + 2 * d + 1] = e1;
              a.idx[(long)d * M + prC] = id;
            }
            bits[tid][d] = id;
          }
        }
      }
    }
    // ================= P6: predicted token = note_embedding(onehot(pitch) | 5 duration bits); next input token =================
    {
      const int row = tid >> 4, e0 = (tid & 15) * 8;
      const bool ok = r0 + row < B;
      const long wr = (long)t * B + min(r0 + row, B - 1);
      const int pch = pidx[row];
      const int last_bit = dur_decision(4, row, (long)n * R + wr);
      float v[8];
      const float4 b0 = *reinterpret_cast<const float4*>(a.b_emb + e0), b1 = *reinterpret_cast<const float4*>(a.b_emb + e0 + 4);
      const float4 w0 = *reinterpret_cast<const float4*>(a.w_embT + (long)pch * FE + e0), w1 = *reinterpret_cast<const float4*>(a.w_embT + (long)pch * FE + e0 + 4);
      v[0] = b0.x + w0.x; v[1] = b0.y + w0.y; v[2] = b0.z + w0.z; v[3] = b0.w + w0.w;
      v[4] = b1.x + w1.x; v[5] = b1.y + w1.y; v[6] = b1.z + w1.z; v[7] = b1.w + w1.w;
#pragma unroll
      for (int d = 0; d < 5; d++) {
        const float4 q0 = *reinterpret_cast<const float4*>(a.w_embT + (long)(FNP + d) * FE + e0), q1 = *reinterpret_cast<const float4*>(a.w_embT + (long)(FNP + d) * FE + e0 + 4);
        const float f = (float)(d < 4 ? bits[row][d] : last_bit);
        v[0] += f * q0.x; v[1] += f * q0.y; v[2] += f * q0.z; v[3] += f * q0.w;
        v[4] += f * q1.x; v[5] += f * q1.y; v[6] += f * q1.z; v[7] += f * q1.w;
      }
      if (ok) {
        float* pp = a.PRED + ((long)(n + 1) * R + wr) * FE + e0;
        *reinterpret_cast<float4*>(pp) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(pp + 4) = make_float4(v[4], v[5], v[6], v[7]);
      }
      if (n < 14) {
        if ((a.coin_mask >> n) & 1u) {                                          // teacher forcing: the ground-truth note n+1
          const float* gp = a.emb + ((long)(n + 1) * R + wr) * FE + e0;
          const float4 g0 = *reinterpret_cast<const float4*>(gp), g1 = *reinterpret_cast<const float4*>(gp + 4);
          v[0] = g0.x; v[1] = g0.y; v[2] = g0.z; v[3] = g0.w; v[4] = g1.x; v[5] = g1.y; v[6] = g1.z; v[7] = g1.w;
        }
        if (a.tok_store && ok) {
          float* tp = a.TOK + ((long)(n + 1) * R + wr) * FE + e0;
          *reinterpret_cast<float4*>(tp) = make_float4(v[0], v[1], v[2], v[3]);
          *reinterpret_cast<float4*>(tp + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
        st_bf16x4_lds(&tok16[row][e0], v[0], v[1], v[2], v[3]);
        st_bf16x4_lds(&tok16[row][e0 + 4], v[4], v[5], v[6], v[7]);
      }
      if (tid < FP && r0 + tid < B) {
        const int rw = r0 + tid;
        long* xr = a.xhat + (((long)rw * 32 + t) * 16 + n + 1) * 6;
        const int pb = pidx[tid];
        xr[0] = pb;
#pragma unroll
        for (int d = 0; d < 4; d++) xr[1 + d] = bits[tid][d];
        xr[5] = dur_decision(4, tid, (long)n * R + (long)t * B + rw);
        int L = a.plen[(long)t * B + rw];
        if (L == 0 && pb == 129) L = n + 1;                                     // first <eos>            (ptvae.py:415-416)
        if (n == 14 && L == 0) L = n + 1;                                       // no <eos> by the end     (ptvae.py:425)
        a.plen[(long)t * B + rw] = L;
      }
    }
    lds_barrier();                                               // B8: the next token is in LDS
    }
  }
}

// =============================================================================================
// re-summarisation of a panel's predicted notes: bi-GRU(128 -> 128) over PRED[0..15], packed by the predicted length
// (ptvae.py:480-486) -> the next time-step token TOKS[t+1] = [fwd final | bwd final].  grid = (panels, 2 directions)
// =============================================================================================
struct ResumArgs {
  const bf16x8* w_ih[2]; const bf16x8* w_hh[2];        // packed [24][4][64] each
  const float* b_ih[2]; const float* b_hh[2];           // [384]
  const float* PRED;                                    // [16][R][128]
  const int* plen;                                      // [R]
  float* XH[2]; __bf16* XG[2];                          // states [17][R][128] (slot 0 = 0), gates [16][4][R][128] (train)
  float* tok_next;                                      // TOKS[t+1]: [B][256]
  int B, t, R, train;
};

__global__ __launch_bounds__(256, 1) void resum_kernel(ResumArgs a) {
  __shared__ __attribute__((aligned(16))) __bf16 x16[16][FP][T16LD];        // the 16 predicted note tokens of the panel
  __shared__ __attribute__((aligned(16))) __bf16 h16[2][FP][T16LD];
  __shared__ __attribute__((aligned(16))) float hf[FP][FHE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int erow = lane >> 2, eq = lane & 3;
  const int B = a.B, R = a.R, t = a.t, dir = blockIdx.y;
  const int r0 = blockIdx.x * FP;
  const int rE = min(r0 + erow, B - 1);
  const bool okE = r0 + erow < B;
  const long wrowE = (long)t * B + rE;
  for (int i = tid; i < 16 * FP * (FE / 4); i += 256) {
    const int c4 = (i % (FE / 4)) * 4, row = (i / (FE / 4)) % FP, s = i / (FP * (FE / 4));
    const float4 v = *reinterpret_cast<const float4*>(a.PRED + ((long)s * R + (long)t * B + min(r0 + row, B - 1)) * FE + c4);
    st_bf16x4_lds(&x16[s][row][c4], v.x, v.y, v.z, v.w);
  }
  for (int i = tid; i < FP * FHE; i += 256) { hf[i / FHE][i % FHE] = 0.f; h16[0][i / FHE][i % FHE] = (__bf16)0.f; }
  const int len = a.plen[wrowE];
  float4 bR[2], bZ[2], bI[2], bH[2];
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int u = (wave * 2 + j) * 16 + eq * 4;
    const float4 ir = *reinterpret_cast<const float4*>(a.b_ih[dir] + u), hr = *reinterpret_cast<const float4*>(a.b_hh[dir] + u);
    const float4 iz = *reinterpret_cast<const float4*>(a.b_ih[dir] + FHE + u), hz = *reinterpret_cast<const float4*>(a.b_hh[dir] + FHE + u);
    bR[j] = make_float4(ir.x + hr.x, ir.y + hr.y, ir.z + hr.z, ir.w + hr.w);
    bZ[j] = make_float4(iz.x + hz.x, iz.y + hz.y, iz.z + hz.z, iz.w + hz.w);
    bI[j] = *reinterpret_cast<const float4*>(a.b_ih[dir] + 2 * FHE + u);
    bH[j] = *reinterpret_cast<const float4*>(a.b_hh[dir] + 2 * FHE + u);
  }
  // round 6: the weights of the 16 steps are the same 48 fragments per wave (192 registers; a wave is alone on its SIMD): loaded ONCE
  // instead of streamed from L2 every step (48 KB per wave and step was 1.9 of a step's 3.2 us); the A operands of a step are read
  // from LDS in one batch ahead of the MFMAs.  Same products in the same k order as gate_products<4, 4>: bit-identical.
  const int ut0 = wave * 2;
  const int tl[6] = {ut0, ut0 + 1, 8 + ut0, 9 + ut0, 16 + ut0, 17 + ut0};
  bf16x8 wih[6][4], whh[6][4];
#pragma unroll
  for (int j = 0; j < 6; j++)
#pragma unroll
    for (int q = 0; q < 4; q++) {
      wih[j][q] = a.w_ih[dir][((long)tl[j] * 4 + q) * 64 + lane];
      whh[j][q] = a.w_hh[dir][((long)tl[j] * 4 + q) * 64 + lane];
    }
  __syncthreads();
  for (int s = 0; s < 16; s++) {
    const int tt = dir ? 15 - s : s, cur = s & 1, nxt = cur ^ 1;
    f32x4 accX[6], accH[6];                                     // (r0, r1, z0, z1, n0, n1): x . W_ih^T and h . W_hh^T
#pragma unroll
    for (int j = 0; j < 6; j++) accX[j] = accH[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (a.train & 2) gate_products<4, 4>(a.w_ih[dir], a.w_hh[dir], tl, &x16[tt][0][0], T16LD, &h16[cur][0][0], T16LD, accX, accH);   // (timing comparisons)
    else {
      const int rl = lane & 15, kq = (lane >> 4) * 8;
      bf16x8 ax[4], ah[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        ax[q] = *reinterpret_cast<const bf16x8*>(&x16[tt][rl][q * 32 + kq]);
        ah[q] = *reinterpret_cast<const bf16x8*>(&h16[cur][rl][q * 32 + kq]);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 4; q++)
#pragma unroll
        for (int j = 0; j < 6; j++) accX[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wih[j][q], ax[q], accX[j], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 4; q++)
#pragma unroll
        for (int j = 0; j < 6; j++) accH[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whh[j][q], ah[q], accH[j], 0, 0, 0);
    }
    const bool live = tt < len;
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int u = (wave * 2 + j) * 16 + eq * 4;
      f32x4 sR, sZ;
#pragma unroll
      for (int e = 0; e < 4; e++) { sR[e] = accX[j][e] + accH[j][e]; sZ[e] = accX[2 + j][e] + accH[2 + j][e]; }
      const f32x4 aR = to_rowmajor_lanes(sR), aZ = to_rowmajor_lanes(sZ), aI = to_rowmajor_lanes(accX[4 + j]), aH = to_rowmajor_lanes(accH[4 + j]);
      const float4 hp4 = *reinterpret_cast<const float4*>(&hf[erow][u]);
      const float hp[4] = {hp4.x, hp4.y, hp4.z, hp4.w};
      const float kR[4] = {bR[j].x, bR[j].y, bR[j].z, bR[j].w}, kZ[4] = {bZ[j].x, bZ[j].y, bZ[j].z, bZ[j].w};
      const float kI[4] = {bI[j].x, bI[j].y, bI[j].z, bI[j].w}, kH[4] = {bH[j].x, bH[j].y, bH[j].z, bH[j].w};
      float r[4], z[4], nn[4], hn[4], h[4];
#pragma unroll
      for (int e = 0; e < 4; e++) {
        r[e] = fsig(aR[e] + kR[e]);
        z[e] = fsig(aZ[e] + kZ[e]);
        hn[e] = aH[e] + kH[e];
        nn[e] = ftanh(aI[e] + kI[e] + r[e] * hn[e]);
        if (!live) { r[e] = 0.f; z[e] = 1.f; nn[e] = 0.f; }                    // masked row: h' = h, zero gate grads
        h[e] = (1.0f - z[e]) * nn[e] + z[e] * hp[e];
      }
      *reinterpret_cast<float4*>(&hf[erow][u]) = make_float4(h[0], h[1], h[2], h[3]);
      st_bf16x4_lds(&h16[nxt][erow][u], h[0], h[1], h[2], h[3]);
      if (okE) {
        if (a.train & 1) {
          *reinterpret_cast<float4*>(a.XH[dir] + ((long)(s + 1) * R + wrowE) * FHE + u) = make_float4(h[0], h[1], h[2], h[3]);
          __bf16* gp = a.XG[dir] + (((long)s * 4) * R + wrowE) * FHE + u;
          const long pl = (long)R * FHE;
          st_bf16x4_lds(gp, r[0], r[1], r[2], r[3]);
          st_bf16x4_lds(gp + pl, z[0], z[1], z[2], z[3]);
          st_bf16x4_lds(gp + 2 * pl, nn[0], nn[1], nn[2], nn[3]);
          st_bf16x4_lds(gp + 3 * pl, hn[0], hn[1], hn[2], hn[3]);
        }
        if (s == 15) *reinterpret_cast<float4*>(a.tok_next + (long)rE * (2 * FHE) + dir * FHE + u) = make_float4(h[0], h[1], h[2], h[3]);
      }
    }
    lds_barrier();
  }
}

}  // namespace ptv

using namespace ptv;

extern "C" long ptv_pack_mfma_b_size(int N, int K) { return (long)((N + 15) / 16) * ((K + 31) / 32) * 512; }

// generalised: the source may be TRANSPOSED (element (n, k) at W[k*ld + n]) and the k-blocks written may be a sub-range [kb0, kb0 + ceil(K/32))
// of a packed buffer with KBtot k-blocks per tile (two sources side by side along K: csrc/heads.hip)
__global__ void pack_b2_kernel(const float* __restrict__ W, long ld, int N, int K, __bf16* __restrict__ out, int NT, int KB, int pairs, int trans,
                               int kb0, int KBtot) {
  const long total = (long)NT * KB * 64;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63);
    const long f = i >> 6;
    const int kb = (int)(f % KB), nt = (int)(f / KB);
    const int c = lane & 15;
    const int n = pairs ? (nt >> 1) * 32 + (c >> 2) * 8 + (nt & 1) * 4 + (c & 3) : nt * 16 + c;
    const int k0 = kb * 32 + (lane >> 4) * 8;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = (__bf16)((n < N && k0 + e < K) ? (trans ? W[(long)(k0 + e) * ld + n] : W[(long)n * ld + k0 + e]) : 0.f);
    *reinterpret_cast<bf16x8*>(out + (((long)nt * KBtot + kb0 + kb) * 64 + lane) * 8) = v;
  }
}

extern "C" int ptv_pack_mfma_b2(const float* W, long ld, int N, int K, void* out, int pairs, int trans, int NT, int kb0, int KBtot, void* stream) {
  if (!W || !out || N <= 0 || K <= 0 || NT * 16 < N || (pairs && (NT & 1)) || kb0 < 0 || kb0 + (K + 31) / 32 > KBtot) return PTV_ERR_ARG;
  if ((trans && ld < N) || (!trans && ld < K)) return PTV_ERR_ARG;
  const int KB = (K + 31) / 32;
  long nb = ((long)NT * KB * 64 + 255) / 256; if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(pack_b2_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, W, ld, N, K, (__bf16*)out, NT, KB, pairs, trans, kb0, KBtot);
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

// several packs in ONE launch (blockIdx.y = job): the per-step re-packs of a module's weights were 4-7 launches of a few microseconds each on
// the critical path and ~100 us of host time
struct PackJobs { const float* W[8]; long ld[8]; int N[8], K[8]; __bf16* out[8]; int pairs[8], trans[8], NT[8], kb0[8], KBtot[8]; };
__global__ void pack_multi_kernel(PackJobs j) {
  const int q = blockIdx.y;
  const float* __restrict__ W = j.W[q];
  const long ld = j.ld[q];
  const int N = j.N[q], K = j.K[q], NT = j.NT[q], KB = (K + 31) / 32, pairs = j.pairs[q], trans = j.trans[q], kb0 = j.kb0[q], KBtot = j.KBtot[q];
  __bf16* __restrict__ out = j.out[q];
  const long total = (long)NT * KB * 64;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int lane = (int)(i & 63);
    const long f = i >> 6;
    const int kb = (int)(f % KB), nt = (int)(f / KB);
    const int c = lane & 15;
    const int n = pairs ? (nt >> 1) * 32 + (c >> 2) * 8 + (nt & 1) * 4 + (c & 3) : nt * 16 + c;
    const int k0 = kb * 32 + (lane >> 4) * 8;
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = (__bf16)((n < N && k0 + e < K) ? (trans ? W[(long)(k0 + e) * ld + n] : W[(long)n * ld + k0 + e]) : 0.f);
    *reinterpret_cast<bf16x8*>(out + (((long)nt * KBtot + kb0 + kb) * 64 + lane) * 8) = v;
  }
}

// jobs: n <= 8 rows of 10 longs {W, ld, N, K, out, pairs, trans, NT, kb0, KBtot} (ptv_pack_mfma_b2's arguments)
extern "C" int ptv_pack_mfma_multi(const long* jobs, int n, void* stream) {
  if (!jobs || n < 1 || n > 8) return PTV_ERR_ARG;
  PackJobs j{};
  long most = 0;
  for (int q = 0; q < n; q++) {
    const long* r = jobs + 10 * q;
    j.W[q] = (const float*)r[0]; j.ld[q] = r[1]; j.N[q] = (int)r[2]; j.K[q] = (int)r[3]; j.out[q] = (__bf16*)r[4];
    j.pairs[q] = (int)r[5]; j.trans[q] = (int)r[6]; j.NT[q] = (int)r[7]; j.kb0[q] = (int)r[8]; j.KBtot[q] = (int)r[9];
    if (!j.W[q] || !j.out[q] || j.N[q] <= 0 || j.K[q] <= 0 || j.NT[q] * 16 < j.N[q] || (j.pairs[q] && (j.NT[q] & 1)) || j.kb0[q] < 0 ||
        j.kb0[q] + (j.K[q] + 31) / 32 > j.KBtot[q] || (j.trans[q] ? j.ld[q] < j.N[q] : j.ld[q] < j.K[q])) return PTV_ERR_ARG;
    const long t = (long)j.NT[q] * ((j.K[q] + 31) / 32) * 64;
    if (t > most) most = t;
  }
  long nb = (most + 255) / 256; if (nb > 512) nb = 512;
  hipLaunchKernelGGL(pack_multi_kernel, dim3((int)nb, n), dim3(256), 0, (hipStream_t)stream, j);
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

extern "C" int ptv_pack_mfma_b(const float* W, long ld, int N, int K, void* out, int pairs, void* stream) {
  if (!W || !out || N <= 0 || K <= 0 || ld < K || (pairs && (N & 31))) return PTV_ERR_ARG;
  const int NT = (N + 15) / 16, KB = (K + 31) / 32;
  long nb = ((long)NT * KB * 64 + 255) / 256; if (nb > 2048) nb = 2048;
  hipLaunchKernelGGL(pack_b_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, W, ld, N, K, (__bf16*)out, NT, KB, pairs);
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

// w[16]: wg_h, wg_t, wp, wd_h, wd_p, wdur (packed bf16), b_hh_n, b_p, b_dh, b_hh_d, tab0, tab, w_out, b_out, w_embT, b_emb
// io[21]: gc, emb, HN, gates_n, pitch, HD, gates_d, dur, idx, TOK, PRED, xhat, plen, force_pitch, force_dur, HN16, HD16, dbg words, h0gc,
//         xch, cnt (cluster mode, train bits 18-20 = S in {2, 4}: S workgroups per 16-sample panel, ptvae_hip.h)
extern "C" int ptv_free_note_loop(const void* const* w, const void* const* io, long ld_pitch, int B, int t, unsigned coin_mask, int train,
                                  void* stream) {
  if (!w || !io || B <= 0 || t < 0 || t >= 32) return PTV_ERR_ARG;
  for (int i = 0; i < 16; i++) if (!w[i]) return PTV_ERR_ARG;
  if ((!io[0] && !io[18]) || !io[2] || !io[4] || !io[7] || !io[8] || !io[9] || !io[10] || !io[11] || !io[12]) return PTV_ERR_ARG;
  if ((train & 3) == 1 && (!io[3] || !io[5] || !io[6])) return PTV_ERR_ARG;
  if (coin_mask && !io[1]) return PTV_ERR_ARG;
  NoteLoopArgs a{};
  a.wg_h = (const bf16x8*)w[0]; a.wg_t = (const bf16x8*)w[1]; a.wp = (const bf16x8*)w[2]; a.wd_h = (const bf16x8*)w[3];
  a.wd_p = (const bf16x8*)w[4]; a.wdur = (const bf16x8*)w[5];
  a.b_hh_n = (const float*)w[6]; a.b_p = (const float*)w[7]; a.b_dh = (const float*)w[8]; a.b_hh_d = (const float*)w[9];
  a.tab0 = (const float*)w[10]; a.tab = (const float*)w[11]; a.w_out = (const float*)w[12]; a.b_out = (const float*)w[13];
  a.w_embT = (const float*)w[14]; a.b_emb = (const float*)w[15];
  // io[18] (or NULL): [h0 | gc] of this time step as one fp32 [B][2048] matrix (one product for both, dec_time_to_notes_hid and the
  // hoisted part of W_ih stacked); then io[0] is ignored and slot 0 of HN is written here
  a.h0 = (const float*)io[18]; a.ld_h0 = 4 * FHN;
  a.gc = a.h0 ? a.h0 + FHN : (const float*)io[0]; a.ld_gc = a.h0 ? 4 * FHN : 3 * FHN;
  a.emb = (const float*)io[1]; a.HN = (float*)io[2]; a.gates_n = (__bf16*)io[3];
  a.pitch = (float*)io[4]; a.ld_pitch = ld_pitch; a.HD = (float*)io[5]; a.gates_d = (__bf16*)io[6]; a.dur = (float*)io[7];
  a.idx = (int*)io[8]; a.TOK = (float*)io[9]; a.PRED = (float*)io[10]; a.xhat = (long*)io[11]; a.plen = (int*)io[12];
  a.force_pitch = (const int*)io[13]; a.force_dur = (const int*)io[14]; a.HN16 = (__bf16*)io[15]; a.HD16 = (__bf16*)io[16];
  a.dbg_out = (train >> 8) & 64 ? (long*)io[17] : nullptr;
  a.B = B; a.t = t; a.R = 32 * B; a.M = 15 * 32 * B; a.coin_mask = coin_mask; a.train = (train & 3) == 1; a.tok_store = (train & 3) != 0; a.dbg = (train >> 8) & 0xff;
  // two kernels: 4 waves walking the phases one after the other, or producers / heads split over 8 waves (the state products of
  // the next note step under the heads of the current one).  Measured (round 3, profiles/LOG.md): with few panels (B = 512: 32
  // workgroups) a note step is bound by ONE CU's L2 port either way and the heads' small dependent loads queue behind the producers'
  // deep prefetch (30.4 vs 32.0 us per note step); with many panels (B = 2048) the chip's L2 is the limit and the overlap wins
  // (47.0 -> 40.8 us).  train bit 16 / 17 force the 4-wave / 8-wave kernel.
  const int panels = (B + FP - 1) / FP;
  // cluster mode: S members per panel, all co-resident (they wait for each other once per note step): at most one member per CU
  const int S = (train & 0x400000) ? 8 : ((train >> 18) & 7);     // (bit 22: eight members)
  const bool split = (train & 0x20000) || (!(train & 0x10000) && panels >= 96 && S <= 1);      // (a cluster request means the 4-wave kernel)
  a.S = 1;
  if (S > 1) {
    // (round 4: up to one member per CU -- 64 panels x 4 members at B = 1024, the per-GPU batch of BASELINE configs[4]; the launch takes
    // its turn among the persistent launches, so nothing else that spins is resident beside it)
    static int ncu = 0;
    if (ncu == 0) { int dev = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) ncu = 128; }
    if ((S != 2 && S != 4 && S != 8) || split || !io[19] || !io[20] || panels * S > ncu) return PTV_ERR_UNSUPPORTED;
    a.S = S; a.xch = (__bf16*)io[19]; a.cnt = (unsigned*)io[20];
  }
  // bench.py's roofline record of the step loop (tag 7): 15 note steps per launch; algorithmic MFMA work of a launch = 15 note steps x B rows x
  // (gate products 2*3Hn*(Hn+E) + pitch head 2*130*Hn + dur_hid 2*64*(Hn+130) + 5 duration steps 2*3*64*64 + token embedding 2*128*135)
  const int pi = prof::want(7, B, FHN) ? prof::begin((hipStream_t)stream) : -1;
  if (!split) {
    const dim3 grid(a.S > 1 ? (panels + 7) / 8 * 8 * a.S : panels);
    if (a.S == 8) hipLaunchKernelGGL((note_loop_kernel<1, 1>), grid, dim3(256), 0, (hipStream_t)stream, a);
    else if (train & 0x200000) hipLaunchKernelGGL((note_loop_kernel<0, 2>), grid, dim3(256), 0, (hipStream_t)stream, a);   // bit 21: head weights streamed (timing comparisons)
    else hipLaunchKernelGGL((note_loop_kernel<1, 2>), grid, dim3(256), 0, (hipStream_t)stream, a);
  }
  else hipLaunchKernelGGL(note_loop2_kernel, dim3(panels), dim3(512), 0, (hipStream_t)stream, a);
  if (pi >= 0) prof::end(pi, (hipStream_t)stream, 15.0 * B * (2.0 * 3 * FHN * (FHN + 128) + 2.0 * 130 * FHN + 2.0 * 64 * (FHN + 130) + 5 * 2.0 * 3 * 64 * 64 + 2.0 * 128 * 135));
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

// w[8]: w_ih, w_hh, w_ih_r, w_hh_r (packed bf16), b_ih, b_hh, b_ih_r, b_hh_r;  io[7]: PRED, plen, XH0, XH1, XG0, XG1, tok_next
extern "C" int ptv_free_resummarize(const void* const* w, const void* const* io, int B, int t, int train, void* stream) {
  if (!w || !io || B <= 0 || t < 0 || t >= 32) return PTV_ERR_ARG;
  for (int i = 0; i < 8; i++) if (!w[i]) return PTV_ERR_ARG;
  if (!io[0] || !io[1] || !io[6]) return PTV_ERR_ARG;
  if ((train & 1) && (!io[2] || !io[3] || !io[4] || !io[5])) return PTV_ERR_ARG;
  ResumArgs a{};
  a.w_ih[0] = (const bf16x8*)w[0]; a.w_hh[0] = (const bf16x8*)w[1]; a.w_ih[1] = (const bf16x8*)w[2]; a.w_hh[1] = (const bf16x8*)w[3];
  a.b_ih[0] = (const float*)w[4]; a.b_hh[0] = (const float*)w[5]; a.b_ih[1] = (const float*)w[6]; a.b_hh[1] = (const float*)w[7];
  a.PRED = (const float*)io[0]; a.plen = (const int*)io[1]; a.XH[0] = (float*)io[2]; a.XH[1] = (float*)io[3];
  a.XG[0] = (__bf16*)io[4]; a.XG[1] = (__bf16*)io[5]; a.tok_next = (float*)io[6];
  a.B = B; a.t = t; a.R = 32 * B; a.train = train;
  hipLaunchKernelGGL(resum_kernel, dim3((B + FP - 1) / FP, 2), dim3(256), 0, (hipStream_t)stream, a);
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

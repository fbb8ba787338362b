// notes_roles.hip -- the teacher-forced notes GRU forward (dec_notes_gru, ptvae.py:395-398 as ONE 15-step sequence over 32*B rows) with
// WAVE ROLES: the round-5 rebuild of notes_persist.hip's row_gru_fwd_kernel<512>.
//
// Why: in the 4-wave kernel every wave did everything -- streamed W_hh | W_x from L2, issued the MFMAs, ran the cell arithmetic and moved
// the panel's HBM operands / results.  A wave's vector-memory queue retires in order, so every wait for a (fast, L2-resident) weight
// fragment also waited for the (slow, HBM) loads and stores issued before it, and with one wave per SIMD the three phases of a note step
// (weight stream ~20 us, products 13 us, cell arithmetic + activation traffic ~25 us) ran back to back: 62 us per step, 0.36 of the HBM
// roofline for two rounds ("a wave's memory queue is in order").
//
// Here a workgroup is 8 waves, two per SIMD, with disjoint jobs:
//   * PRODUCT waves (0-3): stream the weight fragments L2 -> registers (a ring of D k-blocks, nothing else in their vector-memory queue),
//     read the bf16 state / token operand from LDS, issue the MFMAs.  A product wave owns 128 units; it walks them in 8 mini-passes of
//     ONE 16-unit tile x 64 rows (r, z, W_hn h, W_in x accumulators: 64 registers), and hands the finished accumulators to its partner
//     through a 16-KB LDS slot.
//   * CELL waves (4-7): take the accumulators out of the slot in whatever lane layout suits the memory system, fetch the hoisted input
//     part GC and the fp32 state (requested one mini-pass ahead), run the sigmoid / tanh / blend, store state and gate planes, and keep
//     the new bf16 state in registers until the step's products are done (the operand copy in LDS is single-buffered: that is what
//     pays for the slots).  They also move the fed tokens and the bf16 state copy HN16.
// The two run side by side on every SIMD: MFMA beside VALU, the L2 weight stream beside the HBM activation streams.
// Hand-off: two monotonic counters per pair in LDS (filled / drained), polled with s_sleep; LDS serves a wave's requests in order, so
// data-then-flag needs no wait.  Two workgroup barriers per note step bracket the rewrite of the operand copy.
//
// Layouts private to this kernel and its BPTT twin: GC and the four gate planes are UNIT-BLOCKED BY 16 ([u / 16][row][16] bf16): a cell
// lane owns 8 consecutive units of two rows, so one wave access is one contiguous kilobyte.  Weights: ptv_pack_mfma_b with pairs = 0.
#include "common.hpp"
#include "gemm_core.hpp"
#include "prof.hpp"
#include "../../include/ptvae_hip.h"
#include "../../include/ptvae_hip_debug.h"

namespace ptv {

extern int g_gemm_prio;

namespace nr {

constexpr int H = 512, E = 128, ROWS = 64, NUT = H / 16, KBH = H / 32, KBX = E / 32, KT = KBH + KBX;
constexpr int HLD = H + 16, TLD = E + 16;                 // bf16 LDS row strides (+16: conflict-free b128 fragment reads)
constexpr int MPS = 8;                                    // mini-passes per product wave and step: 4 waves x 8 x 16 units = 512
constexpr int SLOT_BYTES = 4 * ROWS * 16 * 4;             // [gate r z hn in][64 rows][16 units] fp32
constexpr int OFF_H16 = 0;
constexpr int OFF_TOK = OFF_H16 + ROWS * HLD * 2;
constexpr int OFF_SLOT = OFF_TOK + ROWS * TLD * 2;
constexpr int OFF_FLAG = OFF_SLOT + 4 * SLOT_BYTES;
constexpr int OFF_BIAS = OFF_FLAG + 64;                    // b_hh fp32 [1536]: the product waves' vector-memory queue holds weights only
constexpr int LDS_BYTES = OFF_BIAS + 3 * H * 4;

typedef __attribute__((ext_vector_type(4))) float f4v;
typedef __attribute__((ext_vector_type(4))) unsigned u4v;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, long bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)(bytes > 0x7fffffffL ? 0x7fffffffL : bytes), 0x00020000);
}
__device__ __forceinline__ u4v as_u4(const bf16x8& v) { return __builtin_bit_cast(u4v, v); }
// two fp32 -> packed bf16 pair (RNE: v_cvt_pk_bf16_f32)
__device__ __forceinline__ unsigned pk2(float a, float b) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  bf16x2 v; v[0] = (__bf16)a; v[1] = (__bf16)b;
  return __builtin_bit_cast(unsigned, v);
}
// element e (0..7) of eight packed bf16 as fp32
__device__ __forceinline__ float bf_at(const u4v& v, int e) { const unsigned w = v[e >> 1]; return __uint_as_float((e & 1) ? (w & 0xffff0000u) : (w << 16)); }

__device__ __forceinline__ float nsig(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float ntanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

// element (row, unit u) of a unit-blocked-by-16 plane of R rows
__device__ __forceinline__ long blk16(long row, int u, long R) { return ((long)(u >> 4) * R + row) * 16 + (u & 15); }

__device__ __forceinline__ void flag_wait(volatile int* f, int v) {
  while (*f < v) __builtin_amdgcn_s_sleep(1);
  asm volatile("" ::: "memory");
}
__device__ __forceinline__ void flag_set(volatile int* f, int v) {
  asm volatile("" ::: "memory");
  *f = v;
  asm volatile("" ::: "memory");
}

struct Args {
  const bf16x8 *w_hh, *w_x;        // pairs = 0 packing: W_hh [96 tiles][16 kb][64], W_x [96][4][64]
  const float* b_hh;               // [1536]
  const __bf16* gc;                // [96][R][16]: the hoisted input part (b_ih folded in), unit-blocked by 16
  const float* x; long x_step;     // fed tokens fp32: x + t * x_step + row * 128
  const float* h0;                 // [R][512] fp32: the initial state
  __bf16* HN16;                    // [T + 1][R][512] bf16: every state, slot 0 included, written here
  __bf16* gates;                   // [T][4][32][R][16] (r, z, n, W_hn h + b_hn) or null
  int R, T, dbg;
  unsigned long long* trace;       // timing experiments: per-wave event stamps of workgroup 0 (s_memtime), or null
  const int* live_top;             // or null: device int -- only the note steps 0 .. *live_top are wanted by the caller; later HN16 slots
                                   // and gate planes stay unwritten
  const int* row_len;              // or null: [R] live note steps per row, rows sorted by DESCENDING length (ptv_rows_by_length): a panel runs the
                                   // steps its longest (= first) row has; the HN16 slots of its dead steps up to the launch-wide limit are
                                   // zero-filled (finite operands for the weight-gradient products), their gate planes stay unwritten
  int nofill;                      // (T bit 24) ... unless the caller's products clip to the same 128-row segments (ptv_wgrad_job.seg_n): unwritten too
};

// slot address of the 16-byte chunk j (units 4j .. 4j+3) of (gate, row): chunks are XOR-swizzled so that both the product wave's
// fragment-shaped stores (16 rows x one chunk per instruction) and the cell wave's reads (32 rows x two chunks) are conflict-free
__device__ __forceinline__ int slot_off(int gate, int row, int j) { return gate * (ROWS * 64) + row * 64 + ((j ^ ((row >> 2) & 1)) << 4); }

template <int DF, int ABL>
__global__ __launch_bounds__(512, 2) void notes_fwd_kernel(Args a) {
  extern __shared__ __attribute__((aligned(16))) char nsm[];
  __bf16* h16 = reinterpret_cast<__bf16*>(nsm + OFF_H16);                 // [64][HLD] the state as MFMA operand (single buffer)
  __bf16* tok16 = reinterpret_cast<__bf16*>(nsm + OFF_TOK);               // [64][TLD] this step's fed tokens
  volatile int* flags = reinterpret_cast<volatile int*>(nsm + OFF_FLAG);  // [pair][0 = filled, 1 = drained]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long R = a.R, RH = R * H;
  const long r0 = (long)blockIdx.x * ROWS;
  const int Tg = a.live_top ? min(a.T, max(__builtin_amdgcn_readfirstlane(*a.live_top), 0) + 1) : a.T;      // launch-wide
  // this panel: the steps the first row of its 128-row BLOCK has -- one granularity of deadness for every kernel of the chain (the heads
  // work on 128-row blocks): inside a live block every row is computed (finite values meet the zero gradients of its shorter rows),
  // a dead block is read by nobody
  const int T = a.row_len ? min(Tg, max(__builtin_amdgcn_readfirstlane(a.row_len[r0 & ~127L]), 0)) : Tg;
  // workgroups of one XCD run in near lockstep and would ask the L2 for the same fragment lines at the same moment: each walks its
  // mini-passes from its own starting point (a k-block rotation on top measured nothing and costs 80 address registers)
  const int rot = (a.dbg & 8) ? 0 : (blockIdx.x >> 3) & (MPS - 1);

  unsigned long long* tr = (a.trace && blockIdx.x == 8) ? a.trace + wave * 2048 : nullptr;
  int tri = 0;
  auto stamp = [&](int code) {
    if (tr && tri < 2040) { const unsigned long long t = __builtin_amdgcn_s_memtime(); if (lane == 0) tr[tri] = (t << 8) | (unsigned)code; tri++; }
  };
  __builtin_amdgcn_s_setprio(3);
  // ---- prologue (all waves): initial state -> bf16 operand copy + HN16 slot 0; step 0's tokens; counters
  if (tid < 16) flags[tid] = 0;
  {
    float* bl = reinterpret_cast<float*>(nsm + OFF_BIAS);
    for (int i = tid; i < 3 * H; i += 512) bl[i] = a.b_hh[i];
  }
  for (int i = tid; i < ROWS * (H / 8); i += 512) {
    const int row = i / (H / 8), c8 = (i % (H / 8)) * 8;
    const long gr = min(r0 + row, R - 1);
    const float4 v0 = *reinterpret_cast<const float4*>(a.h0 + gr * H + c8), v1 = *reinterpret_cast<const float4*>(a.h0 + gr * H + c8 + 4);
    bf16x8 o;
    o[0] = (__bf16)v0.x; o[1] = (__bf16)v0.y; o[2] = (__bf16)v0.z; o[3] = (__bf16)v0.w;
    o[4] = (__bf16)v1.x; o[5] = (__bf16)v1.y; o[6] = (__bf16)v1.z; o[7] = (__bf16)v1.w;
    *reinterpret_cast<bf16x8*>(h16 + row * HLD + c8) = o;
    if (r0 + row < R) *reinterpret_cast<bf16x8*>(a.HN16 + gr * H + c8) = o;
  }
  for (int i = tid; i < ROWS * (E / 4); i += 512) {
    const int row = i / (E / 4), c4 = (i % (E / 4)) * 4;
    const float4 v = *reinterpret_cast<const float4*>(a.x + min(r0 + row, R - 1) * E + c4);
    bf16x4 o; o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
    *reinterpret_cast<bf16x4*>(tok16 + row * TLD + c4) = o;
  }
  __syncthreads();

  if (a.dbg & 16) { if (wave >= 4) __builtin_amdgcn_s_setprio(1); }
  if (a.dbg & 32) { if (wave < 4) __builtin_amdgcn_s_setprio(1); }
  if (wave < 4) {
    // =========================================================================================================================
    // PRODUCT wave
    // =========================================================================================================================
    const int rl = lane & 15, q = lane >> 4;
    char* slot = nsm + OFF_SLOT + wave * SLOT_BYTES;
    volatile int* f_fill = flags + 2 * wave;
    volatile int* f_drain = flags + 2 * wave + 1;
    const bf16x8* wh = a.w_hh + lane;
    const bf16x8* wx = a.w_x + lane;
    // the weight stream: fragment f = 3 k + gate of the current unit tile (60 per mini-pass), a ring of DF fragments, DF - 3 requested
    // ahead of the one being multiplied; 60 % DF == 0 keeps every ring index static across the mini-pass loop
    static_assert((3 * KT) % DF == 0 && DF % 3 == 0, "ring");
    bf16x8 ring[DF];
    auto ldw = [&](bf16x8& d, int ut, int f) {                                 // k < KBH: W_hh block k; else W_x block k - KBH
      if constexpr (ABL & 4) return;
      const int k = f / 3, g = f % 3;
      d = k < KBH ? wh[((long)(g * NUT + ut) * KBH + k) * 64] : wx[((long)(g * NUT + ut) * KBX + (k - KBH)) * 64];
    };
    auto tile_of = [&](int mp) { return wave * MPS + ((mp + rot) & (MPS - 1)); };
    if constexpr (ABL & 4) {
#pragma unroll
      for (int f = 0; f < DF; f++) ring[f] = wh[f * 64];
    }
    {
      const int ut = tile_of(0);
#pragma unroll
      for (int f = 0; f < DF - 3; f++) ldw(ring[f], ut, f);
    }
    int cnt = 0;                                                              // mini-passes handed over so far
    u4v tk[8];                                                                // the next step's fed tokens (fp32), this wave's quarter of the panel
    for (int n = 0; n < T; n++) {
      // next step's tokens of the panel: 32 KB contiguous, rows beyond R read as zero (descriptor bounds)
      const __amdgpu_buffer_rsrc_t rs_x = rsrc(a.x + (long)(n + 1 < T ? n + 1 : n) * a.x_step + r0 * E, min((long)ROWS, R - r0) * E * 4);
#pragma unroll 1
      for (int mp = 0; mp < MPS; mp++) {
        const int ut = tile_of(mp), utn = tile_of((mp + 1) & (MPS - 1));
        // requested in front of the step's last k-loop: every weight fragment that loop waits for is OLDER (the in-order queue never
        // waits for these), and by the loop's end all but the newest DF - 3 requests of the wave have returned -- these among them
        if (mp == MPS - 1 && n + 1 < T) {
#pragma unroll
          for (int j = 0; j < 8; j++) tk[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (unsigned)(wave * 64 + lane) * 16u, (unsigned)j * 4096u, 0);
        }
        const int u = ut * 16 + q * 4;                                         // this lane's 4 units of the tile (MFMA C layout)
        stamp(1);
        f32x4 acc[4][4];                                                       // [M tile][r, z, W_hn h, W_in x]
        {
          const float* bl = reinterpret_cast<const float*>(nsm + OFF_BIAS);
          const f32x4 bR = *reinterpret_cast<const f32x4*>(bl + u), bZ = *reinterpret_cast<const f32x4*>(bl + H + u),
                      bN = *reinterpret_cast<const f32x4*>(bl + 2 * H + u);
#pragma unroll
          for (int i = 0; i < 4; i++) { acc[i][0] = bR; acc[i][1] = bZ; acc[i][2] = bN; acc[i][3] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        }
        // A fragments (state / token operand, LDS) run one k-block ahead of the MFMAs that consume them
        bf16x8 av[2][4];
        auto lda_ = [&](bf16x8 (&d)[4], int k) {
          const bool tokpart = k >= KBH;
          const __bf16* A = tokpart ? tok16 : h16;
          const int lda = tokpart ? TLD : HLD, kb = tokpart ? k - KBH : k;
#pragma unroll
          for (int i = 0; i < 4; i++) d[i] = *reinterpret_cast<const bf16x8*>(A + (i * 16 + rl) * lda + kb * 32 + q * 8);
        };
        lda_(av[0], 0);
#pragma unroll
        for (int k = 0; k < KT; k++) {
          // requests first, pinned (hipcc otherwise sinks every LDS read to just in front of its MFMA and waits ~130 cycles for it)
          if (k + 1 < KT) lda_(av[(k + 1) & 1], k + 1);
#pragma unroll
          for (int g = 0; g < 3; g++) {
            const int fn = 3 * k + g + DF - 3;                                 // (the slot of the previous k-block's fragment g: consumed)
            if (fn < 3 * KT) ldw(ring[fn % DF], ut, fn); else ldw(ring[fn % DF], utn, fn - 3 * KT);
          }
          __builtin_amdgcn_sched_barrier(0);
          const bool tokpart = k >= KBH;
#pragma unroll
          for (int g = 0; g < 3; g++) {
            const int slot_ = g < 2 ? g : (tokpart ? 3 : 2);
#pragma unroll
            for (int i = 0; i < 4; i++) acc[i][slot_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ring[(3 * k + g) % DF], av[k & 1][i], acc[i][slot_], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        // ---- hand the accumulators over
        stamp(2);
        flag_wait(f_drain, cnt);
        stamp(3);
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
          for (int g = 0; g < 4; g++) *reinterpret_cast<f32x4*>(slot + slot_off(g, i * 16 + rl, q)) = acc[i][g];
        cnt++;
        flag_set(f_fill, cnt);
        stamp(4);
      }
      lds_barrier();          // B1: every product wave is done reading this step's operand copy
      stamp(5);
      if (n + 1 < T) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const int li = wave * 64 + lane, row = j * 8 + (li >> 5), c4 = (li & 31) * 4;
          bf16x4 o; o[0] = (__bf16)__uint_as_float(tk[j][0]); o[1] = (__bf16)__uint_as_float(tk[j][1]); o[2] = (__bf16)__uint_as_float(tk[j][2]); o[3] = (__bf16)__uint_as_float(tk[j][3]);
          *reinterpret_cast<bf16x4*>(tok16 + row * TLD + c4) = o;
        }
      }
      lds_barrier();          // B2: the cell waves have rewritten it
      stamp(6);
    }
  } else {
    // =========================================================================================================================
    // CELL wave: lane = (row pair rr / rr + 32, unit half hh): 8 consecutive units of two rows per mini-pass
    // =========================================================================================================================
    const int cw = wave - 4;
    const int rr = lane >> 1, hh = lane & 1;
    const char* slot = nsm + OFF_SLOT + cw * SLOT_BYTES;
    volatile int* f_fill = flags + 2 * cw;
    volatile int* f_drain = flags + 2 * cw + 1;
    long grow[2]; bool ok[2];
#pragma unroll
    for (int c = 0; c < 2; c++) { ok[c] = r0 + rr + 32 * c < R; grow[c] = min(r0 + rr + 32 * c, R - 1); }
    int rotv = rot;                                                           // (re-read per step through an opaque copy: hipcc would hoist
    auto tile_of = [&](int mp) { return cw * MPS + ((mp + rotv) & (MPS - 1)); };   // every per-mini-pass address out of the step loop and spill them)
    // every HBM operand goes through a buffer descriptor: wave-uniform base (+ a scalar offset per mini-pass / gate) and ONE per-lane
    // 32-bit offset per row -- with flat addresses hipcc hoists the 8 x 2 x 8 lane addresses of a step out of the step loop (190 spills)
    const unsigned vo16[2] = {(unsigned)(grow[0] * 32 + hh * 16), (unsigned)(grow[1] * 32 + hh * 16)};       // unit-blocked bf16 planes
    const unsigned plane = (unsigned)(RH * 2);                                  // bytes of one [32][R][16] bf16 plane
    const __amdgpu_buffer_rsrc_t rs_gc = rsrc(a.gc, 3L * RH * 2);
    struct Ops { u4v g[2][3]; };
    auto ldops = [&](Ops& o, int mp) {
      const int ut = tile_of(mp);
      if constexpr (ABL & 2) return;
#pragma unroll
      for (int c = 0; c < 2; c++) {
#pragma unroll
        for (int g = 0; g < 3; g++) o.g[c][g] = __builtin_amdgcn_raw_buffer_load_b128(rs_gc, vo16[c], (unsigned)g * plane + (unsigned)ut * (unsigned)(R * 32), 2);
      }
    };
    int cnt = 0;
    // the fp32 state of this lane's 128 cells lives in registers for the whole sequence: no per-step store + read-back of [R][512] fp32
    // (68 MB per step at B = 512 -- a third of the kernel's HBM traffic; the backward reads the bf16 copy HN16)
    f4v st[MPS][2][2];
    {
      const __amdgpu_buffer_rsrc_t rs_h0 = rsrc(a.h0, RH * 4);
#pragma unroll
      for (int mp = 0; mp < MPS; mp++)
#pragma unroll
        for (int c = 0; c < 2; c++) {
          const unsigned vo = (unsigned)(grow[c] * (H * 4) + hh * 32), so = (unsigned)tile_of(mp) * 64u;
          const u4v v0 = __builtin_amdgcn_raw_buffer_load_b128(rs_h0, vo, so, 0), v1 = __builtin_amdgcn_raw_buffer_load_b128(rs_h0, vo + 16u, so, 0);
#pragma unroll
          for (int e = 0; e < 4; e++) { st[mp][c][0][e] = __uint_as_float(v0[e]); st[mp][c][1][e] = __uint_as_float(v1[e]); }
        }
    }
    for (int n = 0; n < T; n++) {
      asm volatile("" : "+s"(rotv));
      Ops ops[2];
      const __amdgpu_buffer_rsrc_t rs_g = rsrc(a.gates ? a.gates + (long)n * 4 * RH : a.HN16, 4L * RH * 2);
      ldops(ops[0], 0);
      // bf16 state copy of this step's INPUT state h_n: LDS (stable until B1) -> HN16 slot n, whole rows (slot 0: the prologue)
      if (n > 0 && !((ABL & 1) && a.R > 0)) {
#pragma unroll 4
        for (int i = cw * 64 + lane; i < ROWS * (H / 8); i += 256) {
          const int row = i / (H / 8), c8 = (i % (H / 8)) * 8;
          if (r0 + row < R)
            __builtin_nontemporal_store(*reinterpret_cast<const bf16x8*>(h16 + row * HLD + c8),
                                        reinterpret_cast<bf16x8*>(a.HN16 + (long)n * RH + (r0 + row) * H + c8));
        }
      }
#pragma unroll
      for (int mp = 0; mp < MPS; mp++) {
        if (mp + 1 < MPS) ldops(ops[(mp + 1) & 1], mp + 1);
        const Ops& o = ops[mp & 1];
        const int ut = tile_of(mp);
        cnt++;
        stamp(11);
        flag_wait(f_fill, cnt);
        stamp(12);
#pragma unroll
        for (int c = 0; c < 2; c++) {
          const int row = rr + 32 * c;
          // two halves of 4 units: half the live values (the cell waves hold 64 registers of new state for the whole step)
          u4v r8, z8, n8, q8;
#pragma unroll
          for (int hf = 0; hf < 2; hf++) {
            f4v ac[4];
#pragma unroll
            for (int g = 0; g < 4; g++) ac[g] = *reinterpret_cast<const f4v*>(slot + slot_off(g, row, 2 * hh + hf));
            if (c == 1 && hf == 1) flag_set(f_drain, cnt);                     // (queued behind this wave's reads: LDS serves a wave in order)
            float r[4], z[4], nn[4], hn[4];
            f4v nv;
#pragma unroll
            for (int e = 0; e < 4; e++) {
              const float hp = st[mp][c][hf][e];
              r[e] = nsig(ac[0][e] + bf_at(o.g[c][0], 4 * hf + e));
              z[e] = nsig(ac[1][e] + bf_at(o.g[c][1], 4 * hf + e));
              hn[e] = ac[2][e];
              nn[e] = ntanh(ac[3][e] + bf_at(o.g[c][2], 4 * hf + e) + r[e] * hn[e]);
              nv[e] = (1.0f - z[e]) * nn[e] + z[e] * hp;
            }
            // (the state stays in aligned 4-register tuples: 128 scalar pieces scattered over the file left hipcc no room for the 16-byte
            // load / LDS-read destinations -- 164 spills)
            asm volatile("" : "+v"(nv));
            st[mp][c][hf] = nv;
            r8[2 * hf] = pk2(r[0], r[1]); r8[2 * hf + 1] = pk2(r[2], r[3]);
            z8[2 * hf] = pk2(z[0], z[1]); z8[2 * hf + 1] = pk2(z[2], z[3]);
            n8[2 * hf] = pk2(nn[0], nn[1]); n8[2 * hf + 1] = pk2(nn[2], nn[3]);
            q8[2 * hf] = pk2(hn[0], hn[1]); q8[2 * hf + 1] = pk2(hn[2], hn[3]);
            __builtin_amdgcn_sched_barrier(0);
          }
          if (ok[c] && !((ABL & 1) && a.R > 0)) {
            if (a.gates) {
              const unsigned so = (unsigned)ut * (unsigned)(R * 32);
              __builtin_amdgcn_raw_buffer_store_b128(r8, rs_g, vo16[c], so, 2);
              __builtin_amdgcn_raw_buffer_store_b128(z8, rs_g, vo16[c], so + plane, 2);
              __builtin_amdgcn_raw_buffer_store_b128(n8, rs_g, vo16[c], so + 2 * plane, 2);
              __builtin_amdgcn_raw_buffer_store_b128(q8, rs_g, vo16[c], so + 3 * plane, 2);
            }
          }
        }
      }
      stamp(13);
      lds_barrier();          // B1: the products of this step are done with the operand copy
      stamp(14);
#pragma unroll
      for (int mp = 0; mp < MPS; mp++)
#pragma unroll
        for (int c = 0; c < 2; c++)
          *reinterpret_cast<u4v*>(h16 + (rr + 32 * c) * HLD + tile_of(mp) * 16 + hh * 8) =
              u4v{pk2(st[mp][c][0][0], st[mp][c][0][1]), pk2(st[mp][c][0][2], st[mp][c][0][3]), pk2(st[mp][c][1][0], st[mp][c][1][1]), pk2(st[mp][c][1][2], st[mp][c][1][3])};
      stamp(15);
      lds_barrier();          // B2
      stamp(16);
    }
    // the final state's bf16 copy
    for (int i = cw * 64 + lane; i < ROWS * (H / 8); i += 256) {
      const int row = i / (H / 8), c8 = (i % (H / 8)) * 8;
      if (r0 + row < R)
        __builtin_nontemporal_store(*reinterpret_cast<const bf16x8*>(h16 + row * HLD + c8),
                                    reinterpret_cast<bf16x8*>(a.HN16 + (long)T * RH + (r0 + row) * H + c8));
    }
  }
  // (rows sorted by length) the panel's dead steps below the launch-wide limit: zero states -- the weight_hh gradient product meets them
  // with exactly-zero gate gradients, and 0 x whatever-the-allocator-left is not 0
  for (int n = T; n < Tg && !a.nofill; n++)
    for (int i = tid; i < ROWS * (H / 8); i += 512) {
      const int row = i / (H / 8), c8 = (i % (H / 8)) * 8;
      if (r0 + row < R) {
        const bf16x8 z8 = bf16x8{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
        *reinterpret_cast<bf16x8*>(a.HN16 + (long)(n + 1) * RH + (r0 + row) * H + c8) = z8;
      }
    }
}

template <int DF, int ABL>
static int launch(const Args& a, hipStream_t s) {
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(notes_fwd_kernel<DF, ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES) != hipSuccess) return PTV_ERR_LAUNCH;
    attr = true;
  }
  hipLaunchKernelGGL((notes_fwd_kernel<DF, ABL>), dim3((unsigned)((a.R + ROWS - 1) / ROWS)), dim3(512), LDS_BYTES, s, a);
  return PTV_OK;
}

}  // namespace nr
}  // namespace ptv

using namespace ptv;

static unsigned long long* g_notes_trace = nullptr;
// timing experiments (round-5 probe, retired): device buffer of 8 x 2048 uint64 that workgroup 8 of the next launches fills with event stamps
extern "C" int ptv_debug_notes_trace(void* buf) { g_notes_trace = (unsigned long long*)buf; return PTV_OK; }

extern "C" int ptv_notes_gru_persist_fwd_top(const void* wg_h, const void* wg_t, const float* b_hh, const void* gc, const float* emb,
                                             const float* h0, void* HN16, void* gates, long R, int T, const int* live_top, void* stream);
extern "C" int ptv_notes_gru_persist_fwd(const void* wg_h, const void* wg_t, const float* b_hh, const void* gc, const float* emb,
                                       const float* h0, void* HN16, void* gates, long R, int T, void* stream) {
  return ptv_notes_gru_persist_fwd_top(wg_h, wg_t, b_hh, gc, emb, h0, HN16, gates, R, T, nullptr, stream);
}

// T: bits 0-7 = steps, bits 8-15 = debug flags (8: no stagger), bits 16-23 = ring depth (0 = default), bit 24 = leave the HN16 slots of a
// panel's dead steps unwritten (rows variant: the caller's products clip to the same segments)
extern "C" int ptv_notes_gru_persist_fwd_rows(const void* wg_h, const void* wg_t, const float* b_hh, const void* gc, const float* emb,
                                              const float* h0, void* HN16, void* gates, long R, int T, const int* live_top, const int* row_len,
                                              void* stream);
extern "C" int ptv_notes_gru_persist_fwd_top(const void* wg_h, const void* wg_t, const float* b_hh, const void* gc, const float* emb,
                                             const float* h0, void* HN16, void* gates, long R, int T, const int* live_top, void* stream) {
  return ptv_notes_gru_persist_fwd_rows(wg_h, wg_t, b_hh, gc, emb, h0, HN16, gates, R, T, live_top, nullptr, stream);
}
extern "C" int ptv_notes_gru_persist_fwd_rows(const void* wg_h, const void* wg_t, const float* b_hh, const void* gc, const float* emb,
                                              const float* h0, void* HN16, void* gates, long R, int T, const int* live_top, const int* row_len,
                                              void* stream) {
  if (!wg_h || !wg_t || !b_hh || !gc || !emb || !h0 || !HN16 || R <= 0 || (T & 0xff) <= 0) return PTV_ERR_ARG;
  nr::Args a{(const bf16x8*)wg_h, (const bf16x8*)wg_t, b_hh, (const __bf16*)gc, emb, R * nr::E, h0, (__bf16*)HN16, (__bf16*)gates,
             (int)R, T & 0xff, (T >> 8) & 0xff, g_notes_trace, live_top, row_len, (T >> 24) & 1};
  const int depth = (T >> 16) & 0xff, abl = a.dbg & 7;
  const int pi = prof::want(3, (int)R, nr::H) ? prof::begin((hipStream_t)stream) : -1;
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if (abl == 1) rc = nr::launch<15, 1>(a, s);
  else if (abl == 2) rc = nr::launch<15, 2>(a, s);
  else if (abl == 3) rc = nr::launch<15, 3>(a, s);
  else if (abl == 4) rc = nr::launch<15, 4>(a, s);
  else if (abl == 7) rc = nr::launch<15, 7>(a, s);
  else if (depth == 12) rc = nr::launch<12, 0>(a, s);
  else if (depth == 30) rc = nr::launch<30, 0>(a, s);
  else rc = nr::launch<15, 0>(a, s);
  if (rc != PTV_OK) return rc;
  if (pi >= 0) prof::end(pi, (hipStream_t)stream, 2.0 * R * 3.0 * nr::H * (nr::H + nr::E) * (T & 0xff));
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

// =============================================================================================================================
// BPTT of the notes GRU, round 5: 8 waves per 64-row workgroup (two per SIMD), every wave owns 64 output units for both phases of a step
//   products  dh = dgh_{s+1} . W_hh (K = 1536): the weight stream L2 -> registers (a ring of 3 k-blocks x 4 tiles); the A operand -- the gate
//             gradients the workgroup itself produced one step earlier -- comes out of LDS for its first 1152 k (144 KB, K-blocked:
//             chunk of 8 k x 64 rows) and out of a K-blocked global scratch for the rest (48 KB per step: LDS holds 160 KB)
//   cells     in the MFMA lane layout (pair-interleaved tiles: 8 consecutive units of a row per lane): saved gates (unit-blocked by 16),
//             arriving gradient (blocked by 32), previous state (the bf16 copy) requested one item ahead; the carry dh (x) z lives in
//             REGISTERS (64 per lane) -- it took 128 KB of LDS in the 4-wave kernel, which is what makes room for the A operand
// The 4-wave kernel (notes_persist.hip) re-read the whole A operand from L2 in every wave (768 KB per CU and step next to the 1.5 MB of
// weights) and ran one wave per SIMD; products and cells cannot overlap across the step boundary (every product of step s - 1 needs every
// gate gradient of step s), so the step is the sum of a weight-bound and an HBM-bound phase: what the second wave per SIMD and the
// LDS-resident operand buy is a cleaner version of each.
// =============================================================================================================================
namespace ptv {
namespace nb {

constexpr int H = 512, ROWS = 64, KT = 3 * H / 32, KL = 36;           // k-blocks: all / held in LDS
constexpr int LDS_A = KL * 4 * ROWS * 16;                             // 147456 bytes
constexpr int SCR_CH = (KT - KL) * 4;                                 // chunks per step that go through the global scratch: 48
constexpr int DW = 3;                                                  // weight ring depth in k-blocks

typedef __attribute__((ext_vector_type(4))) float f4v;
typedef __attribute__((ext_vector_type(4))) unsigned u4v;

struct Args {
  const bf16x8* wt;                // pair-interleaved packing of W_hh^T: [32 tiles of output units][48 kb][64]
  const __bf16* HN16;              // [T + 1][R][512] bf16 states
  const __bf16* gates;             // [T][4][32][R][16]
  const __bf16* ext;               // [16][T * R][32]: gradient arriving at the state after step s, column-blocked by 32
  __bf16* dgi; __bf16* dgh;        // [T][R][1536], [T][R][512] (the n third)
  float* dh0;                      // [R][512] or null
  __bf16* scratch;                 // [grid][2][48][64][8]
  int* top_step;
  int R, T, skip;
  const int* bound;                // or null: device int, NO gradient arrives after note step *bound (the caller knows: the forward stopped there /
                                   // the loss says so) and the consumers of dgi / dgh stop at top_step <= *bound -- steps beyond it are not touched
  const int* row_len;              // or null (needs bound): [R] live note steps per row, rows sorted by DESCENDING length: no gradient arrives at
                                   // this panel after step row_len[first row] - 1 -- `ext` is not read there (it may be unwritten), the steps
                                   // between that and *bound get zero rows of dgi / dgh (their consumers know only the launch-wide limit)
  int nofill;                      // (T bit 16) ... unless the caller's consumers clip to the same 128-row segments (ptv_wgrad_job.seg_n,
                                   // ptv_sum_steps_seg, ptv_gemm_mtop_seg): then those rows stay unwritten
};

__device__ __forceinline__ float bfv(const u4v& v, int e) { const unsigned w = v[e >> 1]; return __uint_as_float((e & 1) ? (w & 0xffff0000u) : (w << 16)); }
__device__ __forceinline__ unsigned pk2b(float a, float b) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  bf16x2 v; v[0] = (__bf16)a; v[1] = (__bf16)b;
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ u4v pk8(const float (&v)[8]) { return u4v{pk2b(v[0], v[1]), pk2b(v[2], v[3]), pk2b(v[4], v[5]), pk2b(v[6], v[7])}; }

template <int ABL>                 // timing experiments (scripts/bench_notes.py): 2 = no products, 4 = no cell loads / stores
__global__ __launch_bounds__(512, 2) void notes_bwd_kernel(Args a) {
  __builtin_amdgcn_s_setprio(3);
  extern __shared__ __attribute__((aligned(16))) char bsm[];           // A operand chunks 0 .. 143: (chunk * 64 + row) * 16 bytes
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rl = lane & 15, q = lane >> 4;                              // fragment / epilogue coordinates: row in tile, k quad = unit octet
  const long R = a.R, RH = R * H, R3H = 3 * RH;
  const long r0 = (long)blockIdx.x * ROWS;
  const int T = a.T;
  __bf16* sc = a.scratch + (long)blockIdx.x * 2 * (SCR_CH * ROWS * 8);
  long grow[4]; bool ok[4];
#pragma unroll
  for (int i = 0; i < 4; i++) { ok[i] = r0 + i * 16 + rl < R; grow[i] = min(r0 + i * 16 + rl, R - 1); }

  // ---- zero-skip: steps at which no gradient arrives for any of the 64 rows, with nothing arriving from later steps either, produce exact
  // zeros (the loss ignores the padded note slots, ptvae.py:498-511): tested on the arriving gradient itself, panel by panel
  // (with a caller-given bound the search starts there: until round 6 every panel read 64 KB and wrote 256 KB of zeros for each of the 8 dead
  // steps of the benchmark batch -- 0.13 GB read, 0.52 GB written per launch for rows nobody reads, a third of the launch's time)
  int s_top = T - 1;
  if (a.skip && a.bound) s_top = min(T - 1, max(*a.bound, -1));
  const int s_panel = (a.skip && a.bound && a.row_len) ? min(s_top, a.row_len[r0 & ~127L] - 1) : s_top;     // (sorted rows: the last live step of the panel's 128-row block)
  for (; a.skip && s_top >= 0; s_top--) {
    unsigned nz = 0;
    if (s_top <= s_panel)
    for (int i = tid; i < ROWS * (H / 8); i += 512) {
      const int row = i / (H / 8), c8 = (i % (H / 8)) * 8;
      if (r0 + row < R) {
        const u4v w = *reinterpret_cast<const u4v*>(a.ext + ((long)(c8 >> 5) * ((long)T * R) + (long)s_top * R + r0 + row) * 32 + (c8 & 31));
        nz |= (w[0] | w[1] | w[2] | w[3]) & 0x7fff7fffu;                // -0.0 is zero too
      }
    }
    if (__syncthreads_or(nz != 0)) break;
    if (a.nofill && s_top > s_panel) continue;                    // (block-uniform) a dead block of a segment-aware caller: nobody reads these rows
    for (int i = tid; i < ROWS * (H / 8); i += 512) {
      const int row = i / (H / 8), c8 = (i % (H / 8)) * 8;
      if (r0 + row < R) {
        const u4v zz = {0u, 0u, 0u, 0u};
        __bf16* pi = a.dgi + (long)s_top * R3H + (r0 + row) * (3 * H) + c8;
        *reinterpret_cast<u4v*>(pi) = zz; *reinterpret_cast<u4v*>(pi + H) = zz; *reinterpret_cast<u4v*>(pi + 2 * H) = zz;
        *reinterpret_cast<u4v*>(a.dgh + (long)s_top * RH + (r0 + row) * H + c8) = zz;
      }
    }
  }
  if (a.top_step && tid == 0 && s_top >= 0) atomicMax(a.top_step, s_top);    // consumers of dgi / dgh may stop after this step's rows

  // ONE lane offset per M tile and operand family (the unit-tile part of every address is a scalar offset of the buffer instruction)
  unsigned g_off[4], e_off[4], h_off[4], d_off[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    g_off[i] = (unsigned)(grow[i] * 32 + (q & 1) * 16 + (long)(q >> 1) * R * 32);      // unit-blocked by 16: [u / 16][R][16] bf16
    e_off[i] = (unsigned)(grow[i] * 64 + q * 16);                                       // column-blocked by 32
    h_off[i] = (unsigned)(grow[i] * (H * 2) + q * 16);                                  // row-major [R][512] bf16 (HN16, dgh)
    d_off[i] = (unsigned)(grow[i] * (3 * H * 2) + q * 16);                              // row-major [R][1536] bf16 (dgi)
  }
  // acc[M tile][unit tile]: the accumulators of the products START each step at the carry dh (x) z the cells left in them (same lane
  // layout: a cell's carry is its own accumulator element) -- d = carry + dgh . W_hh comes out of the MFMAs, and the carry costs no
  // registers of its own (it took 128 KB of LDS in the 4-wave kernel, 64 registers in the first version of this one: 118 spills)
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int lds_w = ((wave * 8 + q) * ROWS + rl) * 16;                  // this lane's place in the A operand: chunk wave * 8 + q, row rl
  // this wave's 4 tiles of W_hh^T through a buffer descriptor: tile j, k-block kb at byte (j * KT + kb) * 1024 + lane * 16 -- a scalar
  // offset per fragment and ONE lane register (flat addresses: hipcc hoists the 192 lane addresses of a step out of the step loop, 280 spills)
  const __amdgpu_buffer_rsrc_t rs_w = nr::rsrc(a.wt + (long)(wave * 4) * KT * 64, 4L * KT * 1024);
  const unsigned lane16 = (unsigned)lane * 16u;
  const unsigned plane = (unsigned)(RH * 2);
  bool first = true;                                                     // no later step has handed a dgh over yet

  for (int s = s_top; s >= (a.dh0 ? -1 : 0); s--) {
    const __bf16* scr = sc + ((s + 1) & 1) * (SCR_CH * ROWS * 8);        // dgh_{s+1} chunks 144 .. 191, written one iteration ago
    __bf16* scw = sc + (s & 1) * (SCR_CH * ROWS * 8);
    // (the operands of the first two cell items are requested inside the LAST two k-blocks of the products, as the weight ring drains: in
    // front of the products they would be 48 more live registers for the whole k-loop -- 484 spills)
    struct Ops { u4v g[4]; u4v ex; u4v hp; };
    Ops ops[2];                                                          // (a ring of 3 -- two items ahead -- spills: 866 vs 810 us)
    const __amdgpu_buffer_rsrc_t rs_g = nr::rsrc(a.gates + (long)(s < 0 ? 0 : s) * 4 * RH, 4L * RH * 2);
    const __amdgpu_buffer_rsrc_t rs_e = nr::rsrc(a.ext, (long)T * RH * 2);
    const __amdgpu_buffer_rsrc_t rs_h = nr::rsrc(a.HN16 + (long)(s < 0 ? 0 : s) * RH, RH * 2);
    auto ldops = [&](Ops& o, int it) {
      const int pr = it >> 2, i = it & 3;
      if constexpr (ABL & 4) return;
      const unsigned ut = (unsigned)(wave * 4 + pr * 2);                  // first unit tile of the item (scalar)
#pragma unroll
      for (int p = 0; p < 4; p++) o.g[p] = __builtin_amdgcn_raw_buffer_load_b128(rs_g, g_off[i], (unsigned)p * plane + ut * (unsigned)(R * 32), 2);
      o.ex = __builtin_amdgcn_raw_buffer_load_b128(rs_e, e_off[i], (unsigned)(((long)(ut >> 1) * ((long)T * R) + (long)s * R) * 64), 2);
      o.hp = __builtin_amdgcn_raw_buffer_load_b128(rs_h, h_off[i], ut * 32u, 0);
    };
    if ((first || (ABL & 2)) && s >= 0) { ldops(ops[0], 0); ldops(ops[1], 1); }
    const __amdgpu_buffer_rsrc_t rs_scr = nr::rsrc(scr, (long)SCR_CH * ROWS * 16), rs_scw = nr::rsrc(scw, (long)SCR_CH * ROWS * 16);
    const __amdgpu_buffer_rsrc_t rs_di = nr::rsrc(a.dgi + (long)(s < 0 ? 0 : s) * R3H, R3H * 2), rs_dh = nr::rsrc(a.dgh + (long)(s < 0 ? 0 : s) * RH, RH * 2);
    if (!first && !(ABL & 2)) {
      // ---- dh = dgh_{s+1} . W_hh: 48 k-blocks, this wave's 4 unit tiles x 4 M tiles
      bf16x8 bw[DW][4], aw[2][4];
      auto ldw = [&](int kb) {
#pragma unroll
        for (int j = 0; j < 4; j++) bw[kb % DW][j] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, lane16, (unsigned)((j * KT + kb) * 1024), 0));
      };
      auto lda_ = [&](int kb) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
          if (kb < KL) aw[kb & 1][i] = *reinterpret_cast<const bf16x8*>(bsm + (((kb * 4 + q) * ROWS) + i * 16 + rl) * 16);
          else aw[kb & 1][i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_scr, (unsigned)((q * ROWS + i * 16 + rl) * 16), (unsigned)((kb - KL) * 4 * ROWS * 16), 0));
        }
      };
#pragma unroll
      for (int kb = 0; kb < DW - 1; kb++) ldw(kb);
      lda_(0);
#pragma unroll
      for (int kb = 0; kb < KT; kb++) {
        if (kb + DW - 1 < KT) ldw(kb + DW - 1);
        if (kb + 1 < KT) lda_(kb + 1);
        if (kb == KT - 2 && s >= 0) ldops(ops[0], 0);
        if (kb == KT - 1 && s >= 0) ldops(ops[1], 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
          for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bw[kb % DW][j], aw[kb & 1][i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    lds_barrier();                                                        // every wave is done with dgh_{s+1} in LDS: the cells may overwrite it
    first = false;
    // ---- cells: 8 items of 16 rows x 32 units, MFMA lane layout
#pragma unroll
    for (int it = 0; it < 8; it++) {
      const int pr = it >> 2, i = it & 3;
      const int u = (wave * 4 + pr * 2) * 16 + q * 8;
      if (s < 0) {                                                        // dh0 = carry + dgh_0 . W_hh
        if (ok[i]) {
          float* p = a.dh0 + grow[i] * H + u;
          *reinterpret_cast<f32x4*>(p) = acc[i][2 * pr];
          *reinterpret_cast<f32x4*>(p + 4) = acc[i][2 * pr + 1];
        }
        continue;
      }
      Ops& o = ops[it & 1];
      // two halves of 4 units (half the live values: the wave carries 128 registers of accumulators and carry through this loop)
      u4v pr_, pz_, pn_, pq_;
#pragma unroll
      for (int hf = 0; hf < 2; hf++) {
        float dr[4], dzz[4], dn[4], dnr[4];
        f4v cz;
#pragma unroll
        for (int e4 = 0; e4 < 4; e4++) {
          const int e = 4 * hf + e4;
          const float gr = bfv(o.g[0], e), gz = bfv(o.g[1], e), gn = bfv(o.g[2], e), gh = bfv(o.g[3], e), hp = bfv(o.hp, e);
          const float d = acc[i][2 * pr + hf][e4] + bfv(o.ex, e);
          dn[e4] = d * (1.0f - gz) * (1.0f - gn * gn);
          dzz[e4] = d * (hp - gn) * gz * (1.0f - gz);
          dr[e4] = dn[e4] * gh * gr * (1.0f - gr);
          dnr[e4] = dn[e4] * gr;
          cz[e4] = d * gz;
        }
        asm volatile("" : "+v"(cz));                                     // (4-register tuples: scalar pieces fragment the register file)
        acc[i][2 * pr + hf] = f32x4{cz[0], cz[1], cz[2], cz[3]};
        pr_[2 * hf] = pk2b(dr[0], dr[1]); pr_[2 * hf + 1] = pk2b(dr[2], dr[3]);
        pz_[2 * hf] = pk2b(dzz[0], dzz[1]); pz_[2 * hf + 1] = pk2b(dzz[2], dzz[3]);
        pn_[2 * hf] = pk2b(dn[0], dn[1]); pn_[2 * hf + 1] = pk2b(dn[2], dn[3]);
        pq_[2 * hf] = pk2b(dnr[0], dnr[1]); pq_[2 * hf + 1] = pk2b(dnr[2], dnr[3]);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (it + 2 < 8) ldops(ops[it & 1], it + 2);                          // (this item's operands are consumed: request the one after next)
      // the next step's A operand: chunk (gate * 512 + u) / 8 of this row -- LDS for chunks 0 .. 143, the scratch beyond
      // (chunk = wave * 8 + pr * 4 + q, row = i * 16 + rl: ONE lane address per gate, the item's part is an immediate; the dn (x) r chunks of
      // waves 2 .. 7 lie beyond the LDS-resident 144)
      char* la = bsm + lds_w + (pr * 4 * ROWS + i * 16) * 16;
      *reinterpret_cast<u4v*>(la) = pr_;
      *reinterpret_cast<u4v*>(la + 64 * ROWS * 16) = pz_;
      if (wave < 2) *reinterpret_cast<u4v*>(la + 128 * ROWS * 16) = pq_;
      else __builtin_amdgcn_raw_buffer_store_b128(pq_, rs_scw, (unsigned)((q * ROWS + rl) * 16), (unsigned)(((wave * 8 + pr * 4 - 16) * ROWS + i * 16) * 16), 0);
      if (ok[i] && !(ABL & 4)) {
        const unsigned so = (unsigned)((wave * 4 + pr * 2) * 32);
        __builtin_amdgcn_raw_buffer_store_b128(pr_, rs_di, d_off[i], so, 2);
        __builtin_amdgcn_raw_buffer_store_b128(pz_, rs_di, d_off[i], so + H * 2, 2);
        __builtin_amdgcn_raw_buffer_store_b128(pn_, rs_di, d_off[i], so + 2 * H * 2, 2);
        __builtin_amdgcn_raw_buffer_store_b128(pq_, rs_dh, h_off[i], so, 2);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();                                                      // dgh_s complete (LDS and scratch) before the next products
  }
}

}  // namespace nb
}  // namespace ptv

extern "C" int ptv_notes_bwd8(const void* wt, const void* HN16, const void* gates, const void* ext, void* dgi, void* dgh, float* dh0, void* scratch,
                              long R, int T, const int* bound, const int* row_len, int* top_step, void* stream) {
  if (!wt || !HN16 || !gates || !ext || !dgi || !dgh || !scratch || R <= 0 || (T & 0xff) <= 0) return PTV_ERR_ARG;
  if ((bound && !top_step) || (row_len && !bound)) return PTV_ERR_ARG;   // (rows beyond the bound stay unwritten: the consumers need the limit)
  nb::Args a{(const bf16x8*)wt, (const __bf16*)HN16, (const __bf16*)gates, (const __bf16*)ext, (__bf16*)dgi, (__bf16*)dgh, dh0, (__bf16*)scratch,
             top_step, (int)R, T & 0xff, g_zero_skip, bound, row_len, (T >> 16) & 1};
  const int abl = (T >> 8) & 6;
  const int pi = prof::want(4, (int)R, 512) ? prof::begin((hipStream_t)stream) : -1;
  const dim3 grid((unsigned)((R + nb::ROWS - 1) / nb::ROWS));
#define NB_LAUNCH(A_)                                                                                                                   \
  do {                                                                                                                                  \
    static bool attr = false;                                                                                                           \
    if (!attr) {                                                                                                                        \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(nb::notes_bwd_kernel<A_>), hipFuncAttributeMaxDynamicSharedMemorySize, nb::LDS_A) != hipSuccess) return PTV_ERR_LAUNCH; \
      attr = true;                                                                                                                      \
    }                                                                                                                                   \
    hipLaunchKernelGGL(nb::notes_bwd_kernel<A_>, grid, dim3(512), nb::LDS_A, (hipStream_t)stream, a);                                   \
  } while (0)
  if (abl == 2) NB_LAUNCH(2); else if (abl == 4) NB_LAUNCH(4); else if (abl == 6) NB_LAUNCH(6); else NB_LAUNCH(0);
#undef NB_LAUNCH
  if (pi >= 0) prof::end(pi, (hipStream_t)stream, 2.0 * R * 3.0 * 512 * 512 * ((T & 0xff) - 1 + (dh0 ? 1 : 0)));
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

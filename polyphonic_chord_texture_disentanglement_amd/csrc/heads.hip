// heads.hip -- the per-note heads of the PianoTree decoder fused into one kernel per direction (round 4).
//
// Reference (ptvae.py:343-352, decode_note): est_pitch = pitch_out_linear(note_summary);  dur_hid = dur_hid_linear([note_summary |
// est_pitch]) -- for all 15*32*B note summaries at once in the teacher-forced restructuring.  As three ptv_gemm launches
// the forward read the [M, 512] bf16 summaries twice and the fresh [M, 130] logits once more (160 + 117 + 84 us at B = 512, all
// HBM-bound with N <= 130), plus a 30-us cast of the initial duration state; the backward (dP += dHD0 . W_dh[:, 512:],
// dNSUM = dHD0 . W_dh[:, :512] + dP . W_p) took 109 + 65 + 179 us re-reading dP and dHD0.  Here a workgroup owns 128 rows:
//   forward   acc[rows][130 | 64] over K = 512 (weights staged through LDS once per workgroup, 52 KB per 128-k chunk, double
//             buffered; the summaries stream straight into A fragments), the logits leave as fp32 AND stay in wave-private LDS as
//             the bf16 A operand of the second product (K = 130) -- one pass over the summaries, the logits never read back
//   backward  dP' = dP + dHD0 . W_dh[:, 512:] in accumulators (written back: the weight-gradient products read it), then
//             dNSUM = [dP' | dHD0] . [W_p ; W_dh[:, :512]] for 4 x 128 output columns with the weight panels staged through LDS;
//             dNSUM leaves as bf16, row-major or column-blocked by 32 (what the row-partitioned BPTT reads, notes_persist.hip);
//             rows from (*m_top + 1) * m_unit on are known to be zero (ptv_last_nonzero_unit): their workgroups write zeros
// MFMA operands bf16, accumulation fp32 -- the arithmetic of the ptv_gemm calls this replaces (bf16 precision mode only).
// Geometry of init_model(): Hn = 512, 130 pitch classes (rows of 136 floats), Hd = 64.
#include "common.hpp"
#include "gemm_core.hpp"
#include "../../include/ptvae_hip.h"

namespace ptv {

constexpr int HHN = 512, HNP = 130, HHD = 64;
constexpr int HPT = 9;                         // pitch tiles (144 columns, 130 live)
constexpr int HDT = 4;                         // duration-state tiles
constexpr int HNT = HPT + HDT;                 // 13 output tiles of the first product
constexpr int HKB = HHN / 32;                  // 16 k-blocks
constexpr int HCH = 4;                         // k-blocks per staged weight chunk
constexpr int HPK = 5;                         // k-blocks of the logits as an operand (130 -> 160)
constexpr int HPLD = 168;                      // LDS row stride of the staged logits (bf16)

struct HeadsFwdArgs {
  const __bf16* hn;                            // [M][512] bf16
  const bf16x8 *wp, *wdh, *wdp;                // packed (ptv_pack_mfma_b): W_p [9][16][64], W_dh[:, :512] [4][16][64], W_dh[:, 512:] [4][5][64]
  const float *b_p, *b_dh;
  float* pitch; long ldp;                      // [M][ldp] fp32
  float* hd0; __bf16* hd16;                    // [M][64] fp32, bf16 copy or null
  long M;
  const int* m_top; long m_unit;               // or null: only the rows below (*m_top + 1) * m_unit are wanted; the others stay unwritten
  const int* row_len;                          // or null (needs m_top): [m_unit] live note steps per row, rows of a step sorted by descending length
};

__global__ __launch_bounds__(256, 1) void heads_fwd_kernel(HeadsFwdArgs a) {
  if (a.m_top && (long)blockIdx.x * 128 >= (long)(max(*a.m_top, 0) + 1) * a.m_unit) return;
  if (a.row_len) {
    // rows sorted by length: a block whose FIRST row has no target at this note step holds only dead rows.  Its logits are an operand of a
    // weight-gradient product that knows only the launch-wide limit: zero rows (finite), nothing else is written
    const long rb = (long)blockIdx.x * 128, n = rb / a.m_unit, r = rb % a.m_unit;
    if (a.row_len[r] <= n) {
      for (int i = threadIdx.x; i < 128 * (int)(a.ldp / 4); i += 256) {
        const long row = rb + i / (a.ldp / 4);
        if (row < a.M) reinterpret_cast<float4*>(a.pitch + row * a.ldp)[i % (a.ldp / 4)] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      return;
    }
  }
  extern __shared__ __attribute__((aligned(16))) char hsm[];
  bf16x8* Bs = reinterpret_cast<bf16x8*>(hsm);                       // [2][HCH][HNT][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rl = lane & 15, kq = lane >> 4;
  const long r0 = (long)blockIdx.x * 128 + wave * 32;
  long row[2];
#pragma unroll
  for (int i = 0; i < 2; i++) row[i] = min(r0 + i * 16 + rl, a.M - 1);

  // weight chunk c (k-blocks 4c .. 4c+3 of all 13 tiles): 13 x 16 bytes per thread
  bf16x8 nb[HNT];
  auto fetch = [&](int c) {
#pragma unroll
    for (int i = 0; i < HNT; i++) {
      const int p = tid + 256 * i, f = p >> 6, ln = p & 63, kbi = f / HNT, tile = f - kbi * HNT;
      nb[i] = tile < HPT ? a.wp[((long)tile * HKB + c * HCH + kbi) * 64 + ln] : a.wdh[((long)(tile - HPT) * HKB + c * HCH + kbi) * 64 + ln];
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int i = 0; i < HNT; i++) Bs[(long)buf * (HCH * HNT * 64) + tid + 256 * i] = nb[i];
  };
  // A fragments of a whole chunk (2 M tiles x 4 k-blocks), requested one chunk ahead: the summaries are the HBM stream of this kernel
  bf16x8 an[2][HCH], ac[2][HCH];
  auto fetch_a = [&](int c) {
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int k = 0; k < HCH; k++)
        an[i][k] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(a.hn + row[i] * HHN + (c * HCH + k) * 32 + kq * 8));
  };

  f32x4 acc[2][HNT];
#pragma unroll
  for (int j = 0; j < HNT; j++) {
    // accumulators start at the biases: lane (row rl, quad kq) holds columns 16 j + 4 kq .. + 3
    f32x4 b = f32x4{0.f, 0.f, 0.f, 0.f};
    if (j < HPT) {
#pragma unroll
      for (int e = 0; e < 4; e++) { const int c = j * 16 + kq * 4 + e; b[e] = c < HNP ? a.b_p[c] : 0.f; }
    } else {
      const float4 t = *reinterpret_cast<const float4*>(a.b_dh + (j - HPT) * 16 + kq * 4);
      b = f32x4{t.x, t.y, t.z, t.w};
    }
    acc[0][j] = b; acc[1][j] = b;
  }
  fetch(0);
  fetch_a(0);
  stash(0);
  __syncthreads();
  for (int c = 0; c < HKB / HCH; c++) {
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int k = 0; k < HCH; k++) ac[i][k] = an[i][k];
    if (c + 1 < HKB / HCH) { fetch(c + 1); fetch_a(c + 1); }
    const bf16x8* bs = Bs + (long)(c & 1) * (HCH * HNT * 64);
#pragma unroll
    for (int k = 0; k < HCH; k++) {
#pragma unroll
      for (int j = 0; j < HNT; j++) {
        const bf16x8 b = bs[(k * HNT + j) * 64 + lane];
        acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, ac[0][k], acc[0][j], 0, 0, 0);
        acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, ac[1][k], acc[1][j], 0, 0, 0);
      }
    }
    if (c + 1 < HKB / HCH) stash((c + 1) & 1);
    __syncthreads();
  }
  // ---- logits: fp32 out, bf16 copy into this wave's LDS rows (the weight buffers are free: every wave passed the last barrier)
  __bf16* ps = reinterpret_cast<__bf16*>(hsm) + wave * (32 * HPLD);
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const long gr = r0 + i * 16 + rl;
    const bool ok = gr < a.M;
#pragma unroll
    for (int j = 0; j < HPT; j++) {
      const int c = j * 16 + kq * 4;
      const f32x4 v = acc[i][j];
      bf16x4 w;
#pragma unroll
      for (int e = 0; e < 4; e++) w[e] = (__bf16)(c + e < HNP ? v[e] : 0.f);
      *reinterpret_cast<bf16x4*>(ps + (i * 16 + rl) * HPLD + c) = w;
      if (ok) {
        float* pp = a.pitch + gr * a.ldp + c;
        if (c + 4 <= HNP) *reinterpret_cast<float4*>(pp) = make_float4(v[0], v[1], v[2], v[3]);
        else if (c < HNP) { pp[0] = v[0]; if (c + 1 < HNP) pp[1] = v[1]; }
      }
    }
    // columns 144 .. 159 of the operand: zero
    *reinterpret_cast<bf16x4*>(ps + (i * 16 + rl) * HPLD + HPT * 16 + kq * 4) = bf16x4{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  // ---- second product: duration state += logits . W_dh[:, 512:]^T
#pragma unroll
  for (int k = 0; k < HPK; k++) {
    bf16x8 b2[HDT];
#pragma unroll
    for (int j = 0; j < HDT; j++) b2[j] = a.wdp[((long)j * HPK + k) * 64 + lane];
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const bf16x8 av = *reinterpret_cast<const bf16x8*>(ps + (i * 16 + rl) * HPLD + k * 32 + kq * 8);
#pragma unroll
      for (int j = 0; j < HDT; j++) acc[i][HPT + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b2[j], av, acc[i][HPT + j], 0, 0, 0);
    }
  }
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const long gr = r0 + i * 16 + rl;
    if (gr >= a.M) continue;
#pragma unroll
    for (int j = 0; j < HDT; j++) {
      const f32x4 v = acc[i][HPT + j];
      const long o = gr * HHD + j * 16 + kq * 4;
      *reinterpret_cast<float4*>(a.hd0 + o) = make_float4(v[0], v[1], v[2], v[3]);
      if (a.hd16) *reinterpret_cast<bf16x4*>(a.hd16 + o) = bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    }
  }
}

// =============================================================================================
// backward
// =============================================================================================
constexpr int HBK = HPK + 2;                   // k-blocks of [dP' (160) | dHD0 (64)]
constexpr int HGT = 8;                         // unit tiles per output group (128 columns)

struct HeadsBwdArgs {
  float* dp; long ldp;                         // [M][ldp] fp32, in/out
  const float* dhd0;                           // [M][64] fp32
  const bf16x8 *wdpT;                          // W_dh[:, 512:]^T packed [9][2][64]   (N = 130 -> 144, K = 64)
  const bf16x8 *wcat;                          // [W_p ; W_dh[:, :512]]^T packed, PAIR-interleaved [32][7][64]: k-blocks 0-4 = the 130 (-> 160) logit rows, 5-6 = the 64 duration rows
  __bf16* dnsum; int blocked;                  // [M][512] bf16, or column-blocked by 32: [16][M][32]
  __bf16* dy16;                                // [M][200] bf16 = [dP' (130) | 0 (6) | dHD0 (64)]: ONE operand for both heads' weight gradients, or null
  const int* m_top; long m_unit;               // rows from (*m_top + 1) * m_unit on are zero (or null)
  long M;
  const int* row_len;                          // or null (needs m_top): rows of a step sorted by descending length, [m_unit] live note steps per row
};

__global__ __launch_bounds__(256, 1) void heads_bwd_kernel(HeadsBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) char hsm[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int rl = lane & 15, kq = lane >> 4;
  const long rb = (long)blockIdx.x * 128;
  const long r0 = rb + wave * 32;
  long live = a.M;
  if (a.m_top) live = min(a.M, ((long)*a.m_top + 1) * a.m_unit);
  if (a.row_len && rb < live) {
    // rows sorted by length: a block whose first row has no target at this note step received nothing.  Its dNSUM rows stay unwritten (the
    // BPTT has the same row lengths as its per-panel bound); its rows of dy16 -- an operand of the heads' weight-gradient product, which
    // knows only the launch-wide limit -- are written as zeros
    const long n = rb / a.m_unit, r = rb % a.m_unit;
    if (a.row_len[r] <= n) {
      if (a.dy16) {
        const bf16x8 z = bf16x8{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
        for (int i = tid; i < 128 * 25; i += 256) {
          const long row = rb + i / 25;
          if (row < a.M) *reinterpret_cast<bf16x8*>(a.dy16 + row * 200 + (i % 25) * 8) = z;
        }
      }
      return;
    }
  }
  if (rb >= live) {                                                       // nothing arrived at these rows: dNSUM = 0, dP stays (zero)
    if (a.blocked & 2) return;                                            // ... and the consumer knows the limit too: the rows stay unwritten
    const bf16x8 z = bf16x8{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
    for (int i = tid; i < 128 * (HHN / 8); i += 256) {
      if (a.blocked & 1) {
        const int blk = i / (128 * 4), r = (i / 4) % 128, q = i % 4;
        if (rb + r < a.M) *reinterpret_cast<bf16x8*>(a.dnsum + ((long)blk * a.M + rb + r) * 32 + q * 8) = z;
      } else {
        const int r = i / (HHN / 8), c8 = (i % (HHN / 8)) * 8;
        if (rb + r < a.M) *reinterpret_cast<bf16x8*>(a.dnsum + (rb + r) * HHN + c8) = z;
      }
    }
    return;
  }
  bf16x8* Bs = reinterpret_cast<bf16x8*>(hsm);                           // [2][HGT][HBK][64] weight panels of an output group (2 x 56 KB)
  __bf16* ps = reinterpret_cast<__bf16*>(hsm + 2 * HGT * HBK * 64 * 16) + wave * (32 * HPLD);     // this wave's [32][168] bf16 rows of dP'
  long row[2];
#pragma unroll
  for (int i = 0; i < 2; i++) row[i] = min(r0 + i * 16 + rl, a.M - 1);

  // weight panel of group 0 on its way while the first product runs
  bf16x8 nb[14];
  auto fetch = [&](int g) {
#pragma unroll
    for (int i = 0; i < 14; i++) { const int p = tid + 256 * i; nb[i] = a.wcat[(long)g * (HGT * HBK * 64) + p]; }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 14; i++) Bs[(long)buf * (HGT * HBK * 64) + tid + 256 * i] = nb[i];
  };
  fetch(0);
  // dHD0 rows as A fragments (K = 64: two k-blocks), kept for both products.  Rows at or beyond `live` inside a partly live
  // workgroup count as zero (their producers passed over them)
  bf16x8 ad[2][2];
  bool rlive[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    rlive[i] = r0 + i * 16 + rl < live;
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const float* p = a.dhd0 + row[i] * HHD + k * 32 + kq * 8;
      float4 lo = *reinterpret_cast<const float4*>(p), hi = *reinterpret_cast<const float4*>(p + 4);
      if (!rlive[i]) lo = hi = make_float4(0.f, 0.f, 0.f, 0.f);
      ad[i][k] = bf16x8{(__bf16)lo.x, (__bf16)lo.y, (__bf16)lo.z, (__bf16)lo.w, (__bf16)hi.x, (__bf16)hi.y, (__bf16)hi.z, (__bf16)hi.w};
    }
  }
  if (a.dy16) {
#pragma unroll
    for (int i = 0; i < 2; i++)
      if (r0 + i * 16 + rl < a.M)
#pragma unroll
        for (int k = 0; k < 2; k++) *reinterpret_cast<bf16x8*>(a.dy16 + (r0 + i * 16 + rl) * 200 + 136 + k * 32 + kq * 8) = ad[i][k];
  }
  // ---- dP' = dP + dHD0 . W_dh[:, 512:]   (accumulators start at dP: lane (row rl, quad kq) holds columns 16 j + 4 kq .. + 3)
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const long gr = r0 + i * 16 + rl;
    const bool ok = gr < a.M;
    f32x4 dpv[HPT];
#pragma unroll
    for (int j = 0; j < HPT; j++) {
      const int c = j * 16 + kq * 4;
      const float* pp = a.dp + row[i] * a.ldp + c;
      f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
      if (c + 4 <= HNP) { const float4 t = *reinterpret_cast<const float4*>(pp); v = f32x4{t.x, t.y, t.z, t.w}; }
      else if (c < HNP) { v[0] = pp[0]; if (c + 1 < HNP) v[1] = pp[1]; }
      if (!rlive[i]) v = f32x4{0.f, 0.f, 0.f, 0.f};
      dpv[j] = v;
    }
#pragma unroll
    for (int k = 0; k < 2; k++)
#pragma unroll
      for (int j = 0; j < HPT; j++)
        dpv[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.wdpT[((long)j * 2 + k) * 64 + lane], ad[i][k], dpv[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < HPT; j++) {
      const int c = j * 16 + kq * 4;
      const f32x4 v = dpv[j];
      bf16x4 w;
#pragma unroll
      for (int e = 0; e < 4; e++) w[e] = (__bf16)(c + e < HNP ? v[e] : 0.f);
      *reinterpret_cast<bf16x4*>(ps + (i * 16 + rl) * HPLD + c) = w;
      if (ok && a.dy16 && c < 136) *reinterpret_cast<bf16x4*>(a.dy16 + gr * 200 + c) = w;       // (columns 130 .. 135: zeros)
      if (ok && rlive[i]) {
        float* pp = a.dp + gr * a.ldp + c;
        if (c + 4 <= HNP) *reinterpret_cast<float4*>(pp) = make_float4(v[0], v[1], v[2], v[3]);
        else if (c < HNP) { pp[0] = v[0]; if (c + 1 < HNP) pp[1] = v[1]; }
      }
    }
    *reinterpret_cast<bf16x4*>(ps + (i * 16 + rl) * HPLD + HPT * 16 + kq * 4) = bf16x4{(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
  }
  stash(0);
  __syncthreads();                                                        // (also orders this wave's staged rows before its reads)
  // A fragments of the second product: dP' (5 k-blocks, from LDS) and dHD0 (2 k-blocks, registers)
  bf16x8 ap[2][HPK];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int k = 0; k < HPK; k++) ap[i][k] = *reinterpret_cast<const bf16x8*>(ps + (i * 16 + rl) * HPLD + k * 32 + kq * 8);
  // ---- dNSUM: four groups of 128 columns
  for (int g = 0; g < 4; g++) {
    if (g + 1 < 4) fetch(g + 1);
    const bf16x8* bs = Bs + (long)(g & 1) * (HGT * HBK * 64);
    f32x4 acc[2][HGT];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < HGT; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < HBK; k++) {
#pragma unroll
      for (int j = 0; j < HGT; j++) {
        const bf16x8 b = bs[(j * HBK + k) * 64 + lane];
#pragma unroll
        for (int i = 0; i < 2; i++)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, k < HPK ? ap[i][k] : ad[i][k - HPK], acc[i][j], 0, 0, 0);
      }
    }
    // pair-interleaved tiles: lane (row rl, quad kq) holds units 32 (j / 2) + 8 kq .. + 7 of the group in acc[.][j], acc[.][j + 1]
#pragma unroll
    for (int i = 0; i < 2; i++) {
      const long gr = r0 + i * 16 + rl;
      if (gr < a.M) {
#pragma unroll
        for (int j = 0; j < HGT; j += 2) {
          const f32x4 lo = acc[i][j], hi = acc[i][j + 1];
          const bf16x8 o = bf16x8{(__bf16)lo[0], (__bf16)lo[1], (__bf16)lo[2], (__bf16)lo[3], (__bf16)hi[0], (__bf16)hi[1], (__bf16)hi[2], (__bf16)hi[3]};
          const int ub = g * 4 + (j >> 1);                                 // 32-unit block
          if (a.blocked & 1) *reinterpret_cast<bf16x8*>(a.dnsum + ((long)ub * a.M + gr) * 32 + kq * 8) = o;
          else *reinterpret_cast<bf16x8*>(a.dnsum + gr * HHN + ub * 32 + kq * 8) = o;
        }
      }
    }
    if (g + 1 < 4) stash((g + 1) & 1);
    __syncthreads();
  }
}

}  // namespace ptv

using namespace ptv;

extern "C" int ptv_heads_fwd_top(const void* hn16, const void* wp_packed, const void* wdh_packed, const void* wdp_packed, const float* b_p,
                                 const float* b_dh, float* pitch, long ldp, float* hd0, void* hd16, long M, const int* m_top, long m_unit,
                                 void* stream);
extern "C" int ptv_heads_fwd(const void* hn16, const void* wp_packed, const void* wdh_packed, const void* wdp_packed, const float* b_p,
                             const float* b_dh, float* pitch, long ldp, float* hd0, void* hd16, long M, void* stream) {
  return ptv_heads_fwd_top(hn16, wp_packed, wdh_packed, wdp_packed, b_p, b_dh, pitch, ldp, hd0, hd16, M, nullptr, 0, stream);
}

extern "C" int ptv_heads_fwd_rows(const void* hn16, const void* wp_packed, const void* wdh_packed, const void* wdp_packed, const float* b_p,
                                  const float* b_dh, float* pitch, long ldp, float* hd0, void* hd16, long M, const int* m_top, long m_unit,
                                  const int* row_len, void* stream);
extern "C" int ptv_heads_fwd_top(const void* hn16, const void* wp_packed, const void* wdh_packed, const void* wdp_packed, const float* b_p,
                                 const float* b_dh, float* pitch, long ldp, float* hd0, void* hd16, long M, const int* m_top, long m_unit,
                                 void* stream) {
  return ptv_heads_fwd_rows(hn16, wp_packed, wdh_packed, wdp_packed, b_p, b_dh, pitch, ldp, hd0, hd16, M, m_top, m_unit, nullptr, stream);
}
extern "C" int ptv_heads_fwd_rows(const void* hn16, const void* wp_packed, const void* wdh_packed, const void* wdp_packed, const float* b_p,
                                  const float* b_dh, float* pitch, long ldp, float* hd0, void* hd16, long M, const int* m_top, long m_unit,
                                  const int* row_len, void* stream) {
  if (row_len && !m_top) return PTV_ERR_ARG;
  if (m_top && (m_unit <= 0 || (m_unit & 127))) return PTV_ERR_ARG;        // (whole 128-row blocks on either side of the limit)
  if (!hn16 || !wp_packed || !wdh_packed || !wdp_packed || !b_p || !b_dh || !pitch || !hd0 || M <= 0 || ldp < HNP || (ldp & 3)) return PTV_ERR_ARG;
  if ((reinterpret_cast<uintptr_t>(pitch) & 15) || (reinterpret_cast<uintptr_t>(hn16) & 15)) return PTV_ERR_ARG;
  HeadsFwdArgs a{(const __bf16*)hn16, (const bf16x8*)wp_packed, (const bf16x8*)wdh_packed, (const bf16x8*)wdp_packed, b_p, b_dh,
                 pitch, ldp, hd0, (__bf16*)hd16, M, m_top, m_unit, row_len};
  const int lds = 2 * HCH * HNT * 64 * 16;                                // 104 KB (the staged logits, 42 KB, reuse it)
  static bool attr = false;
  if (!attr) { if (hipFuncSetAttribute((const void*)heads_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return PTV_ERR_LAUNCH; attr = true; }
  hipLaunchKernelGGL(heads_fwd_kernel, dim3((unsigned)((M + 127) / 128)), dim3(256), lds, (hipStream_t)stream, a);
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

extern "C" int ptv_heads_bwd_rows(float* dp, long ldp, const float* dhd0, const void* wdpT_packed, const void* wcat_packed, void* dnsum16,
                                  int blocked, void* dy16, const int* m_top, long m_unit, const int* row_len, long M, void* stream);
extern "C" int ptv_heads_bwd(float* dp, long ldp, const float* dhd0, const void* wdpT_packed, const void* wcat_packed, void* dnsum16,
                             int blocked, void* dy16, const int* m_top, long m_unit, long M, void* stream) {
  return ptv_heads_bwd_rows(dp, ldp, dhd0, wdpT_packed, wcat_packed, dnsum16, blocked, dy16, m_top, m_unit, nullptr, M, stream);
}
extern "C" int ptv_heads_bwd_rows(float* dp, long ldp, const float* dhd0, const void* wdpT_packed, const void* wcat_packed, void* dnsum16,
                                  int blocked, void* dy16, const int* m_top, long m_unit, const int* row_len, long M, void* stream) {
  if (row_len && (!m_top || (m_unit & 127))) return PTV_ERR_ARG;
  if (!dp || !dhd0 || !wdpT_packed || !wcat_packed || !dnsum16 || M <= 0 || ldp < HNP || (ldp & 3) || (m_top && m_unit <= 0)) return PTV_ERR_ARG;
  if (reinterpret_cast<uintptr_t>(dp) & 15) return PTV_ERR_ARG;
  HeadsBwdArgs a{dp, ldp, dhd0, (const bf16x8*)wdpT_packed, (const bf16x8*)wcat_packed, (__bf16*)dnsum16, blocked, (__bf16*)dy16, m_top, m_unit, M, row_len};
  const int lds = 2 * HGT * HBK * 64 * 16 + 4 * 32 * HPLD * 2;            // 112 KB + 42 KB
  static bool attr = false;
  if (!attr) { if (hipFuncSetAttribute((const void*)heads_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return PTV_ERR_LAUNCH; attr = true; }
  hipLaunchKernelGGL(heads_bwd_kernel, dim3((unsigned)((M + 127) / 128)), dim3(256), lds, (hipStream_t)stream, a);
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

// dur_bwd.hip -- backward of the 5-step duration GRU (dur.hip's forward; ptvae.py:353-367) as ONE kernel.
//
// The generic path runs 5 BPTT step kernels (each streams dgh, dhz and the gate planes through HBM), a
// split-K product for dW_hh over 5*M rows, and column sums for the bias and token gradients.  Here a wave
// owns 16 rows for all 5 steps, walking them backwards:
//   * dh stays in registers (fp32, MFMA C layout: lane = row, 16 units); per step only the saved gate
//     planes and h_{d} are read -- the operands of step d-1 are requested before step d is computed
//   * dgh_d . W_hh (the carry into step d-1) is 24 v_mfma_f32_16x16x32_bf16 per wave against W_hh^T
//     resident in LDS
//   * the parameter gradients are accumulated in-kernel: the block's 64 rows of [dr dz dnr dn] and of
//     [h_d | onehot(token class)] go to LDS TRANSPOSED (K = rows), and  [256 x 64rows] . [64rows x 80]
//     per step lands in 80 accumulator registers per wave that live for the whole kernel:
//         columns 0..63  -> dW_hh (rows dr, dz, dnr)
//         columns 64..66 -> per-token-class sums of the gate gradients (class 0 = <sos> step, 1/2 = one-hot
//                           token 0/1): bias gradients and the input-weight / <sos>-token gradients
//     each block writes one [256 x 80] partial; a column-sum over blocks and ptv_dur_bwd_finalize fold them
//     into the parameter gradients.
// Nothing but dh0 [M, 64] and the per-block partials is written.  bf16 precision, H = 64 only.
//
// RC (recompute, round 4): the forward no longer writes the four gate planes of its 5 steps (629 MB at B = 512, two thirds of
// everything it wrote); the backward rebuilds r, z, n, hn of step d from h_{d-1} -- which it reads anyway -- with the forward's own
// 24 MFMAs per wave (same operands, same order, same fp32 expressions: the values are the forward's, before their bf16 rounding) against
// W_hh resident in LDS.  Saved planes are still accepted (gates != null: the free-running note loop writes them).
#include "common.hpp"
#include "gemm_core.hpp"
#include "../../include/ptvae_hip.h"

namespace ptv {

constexpr int BH = 64;                  // hidden size
constexpr int KLD = 80;                 // LDS row stride (bf16) of the K = 64-rows operands: 160 B, conflict-free b128 reads
constexpr int WLD = 208;                // LDS row stride (bf16) of the K = 192 operands: 416 B
constexpr int PART_COLS = 80;           // 64 h units + 3 token classes + zero pad
constexpr int PART = 256 * PART_COLS;   // floats per block partial

struct DurBwdArgs {
  const __bf16* gates; long plane_g, step_g;     // gate plane p of step d at gates + d*step_g + p*plane_g (+ row*64 + unit); null: recompute
  const float* b_hh; const float* tab0; const float* tab;   // recompute mode: [192], [192] = W_i sos + b_i, [2][192] = W_i onehot + b_i
  const void* hall; long plane_h; int h_bf16;    // h_d at hall + d*plane_h (+ row*64 + unit), d = 0..4; fp32 or bf16
  const float* ddur; long ld_dd;                 // [M, 10]: d loss / d est_dur
  const float* w_hh; const float* w_out;         // [192, 64], [2, 64]
  const int* idx; long idx_stride;               // idx[d*idx_stride + row]: token fed to step d+1
  float* dh0;                                    // [M, 64]
  float* part;                                   // [gridDim.x][PART]
  long M;
  int skip;                                      // pass over tiles that receive no gradient
};

struct DurOps { bf16x4 g[4][4]; float4 hp[4]; };   // [plane][fragment]

template <bool HB, bool RC>
__device__ __forceinline__ void dur_load(const DurBwdArgs& a, int d, long row, int ug, DurOps& o) {
  if constexpr (!RC) {
    const __bf16* gp = a.gates + d * a.step_g + row * BH + ug;
#pragma unroll
    for (int p = 0; p < 4; p++)
#pragma unroll
      for (int f = 0; f < 4; f++) o.g[p][f] = *reinterpret_cast<const bf16x4*>(gp + p * a.plane_g + f * 16);
  }
  if constexpr (HB) {
    const __bf16* hp = reinterpret_cast<const __bf16*>(a.hall) + d * a.plane_h + row * BH + ug;
#pragma unroll
    for (int f = 0; f < 4; f++) {
      const bf16x4 v = *reinterpret_cast<const bf16x4*>(hp + f * 16);
      o.hp[f] = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
    }
  } else {
    const float* hp = reinterpret_cast<const float*>(a.hall) + d * a.plane_h + row * BH + ug;
#pragma unroll
    for (int f = 0; f < 4; f++) o.hp[f] = *reinterpret_cast<const float4*>(hp + f * 16);
  }
}

template <bool HB, bool RC>
__global__ __launch_bounds__(256, 1) void dur_gru_bwd_kernel(DurBwdArgs a) {
  __builtin_amdgcn_s_setprio(3);                                         // a launch of the latency chain: wins instruction issue against sibling-stream products
  __shared__ __attribute__((aligned(16))) __bf16 WT[BH * WLD];          // W_hh^T: WT[unit][gate-unit]
  __shared__ __attribute__((aligned(16))) __bf16 DG[4][16 * WLD];       // per-wave dgh rows [16][192]
  __shared__ __attribute__((aligned(16))) __bf16 AT[256 * KLD];         // [dr dz dnr dn][block row]
  __shared__ __attribute__((aligned(16))) __bf16 HT[PART_COLS * KLD];   // [h unit | class | 0][block row]
  __shared__ float wo[2 * BH];
  __shared__ __attribute__((aligned(16))) __bf16 Wn[RC ? 3 * BH * KLD : 8];   // recompute: W_hh [gate unit][k], the forward's operand
  __shared__ __attribute__((aligned(16))) __bf16 HP[RC ? 4 * 16 * KLD : 8];   //            per-wave h_{d-1} tile
  __shared__ float tabs[RC ? 3 : 1][RC ? 3 * BH : 1];
  __shared__ float bh[RC ? 3 * BH : 1];
  if constexpr (RC) {
    for (int i = threadIdx.x; i < 3 * BH * BH; i += 256) Wn[(i / BH) * KLD + (i % BH)] = (__bf16)a.w_hh[i];
    for (int i = threadIdx.x; i < 3 * BH; i += 256) {
      tabs[0][i] = a.tab0[i]; tabs[1][i] = a.tab[i]; tabs[2][i] = a.tab[3 * BH + i];
      bh[i] = a.b_hh[i];
    }
  }
  for (int i = threadIdx.x; i < 3 * BH * BH; i += 256) WT[(i % BH) * WLD + (i / BH)] = (__bf16)a.w_hh[i];
  for (int i = threadIdx.x; i < (PART_COLS - BH) * KLD; i += 256) HT[BH * KLD + i] = (__bf16)0.f;
  for (int i = threadIdx.x; i < 2 * BH; i += 256) wo[i] = a.w_out[i];
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rl = lane & 15, kq = lane >> 4, ug = kq * 4;
  const int brow = wave * 16 + rl;                                     // row within the block's 64
  __bf16* dg = DG[wave];
  f32x4 C[4][5];                                                       // gate-unit tiles 4*wave..+3  x  column tiles 0..4
#pragma unroll
  for (int t = 0; t < 4; t++)
#pragma unroll
    for (int hh = 0; hh < 5; hh++) C[t][hh] = f32x4{0.f, 0.f, 0.f, 0.f};

  const long tiles = (a.M + 63) / 64;
  for (long tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const long row = tile * 64 + brow;
    const bool ok = row < a.M;
    const long rowc = ok ? row : a.M - 1;                              // clamped: loads stay in bounds
    float dd[10];
#pragma unroll
    for (int i = 0; i < 5; i++) {
      const float2 v = *reinterpret_cast<const float2*>(a.ddur + rowc * a.ld_dd + 2 * i);
      dd[2 * i] = ok ? v.x : 0.f; dd[2 * i + 1] = ok ? v.y : 0.f;
    }
    // a tile whose 64 rows receive no gradient (padded note slots: CrossEntropyLoss(ignore_index=2), ptvae.py:505-510 -- the late
    // note steps of every row, about half of all tiles on this data) has exactly zero state and parameter gradients
    {
      bool nzr = false;
#pragma unroll
      for (int i = 0; i < 10; i++) nzr |= dd[i] != 0.f;
      if (a.skip && !__syncthreads_or(nzr)) {
        if (ok) {
#pragma unroll
          for (int f = 0; f < 4; f++) *reinterpret_cast<float4*>(a.dh0 + row * BH + f * 16 + ug) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        continue;
      }
    }
    int cls[5];
    cls[0] = 0;
#pragma unroll
    for (int d = 1; d < 5; d++) cls[d] = 1 + a.idx[(d - 1) * a.idx_stride + rowc];
    float carry[4][4];
#pragma unroll
    for (int f = 0; f < 4; f++)
#pragma unroll
      for (int e = 0; e < 4; e++) carry[f][e] = 0.f;
    DurOps ops[2];
    dur_load<HB, RC>(a, 4, rowc, ug, ops[0]);
#pragma unroll
    for (int d = 4; d >= 0; d--) {
      const DurOps& o = ops[(4 - d) & 1];
      if (d > 0) dur_load<HB, RC>(a, d - 1, rowc, ug, ops[(5 - d) & 1]);       // next step's operands in flight under this one
      f32x4 gh[RC ? 12 : 1];
      if constexpr (RC) {
        // the forward's recurrent product again: gh = h_{d-1} . W_hh^T (dur.hip: same fragments, same k order)
        __bf16* hpt = HP + wave * 16 * KLD;
#pragma unroll
        for (int f = 0; f < 4; f++) {
          bf16x4 p;
          p[0] = (__bf16)o.hp[f].x; p[1] = (__bf16)o.hp[f].y; p[2] = (__bf16)o.hp[f].z; p[3] = (__bf16)o.hp[f].w;
          *reinterpret_cast<bf16x4*>(hpt + rl * KLD + f * 16 + ug) = p;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int n = 0; n < 12; n++) gh[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < BH; ks += 32) {
          const bf16x8 hb = *reinterpret_cast<const bf16x8*>(hpt + rl * KLD + ks + kq * 8);
#pragma unroll
          for (int n = 0; n < 12; n++) {
            const bf16x8 wb = *reinterpret_cast<const bf16x8*>(Wn + (n * 16 + rl) * KLD + ks + kq * 8);
            gh[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb, hb, gh[n], 0, 0, 0);
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
      const float* gi = tabs[RC ? cls[d] : 0];
      float dhz[4][4];
#pragma unroll
      for (int f = 0; f < 4; f++) {
        const float hp[4] = {o.hp[f].x, o.hp[f].y, o.hp[f].z, o.hp[f].w};
        bf16x4 pr, pz, pq;
#pragma unroll
        for (int e = 0; e < 4; e++) {
          const int u = f * 16 + ug + e;
          float dh = carry[f][e] + dd[2 * d] * wo[u] + dd[2 * d + 1] * wo[BH + u];
          if (!ok) dh = 0.f;
          float r, z, n, hn;
          if constexpr (RC) {
            r = sigmoid_fast(gi[u] + gh[f][e] + bh[u]);
            z = sigmoid_fast(gi[BH + u] + gh[4 + f][e] + bh[BH + u]);
            hn = gh[8 + f][e] + bh[2 * BH + u];
            n = tanh_fast(gi[2 * BH + u] + r * hn);
          } else {
            r = (float)o.g[0][f][e]; z = (float)o.g[1][f][e]; n = (float)o.g[2][f][e]; hn = (float)o.g[3][f][e];
          }
          const float dn = dh * (1.0f - z) * (1.0f - n * n);
          const float dz = dh * (hp[e] - n) * z * (1.0f - z);
          const float dr = dn * hn * r * (1.0f - r);
          const float dnr = dn * r;
          dhz[f][e] = dh * z;
          pr[e] = (__bf16)dr; pz[e] = (__bf16)dz; pq[e] = (__bf16)dnr;
          // transposed copies (K = block rows) for the parameter-gradient products
          AT[(0 * BH + u) * KLD + brow] = pr[e];
          AT[(1 * BH + u) * KLD + brow] = pz[e];
          AT[(2 * BH + u) * KLD + brow] = pq[e];
          AT[(3 * BH + u) * KLD + brow] = (__bf16)dn;
          HT[u * KLD + brow] = (__bf16)(ok ? hp[e] : 0.f);
        }
        *reinterpret_cast<bf16x4*>(dg + rl * WLD + 0 * BH + f * 16 + ug) = pr;
        *reinterpret_cast<bf16x4*>(dg + rl * WLD + 1 * BH + f * 16 + ug) = pz;
        *reinterpret_cast<bf16x4*>(dg + rl * WLD + 2 * BH + f * 16 + ug) = pq;
      }
      if (kq == 0) {
#pragma unroll
        for (int k = 0; k < 3; k++) HT[(BH + k) * KLD + brow] = (__bf16)((ok && cls[d] == k) ? 1.0f : 0.0f);
      }
      __builtin_amdgcn_wave_barrier();
      // carry into step d-1: dgh_d . W_hh  (K = 192 gate units), + dh (x) z
      f32x4 acc[4];
#pragma unroll
      for (int n = 0; n < 4; n++) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 3 * BH; ks += 32) {
        const bf16x8 hb = *reinterpret_cast<const bf16x8*>(dg + rl * WLD + ks + kq * 8);
#pragma unroll
        for (int n = 0; n < 4; n++) {
          const bf16x8 wb = *reinterpret_cast<const bf16x8*>(WT + (n * 16 + rl) * WLD + ks + kq * 8);
          acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb, hb, acc[n], 0, 0, 0);
        }
      }
#pragma unroll
      for (int f = 0; f < 4; f++)
#pragma unroll
        for (int e = 0; e < 4; e++) carry[f][e] = acc[f][e] + dhz[f][e];
      __syncthreads();                                                  // the block's AT / HT are complete
#pragma unroll
      for (int ks = 0; ks < 64; ks += 32) {
        bf16x8 at[4];
#pragma unroll
        for (int t = 0; t < 4; t++) at[t] = *reinterpret_cast<const bf16x8*>(AT + ((wave * 4 + t) * 16 + rl) * KLD + ks + kq * 8);
#pragma unroll
        for (int hh = 0; hh < 5; hh++) {
          const bf16x8 hb = *reinterpret_cast<const bf16x8*>(HT + (hh * 16 + rl) * KLD + ks + kq * 8);
#pragma unroll
          for (int t = 0; t < 4; t++) C[t][hh] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(hb, at[t], C[t][hh], 0, 0, 0);
        }
      }
      __syncthreads();                                                  // AT / HT free for the next step
    }
    if (ok) {
#pragma unroll
      for (int f = 0; f < 4; f++)
        *reinterpret_cast<float4*>(a.dh0 + row * BH + f * 16 + ug) = make_float4(carry[f][0], carry[f][1], carry[f][2], carry[f][3]);
    }
  }
  // block partial: P[gate-unit][col], gate-unit = (4*wave + t)*16 + (lane & 15), col = hh*16 + (lane >> 4)*4 + e
  float* P = a.part + (long)blockIdx.x * PART;
#pragma unroll
  for (int t = 0; t < 4; t++)
#pragma unroll
    for (int hh = 0; hh < 5; hh++)
      *reinterpret_cast<float4*>(P + ((wave * 4 + t) * 16 + rl) * PART_COLS + hh * 16 + ug) = make_float4(C[t][hh][0], C[t][hh][1], C[t][hh][2], C[t][hh][3]);
}

// S [256 x 80] = column-summed block partials -> parameter gradients (all accumulate)
//   T[j][c] = class sums of dgi (rows dr, dz, dn):  j < 128: S[j][64+c];  j >= 128: S[64+j][64+c]
__global__ void dur_bwd_finalize_kernel(const float* __restrict__ S, float* g_whh, float* g_bhh, float* g_bih, float* g_wih,
                                        float* g_sos, const float* __restrict__ w_ih, const float* __restrict__ sos, int I) {
  __shared__ float red[8][192];
  const int j = threadIdx.x;                                           // 192 threads: one gate unit each
  for (int k = 0; k < BH; k++) g_whh[j * BH + k] += S[j * PART_COLS + k];
  g_bhh[j] += S[j * PART_COLS + 64] + S[j * PART_COLS + 65] + S[j * PART_COLS + 66];
  const float* t = S + (j < 128 ? j : 64 + j) * PART_COLS + 64;
  const float t0 = t[0], t1 = t[1], t2 = t[2];
  g_bih[j] += t0 + t1 + t2;
  for (int k = 0; k < I; k++) g_wih[j * I + k] += t0 * sos[k] + (k == 0 ? t1 : 0.f) + (k == 1 ? t2 : 0.f);
  for (int k = 0; k < I && k < 8; k++) red[k][j] = t0 * w_ih[j * I + k];
  __syncthreads();
  if (j < I && j < 8) {
    float s = 0.f;
    for (int q = 0; q < 192; q++) s += red[j][q];
    g_sos[j] += s;
  }
}

}  // namespace ptv

using namespace ptv;

extern "C" int ptv_dur_gru_bwd_part_size(void) { return PART; }

extern "C" int ptv_dur_gru_bwd(int H, long M, const void* gates, long plane_g, long step_g, const void* hall, long plane_h, int h_bf16,
                               const float* ddur, long ld_dd, const float* w_hh, const float* w_out,
                               const int* idx, long idx_stride, float* dh0, float* part, int nblocks,
                               const float* b_hh, const float* tab0, const float* tab, void* stream) {
  if (H != BH) return PTV_ERR_ARG;
  if (M <= 0 || !hall || !ddur || !w_hh || !w_out || !idx || !dh0 || !part || nblocks <= 0) return PTV_ERR_ARG;
  if (!gates && (!b_hh || !tab0 || !tab)) return PTV_ERR_ARG;            // recompute mode needs what the forward's gates were built from
  if ((plane_g & 3) || (step_g & 3) || (plane_h & 3) || (ld_dd & 1)) return PTV_ERR_ARG;
  DurBwdArgs a{(const __bf16*)gates, plane_g, step_g, b_hh, tab0, tab, hall, plane_h, h_bf16, ddur, ld_dd, w_hh, w_out, idx, idx_stride,
               dh0, part, M, g_zero_skip};
  hipStream_t s = (hipStream_t)stream;
  if (!gates) {
    if (h_bf16) hipLaunchKernelGGL((dur_gru_bwd_kernel<true, true>), dim3(nblocks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((dur_gru_bwd_kernel<false, true>), dim3(nblocks), dim3(256), 0, s, a);
  } else if (h_bf16) hipLaunchKernelGGL((dur_gru_bwd_kernel<true, false>), dim3(nblocks), dim3(256), 0, s, a);
  else hipLaunchKernelGGL((dur_gru_bwd_kernel<false, false>), dim3(nblocks), dim3(256), 0, s, a);
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

extern "C" int ptv_dur_bwd_finalize(const float* S, float* g_whh, float* g_bhh, float* g_bih, float* g_wih, float* g_sos,
                                    const float* w_ih, const float* sos, int I, void* stream) {
  if (!S || !g_whh || !g_bhh || !g_bih || !g_wih || !g_sos || !w_ih || !sos || I < 2 || I > 8) return PTV_ERR_ARG;
  hipLaunchKernelGGL(dur_bwd_finalize_kernel, dim3(1), dim3(192), 0, (hipStream_t)stream, S, g_whh, g_bhh, g_bih, g_wih, g_sos, w_ih, sos, I);
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

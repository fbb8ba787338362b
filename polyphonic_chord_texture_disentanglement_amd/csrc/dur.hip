// dur.hip -- the 5-step duration GRU of PtvaeDecoder.decode_note (ptvae.py:353-367) as ONE kernel.
//
//   token_0 = dur_sos_token;  for d in 0..4:  h = GRU(token_d, h);  est_dur_d = dur_out_linear(h);
//                                             token_{d+1} = onehot5[argmax est_dur_d]
//
// The input token is one of three vectors, so W_i token + b_i is a 3-row table; the recurrence is
// 64-wide.  Per step the generic path launches a GRU-step kernel and a token kernel and round-trips the
// state through HBM; here a wave owns 16 rows for all 5 steps:
//   * W_hh (bf16, 192 x 64) and the gate tables live in LDS for the whole kernel
//   * the hidden state stays in registers (fp32, MFMA C-layout: lane = row, 16 units) and is mirrored as
//     bf16 into a per-wave LDS tile that feeds the next step's MFMA A operand
//   * h . W_hh^T is 24 v_mfma_f32_16x16x32_bf16 per step per wave; the 2-wide output layer and the argmax
//     are a 4-lane shuffle reduction
//   * what the backward needs (states, gate planes) is streamed out once; nothing is read back.
// bf16 precision, H = 64 only (the init_model() configuration); other shapes use the per-step kernels.
#include "common.hpp"
#include "gemm_core.hpp"
#include "../../include/ptvae_hip.h"

namespace ptv {

constexpr int DH = 64;                 // hidden size
constexpr int DLD = DH + 16;           // LDS row stride (bf16 elements): 160 B = 32 mod 64, conflict-free b128 fragment reads

__device__ __forceinline__ void st8f(void* p, long i, bool bf, const float (&v)[8]) {
  if (bf) {
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; e++) o[e] = (__bf16)v[e];
    *reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(p) + i) = o;
  } else {
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(p) + i) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(p) + i + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
}

struct DurArgs {
  const float* h0; long ld_h0;         // [M, 64] initial state (dur_hid_linear output)
  const float* w_hh; const float* b_hh;   // [192, 64], [192]
  const float* tab0; const float* tab;    // [192] = W_i sos + b_i ; [2][192] = W_i onehot(0/1) + b_i
  const float* w_out; const float* b_out; // [2, 64], [2]
  float* hall; long plane_h;           // state after step d at hall + d*plane_h (+ row*64); may be null
  __bf16* hall16;                      // bf16 copy, same indexing; may be null
  void* gates; long plane_g; long step_g; int gates_bf16;   // gate plane p of step d at gates + d*step_g + p*plane_g
  float* dur_out; long ld_out;         // est_dur_d at dur_out[row*ld_out + 2d .. 2d+1]
  int* idx; long idx_stride;           // idx[d*idx_stride + row]
  const int* force; long force_stride; // replay mode (tests), may be null
  long M;
  const int* m_top; long m_unit;       // or null: only the rows below (*m_top + 1) * m_unit are wanted; the others stay unwritten
  const int* row_len;                  // or null (needs m_top): rows of a step sorted by descending length, [m_unit] live note steps per row -- a
                                       // tile inside a 128-row block whose first row has no target at its note step is passed over
};

__global__ __launch_bounds__(256, 2) void dur_gru_fwd_kernel(DurArgs a) {
  __builtin_amdgcn_s_setprio(3);                                         // a launch of the latency chain: wins instruction issue against sibling-stream products
  __shared__ __attribute__((aligned(16))) __bf16 Ws[3 * DH * DLD];       // W_hh as bf16
  __shared__ __attribute__((aligned(16))) __bf16 Hs[4][16 * DLD];        // per-wave state tile
  __shared__ float tabs[3][3 * DH];                                     // gate tables: sos, idx0, idx1
  __shared__ float bh[3 * DH];
  __shared__ float wo[2 * DH + 2];
  // W_hh rows are STORED in the order the MFMA tiles consume them (pair-interleaved units, below): the fragment reads stay on 16
  // consecutive LDS rows (conflict-free); unit u of gate g -> tile f = 2*(u/32) + (u%8)/4, operand row 4*((u%32)/8) + u%4
  for (int i = threadIdx.x; i < 3 * DH * DH; i += 256) {
    const int r = i / DH, g = r / DH, u = r % DH;
    const int lr = g * DH + (2 * (u >> 5) + ((u & 7) >> 2)) * 16 + 4 * ((u & 31) >> 3) + (u & 3);
    Ws[lr * DLD + (i % DH)] = (__bf16)a.w_hh[i];
  }
  for (int i = threadIdx.x; i < 3 * DH; i += 256) {
    tabs[0][i] = a.tab0[i]; tabs[1][i] = a.tab[i]; tabs[2][i] = a.tab[3 * DH + i];
    bh[i] = a.b_hh[i];
  }
  for (int i = threadIdx.x; i < 2 * DH; i += 256) wo[i] = a.w_out[i];
  if (threadIdx.x < 2) wo[2 * DH + threadIdx.x] = a.b_out[threadIdx.x];
  __syncthreads();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rl = lane & 15, q8 = (lane >> 4) * 8;                        // row in tile; first of this lane's 8 units in every tile PAIR
  // Pair-interleaved units: MFMA tile f of a gate is fed the W_hh rows of units 32*(f/2) + 8*(i/4) + 4*(f%2) + i%4 (i = row of the
  // operand tile), so the accumulators h[2m][0..3], h[2m+1][0..3] of a lane are the 8 CONSECUTIVE units 32m + q8 .. +7 of its
  // row: states and gate planes leave in 16-byte pieces (4 lanes = 64 contiguous bytes of a row) instead of 8-byte ones
  __bf16* hs = Hs[wave];
  const long m_live = a.m_top ? min(a.M, (long)(max(*a.m_top, 0) + 1) * a.m_unit) : a.M;
  const long tiles = (m_live + 15) / 16;
  for (long tile = (long)blockIdx.x * 4 + wave; tile < tiles; tile += (long)gridDim.x * 4) {
    if (a.row_len) { const long rb = tile * 16 / 128 * 128; if (a.row_len[rb % a.m_unit] <= rb / a.m_unit) continue; }     // (deadness per 128-row block, as the heads)
    const long row = tile * 16 + rl;
    const bool ok = row < a.M;
    float h[4][4];                                                      // h[f][e]: unit f*16 + ug + e
#pragma unroll
    for (int f = 0; f < 4; f++) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok) v = *reinterpret_cast<const float4*>(a.h0 + row * a.ld_h0 + 32 * (f >> 1) + q8 + 4 * (f & 1));
      h[f][0] = v.x; h[f][1] = v.y; h[f][2] = v.z; h[f][3] = v.w;
    }
    int tok = 0;                                                        // table row: 0 = sos, 1 + idx afterwards
#pragma unroll 1
    for (int d = 0; d < 5; d++) {
      // state -> LDS (bf16) as the MFMA operand
#pragma unroll
      for (int m = 0; m < 2; m++) {
        bf16x8 p;
#pragma unroll
        for (int e = 0; e < 4; e++) { p[e] = (__bf16)h[2 * m][e]; p[4 + e] = (__bf16)h[2 * m + 1][e]; }
        *reinterpret_cast<bf16x8*>(hs + rl * DLD + 32 * m + q8) = p;
      }
      __builtin_amdgcn_wave_barrier();
      f32x4 acc[12];
#pragma unroll
      for (int n = 0; n < 12; n++) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < DH; ks += 32) {
        const bf16x8 hb = *reinterpret_cast<const bf16x8*>(hs + rl * DLD + ks + (lane >> 4) * 8);
#pragma unroll
        for (int n = 0; n < 12; n++) {
          const bf16x8 wb = *reinterpret_cast<const bf16x8*>(Ws + (n * 16 + rl) * DLD + ks + (lane >> 4) * 8);
          acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb, hb, acc[n], 0, 0, 0);
        }
      }
      __builtin_amdgcn_wave_barrier();
      const float* gi = tabs[tok];
      float o0 = 0.f, o1 = 0.f;
#pragma unroll
      for (int m = 0; m < 2; m++) {
        float r[8], z[8], n[8], hn[8], hv[8];
#pragma unroll
        for (int e8 = 0; e8 < 8; e8++) {
          const int f = 2 * m + (e8 >> 2), e = e8 & 3;
          const int j = 32 * m + q8 + e8;
          r[e8] = sigmoid_fast(gi[j] + acc[f][e] + bh[j]);
          z[e8] = sigmoid_fast(gi[DH + j] + acc[4 + f][e] + bh[DH + j]);
          hn[e8] = acc[8 + f][e] + bh[2 * DH + j];
          n[e8] = tanh_fast(gi[2 * DH + j] + r[e8] * hn[e8]);
          h[f][e] = (1.0f - z[e8]) * n[e8] + z[e8] * h[f][e];
          hv[e8] = h[f][e];
          o0 += wo[j] * h[f][e];
          o1 += wo[DH + j] * h[f][e];
        }
        if (ok) {
          const long off = row * DH + 32 * m + q8;
          if (a.hall) {
            *reinterpret_cast<float4*>(a.hall + d * a.plane_h + off) = make_float4(hv[0], hv[1], hv[2], hv[3]);
            *reinterpret_cast<float4*>(a.hall + d * a.plane_h + off + 4) = make_float4(hv[4], hv[5], hv[6], hv[7]);
          }
          if (a.hall16) st8f(a.hall16, d * a.plane_h + off, true, hv);
          if (a.gates) {
            const long g0 = d * a.step_g + off;
            st8f(a.gates, g0 + 0 * a.plane_g, a.gates_bf16, r);
            st8f(a.gates, g0 + 1 * a.plane_g, a.gates_bf16, z);
            st8f(a.gates, g0 + 2 * a.plane_g, a.gates_bf16, n);
            st8f(a.gates, g0 + 3 * a.plane_g, a.gates_bf16, hn);
          }
        }
      }
      // est_dur = dur_out_linear(h): reduce the 4 lanes that share this row (lanes rl, rl+16, rl+32, rl+48)
      o0 += __shfl_xor(o0, 16, 64); o1 += __shfl_xor(o1, 16, 64);
      o0 += __shfl_xor(o0, 32, 64); o1 += __shfl_xor(o1, 32, 64);
      o0 += wo[2 * DH]; o1 += wo[2 * DH + 1];
      int id = o1 > o0 ? 1 : 0;                                           // first max wins ties (torch.max)
      if (a.force && ok) id = a.force[d * a.force_stride + row];
      if (ok && lane < 16) {
        a.dur_out[row * a.ld_out + 2 * d] = o0;
        a.dur_out[row * a.ld_out + 2 * d + 1] = o1;
        a.idx[d * a.idx_stride + row] = id;
      }
      tok = 1 + id;
    }
  }
}

}  // namespace ptv

using namespace ptv;

extern "C" int ptv_dur_gru_fwd_top(int H, long M, const float* h0, long ld_h0, const float* w_hh, const float* b_hh,
                                   const float* tab0, const float* tab, const float* w_out, const float* b_out,
                                   float* hall, long plane_h, void* hall16, void* gates, long plane_g, long step_g, int gates_bf16,
                                   float* dur_out, long ld_out, int* idx, long idx_stride, const int* force, long force_stride,
                                   const int* m_top, long m_unit, void* stream);
extern "C" int ptv_dur_gru_fwd(int H, long M, const float* h0, long ld_h0, const float* w_hh, const float* b_hh,
                               const float* tab0, const float* tab, const float* w_out, const float* b_out,
                               float* hall, long plane_h, void* hall16, void* gates, long plane_g, long step_g, int gates_bf16,
                               float* dur_out, long ld_out, int* idx, long idx_stride, const int* force, long force_stride,
                               void* stream) {
  return ptv_dur_gru_fwd_top(H, M, h0, ld_h0, w_hh, b_hh, tab0, tab, w_out, b_out, hall, plane_h, hall16, gates, plane_g, step_g, gates_bf16,
                             dur_out, ld_out, idx, idx_stride, force, force_stride, nullptr, 0, stream);
}

extern "C" int ptv_dur_gru_fwd_rows(int H, long M, const float* h0, long ld_h0, const float* w_hh, const float* b_hh,
                                    const float* tab0, const float* tab, const float* w_out, const float* b_out,
                                    float* hall, long plane_h, void* hall16, void* gates, long plane_g, long step_g, int gates_bf16,
                                    float* dur_out, long ld_out, int* idx, long idx_stride, const int* force, long force_stride,
                                    const int* m_top, long m_unit, const int* row_len, void* stream);
extern "C" int ptv_dur_gru_fwd_top(int H, long M, const float* h0, long ld_h0, const float* w_hh, const float* b_hh,
                                   const float* tab0, const float* tab, const float* w_out, const float* b_out,
                                   float* hall, long plane_h, void* hall16, void* gates, long plane_g, long step_g, int gates_bf16,
                                   float* dur_out, long ld_out, int* idx, long idx_stride, const int* force, long force_stride,
                                   const int* m_top, long m_unit, void* stream) {
  return ptv_dur_gru_fwd_rows(H, M, h0, ld_h0, w_hh, b_hh, tab0, tab, w_out, b_out, hall, plane_h, hall16, gates, plane_g, step_g, gates_bf16,
                              dur_out, ld_out, idx, idx_stride, force, force_stride, m_top, m_unit, nullptr, stream);
}
extern "C" int ptv_dur_gru_fwd_rows(int H, long M, const float* h0, long ld_h0, const float* w_hh, const float* b_hh,
                                   const float* tab0, const float* tab, const float* w_out, const float* b_out,
                                   float* hall, long plane_h, void* hall16, void* gates, long plane_g, long step_g, int gates_bf16,
                                   float* dur_out, long ld_out, int* idx, long idx_stride, const int* force, long force_stride,
                                   const int* m_top, long m_unit, const int* row_len, void* stream) {
  if (row_len && (!m_top || (m_unit & 127))) return PTV_ERR_ARG;          // (deadness per 128-row block)
  if (m_top && (m_unit <= 0 || (m_unit & 15))) return PTV_ERR_ARG;         // (whole 16-row tiles on either side of the limit)
  if (H != DH) return PTV_ERR_ARG;                 // callers fall back to the per-step kernels for other sizes
  if (M <= 0 || !h0 || !w_hh || !b_hh || !tab0 || !tab || !w_out || !b_out || !dur_out || !idx) return PTV_ERR_ARG;
  if ((ld_h0 & 3) || (plane_h & 7) || (plane_g & 7) || (step_g & 7)) return PTV_ERR_ARG;      // 16-byte bf16 pieces
  DurArgs a{h0, ld_h0, w_hh, b_hh, tab0, tab, w_out, b_out, hall, plane_h, (__bf16*)hall16, gates, plane_g, step_g, gates_bf16,
            dur_out, ld_out, idx, idx_stride, force, force_stride, M, m_top, m_unit, row_len};
  // grid: whole rounds of resident blocks (3 per CU: 44.5 KB of LDS each) -- 1024 blocks on 256 CUs were 1 1/3 rounds, the last one a third full
  const int cap = 3 * num_cus();                               // (512 / 768 / 1024 / 2048 blocks: 247 / 276 / 247 / 251 us -- it does not matter)
  long nb = ((M + 15) / 16 + 3) / 4; if (nb > cap) nb = cap; if (nb < 1) nb = 1;
  hipLaunchKernelGGL(dur_gru_fwd_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, a);
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

// misc.hip -- duration-head token kernel, chord-decoder token kernel, fused clip + Adam.
#include "common.hpp"
#include "../../include/ptvae_hip.h"

namespace ptv {

// est_dur = dur_out_linear(h) (ptvae.py:361-362), next token index = argmax (ptvae.py:365-367).
// 16 lanes per row (float4 each covers H <= 64 ... loops for larger H), 4 rows per wave.
__global__ void dur_out_token_kernel(const float* __restrict__ h, int H, const float* __restrict__ w_out, const float* __restrict__ b_out,
                                     float* __restrict__ dur_out, long ld_out, int* __restrict__ idx, const int* __restrict__ force_idx, long rows) {
  const int sub = threadIdx.x & 15;
  const long rpb = blockDim.x / 16;
  for (long r = (long)blockIdx.x * rpb + threadIdx.x / 16; r < rows; r += (long)gridDim.x * rpb) {
    float s0 = 0.f, s1 = 0.f;
    for (int j = sub; j < H; j += 16) { float v = h[r * H + j]; s0 += v * w_out[j]; s1 += v * w_out[H + j]; }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) { s0 += __shfl_xor(s0, o, 16); s1 += __shfl_xor(s1, o, 16); }
    if (sub == 0) {
      s0 += b_out[0]; s1 += b_out[1];
      dur_out[r * ld_out + 0] = s0; dur_out[r * ld_out + 1] = s1;
      if (idx) idx[r] = force_idx ? force_idx[r] : (s1 > s0 ? 1 : 0);      // first max wins ties (torch.max)
    }
  }
}

// ---------------------------------------------------------------------------------------------
// gradient global norm + clip_grad_norm_(.,clip) + Adam (module.py:142-144, train.py:50) over the
// flat parameter / gradient buffers.  Two launches, no host sync: the norm stays on the device.
// ---------------------------------------------------------------------------------------------
__global__ void sumsq_kernel(const float* __restrict__ g, long n, float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  const long n4 = n / 4;
  const float4* g4 = reinterpret_cast<const float4*>(g);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    float4 v = g4[i]; s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  for (long i = n4 * 4 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) s += g[i] * g[i];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

__global__ void clip_adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n,
                                 const float* __restrict__ sumsq, float gscale, float clip, float lr, float b1, float b2, float eps,
                                 float bc1, float bc2_sqrt) {
  // grads are first scaled by gscale (1/world_size after a sum all-reduce); sumsq is of the UNSCALED buffer
  const float norm = sqrtf(sumsq[0]) * gscale;
  float coef = clip > 0.f ? clip / (norm + 1e-6f) : 1.f;
  coef = fminf(coef, 1.f) * gscale;
  const float step = lr / bc1;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float gi = g[i] * coef;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi; v[i] = vi;
    p[i] -= step * mi / (sqrtf(vi) / bc2_sqrt + eps);
  }
}

}  // namespace ptv

using namespace ptv;

extern "C" int ptv_dur_out_token(const float* h, int H, const float* w_out, const float* b_out, float* dur_out, long ld_out,
                                 int* idx, const int* force_idx, long rows, void* stream) {
  if (!h || !w_out || !b_out || !dur_out || rows <= 0 || H <= 0) return PTV_ERR_ARG;
  long nb = (rows + 15) / 16; if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(dur_out_token_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, h, H, w_out, b_out, dur_out, ld_out, idx, force_idx, rows);
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

extern "C" int ptv_grad_sumsq(const float* g, long n, float* sumsq, void* stream) {
  if (!g || !sumsq || n <= 0) return PTV_ERR_ARG;
  if (reinterpret_cast<uintptr_t>(g) & 15) return PTV_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(sumsq, 0, sizeof(float), s) != hipSuccess) return PTV_ERR_LAUNCH;
  long nb = (n / 4 + 255) / 256; if (nb > 2048) nb = 2048; if (nb < 1) nb = 1;
  hipLaunchKernelGGL(sumsq_kernel, dim3((int)nb), dim3(256), 0, s, g, n, sumsq);
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

extern "C" int ptv_clip_adam_step(float* p, const float* g, float* m, float* v, long n, const float* sumsq, float gscale, float clip,
                                  float lr, float beta1, float beta2, float eps, int step, void* stream) {
  if (!p || !g || !m || !v || !sumsq || n <= 0 || step < 1) return PTV_ERR_ARG;
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  long nb = (n + 255) / 256; if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(clip_adam_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, sumsq, gscale, clip, lr, beta1, beta2, eps,
                     (float)bc1, (float)sqrt(bc2));
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

// loss.hip -- reconstruction / KL / chord losses of DisentangleVAE.loss_function (model.py:57-90,
// ptvae.py:498-511) and their gradients w.r.t. the logits and the posterior parameters.
//   pl = CE(pitch logits, x[:,:,1:,0], ignore 130)      dl = CE(dur logits, x[:,:,1:,1:], ignore 2)
//   kl_x = mean_{B x Z}(-log s + (s^2+m^2)/2 - 1/2)     root/chroma/bass = plain CE
//   loss = w0*pl + w1*dl + beta*(kl_chd+kl_rhy) + root + chroma + bass
// Row layouts: logits may be batch-major [B,32,15,*] (API tensors) or step-major [15,32,B,*]
// (the decoder's internal layout); targets are materialised once per step in the same order.
#include "common.hpp"
#include "../../include/ptvae_hip.h"

namespace ptv {

// sums:   0 pitch nll, 1 dur nll, 2 kl_chd, 3 kl_rhy, 4 root nll, 5 chroma nll, 6 bass nll
// counts: 0 valid pitch targets, 1 valid dur targets

__global__ void pianotree_targets_kernel(const long* __restrict__ x, int B, int step_major,
                                         int* __restrict__ pitch_t, int* __restrict__ dur_t, int* __restrict__ counts, int* __restrict__ row_live) {
  __shared__ int red[3][4];
  const long rows = (long)B * 480;
  int cp = 0, cd = 0, top = 0;                                           // top: the last note step that holds ANY non-ignored target
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < rows; i += (long)gridDim.x * blockDim.x) {
    int b, t, n;
    if (step_major) { b = (int)(i % B); long q = i / B; t = (int)(q % 32); n = (int)(q / 32); }
    else { n = (int)(i % 15); long q = i / 15; t = (int)(q % 32); b = (int)(q / 32); }
    const long* xr = x + (((long)b * 32 + t) * 16 + n + 1) * 6;
    int p = (int)xr[0];
    pitch_t[i] = p;
    cp += (p != 130);
    int live = (p != 130);
#pragma unroll
    for (int d = 0; d < 5; d++) { int v = (int)xr[1 + d]; dur_t[i * 5 + d] = v; cd += (v != 2); live |= (v != 2); }
    if (live) top = max(top, n);
    if (live && row_live) atomicMax(row_live + (long)t * B + b, n + 1);     // row (t, b): its live note steps are 0 .. row_live - 1
  }
  for (int o = 32; o > 0; o >>= 1) { cp += __shfl_xor(cp, o, 64); cd += __shfl_xor(cd, o, 64); top = max(top, __shfl_xor(top, o, 64)); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = cp; red[1][threadIdx.x >> 6] = cd; red[2][threadIdx.x >> 6] = top; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(counts + 0, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
    atomicAdd(counts + 1, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
    // counts[2] (zero-initialised like the others): an UPPER BOUND of the last note step whose logits can receive a gradient from this
    // loss -- the cross-entropy gradient of an ignored row is exactly zero (ptvae.py:505-510).  The decoder's backward takes it as its
    // zero-skip limit instead of scanning the 134-MB gradient (functional.decoder_bwd_core); blocks that cannot raise it skip the atomic
    const int bt = max(max(red[2][0], red[2][1]), max(red[2][2], red[2][3]));
    if (bt > __hip_atomic_load(counts + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(counts + 2, bt);
  }
}

// c [B,8,36] -> root/bass argmax targets [8B], chroma targets [8B*12]   (model.py:72-74)
__global__ void chord_targets_kernel(const float* __restrict__ c, int B, int step_major,
                                     int* __restrict__ root_t, int* __restrict__ chroma_t, int* __restrict__ bass_t) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * 8) return;
  int b, t;
  if (step_major) { b = (int)(i % B); t = (int)(i / B); } else { t = (int)(i % 8); b = (int)(i / 8); }
  const float* cr = c + ((long)b * 8 + t) * 36;
  int ar = 0, ab = 0; float mr = cr[0], mb = cr[24];
  for (int k = 1; k < 12; k++) { if (cr[k] > mr) { mr = cr[k]; ar = k; } if (cr[24 + k] > mb) { mb = cr[24 + k]; ab = k; } }
  root_t[i] = ar; bass_t[i] = ab;
  for (int k = 0; k < 12; k++) chroma_t[i * 12 + k] = (int)cr[12 + k];
}

// one wave per row, C <= 256
template <bool BWD>
__global__ void ce_wave_kernel(const float* __restrict__ logits, long ld, const int* __restrict__ tgt, long rows, int C, int ignore,
                               float* __restrict__ nll_sum, const float* __restrict__ gscale, float* __restrict__ dlogits, long ldd, OrdScratch sc) {
  __shared__ float red[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  float local = 0.f;
  const float gs = BWD ? gscale[0] : 0.f;
  for (long r = (long)blockIdx.x * 4 + w; r < rows; r += (long)gridDim.x * 4) {
    const float* lr = logits + r * ld;
    const int t = tgt[r];
    float v[4]; float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < 4; k++) { int c = lane + 64 * k; v[k] = c < C ? lr[c] : -INFINITY; m = fmaxf(m, v[k]); }
    m = wave_max(m);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 4; k++) { int c = lane + 64 * k; if (c < C) s += expf(v[k] - m); }
    s = wave_sum(s);
    const bool valid = t != ignore;
    if (!BWD) {
      if (valid && lane == 0) local += -(lr[t] - m - logf(s));
    } else {
      float* dr = dlogits + r * ldd;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        int c = lane + 64 * k;
        if (c < C) dr[c] = valid ? gs * (expf(v[k] - m) / s - (c == t ? 1.f : 0.f)) : 0.f;
      }
    }
  }
  if (!BWD) {
    if (lane == 0) red[w] = local;
    __syncthreads();
    if (threadIdx.x == 0) red[0] = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    ordered_commit(nll_sum, red, 1, sc, 0, blockIdx.x, gridDim.x);
  }
}

// rows padded to a multiple of 4 floats (the 130-wide pitch logits live in 136-float rows): half a wave per row, 16-byte
// loads / stores, two rows per wave instruction stream.  C <= 256.  (The scalar kernel above moved 4 bytes per lane.)
template <bool BWD>
__global__ void ce_vec_kernel(const float* __restrict__ logits, long ld, const int* __restrict__ tgt, long rows, int C, int ignore,
                              float* __restrict__ nll_sum, const float* __restrict__ gscale, float* __restrict__ dlogits, long ldd, OrdScratch sc) {
  __builtin_amdgcn_s_setprio(3);                                         // always part of a latency chain
  __shared__ float red[8];
  const int lane = threadIdx.x & 63, sub = lane & 31, hw = threadIdx.x >> 5;       // 8 half-waves per block
  float local = 0.f;
  const float gs = BWD ? gscale[0] : 0.f;
  const int nch = (C + 3) >> 2;
  for (long r0 = (long)blockIdx.x * 8; r0 < rows; r0 += (long)gridDim.x * 8) {
    const long r = r0 + hw;
    const bool live = r < rows;
    const long rc = live ? r : rows - 1;
    const float* lr = logits + rc * ld;
    const int t = tgt[rc];
    // rows whose target is ignore_index (the padded note slots: more than half of the pitch rows) contribute nothing forward and a zero
    // row backward: their logits are never read.  (t is uniform over the half-wave that owns the row; the shuffles stay inside it.)
    if (!live || t == ignore) {
      if (BWD && live) {
        float* dr = dlogits + r * ldd;
#pragma unroll
        for (int k = 0; k < 2; k++) {
          const int j = sub + 32 * k;
          if (j < nch) *reinterpret_cast<float4*>(dr + 4 * j) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
      continue;
    }
    float4 v[2]; float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const int j = sub + 32 * k;
      v[k] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
      if (j < nch) {
        v[k] = *reinterpret_cast<const float4*>(lr + 4 * j);
        if (4 * j + 1 >= C) v[k].y = -INFINITY;
        if (4 * j + 2 >= C) v[k].z = -INFINITY;
        if (4 * j + 3 >= C) v[k].w = -INFINITY;
      }
      m = fmaxf(m, fmaxf(fmaxf(v[k].x, v[k].y), fmaxf(v[k].z, v[k].w)));
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    float4 e[2]; float s = 0.f;
#pragma unroll
    for (int k = 0; k < 2; k++) {
      e[k] = make_float4(expf(v[k].x - m), expf(v[k].y - m), expf(v[k].z - m), expf(v[k].w - m));    // exp(-inf) = 0 on masked slots
      s += e[k].x + e[k].y + e[k].z + e[k].w;
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const bool valid = t != ignore;
    if (!BWD) {
      if (live && valid && sub == 0) local += -(lr[t] - m - logf(s));
    } else if (live) {
      float* dr = dlogits + r * ldd;
      const float inv = gs / s;
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const int j = sub + 32 * k;
        if (j < nch) {
          float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
          if (valid) {
            d = make_float4(e[k].x * inv, e[k].y * inv, e[k].z * inv, e[k].w * inv);
            const int q = t - 4 * j;
            if (q == 0) d.x -= gs; else if (q == 1) d.y -= gs; else if (q == 2) d.z -= gs; else if (q == 3) d.w -= gs;
          }
          *reinterpret_cast<float4*>(dr + 4 * j) = d;            // slots >= C fall into the row padding
        }
      }
    }
  }
  if (!BWD) {
    if (sub == 0) red[hw] = local;
    __syncthreads();
    if (threadIdx.x == 0) { float t = 0.f; for (int i = 0; i < 8; i++) t += red[i]; red[0] = t; }
    __syncthreads();
    ordered_commit(nll_sum, red, 1, sc, 0, blockIdx.x, gridDim.x);
  }
}

static inline bool ce_vec_ok(const float* p, long ld, int C) {
  return C > 16 && (ld & 3) == 0 && ld >= ((C + 3) & ~3) && (reinterpret_cast<uintptr_t>(p) & 15) == 0;
}

// one thread per row, C <= 16 (duration bits C=2, chord heads C=12 / 2)
template <bool BWD>
__global__ void ce_small_kernel(const float* __restrict__ logits, long ld, const int* __restrict__ tgt, long rows, int C, int ignore,
                                float* __restrict__ nll_sum, const float* __restrict__ gscale, float* __restrict__ dlogits, long ldd, OrdScratch sc) {
  __shared__ float red[4];
  float local = 0.f;
  const float gs = BWD ? gscale[0] : 0.f;
  for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (long)gridDim.x * blockDim.x) {
    const float* lr = logits + r * ld;
    const int t = tgt[r];
    float v[16]; float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < 16; k++) { v[k] = k < C ? lr[k] : -INFINITY; m = fmaxf(m, v[k]); }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; k++) if (k < C) s += expf(v[k] - m);
    const bool valid = t != ignore;
    if (!BWD) {
      if (valid) {
        float vt = 0.f;
#pragma unroll
        for (int k = 0; k < 16; k++) if (k == t) vt = v[k];
        local += -(vt - m - logf(s));
      }
    } else {
      float* dr = dlogits + r * ldd;
#pragma unroll
      for (int k = 0; k < 16; k++) if (k < C) dr[k] = valid ? gs * (expf(v[k] - m) / s - (k == t ? 1.f : 0.f)) : 0.f;
    }
  }
  if (!BWD) {
    local = wave_sum(local);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) red[0] = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    ordered_commit(nll_sum, red, 1, sc, 0, blockIdx.x, gridDim.x);
  }
}

// grouped form for the weighted duration loss (ptvae.py:512-527): row r belongs to group r % G (the duration bit position),
// every group is its own CrossEntropyLoss(ignore_index) mean -> per-group nll sums / valid counts, per-group gradient scales
template <bool BWD>
__global__ void ce_group_kernel(const float* __restrict__ logits, long ld, const int* __restrict__ tgt, long rows, int C, int ignore, int G,
                                float* __restrict__ nll_sum, int* __restrict__ cnt, const float* __restrict__ gscale,
                                float* __restrict__ dlogits, long ldd) {
  float local[8]; int lc[8];
#pragma unroll
  for (int g = 0; g < 8; g++) { local[g] = 0.f; lc[g] = 0; }
  for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (long)gridDim.x * blockDim.x) {
    const float* lr = logits + r * ld;
    const int t = tgt[r], grp = (int)(r % G);
    float v[16]; float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < 16; k++) { v[k] = k < C ? lr[k] : -INFINITY; m = fmaxf(m, v[k]); }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; k++) if (k < C) s += expf(v[k] - m);
    const bool valid = t != ignore;
    if (!BWD) {
      if (valid) {
        float vt = 0.f;
#pragma unroll
        for (int k = 0; k < 16; k++) if (k == t) vt = v[k];
        const float nll = -(vt - m - logf(s));
#pragma unroll
        for (int g = 0; g < 8; g++) if (g == grp) { local[g] += nll; lc[g] += 1; }
      }
    } else {
      const float gs = gscale[grp];
      float* dr = dlogits + r * ldd;
#pragma unroll
      for (int k = 0; k < 16; k++) if (k < C) dr[k] = valid ? gs * (expf(v[k] - m) / s - (k == t ? 1.f : 0.f)) : 0.f;
    }
  }
  if (!BWD) {
#pragma unroll
    for (int g = 0; g < 8; g++) {
      if (g < G) {
        const float t = wave_sum(local[g]);
        int c = lc[g];
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
        if ((threadIdx.x & 63) == 0 && c != 0) { atomicAdd(nll_sum + g, t); atomicAdd(cnt + g, c); }
      }
    }
  }
}

// dl = sum_d w[d] * nll[d] / cnt[d] written as (sums1 = dl, counts1 = 1) so that ptv_loss_finalize / ptv_loss_bwd_scales apply
__global__ void wdur_finalize_kernel(const float* __restrict__ gsum, const int* __restrict__ gcnt, float w0, float w1, float w2, float w3,
                                     float w4, float* __restrict__ sums1, int* __restrict__ counts1) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float w[5] = {w0, w1, w2, w3, w4};
  float dl = 0.f;
  for (int d = 0; d < 5; d++) dl += w[d] * (gsum[d] / (float)gcnt[d]);
  sums1[0] = dl; counts1[0] = 1;
}
__global__ void wdur_scales_kernel(const float* __restrict__ gs1, const int* __restrict__ gcnt, float w0, float w1, float w2, float w3,
                                   float w4, float* __restrict__ out5) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  const float w[5] = {w0, w1, w2, w3, w4};
  for (int d = 0; d < 5; d++) out5[d] = gs1[0] * w[d] / (float)gcnt[d];
}

__global__ void kl_fwd_kernel(const float* __restrict__ mu, const float* __restrict__ sd, long n, float* __restrict__ kl_sum, OrdScratch sc) {
  __shared__ float red[4];
  float s = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float m = mu[i], d = sd[i];
    s += -logf(d) + (d * d + m * m) * 0.5f - 0.5f;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) red[0] = red[0] + red[1] + red[2] + red[3];
  __syncthreads();
  ordered_commit(kl_sum, red, 1, sc, 0, blockIdx.x, gridDim.x);
}

__global__ void kl_bwd_kernel(const float* __restrict__ mu, const float* __restrict__ sd, long n, const float* __restrict__ gscale,
                              float* __restrict__ dmu, float* __restrict__ dsd) {
  const float gs = gscale[0];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float m = mu[i], d = sd[i];
    dmu[i] = gs * m;
    dsd[i] = gs * (d - 1.0f / d);
  }
}

// out: loss, recon, pl, dl, kl, kl_chd, kl_rhy, chord, root, chroma, bass   (train.py:54-55 order)
__global__ void loss_finalize_kernel(const float* __restrict__ sums, const int* __restrict__ counts, float beta, float w0, float w1,
                                     float n_kl, float n_root, float n_chroma, float* __restrict__ out, const float* __restrict__ sp) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (sp) beta = sp[0];                                                   // (ptv_step_params: beta of a graph-replayed step)
  float pl = sums[0] / (float)counts[0];
  float dl = sums[1] / (float)counts[1];
  float klc = sums[2] / n_kl, klr = sums[3] / n_kl;
  float root = sums[4] / n_root, chroma = sums[5] / n_chroma, bass = sums[6] / n_root;
  float recon = w0 * pl + w1 * dl;
  float kl = klc + klr;
  float chord = root + chroma + bass;
  out[0] = recon + beta * kl + chord;
  out[1] = recon; out[2] = pl; out[3] = dl; out[4] = kl; out[5] = klc; out[6] = klr;
  out[7] = chord; out[8] = root; out[9] = chroma; out[10] = bass;
}

// upstream grads of the 11 outputs (null entries = 0) -> per-component scale factors gs[7]
__global__ void loss_bwd_scales_kernel(const float* __restrict__ g, const int* __restrict__ counts, float beta, float w0, float w1,
                                       float n_kl, float n_root, float n_chroma, float* __restrict__ gs, const float* __restrict__ sp) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (sp) beta = sp[0];
  float g_recon = g[0] + g[1];
  gs[0] = (g_recon * w0 + g[2]) / (float)counts[0];
  gs[1] = (g_recon * w1 + g[3]) / (float)counts[1];
  float g_kl = g[0] * beta + g[4];
  gs[2] = (g_kl + g[5]) / n_kl;
  gs[3] = (g_kl + g[6]) / n_kl;
  float g_chord = g[0] + g[7];
  gs[4] = (g_chord + g[8]) / n_root;
  gs[5] = (g_chord + g[9]) / n_chroma;
  gs[6] = (g_chord + g[10]) / n_root;
}

static inline int grid_rows(long n, int per_block, int cap = 8192) {
  long b = (n + per_block - 1) / per_block; if (b > cap) b = cap; if (b < 1) b = 1; return (int)b;
}

}  // namespace ptv

using namespace ptv;

extern "C" int ptv_pianotree_targets_rows(const long* x, int B, int step_major, int* pitch_t, int* dur_t, int* counts, int* row_live, void* stream);
extern "C" int ptv_pianotree_targets(const long* x, int B, int step_major, int* pitch_t, int* dur_t, int* counts, void* stream) {
  return ptv_pianotree_targets_rows(x, B, step_major, pitch_t, dur_t, counts, nullptr, stream);
}
extern "C" int ptv_pianotree_targets_rows(const long* x, int B, int step_major, int* pitch_t, int* dur_t, int* counts, int* row_live, void* stream) {
  if (!x || !pitch_t || !dur_t || !counts || B <= 0) return PTV_ERR_ARG;
  hipLaunchKernelGGL(pianotree_targets_kernel, dim3(grid_rows((long)B * 480, 256, 256)), dim3(256), 0, (hipStream_t)stream, x, B, step_major, pitch_t, dur_t, counts, row_live);   // (two atomics on the count words per block)
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

extern "C" int ptv_chord_targets(const float* c, int B, int step_major, int* root_t, int* chroma_t, int* bass_t, void* stream) {
  if (!c || !root_t || !chroma_t || !bass_t || B <= 0) return PTV_ERR_ARG;
  hipLaunchKernelGGL(chord_targets_kernel, dim3(cdiv((long)B * 8, 256)), dim3(256), 0, (hipStream_t)stream, c, B, step_major, root_t, chroma_t, bass_t);
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

extern "C" int ptv_ce_fwd(const float* logits, long ld, const int* targets, long rows, int C, int ignore_index, float* nll_sum, void* stream) {
  if (!logits || !targets || !nll_sum || rows <= 0 || C <= 0 || C > 256) return PTV_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const OrdScratch sc = ord_scratch(s, 2048, 1);
  if (C <= 16) hipLaunchKernelGGL((ce_small_kernel<false>), dim3(grid_rows(rows, 256, 2048)), dim3(256), 0, s, logits, ld, targets, rows, C, ignore_index, nll_sum, nullptr, nullptr, 0L, sc);
  else if (ce_vec_ok(logits, ld, C)) hipLaunchKernelGGL((ce_vec_kernel<false>), dim3(grid_rows(rows, 8, 2048)), dim3(256), 0, s, logits, ld, targets, rows, C, ignore_index, nll_sum, nullptr, nullptr, 0L, sc);   // (one atomicAdd on nll_sum per block: 16384 blocks spent 210 us queueing on that one address, 2048: 51 us)
  else hipLaunchKernelGGL((ce_wave_kernel<false>), dim3(grid_rows(rows, 4, 2048)), dim3(256), 0, s, logits, ld, targets, rows, C, ignore_index, nll_sum, nullptr, nullptr, 0L, sc);
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

extern "C" int ptv_ce_bwd(const float* logits, long ld, const int* targets, long rows, int C, int ignore_index, const float* gscale, float* dlogits, long ldd, void* stream) {
  if (!logits || !targets || !gscale || !dlogits || rows <= 0 || C <= 0 || C > 256) return PTV_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (C <= 16) hipLaunchKernelGGL((ce_small_kernel<true>), dim3(grid_rows(rows, 256, 2048)), dim3(256), 0, s, logits, ld, targets, rows, C, ignore_index, nullptr, gscale, dlogits, ldd, OrdScratch{nullptr, nullptr});
  else if (ce_vec_ok(logits, ld, C) && ce_vec_ok(dlogits, ldd, C)) hipLaunchKernelGGL((ce_vec_kernel<true>), dim3(grid_rows(rows, 8, 16384)), dim3(256), 0, s, logits, ld, targets, rows, C, ignore_index, nullptr, gscale, dlogits, ldd, OrdScratch{nullptr, nullptr});
  else hipLaunchKernelGGL((ce_wave_kernel<true>), dim3(grid_rows(rows, 4, 16384)), dim3(256), 0, s, logits, ld, targets, rows, C, ignore_index, nullptr, gscale, dlogits, ldd, OrdScratch{nullptr, nullptr});
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

extern "C" int ptv_kl_fwd(const float* mu, const float* sd, long n, float* kl_sum, void* stream) {
  if (!mu || !sd || !kl_sum || n <= 0) return PTV_ERR_ARG;
  hipLaunchKernelGGL(kl_fwd_kernel, dim3(grid_rows(n, 256, 256)), dim3(256), 0, (hipStream_t)stream, mu, sd, n, kl_sum, ord_scratch((hipStream_t)stream, 256, 1));
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

extern "C" int ptv_kl_bwd(const float* mu, const float* sd, long n, const float* gscale, float* dmu, float* dsd, void* stream) {
  if (!mu || !sd || !gscale || !dmu || !dsd || n <= 0) return PTV_ERR_ARG;
  hipLaunchKernelGGL(kl_bwd_kernel, dim3(grid_rows(n, 256, 1024)), dim3(256), 0, (hipStream_t)stream, mu, sd, n, gscale, dmu, dsd);
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

extern "C" int ptv_loss_finalize(const float* sums, const int* counts, float beta, float w0, float w1, float n_kl, float n_root, float n_chroma, float* out11, void* stream) {
  if (!sums || !counts || !out11) return PTV_ERR_ARG;
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sums, counts, beta, w0, w1, n_kl, n_root, n_chroma, out11, beta != 0.f ? ptv::g_step_params : nullptr);
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

extern "C" int ptv_loss_bwd_scales(const float* gout11, const int* counts, float beta, float w0, float w1, float n_kl, float n_root, float n_chroma, float* gs7, void* stream) {
  if (!gout11 || !counts || !gs7) return PTV_ERR_ARG;
  hipLaunchKernelGGL(loss_bwd_scales_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, gout11, counts, beta, w0, w1, n_kl, n_root, n_chroma, gs7, beta != 0.f ? ptv::g_step_params : nullptr);
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

extern "C" int ptv_ce_group_fwd(const float* logits, long ld, const int* targets, long rows, int C, int ignore_index, int G,
                                float* nll_sum, int* count, void* stream) {
  if (!logits || !targets || !nll_sum || !count || rows <= 0 || C <= 0 || C > 16 || G <= 0 || G > 8) return PTV_ERR_ARG;
  hipLaunchKernelGGL((ce_group_kernel<false>), dim3(grid_rows(rows, 256, 2048)), dim3(256), 0, (hipStream_t)stream, logits, ld, targets, rows, C,
                     ignore_index, G, nll_sum, count, nullptr, nullptr, 0L);
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

extern "C" int ptv_ce_group_bwd(const float* logits, long ld, const int* targets, long rows, int C, int ignore_index, int G,
                                const float* gscale, float* dlogits, long ldd, void* stream) {
  if (!logits || !targets || !gscale || !dlogits || rows <= 0 || C <= 0 || C > 16 || G <= 0 || G > 8) return PTV_ERR_ARG;
  hipLaunchKernelGGL((ce_group_kernel<true>), dim3(grid_rows(rows, 256, 2048)), dim3(256), 0, (hipStream_t)stream, logits, ld, targets, rows, C,
                     ignore_index, G, nullptr, nullptr, gscale, dlogits, ldd);
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

extern "C" int ptv_wdur_finalize(const float* gsum5, const int* gcnt5, float w0, float w1, float w2, float w3, float w4, float* sums1,
                                 int* counts1, void* stream) {
  if (!gsum5 || !gcnt5 || !sums1 || !counts1) return PTV_ERR_ARG;
  hipLaunchKernelGGL(wdur_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, gsum5, gcnt5, w0, w1, w2, w3, w4, sums1, counts1);
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

extern "C" int ptv_wdur_scales(const float* gs1, const int* gcnt5, float w0, float w1, float w2, float w3, float w4, float* out5,
                               void* stream) {
  if (!gs1 || !gcnt5 || !out5) return PTV_ERR_ARG;
  hipLaunchKernelGGL(wdur_scales_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, gs1, gcnt5, w0, w1, w2, w3, w4, out5);
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

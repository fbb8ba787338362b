// data.hip -- the reference's per-item data contract as ONE device batch transform.
//
// ArrangementDataset.__getitem__ (dataset.py:88-112) turns a 2-bar accompaniment piano-roll (values 2 = onset, 1 = sustain,
// 0 = silence; converter.py:35-47) and 8 raw chords [root, 12 chroma bits, bass] into the three tensors the train step
// consumes, per item, in numpy, with num_workers = 0 (dataset.py:279-280): at 30k samples/s that loader is the first thing
// that caps a run.  Here the segment bank lives in HBM as uint8 and one kernel does, per sample of the batch:
//   augment_pr        converter.py:65-68    pitch roll by `shift` semitones (np.roll over the 128-pitch axis, wraps)
//   pr_to_onehot_pr + piano_roll_to_target   converter.py:78-113   duration at onset cells: walking time backwards, per pitch,
//                     every non-onset non-silence cell adds 1 to a carry that an onset consumes (carry + 1) and clears
//   target_to_3dtarget converter.py:116-147 (args of dataset.py:98-104)  PianoTree grid [32,16,6]: <sos>, the step's notes in
//                     ascending pitch as [pitch, 5-bit MSB-first binary of dur-1], <eos>, <pad> rows
//   expand_chord      converter.py:150-164  root / bass (x + shift) mod 12 one-hots, chroma rolled by shift
// Integer / index work: bit-exact against the oracle (oracle/data_oracle.py) and the reference-generated fixtures.
// HBM-bound: reads 4 KB + writes 16 KB (pr_mat f32) + 24 KB (grid int64) + 1.1 KB (chord) per sample, all coalesced except
// the grid rows (32 lanes x 96 longs each).
#include "common.hpp"
#include "../../include/ptvae_hip.h"

namespace ptv {

__global__ __launch_bounds__(128) void batch_transform_kernel(const unsigned char* __restrict__ pr, const float* __restrict__ chord14,
                                                              const int* __restrict__ index, const int* __restrict__ shift,
                                                              float* __restrict__ pr_mat, long* __restrict__ x, float* __restrict__ c,
                                                              int* __restrict__ err, int B) {
  __shared__ unsigned char dur[32][128];
  const int b = blockIdx.x, p = threadIdx.x;
  const long src_item = index ? index[b] : b;
  const int sh = shift ? shift[b] : 0;
  const int src = (((p - sh) % 128) + 128) % 128;              // np.roll(pr, sh)[p] = pr[(p - sh) mod 128]
  const unsigned char* col = pr + src_item * 32 * 128 + src;
  int carry = 0;
  for (int t = 31; t >= 0; t--) {
    const int v = col[t * 128];
    const bool onset = v == 2;
    const int cur = ((v != 2 && v != 0) ? 1 : 0) + carry;
    const int d = onset ? cur + 1 : 0;
    dur[t][p] = (unsigned char)d;
    pr_mat[((long)b * 32 + t) * 128 + p] = (float)d;
    carry = onset ? 0 : cur;
  }
  __syncthreads();
  if (p < 32) {
    const int t = p;
    long* g = x + ((long)b * 32 + t) * 16 * 6;
    g[0] = 128;
#pragma unroll
    for (int k = 1; k < 6; k++) g[k] = 2;
    int n = 1;
    bool over = false;
    for (int q = 0; q < 128; q++) {
      const int d = dur[t][q];
      if (d == 0) continue;
      if (n >= 15) { over = true; continue; }               // the reference raises IndexError here (converter.py:141)
      long* r = g + n * 6;
      const int e = d - 1;
      r[0] = q; r[1] = (e >> 4) & 1; r[2] = (e >> 3) & 1; r[3] = (e >> 2) & 1; r[4] = (e >> 1) & 1; r[5] = e & 1;
      n++;
    }
    for (int k = n; k < 16; k++) {
      long* r = g + k * 6;
      r[0] = k == n ? 129 : 130;
      r[1] = r[2] = r[3] = r[4] = r[5] = 2;
    }
    if (over) atomicOr(err, 1);
  }
  // chords: 8 steps x 36 outputs
  for (int i = p; i < 8 * 36; i += 128) {
    const int s = i / 36, k = i % 36;
    const float* cr = chord14 + (src_item * 8 + s) * 14;
    float v;
    if (k < 12) v = ((((int)cr[0] + sh) % 12 + 12) % 12 == k) ? 1.f : 0.f;
    else if (k < 24) v = cr[1 + ((((k - 12) - sh) % 12) + 12) % 12];
    else v = ((((int)cr[13] + sh) % 12 + 12) % 12 == k - 24) ? 1.f : 0.f;
    c[((long)b * 8 + s) * 36 + k] = v;
  }
}

// out[i, :] = slerp path between z1[b] and z2[b] (model.py:216-242 interp_path): directions interpolated on the unit sphere,
// norms interpolated geometrically.  One wave per sample; out [B, n, D].
__global__ __launch_bounds__(64) void slerp_path_kernel(const float* __restrict__ z1, const float* __restrict__ z2, float* __restrict__ out,
                                                        int D, int n) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const float* a = z1 + (long)b * D;
  const float* q = z2 + (long)b * D;
  float na = 0.f, nq = 0.f, dot = 0.f;
  for (int k = lane; k < D; k += 64) { na += a[k] * a[k]; nq += q[k] * q[k]; dot += a[k] * q[k]; }
  na = sqrtf(wave_sum(na)); nq = sqrtf(wave_sum(nq)); dot = wave_sum(dot) / (na * nq);
  const float omega = acosf(dot), so = sinf(omega);
  const float la = logf(na), lq = logf(nq);
  for (int i = 0; i < n; i++) {
    const float t = n > 1 ? (float)i / (float)(n - 1) : 0.f;
    const float wa = sinf((1.0f - t) * omega) / so, wq = sinf(t * omega) / so;
    const float len = expf(la + (lq - la) * t);
    for (int k = lane; k < D; k += 64) out[((long)b * n + i) * D + k] = (wa * a[k] / na + wq * q[k] / nq) * len;
  }
}

}  // namespace ptv

extern "C" int ptv_batch_transform(const unsigned char* pr, const float* chord14, const int* index, const int* shift,
                                   float* pr_mat, long* x, float* c, int* err, int B, void* stream) {
  if (!pr || !chord14 || !pr_mat || !x || !c || !err || B <= 0) return PTV_ERR_ARG;
  hipLaunchKernelGGL(ptv::batch_transform_kernel, dim3(B), dim3(128), 0, (hipStream_t)stream, pr, chord14, index, shift, pr_mat, x, c, err, B);
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

extern "C" int ptv_slerp_path(const float* z1, const float* z2, float* out, int B, int D, int n, void* stream) {
  if (!z1 || !z2 || !out || B <= 0 || D <= 0 || n <= 0) return PTV_ERR_ARG;
  hipLaunchKernelGGL(ptv::slerp_path_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, z1, z2, out, D, n);
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

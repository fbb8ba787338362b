// rng.hip -- sharding-invariant reparameterisation noise (SURVEY.md section 8 d/e).
//
// The reference draws eps for Normal.rsample (train_utils.py:33-34) from torch's global generator: the value a sample gets
// depends on its position in the batch and on how the batch is split over processes.  Here eps[row, j] is a pure function of
// (seed, stream, GLOBAL sample index, j): Philox4x32-10 keyed by the seed, counter = (global row, j / 4, stream lo, stream hi),
// four 32-bit words -> two Box-Muller pairs.  One batch on one GPU and the same batch split over N ranks (row_offset =
// rank * B_local) see identical noise; `stream` separates draws (training step x {chd, rhy}).
#include "common.hpp"
#include "../../include/ptvae_hip.h"

namespace ptv {

__device__ __forceinline__ void philox4x32_10(unsigned (&c)[4], unsigned k0, unsigned k1) {
#pragma unroll
  for (int r = 0; r < 10; r++) {
    const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned)p1;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

__global__ void philox_normal_kernel(float* __restrict__ out, long rows, int Z, unsigned long long seed, unsigned long long stream,
                                     long row_offset) {
  const int q4 = (Z + 3) >> 2;
  const long n = rows * q4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long r = i / q4; const int q = (int)(i - r * q4);
    const unsigned long long g = (unsigned long long)(row_offset + r);
    // counter = (row lo, row hi * 2^16 + column quad, stream lo, stream hi): rows < 2^48, quads < 2^16
    unsigned c[4] = {(unsigned)g, (unsigned)(g >> 32) * 0x10000u + (unsigned)q, (unsigned)stream, (unsigned)(stream >> 32)};
    philox4x32_10(c, (unsigned)seed, (unsigned)(seed >> 32));
    float v[4];
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const float u1 = ((float)c[2 * h] + 0.5f) * 2.3283064365386963e-10f;        // (0, 1]: 2^-32 * (x + 1/2), rounded
      const float u2 = ((float)c[2 * h + 1] + 0.5f) * 2.3283064365386963e-10f;
      const float rad = sqrtf(-2.0f * logf(fminf(fmaxf(u1, 1.1754944e-38f), 1.0f)));
      const float th = 6.283185307179586f * u2;
      v[2 * h] = rad * cosf(th); v[2 * h + 1] = rad * sinf(th);
    }
#pragma unroll
    for (int e = 0; e < 4; e++) if (q * 4 + e < Z) out[r * Z + q * 4 + e] = v[e];
  }
}

}  // namespace ptv

extern "C" int ptv_philox_normal(float* out, long rows, int Z, unsigned long long seed, unsigned long long stream_id, long row_offset,
                                 void* stream) {
  if (!out || rows <= 0 || Z <= 0 || Z > 4 * 65536 || row_offset < 0) return PTV_ERR_ARG;
  long nb = (rows * ((Z + 3) / 4) + 255) / 256; if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(ptv::philox_normal_kernel, dim3((int)nb), dim3(256), 0, (hipStream_t)stream, out, rows, Z, seed, stream_id, row_offset);
  PTV_CHECK_LAUNCH();
  return PTV_OK;
}

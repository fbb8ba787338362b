// How fast can every CU stream the SAME few megabytes from its XCD's L2 (the weight stream of the row-partitioned GRU kernels)?
// 256 workgroups x 4 waves; each wave reads the buffer in 1-KB wave-loads (16 B per lane), U loads in flight, `passes` times.
// Variants: rot = 0 (all workgroups walk in the same order) / 1 (each workgroup starts at its own offset); per-wave quarter or whole buffer.
// hipcc --offload-arch=gfx950 -O3 scripts/micro/l2_bcast.hip -o /tmp/l2_bcast && /tmp/l2_bcast
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int U>
__global__ __launch_bounds__(256, 1) void stream_kernel(const u32x4* __restrict__ buf, long n16, int passes, int rot, int split, unsigned* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // fragments of 64 x 16 B; this wave's range
  const long nfrag = n16 / 64;
  const long per = split ? nfrag / 4 : nfrag;
  const long base = split ? wave * per : 0;
  const long start = rot ? ((long)blockIdx.x * 977 + wave * 131) % per : 0;
  u32x4 acc = {0, 0, 0, 0};
  for (int p = 0; p < passes; p++) {
    for (long f = 0; f < per; f += U) {
      u32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        long g = start + f + u; if (g >= per) g -= per;
        v[u] = buf[(base + g) * 64 + lane];
      }
#pragma unroll
      for (int u = 0; u < U; u++) acc ^= v[u];
    }
  }
  if (acc.x == 0x12345678u) out[0] = acc.y;
}

int main() {
  const long bytes_list[] = {1966080, 524288, 8388608};
  unsigned* out; hipMalloc(&out, 64);
  for (long bytes : bytes_list) {
    u32x4* buf; hipMalloc(&buf, bytes); hipMemset(buf, 1, bytes);
    const long n16 = bytes / 16;
    for (int split = 0; split < 2; split++)
      for (int rot = 0; rot < 2; rot++) {
        const int passes = 30;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        auto run = [&](int U) {
          if (U == 4) hipLaunchKernelGGL(stream_kernel<4>, dim3(256), dim3(256), 0, 0, buf, n16, passes, rot, split, out);
          else if (U == 8) hipLaunchKernelGGL(stream_kernel<8>, dim3(256), dim3(256), 0, 0, buf, n16, passes, rot, split, out);
          else hipLaunchKernelGGL(stream_kernel<16>, dim3(256), dim3(256), 0, 0, buf, n16, passes, rot, split, out);
        };
        for (int U : {4, 8, 16}) {
          run(U); hipDeviceSynchronize();
          hipEventRecord(e0); run(U); hipEventRecord(e1); hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1);
          const double per_cu = (double)bytes * (split ? 1 : 4) * passes / (ms * 1e-3) / 1e9;      // bytes a CU pulled per second
          printf("buffer %7.2f MB  %s  rot %d  U %2d : %7.3f ms  %6.1f GB/s per CU  %6.2f TB/s chip\n", bytes / 1048576.0,
                 split ? "quarter per wave" : "whole per wave  ", rot, U, ms, per_cu, per_cu * 256 / 1e3);
        }
      }
    hipFree(buf);
  }
  return 0;
}

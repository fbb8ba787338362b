// tcp_order.hip -- does a CU's vector-memory pipe return data IN ORDER ACROSS WAVES?  Four "hit" waves per CU stream a 1.9-MB L2-resident
// buffer (the notes GRU's weights) while N "miss" waves of the same workgroup stream once-only data from HBM.  If the L1 returned loads in
// issue order across waves, the hit waves' bandwidth would collapse to the miss stream's latency even though they never touch HBM.
//   hipcc --offload-arch=gfx950 -O3 -o tcp_order tcp_order.hip && ./tcp_order
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int U>
__global__ __launch_bounds__(512) void k(const u32x4* __restrict__ w, long n16, int passes, const u32x4* __restrict__ big, long big16_per_wave, int nmiss,
                                         int miss_depth, unsigned* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  u32x4 acc = {0, 0, 0, 0};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (wave < 4) {                                                   // hit waves: a quarter of the buffer each, U loads in flight
    const long per = n16 / 64 / 4, base = wave * per;
    for (int p = 0; p < passes; p++)
      for (long f = 0; f < per; f += U) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) { long g = f + u; if (g >= per) g -= per; v[u] = w[(base + g) * 64 + lane]; }
#pragma unroll
        for (int u = 0; u < U; u++) acc ^= v[u];
      }
    if (lane == 0) atomicMax((unsigned long long*)(out + 2), __builtin_amdgcn_s_memtime() - t0);       // slowest hit wave, shader cycles
  } else if (wave - 4 < nmiss) {                                    // miss waves: once-only data, miss_depth loads in flight
    const u32x4* p = big + ((long)blockIdx.x * 4 + (wave - 4)) * big16_per_wave;
    for (long i = 0; i + 8 * 64 <= big16_per_wave; i += 8 * 64) {
      u32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) if (u < miss_depth) v[u] = __builtin_nontemporal_load(p + i + u * 64 + lane);
#pragma unroll
      for (int u = 0; u < 8; u++) if (u < miss_depth) acc ^= v[u];
    }
  }
  if (acc.x == 0x12345678u) out[0] = acc.y;
}

int main() {
  const long wbytes = 1966080; const int passes = 8;
  unsigned* out; hipMalloc(&out, 64);
  u32x4* w; hipMalloc(&w, wbytes); hipMemset(w, 1, wbytes);
  const long big_per_wave = 4L << 20;                               // 4 MB per miss wave
  u32x4* big; hipMalloc(&big, 256 * 4 * big_per_wave); hipMemset(big, 2, 256 * 4 * big_per_wave);
  for (int nmiss = 0; nmiss <= 4; nmiss += (nmiss == 0 ? 1 : (nmiss == 1 ? 3 : 1))) {
    for (int depth = 1; depth <= 8; depth *= 8) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      auto run = [&]() { hipLaunchKernelGGL((k<8>), dim3(256), dim3(512), 0, 0, w, wbytes / 16, passes, big, big_per_wave / 16, nmiss, depth, out); };
      run(); hipDeviceSynchronize(); hipMemset(out, 0, 64);
      hipEventRecord(e0); run(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      // (the kernel ends when the slower role ends: report both rates against the whole time)
      unsigned long long cyc = 0; hipMemcpy(&cyc, out + 2, 8, hipMemcpyDeviceToHost);
      printf("miss waves %d depth %d : kernel %7.3f ms   slowest hit wave %9llu cycles (s_memtime ticks)   miss stream %5.2f TB/s chip\n", nmiss, depth, ms, cyc,
             nmiss ? 256.0 * nmiss * big_per_wave * depth / 8 / (ms * 1e-3) / 1e12 : 0.0);
      if (nmiss == 0) break;
    }
  }
  // hit waves alone over the time of the miss waves alone, for reference: miss-only run
  return 0;
}

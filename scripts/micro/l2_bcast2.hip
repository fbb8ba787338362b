// l2_bcast.hip plus the row-GRU kernels' activation streams: per pass over the shared 1.9-MB weight buffer every workgroup also
// writes WR KB and reads RD KB of private streaming data (once-only bytes), interleaved with the weight loads.  Store / load cache
// policy variants.  Question: what does the activation stream do to the weight stream's 110 GB/s per CU?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// POL: 0 plain stores/loads, 1 nontemporal, 2 stores only at the END of the pass (burst), 3 nt + burst
template <int U, int POL>
__global__ __launch_bounds__(256, 1) void mix_kernel(const u32x4* __restrict__ w, long n16, int passes, u32x4* __restrict__ act, long act16_per_wg_pass,
                                                      int wr_per_frag, int rd_per_frag, unsigned* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long nfrag = n16 / 64, per = nfrag / 4, base = wave * per;
  const long start = ((long)blockIdx.x * 977 + wave * 131) % per;
  u32x4 acc = {0, 0, 0, 0}, xprev = {0, 0, 0, 0};
  const u32x4 cst = {1u, 2u, 3u, (unsigned)threadIdx.x};
  for (int p = 0; p < passes; p++) {
    u32x4* a = act + ((long)(blockIdx.x * passes + p) * act16_per_wg_pass) + (long)wave * (act16_per_wg_pass / 4);
    long ai = 0;
    for (long f = 0; f < per; f += U) {
      u32x4 v[U];
#pragma unroll
      for (int u = 0; u < U; u++) {
        long g = start + f + u; if (g >= per) g -= per;
        v[u] = w[(base + g) * 64 + lane];
      }
      // activation traffic interleaved: rd reads + wr writes of 1 KB per U weight fragments.  The read is consumed one group LATER
      // (its HBM latency must not sit in front of the weight loads in the in-order vmcnt queue), the stores carry a constant
      u32x4 xn = {0, 0, 0, 0};
      if (rd_per_frag) { xn = (POL & 1) ? __builtin_nontemporal_load(a + ai * 64 + lane) : a[ai * 64 + lane]; ai++; }
      if (!(POL & 2))
        for (int r = 0; r < wr_per_frag; r++) {
          if (POL & 1) __builtin_nontemporal_store(cst, a + ai * 64 + lane); else a[ai * 64 + lane] = cst;
          ai++;
        }
#pragma unroll
      for (int u = 0; u < U; u++) acc ^= v[u];
      acc ^= xprev; xprev = xn;
    }
    if (POL & 2)
      for (long f = 0; f < per; f += U)
        for (int r = 0; r < wr_per_frag; r++) {
          if (POL & 1) __builtin_nontemporal_store(cst, a + ai * 64 + lane); else a[ai * 64 + lane] = cst;
          ai++;
        }
  }
  if (acc.x == 0x12345678u) out[0] = acc.y;
}

int main() {
  const long wbytes = 1966080;
  const int passes = 15;
  unsigned* out; hipMalloc(&out, 64);
  u32x4* w; hipMalloc(&w, wbytes); hipMemset(w, 1, wbytes);
  const long n16 = wbytes / 16, per = n16 / 64 / 4;
  const int U = 8;
  // per wave per pass: per / U groups; activations per group: (rd + wr) KB
  for (int cfg = 0; cfg < 7; cfg++) {
    // cfg 4-6: stores only / loads only -- which of the two slow streams blocks the weight stream?
    const int rd = cfg == 0 ? 0 : (cfg == 4 ? 0 : (cfg >= 5 ? cfg - 4 : 1)), wr = cfg == 0 ? 0 : (cfg <= 3 ? cfg : (cfg == 4 ? 2 : 0));
    const long groups = (per + U - 1) / U;
    const long act16_per_wave = groups * (rd + wr) * 64;
    const long act16_per_wg_pass = act16_per_wave * 4;
    const size_t act_bytes = (size_t)256 * passes * act16_per_wg_pass * 16 + 1024;
    u32x4* act; hipMalloc(&act, act_bytes); hipMemset(act, 0, act_bytes);
    for (int pol = 0; pol < 4; pol++) {
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      auto run = [&]() {
        if (pol == 0) hipLaunchKernelGGL((mix_kernel<U, 0>), dim3(256), dim3(256), 0, 0, w, n16, passes, act, act16_per_wg_pass, wr, rd, out);
        if (pol == 1) hipLaunchKernelGGL((mix_kernel<U, 1>), dim3(256), dim3(256), 0, 0, w, n16, passes, act, act16_per_wg_pass, wr, rd, out);
        if (pol == 2) hipLaunchKernelGGL((mix_kernel<U, 2>), dim3(256), dim3(256), 0, 0, w, n16, passes, act, act16_per_wg_pass, wr, rd, out);
        if (pol == 3) hipLaunchKernelGGL((mix_kernel<U, 3>), dim3(256), dim3(256), 0, 0, w, n16, passes, act, act16_per_wg_pass, wr, rd, out);
      };
      run(); hipDeviceSynchronize();
      hipEventRecord(e0); run(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double us_pass = ms * 1e3 / passes;
      const double act_kb = (double)act16_per_wg_pass * 16 / 1024;
      printf("act per WG per pass: %6.0f KB (rd %d wr %d per 8 KB of weights)  policy %d : %7.2f us per pass  weights %6.1f GB/s per CU  activations %5.2f TB/s chip\n",
             act_kb, rd, wr, pol, us_pass, wbytes / (us_pass * 1e-6) / 1e9, act_kb * 1024 * 256 / (us_pass * 1e-6) / 1e12);
    }
    hipFree(act);
  }
  return 0;
}

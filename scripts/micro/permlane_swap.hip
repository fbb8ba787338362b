#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ float xor16_sum(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}
__device__ __forceinline__ float xor32_sum(float x) {
  float a = x, b = x;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}
__global__ void k(float* o) {
  float x = o[threadIdx.x];
  float y = xor16_sum(x);
  float z = xor32_sum(y);
  o[64 + threadIdx.x] = y; o[128 + threadIdx.x] = z;
  float s = x + __shfl_xor(x, 16, 64); float s2 = s + __shfl_xor(s, 32, 64);
  o[192 + threadIdx.x] = s; o[256 + threadIdx.x] = s2;
}
int main() {
  float h[320]; for (int i = 0; i < 64; i++) h[i] = 1.0f / (i + 3) + i * 0.37f;
  float* d; hipMalloc(&d, sizeof(h)); hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 64; i++) { if (h[64 + i] != h[192 + i] || h[128 + i] != h[256 + i]) bad++; }
  printf("permlane swap sums: %d mismatches of 64 (y0=%g s0=%g z0=%g s20=%g)\n", bad, h[64], h[192], h[128], h[256]);
  return bad != 0;
}

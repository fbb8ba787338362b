// micro-benchmark: HBM rate of the GRU-cell epilogue's access pattern, without the GEMM.
// per element (row m, unit j): read gi[m, g*H + j] (bf16, g = 0..2), hp[m, j] (fp32);
//                              write h[m, j] (fp32), h16[m, j] (bf16), gates[p][m, j] (bf16, p = 0..3)
// PAT 0: MFMA C layout   (lane&15 = row, lane>>4 = 4-unit group; 8-byte bf16 accesses, 16 rows per instruction)
// PAT 1: row-contiguous, 4 units per lane (16 lanes per 64-unit row segment)
// PAT 2: row-contiguous, 8 units per lane (16-byte bf16 accesses; 8 lanes per 64-unit row segment)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int PAT, int CELLS>
__global__ __launch_bounds__(256) void k(const __bf16* __restrict__ gi, const float* __restrict__ hp, float* __restrict__ h,
                                         __bf16* __restrict__ h16, __bf16* __restrict__ gates, int M, int H, const __bf16* __restrict__ gi2) {
  // block tile: 64 rows x 64 units
  const int ntn = H / 64;
  const int mt = blockIdx.x / ntn, nt = blockIdx.x % ntn;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long plane = (long)M * H;
  if constexpr (PAT == 0) {
    const int m0 = mt * 64 + (wave >> 1) * 32, j0 = nt * 64 + (wave & 1) * 32;
    bf16x4 g[2][2][3]; float4 p[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int f = 0; f < 2; f++) {
        const long m = m0 + i * 16 + (lane & 15); const int j = j0 + f * 16 + (lane >> 4) * 4;
#pragma unroll
        for (int q = 0; q < 3; q++) g[i][f][q] = *(const bf16x4*)(gi + m * 3 * H + q * H + j);
        p[i][f] = *(const float4*)(hp + m * H + j);
      }
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int f = 0; f < 2; f++) {
        const long m = m0 + i * 16 + (lane & 15); const int j = j0 + f * 16 + (lane >> 4) * 4;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] = (float)g[i][f][0][e] + (float)g[i][f][1][e] * (float)g[i][f][2][e] + ((const float*)&p[i][f])[e];
        *(float4*)(h + m * H + j) = make_float4(v[0], v[1], v[2], v[3]);
        bf16x4 o; for (int e = 0; e < 4; e++) o[e] = (__bf16)v[e];
        *(bf16x4*)(h16 + m * H + j) = o;
#pragma unroll
        for (int q = 0; q < 4; q++) *(bf16x4*)(gates + q * plane + m * H + j) = o;
      }
  } else if constexpr (PAT == 1) {
    // 256 threads: 16 rows x 16 lanes per pass, 4 passes
    bf16x4 g[4][3]; float4 p[4];
    const int j = nt * 64 + (threadIdx.x & 15) * 4;
#pragma unroll
    for (int ps = 0; ps < 4; ps++) {
      const long m = mt * 64 + ps * 16 + (threadIdx.x >> 4);
#pragma unroll
      for (int q = 0; q < 3; q++) g[ps][q] = *(const bf16x4*)(gi + m * 3 * H + q * H + j);
      p[ps] = *(const float4*)(hp + m * H + j);
    }
#pragma unroll
    for (int ps = 0; ps < 4; ps++) {
      const long m = mt * 64 + ps * 16 + (threadIdx.x >> 4);
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; e++) v[e] = (float)g[ps][0][e] + (float)g[ps][1][e] * (float)g[ps][2][e] + ((const float*)&p[ps])[e];
      *(float4*)(h + m * H + j) = make_float4(v[0], v[1], v[2], v[3]);
      bf16x4 o; for (int e = 0; e < 4; e++) o[e] = (__bf16)v[e];
      *(bf16x4*)(h16 + m * H + j) = o;
#pragma unroll
      for (int q = 0; q < 4; q++) *(bf16x4*)(gates + q * plane + m * H + j) = o;
    }
  } else if constexpr (PAT == 4) {
    bf16x4 g[4][3], g2[4][3]; float4 p[4];
    const int j = nt * 64 + (wave & 1) * 32 + (lane & 7) * 4;
#pragma unroll
    for (int ps = 0; ps < 4; ps++) {
      const long m = mt * 64 + (wave >> 1) * 32 + ps * 8 + (lane >> 3);
#pragma unroll
      for (int q = 0; q < 3; q++) { g[ps][q] = *(const bf16x4*)(gi + m * 3 * H + q * H + j); g2[ps][q] = *(const bf16x4*)(gi2 + m * 3 * H + q * H + j); }
      p[ps] = *(const float4*)(hp + m * H + j);
    }
#pragma unroll
    for (int ps = 0; ps < 4; ps++) {
      const long m = mt * 64 + (wave >> 1) * 32 + ps * 8 + (lane >> 3);
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; e++) v[e] = (float)g[ps][0][e] + (float)g2[ps][0][e] + ((float)g[ps][1][e] + (float)g2[ps][1][e]) * ((float)g[ps][2][e] + (float)g2[ps][2][e]) + ((const float*)&p[ps])[e];
      *(float4*)(h + m * H + j) = make_float4(v[0], v[1], v[2], v[3]);
      bf16x4 o; for (int e = 0; e < 4; e++) o[e] = (__bf16)v[e];
      *(bf16x4*)(h16 + m * H + j) = o;
#pragma unroll
      for (int q = 0; q < 4; q++) *(bf16x4*)(gates + q * plane + m * H + j) = o;
    }
  } else if constexpr (PAT == 3) {
    // wave tile 32 rows x 32 units (2x2 waves), row-contiguous inside the wave: 8 lanes per row, 8 rows per pass
    bf16x4 g[4][3]; float4 p[4];
    const int j = nt * 64 + (wave & 1) * 32 + (lane & 7) * 4;
#pragma unroll
    for (int ps = 0; ps < 4; ps++) {
      const long m = mt * 64 + (wave >> 1) * 32 + ps * 8 + (lane >> 3);
#pragma unroll
      for (int q = 0; q < 3; q++) g[ps][q] = *(const bf16x4*)(gi + m * 3 * H + q * H + j);
      p[ps] = *(const float4*)(hp + m * H + j);
    }
#pragma unroll
    for (int ps = 0; ps < 4; ps++) {
      const long m = mt * 64 + (wave >> 1) * 32 + ps * 8 + (lane >> 3);
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; e++) v[e] = (float)g[ps][0][e] + (float)g[ps][1][e] * (float)g[ps][2][e] + ((const float*)&p[ps])[e];
      *(float4*)(h + m * H + j) = make_float4(v[0], v[1], v[2], v[3]);
      bf16x4 o; for (int e = 0; e < 4; e++) o[e] = (__bf16)v[e];
      *(bf16x4*)(h16 + m * H + j) = o;
#pragma unroll
      for (int q = 0; q < 4; q++) *(bf16x4*)(gates + q * plane + m * H + j) = o;
    }
  } else {
    // 8 units per lane: 32 rows x 8 lanes per pass, 2 passes
    bf16x8 g[2][3]; float4 p[2][2];
    const int j = nt * 64 + (threadIdx.x & 7) * 8;
#pragma unroll
    for (int ps = 0; ps < 2; ps++) {
      const long m = mt * 64 + ps * 32 + (threadIdx.x >> 3);
#pragma unroll
      for (int q = 0; q < 3; q++) g[ps][q] = *(const bf16x8*)(gi + m * 3 * H + q * H + j);
      p[ps][0] = *(const float4*)(hp + m * H + j); p[ps][1] = *(const float4*)(hp + m * H + j + 4);
    }
#pragma unroll
    for (int ps = 0; ps < 2; ps++) {
      const long m = mt * 64 + ps * 32 + (threadIdx.x >> 3);
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; e++) v[e] = (float)g[ps][0][e] + (float)g[ps][1][e] * (float)g[ps][2][e] + ((const float*)&p[ps][0])[e];
      *(float4*)(h + m * H + j) = make_float4(v[0], v[1], v[2], v[3]);
      *(float4*)(h + m * H + j + 4) = make_float4(v[4], v[5], v[6], v[7]);
      bf16x8 o; for (int e = 0; e < 8; e++) o[e] = (__bf16)v[e];
      *(bf16x8*)(h16 + m * H + j) = o;
#pragma unroll
      for (int q = 0; q < 4; q++) *(bf16x8*)(gates + q * plane + m * H + j) = o;
    }
  }
}

int main(int argc, char** argv) {
  const int M = 16384, H = 512, T = 8;
  __bf16 *gi, *h16, *gates, *gi2; float *hp, *h;
  hipMalloc(&gi2, (size_t)M * 3 * H * 2); hipMemset(gi2, 0, (size_t)M * 3 * H * 2);
  hipMalloc(&gi, (size_t)T * M * 3 * H * 2); hipMalloc(&hp, (size_t)T * M * H * 4); hipMalloc(&h, (size_t)T * M * H * 4);
  hipMalloc(&h16, (size_t)T * M * H * 2); hipMalloc(&gates, (size_t)T * 4 * M * H * 2);
  hipMemset(gi, 0, (size_t)T * M * 3 * H * 2); hipMemset(hp, 0, (size_t)T * M * H * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double bytes = (double)M * H * (6 + 4 + 4 + 2 + 8);
  for (int pat = 0; pat < 5; pat++) {
    float best = 1e9;
    for (int rep = 0; rep < 4; rep++) {
      hipEventRecord(e0);
      for (int t = 0; t < T; t++) {
        const dim3 grid(M / 64 * H / 64);
        const long o = (long)t * M * H;
        if (pat == 0) hipLaunchKernelGGL((k<0, 4>), grid, dim3(256), 0, 0, gi + 3 * o, hp + o, h + o, h16 + o, gates + 4 * o, M, H, gi2);
        if (pat == 1) hipLaunchKernelGGL((k<1, 4>), grid, dim3(256), 0, 0, gi + 3 * o, hp + o, h + o, h16 + o, gates + 4 * o, M, H, gi2);
        if (pat == 3) hipLaunchKernelGGL((k<3, 4>), grid, dim3(256), 0, 0, gi + 3 * o, hp + o, h + o, h16 + o, gates + 4 * o, M, H, gi2);
        if (pat == 4) hipLaunchKernelGGL((k<4, 4>), grid, dim3(256), 0, 0, gi + 3 * o, hp + o, h + o, h16 + o, gates + 4 * o, M, H, gi2);
        if (pat == 2) hipLaunchKernelGGL((k<2, 4>), grid, dim3(256), 0, 0, gi + 3 * o, hp + o, h + o, h16 + o, gates + 4 * o, M, H, gi2);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    const double by = bytes + (pat == 4 ? (double)M * H * 6 : 0);
    printf("pattern %d: %.1f us/launch  %.2f TB/s\n", pat, best * 1e3 / T, by / (best * 1e-3 / T) / 1e12);
  }
  return 0;
}

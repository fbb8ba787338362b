// common.hpp -- shared host/device helpers for the ptvae HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define PTV_OK 0
#define PTV_ERR_ARG (-1)
#define PTV_ERR_LAUNCH (-2)
#define PTV_ERR_UNSUPPORTED (-3)   // shape / device outside what a specialised kernel handles: caller uses the generic path

#define PTV_PREC_F32 0
#define PTV_PREC_BF16 1

// every entry point ends with this: report launch-configuration errors as a status code, never throw
#define PTV_CHECK_LAUNCH()                                                          \
  do {                                                                              \
    hipError_t e__ = hipGetLastError();                                             \
    if (e__ != hipSuccess) {                                                        \
      fprintf(stderr, "[ptvae_hip] %s:%d launch failed: %s\n", __FILE__, __LINE__,  \
              hipGetErrorString(e__));                                              \
      return PTV_ERR_LAUNCH;                                                        \
    }                                                                               \
  } while (0)

#define PTV_TRY(expr)                 \
  do {                                \
    int rc__ = (expr);                \
    if (rc__ != PTV_OK) return rc__;  \
  } while (0)

namespace ptv {

// accurate (ocml) transcendentals: the epilogues are a negligible share of the step and the fp32
// parity path must track the CPU reference through 50+ recurrent steps
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return tanhf(x); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Barrier for exchanges that go through LDS only.  __syncthreads() carries a workgroup-scope fence, for which hipcc also drains the
// wave's outstanding GLOBAL stores (s_waitcnt vmcnt(0)): about a microsecond of write-acknowledge latency at every barrier of a
// kernel that streams results to HBM while it iterates.  Use only where no other wave of the workgroup reads what this wave wrote
// to global memory.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ptv_zero_skip (misc.hip): the backward kernels pass over work whose result is exactly zero (note steps / tiles at which no gradient
// arrives, panel steps beyond the longest sequence); 0 makes them run dense (timing comparisons)
extern int g_zero_skip;
// ptv_step_params (misc.hip): device array of per-step scalars that override by-value arguments (graph-replayed steps), or null
extern const float* g_step_params;

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace ptv

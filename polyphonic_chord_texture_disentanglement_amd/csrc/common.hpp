// common.hpp -- shared host/device helpers for the ptvae HIP library (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

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

// compute units of the current device (MI355X: 256), cached
inline int num_cus() {
  static int n = 0;
  if (n == 0) { int dev = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256; }
  return n;
}

// accurate (ocml) transcendentals: the epilogues are a negligible share of the step and the fp32
// parity path must track the CPU reference through 50+ recurrent steps
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) { return tanhf(x); }
// bf16-precision kernels: hardware exp / reciprocal (~1 ulp each; the operands of these gates were rounded to bf16, 2^-9).  The ocml
// forms cost ~100 VALU instructions per GRU unit and step -- the 5-step duration GRU (236 M gate evaluations at B = 512) spent its whole
// 250 us on them (round 4: 8.3e9 lane-instructions = 240 us of the chip's VALU issue)
__device__ __forceinline__ float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanh_fast(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x)); }

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

// ---- ordered grid reductions (ptv_wgrad_mode / PTV_WGRAD_ORDERED, default on): instead of ending in one fp32 atomicAdd per block
// (summation order = arrival order: the last bits differ from run to run), every block parks its partial in a per-stream scratch
// slot and the LAST block to arrive adds the partials in block order.  slots == nullptr: the atomics.
// Hand-off without fences (an agent-scope release is an L2 write-back on a multi-XCD part, one per block): partials go out as
// write-through sc1 stores, the wave waits for their acknowledgement (vmcnt), one lane bumps the counter (relaxed), and the last block
// reads them with agent-scope loads -- the recipe of the persistent recurrences (gru_persist.hip).
struct OrdScratch { float* slots; unsigned* counters; };
constexpr long ORD_SLOT_FLOATS = 1L << 20;     // 4 MB of partials per stream
constexpr int ORD_COUNTERS = 1024;            // (the duration GRU's partial sums are 320 column blocks wide)
// ordered mode is on but a reduction had to fall back to fp32 atomics (no workspace: inside a capture, pool full, need too large):
// counted, readable through ptv_ordered_fallbacks() -- bit-reproducibility is the advertised default and must not be lost silently
extern int g_ord_fallbacks;
OrdScratch ord_scratch(hipStream_t s, long need_floats, int need_counters);     // misc.hip

// Every thread of the block calls it after the block's partial vector part[0..L) (LDS) is complete and visible (caller synced).
// group = which output vector / counter, idx = this block's position among the n blocks that contribute to it.
__device__ __forceinline__ void ordered_commit(float* out, const float* part, int L, const OrdScratch& sc, int group, int idx, int n) {
  if (!sc.slots) {
    for (int i = threadIdx.x; i < L; i += blockDim.x) { const float v = part[i]; if (v != 0.f) atomicAdd(out + i, v); }
    return;
  }
  __shared__ int s_last;
  __shared__ float s_red[16];
  float* base = sc.slots + (long)group * n * L;
  for (int i = threadIdx.x; i < L; i += blockDim.x) __hip_atomic_store(base + (long)idx * L + i, part[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the write-through (sc1) stores above are acknowledged ...
  __syncthreads();
  if (threadIdx.x == 0) s_last = __hip_atomic_fetch_add(sc.counters + group, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(n - 1);
  __syncthreads();
  if (!s_last) return;
  if (L == 1) {                                                   // a scalar: the whole block adds (fixed strides + fixed tree = fixed order)
    float s = 0.f;
    for (int b = threadIdx.x; b < n; b += blockDim.x) s += __hip_atomic_load(base + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
      float t = 0.f;
      for (int w = 0; w < (int)((blockDim.x + 63) >> 6); w++) t += s_red[w];
      out[0] += t;
    }
  } else {
    for (int i = threadIdx.x; i < L; i += blockDim.x) {
      float s = 0.f;
      int b = 0;
      for (; b + 8 <= n; b += 8) {                                  // eight partials in flight, added in block order
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; q++) v[q] = __hip_atomic_load(base + (long)(b + q) * L + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int q = 0; q < 8; q++) s += v[q];
      }
      for (; b < n; b++) s += __hip_atomic_load(base + (long)b * L + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      out[i] += s;
    }
  }
  if (threadIdx.x == 0) __hip_atomic_store(sc.counters + group, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace ptv

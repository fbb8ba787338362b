// prof.hpp -- optional launch timing (ptv_prof_* in include/ptvae_hip_debug.h): HIP events on the launch stream around the launches of
// one kernel family.  Defined in gru.hip.
#pragma once
#include <hip/hip_runtime.h>
namespace ptv {
namespace prof {
bool want(int tag, int M, int H);          // tag enabled and (M, H) pass the filter
int begin(hipStream_t s);                  // records the start event; returns the slot or -1
void end(int slot, hipStream_t s, double flops);
void aux(int slot, double a, double b, double c = 0.0);   // FLOPs of the slot that sit under a device-side row limit (15-unit / 16-unit products)
}  // namespace prof
}  // namespace ptv

// Raw-HIP reproduction of the nested fork/join capture pattern (side stream s1 forks s2 and joins it back), to see which call dies.
// build: hipcc --offload-arch=gfx950 -o graph_nested graph_nested.hip ; run: ./graph_nested [variant]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("ERR %s line %d: %s\n", #x, __LINE__, hipGetErrorString(e)); return 1; } } while (0)
__global__ void add1(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.f; }
int main(int argc, char** argv) {
  const char* v = argc > 1 ? argv[1] : "nested";
  float *a, *b, *c; const int n = 1 << 20;
  CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&c, n * 4));
  hipStream_t m, s1, s2;
  CK(hipStreamCreateWithFlags(&m, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t e0, e1, e2, e3, e4;
  CK(hipEventCreateWithFlags(&e0, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&e2, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&e3, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&e4, hipEventDisableTiming));
  hipStreamCaptureMode mode = !strcmp(v, "relaxed") ? hipStreamCaptureModeRelaxed : hipStreamCaptureModeGlobal;
  CK(hipStreamBeginCapture(m, mode));
  add1<<<n / 256, 256, 0, m>>>(a, n);
  CK(hipEventRecord(e0, m)); CK(hipStreamWaitEvent(s1, e0, 0));               // fork s1 from m
  add1<<<n / 256, 256, 0, s1>>>(b, n);
  CK(hipEventRecord(e1, s1)); CK(hipStreamWaitEvent(s2, e1, 0));              // fork s2 from s1
  add1<<<n / 256, 256, 0, s2>>>(c, n);
  if (!strcmp(v, "work_between")) add1<<<n / 256, 256, 0, s1>>>(b, n);
  CK(hipEventRecord(e2, s2)); CK(hipStreamWaitEvent(s1, e2, 0));              // join s2 into s1
  add1<<<n / 256, 256, 0, s1>>>(b, n);
  CK(hipEventRecord(e3, s1)); CK(hipStreamWaitEvent(m, e3, 0));               // join s1 into m
  if (!strcmp(v, "join_both")) { CK(hipEventRecord(e4, s2)); CK(hipStreamWaitEvent(m, e4, 0)); }
  add1<<<n / 256, 256, 0, m>>>(a, n);
  hipGraph_t g;
  printf("ending capture\n"); fflush(stdout);
  CK(hipStreamEndCapture(m, &g));
  printf("capture ended\n"); fflush(stdout);
  hipGraphExec_t ge;
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  printf("instantiated\n"); fflush(stdout);
  CK(hipGraphLaunch(ge, m)); CK(hipStreamSynchronize(m));
  printf("OK %s\n", v);
  return 0;
}

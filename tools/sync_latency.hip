// How long does the host wait for a tiny launch?  hipStreamSynchronize against polling an event and against spinning on a word in mapped
// host memory that a kernel / a stream write sets.   hipcc --offload-arch=gfx950 -O2 -o build/sync_latency tools/sync_latency.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void tiny(volatile uint32_t* flag, uint32_t v, float* sink) {
  if (threadIdx.x == 0 && blockIdx.x == 0) { sink[0] += 1.0f; if (flag) { __threadfence_system(); *flag = v; } }
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  float* sink; CK(hipMalloc(&sink, 4));
  uint32_t* flag_h; CK(hipHostMalloc((void**)&flag_h, 64, hipHostMallocMapped));
  uint32_t* flag_d; CK(hipHostGetDevicePointer((void**)&flag_d, flag_h, 0));
  hipEvent_t ev; CK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  const int K = 5000;
  for (int mode = 0; mode < 4; ++mode) {
    *flag_h = 0;
    for (int w = 0; w < 200; ++w) { hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st, nullptr, 0u, sink); CK(hipStreamSynchronize(st)); }
    const double t0 = now();
    for (int k = 1; k <= K; ++k) {
      if (mode == 0) { hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st, nullptr, 0u, sink); CK(hipStreamSynchronize(st)); }
      if (mode == 1) { hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st, nullptr, 0u, sink); CK(hipEventRecord(ev, st)); while (hipEventQuery(ev) == hipErrorNotReady) {} }
      if (mode == 2) { hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st, flag_d, (uint32_t)k, sink); while (*(volatile uint32_t*)flag_h != (uint32_t)k) {} }
      if (mode == 3) { hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st, nullptr, 0u, sink); hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, st, nullptr, 0u, sink); CK(hipStreamSynchronize(st)); }
    }
    const double dt = (now() - t0) / K;
    CK(hipStreamSynchronize(st));
    const char* what[] = {"launch + hipStreamSynchronize", "launch + hipEventRecord + spin on hipEventQuery", "launch + spin on a word the kernel sets in mapped host memory",
                          "two launches + hipStreamSynchronize"};
    printf("%-66s %7.2f us per iteration\n", what[mode], dt);
  }
  return 0;
}

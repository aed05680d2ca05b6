// hbm_calib.hip -- measured HBM copy bandwidth on this GPU and calibration kernels for the rocprofv3
// FETCH_SIZE / WRITE_SIZE counters (MI355X_MICROARCH.md "HBM": FETCH_SIZE reads 1/2 of the bytes of a
// 16 B/lane stream on gfx950; other access widths must be calibrated on a known byte count).
// The three kernels move a known number of bytes with the access widths the step kernel uses:
//   copy16: 16 B/lane loads+stores (float4)   copy8: 8 B/lane (double planes)   copy4: 4 B/lane (float planes)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/hbm_calib tools/hbm_calib.hip ; run: tools/hbm_calib [MiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <typename T>
__global__ __launch_bounds__(256) void copy_kernel(const T* __restrict__ a, T* __restrict__ b, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) b[i] = a[i];
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <typename T>
int run(const char* name, void* a, void* b, size_t bytes, int reps) {
  size_t n = bytes / sizeof(T);
  dim3 grid((unsigned)((n + 255) / 256)), block(256);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(copy_kernel<T>, grid, block, 0, 0, (const T*)a, (T*)b, n);
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(copy_kernel<T>, grid, block, 0, 0, (const T*)a, (T*)b, n);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  double gbps = 2.0 * bytes * reps / (ms * 1e-3) / 1e9;
  printf("{\"kernel\": \"%s\", \"bytes_read_per_launch\": %zu, \"bytes_written_per_launch\": %zu, \"launches\": %d, "
         "\"avg_us\": %.2f, \"GBps_read_plus_write\": %.1f}\n", name, bytes, bytes, reps + 3, ms * 1e3 / reps, gbps);
  return 0;
}

int main(int argc, char** argv) {
  size_t mib = argc > 1 ? (size_t)atol(argv[1]) : 1024;
  size_t bytes = mib << 20;
  void *a, *b;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
  CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
  if (run<float4>("copy16", a, b, bytes, 20)) return 1;
  if (run<double>("copy8", a, b, bytes, 20)) return 1;
  if (run<float>("copy4", a, b, bytes, 20)) return 1;
  return 0;
}

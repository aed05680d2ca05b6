// rtz_check.hip -- which field of MODE.FP_ROUND does v_cvt_f32_f64 obey on gfx950, and does a round-toward-zero conversion equal
// quad_core.hpp's split_hi (round to nearest, then one ulp back toward zero if it rounded away) bit for bit?
//   hipcc --offload-arch=gfx950 -O2 -o tools/rtz_check tools/rtz_check.hip && tools/rtz_check
// MODE[1:0] = single-precision round mode, MODE[3:2] = double / half precision round mode (0 nearest even, 1 +inf, 2 -inf, 3 zero).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

__device__ __forceinline__ float split_hi_ref(double v) {
  const float h = (float)v;
  uint32_t hb = __builtin_bit_cast(uint32_t, h);
  if (fabs((double)h) > fabs(v)) hb -= 1u;
  return __builtin_bit_cast(float, hb);
}

template <int FIELD>   // 0: set MODE[1:0] = 3; 1: set MODE[3:2] = 3
__device__ __forceinline__ float cvt_rtz(double v) {
  float h;
  if constexpr (FIELD == 0)
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n\tv_cvt_f32_f64 %0, %1\n\ts_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0" : "=v"(h) : "v"(v));
  else
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3\n\tv_cvt_f32_f64 %0, %1\n\ts_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 0" : "=v"(h) : "v"(v));
  return h;
}

__global__ void check(const double* in, int n, float* ref, float* a, float* b, float* after) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  ref[i] = split_hi_ref(in[i]);
  a[i] = cvt_rtz<0>(in[i]);
  b[i] = cvt_rtz<1>(in[i]);
  after[i] = (float)in[i];          // an ordinary conversion after the mode was restored: round to nearest again
}

int main() {
  const int n = 1 << 20;
  std::vector<double> h(n);
  uint64_t s = 88172645463325252ull;
  for (int i = 0; i < n; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    double v = (double)(int64_t)(s >> 11) / 9007199254740992.0;      // [0, 1)
    const int e = (int)((s >> 3) % 60) - 40;
    v = std::ldexp(v * 2.0 - 1.0, e);
    if (i % 97 == 0) v = 0.0;
    if (i % 101 == 0) v = (double)(float)v;                           // exactly representable: nothing to round
    if (i % 103 == 0) v = std::ldexp(v, -140);                         // float denormals / underflow
    h[i] = v;
  }
  double* d; float *r, *a, *b, *c;
  hipMalloc(&d, n * 8); hipMalloc(&r, n * 4); hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&c, n * 4);
  hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(check, dim3(n / 256), dim3(256), 0, 0, d, n, r, a, b, c);
  std::vector<float> hr(n), ha(n), hb(n), hc(n);
  hipMemcpy(hr.data(), r, n * 4, hipMemcpyDeviceToHost); hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost);
  hipMemcpy(hb.data(), b, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hc.data(), c, n * 4, hipMemcpyDeviceToHost);
  long bad_a = 0, bad_b = 0, bad_c = 0, differ_rn = 0;
  for (int i = 0; i < n; ++i) {
    bad_a += std::memcmp(&hr[i], &ha[i], 4) != 0;
    bad_b += std::memcmp(&hr[i], &hb[i], 4) != 0;
    const float rn = (float)h[i];
    bad_c += std::memcmp(&rn, &hc[i], 4) != 0;
    differ_rn += std::memcmp(&rn, &hr[i], 4) != 0;
  }
  printf("{\"n\": %d, \"truncation_differs_from_nearest\": %ld, \"mismatch_with_MODE_1_0\": %ld, \"mismatch_with_MODE_3_2\": %ld, "
         "\"nearest_after_restore_mismatch\": %ld}\n", n, differ_rn, bad_a, bad_b, bad_c);
  return 0;
}

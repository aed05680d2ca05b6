// Are the fp64 operations the step kernel relies on correctly rounded on this device / with this compiler?
// Compares sqrt, division, fma and fmin/fmax on random doubles against the host's IEEE results, bit for bit.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <random>
#include <vector>
__global__ void k(const double* a, const double* b, const double* c, double* o, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  o[0 * n + i] = sqrt(a[i]);
  o[1 * n + i] = a[i] / b[i];
  o[2 * n + i] = fma(a[i], b[i], c[i]);
  o[3 * n + i] = a[i] * b[i] + c[i];            // contraction is the compiler's choice
  o[4 * n + i] = 1.0 / b[i];
  o[5 * n + i] = (double)sqrtf((float)a[i]);
}
int main() {
  const int n = 1 << 20;
  std::mt19937_64 g(1);
  std::uniform_real_distribution<double> u(0.0, 1.0), w(-40.0, 40.0);
  std::vector<double> a(n), b(n), c(n), o(6 * n);
  for (int i = 0; i < n; ++i) { a[i] = u(g); b[i] = 0.01 + u(g); c[i] = w(g); }
  double *da, *db, *dc, *dout;
  hipMalloc(&da, n * 8); hipMalloc(&db, n * 8); hipMalloc(&dc, n * 8); hipMalloc(&dout, 6 * n * 8);
  hipMemcpy(da, a.data(), n * 8, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 8, hipMemcpyHostToDevice);
  hipMemcpy(dc, c.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, da, db, dc, dout, n);
  hipMemcpy(o.data(), dout, 6 * n * 8, hipMemcpyDeviceToHost);
  long bad[6] = {0, 0, 0, 0, 0, 0}; double worst[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    const double r[6] = {std::sqrt(a[i]), a[i] / b[i], std::fma(a[i], b[i], c[i]), a[i] * b[i] + c[i], 1.0 / b[i],
                         (double)std::sqrt((float)a[i])};
    for (int j = 0; j < 6; ++j) {
      const double d = o[(size_t)j * n + i];
      if (std::memcmp(&d, &r[j], 8) != 0) { bad[j]++; const double e = std::fabs(d - r[j]) / std::fabs(r[j]); if (e > worst[j]) worst[j] = e; }
    }
  }
  const char* nm[6] = {"sqrt", "div", "fma", "a*b+c (uncontracted on host)", "1/x", "sqrtf"};
  for (int j = 0; j < 6; ++j) std::printf("%-30s mismatches %ld / %d   worst rel %.3g\n", nm[j], bad[j], n, worst[j]);
  return 0;
}

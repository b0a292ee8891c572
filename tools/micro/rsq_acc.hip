// Accuracy of v_rsq_f64 and of one / two Newton steps on it (is the second step of chol.hip's rsqrt_nr needed?).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double* d, double* y0, double* y1, double* y2, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double x = d[i];
  double y = __builtin_amdgcn_rsq(x);
  y0[i] = y;
  double e = fma(-x * y, y, 1.0);
  y = fma(0.5 * y, e, y);
  y1[i] = y;
  e = fma(-x * y, y, 1.0);
  y = fma(0.5 * y, e, y);
  y2[i] = y;
}
int main() {
  const int n = 1 << 20;
  std::vector<double> h(n), a(n), b(n), c(n);
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; i++) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    double u = (double)(s >> 11) / 9007199254740992.0;          // [0,1)
    h[i] = std::ldexp(1.0 + u, (int)(s % 200) - 100);            // 2^-100 .. 2^100
  }
  double *d, *y0, *y1, *y2;
  hipMalloc(&d, n * 8); hipMalloc(&y0, n * 8); hipMalloc(&y1, n * 8); hipMalloc(&y2, n * 8);
  hipMemcpy(d, h.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, y0, y1, y2, n);
  hipMemcpy(a.data(), y0, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(b.data(), y1, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(c.data(), y2, n * 8, hipMemcpyDeviceToHost);
  double m0 = 0, m1 = 0, m2 = 0;
  for (int i = 0; i < n; i++) {
    long double ex = 1.0L / sqrtl((long double)h[i]);
    m0 = fmax(m0, (double)fabsl((a[i] - ex) / ex));
    m1 = fmax(m1, (double)fabsl((b[i] - ex) / ex));
    m2 = fmax(m2, (double)fabsl((c[i] - ex) / ex));
  }
  printf("max relative error over %d inputs: v_rsq_f64 %.3e (2^%.1f), +1 Newton %.3e (%.2f ulp), +2 Newton %.3e (%.2f ulp)\n",
         n, m0, log2(m0), m1, m1 / 1.11e-16, m2, m2 / 1.11e-16);
  return 0;
}

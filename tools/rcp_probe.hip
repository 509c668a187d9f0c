// Accuracy of v_rcp_f64 on gfx950 and of 1 / 2 Newton steps on top of it (diagnostic).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double *x, double *r0, double *r1, double *r2, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double d = x[i];
  double a = __builtin_amdgcn_rcp(d);
  r0[i] = a;
  a = fma(fma(-d, a, 1.0), a, a);
  r1[i] = a;
  a = fma(fma(-d, a, 1.0), a, a);
  r2[i] = a;
}
int main() {
  const int n = 1 << 20;
  std::vector<double> x(n), r0(n), r1(n), r2(n);
  unsigned long long s = 88172645463325252ULL;
  for (int i = 0; i < n; i++) {
    s ^= s << 13, s ^= s >> 7, s ^= s << 17;
    double m = 1.0 + (double)(s >> 11) / 9007199254740992.0;  // [1,2)
    int e = (int)((s >> 3) % 80) - 40;
    x[i] = ldexp(m, e) * ((s & 1) ? -1 : 1);
  }
  double *dx, *d0, *d1, *d2;
  hipMalloc(&dx, n * 8), hipMalloc(&d0, n * 8), hipMalloc(&d1, n * 8), hipMalloc(&d2, n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, d0, d1, d2, n);
  hipMemcpy(r0.data(), d0, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(r1.data(), d1, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(r2.data(), d2, n * 8, hipMemcpyDeviceToHost);
  double e0 = 0, e1 = 0, e2 = 0;
  for (int i = 0; i < n; i++) {
    long double t = 1.0L / (long double)x[i];
    e0 = fmax(e0, (double)fabsl(((long double)r0[i] - t) / t));
    e1 = fmax(e1, (double)fabsl(((long double)r1[i] - t) / t));
    e2 = fmax(e2, (double)fabsl(((long double)r2[i] - t) / t));
  }
  printf("max rel err: rcp %.3e (2^%.1f)  +1 NR %.3e (%.2f ulp)  +2 NR %.3e (%.2f ulp)\n", e0, log2(e0), e1,
         e1 / 1.11e-16, e2, e2 / 1.11e-16);
  return 0;
}

// GPU box: accuracy of v_rcp_f64 on gfx950 and of one / two Newton steps on top of it (fast_rcp of csrc/dsge_device.hpp uses two).
//   hipcc --offload-arch=gfx950 -O2 -o tools/latency_probe/rcp_probe tools/latency_probe/rcp_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double* x, double* y0, double* y1, double* y2, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double v = x[i];
  double y = __builtin_amdgcn_rcp(v);
  y0[i] = y;
  y = y * fma(-v, y, 2.0);
  y1[i] = y;
  y = y * fma(-v, y, 2.0);
  y2[i] = y;
}
int main() {
  const int n = 1 << 20;
  std::vector<double> x(n), r0(n), r1(n), r2(n);
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const double u = (double)(s >> 11) / 9007199254740992.0;
    x[i] = ldexp(1.0 + u, (int)(s % 120) - 60) * ((s >> 3) & 1 ? 1.0 : -1.0);
  }
  double *dx, *d0, *d1, *d2;
  hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, n);
  hipMemcpy(r0.data(), d0, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(r1.data(), d1, n * 8, hipMemcpyDeviceToHost);
  hipMemcpy(r2.data(), d2, n * 8, hipMemcpyDeviceToHost);
  double e0 = 0, e1 = 0, e2 = 0;
  for (int i = 0; i < n; ++i) {
    const long double t = 1.0L / (long double)x[i];
    e0 = fmax(e0, (double)fabsl(((long double)r0[i] - t) / t));
    e1 = fmax(e1, (double)fabsl(((long double)r1[i] - t) / t));
    e2 = fmax(e2, (double)fabsl(((long double)r2[i] - t) / t));
  }
  printf("max relative error of 1/x over %d values: v_rcp_f64 %.3e, + one Newton step %.3e, + two %.3e (eps = 1.1e-16)\n", n, e0, e1, e2);
  return 0;
}

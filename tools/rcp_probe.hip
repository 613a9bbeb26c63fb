// Accuracy of v_rcp_f64 / v_rsq_f64 and of 1 vs 2 Newton steps on gfx950 (max error in ulp against the correctly rounded
// result computed in long double on the host).  build: hipcc --offload-arch=gfx950 -O3 -o tools/rcp_probe.bin tools/rcp_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void k(const double* x, double* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double v = x[i];
  double r0 = __builtin_amdgcn_rcp(v);
  double r1 = fma(fma(-v, r0, 1.0), r0, r0);
  double r2 = fma(fma(-v, r1, 1.0), r1, r1);
  double y0 = __builtin_amdgcn_rsq(v);
  double e = fma(-(v * y0), y0, 1.0);
  double y1 = fma(0.5 * y0, e, y0);
  e = fma(-(v * y1), y1, 1.0);
  double y2 = fma(0.5 * y1, e, y1);
  out[i] = r0; out[n + i] = r1; out[2 * n + i] = r2; out[3 * n + i] = y0; out[4 * n + i] = y1; out[5 * n + i] = y2;
}

static double ulp_err(double got, long double exact) {
  const double ex = (double)exact;
  const double u = std::nextafter(std::fabs(ex), INFINITY) - std::fabs(ex);
  return (double)(std::fabs((long double)got - exact) / u);
}

int main() {
  const int n = 1 << 22;
  std::vector<double> x(n), out(6 * (size_t)n);
  uint64_t s = 0x9E3779B97F4A7C15ull;
  for (int i = 0; i < n; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const double m = 1.0 + (double)(s >> 11) / 9007199254740992.0;       // [1, 2)
    const int e = (int)((s >> 3) % 80) - 40;
    x[i] = std::ldexp(m, e);
  }
  double *dx, *dout;
  hipMalloc(&dx, n * 8); hipMalloc(&dout, 6 * (size_t)n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, dout, n);
  hipMemcpy(out.data(), dout, 6 * (size_t)n * 8, hipMemcpyDeviceToHost);
  double mx[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < n; ++i) {
    const long double r = 1.0L / (long double)x[i], y = 1.0L / sqrtl((long double)x[i]);
    for (int c = 0; c < 3; ++c) mx[c] = std::fmax(mx[c], ulp_err(out[(size_t)c * n + i], r));
    for (int c = 3; c < 6; ++c) mx[c] = std::fmax(mx[c], ulp_err(out[(size_t)c * n + i], y));
  }
  printf("max error in ulp over %d samples:  rcp raw %.3g  +1 Newton %.3g  +2 Newton %.3g   |  rsq raw %.3g  +1 Newton %.3g  +2 Newton %.3g\n", n,
         mx[0], mx[1], mx[2], mx[3], mx[4], mx[5]);
  return 0;
}

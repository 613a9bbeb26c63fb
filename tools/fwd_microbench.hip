// Micro-benchmark (development tool, not part of the product): where does one closed-loop Cartpole rollout spend
// its time?  Variants switch off pieces of the forward-pass inner loop.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../trajectoryoptimization.jl_amd/csrc/models.h"
using namespace to;
#define EL(p, e) (p)[(size_t)(e) * 64]
template <int LOADS, int STORES, int COST, int DYN>
__global__ void __launch_bounds__(64) k(const double* X, const double* U, const double* K, const double* d, double* Xn, double* Un, double* out,
                                       const double* Qs, int N, double h, const double* mp) {
  using M = CartpoleModel; constexpr int n = 4, m = 1, ne = 4;
  const int tile = blockIdx.x, lane = threadIdx.x;
  const double* pX = X + (size_t)tile * N * n * 64 + lane; const double* pU = U + (size_t)tile * (N - 1) * 64 + lane;
  const double* pK = K + (size_t)tile * (N - 1) * 4 * 64 + lane; const double* pd = d + (size_t)tile * (N - 1) * 64 + lane;
  double* pXn = Xn + (size_t)(blockIdx.y * gridDim.x + tile) * N * n * 64 + lane; double* pUn = Un + (size_t)(blockIdx.y * gridDim.x + tile) * (N - 1) * 64 + lane;
  double P[16]; for (int i = 0; i < 4; ++i) P[i] = mp[i];
  double xb[n] = {0.01 * lane, 0.02, 0, 0}, J = 0;
  double xk[n], Kk[ne], dk, uk;
  for (int i = 0; i < n; ++i) xk[i] = LOADS ? EL(pX, i) : 0.0;
  for (int i = 0; i < ne; ++i) Kk[i] = LOADS ? EL(pK, i) : 0.1;
  dk = LOADS ? EL(pd, 0) : 0.1; uk = LOADS ? EL(pU, 0) : 0.01;
  for (int k = 0; k < N - 1; ++k) {
    double cx[n], cK[ne], cd = dk, cu = uk;
    for (int i = 0; i < n; ++i) { cx[i] = xk[i]; cK[i] = Kk[i]; }
    if (LOADS && k + 1 < N - 1) {
      for (int i = 0; i < n; ++i) xk[i] = EL(pX, (k + 1) * n + i);
      for (int i = 0; i < ne; ++i) Kk[i] = EL(pK, (k + 1) * ne + i);
      dk = EL(pd, k + 1); uk = EL(pU, k + 1);
    }
    double dx[ne], ub[m], xn[n];
    for (int i = 0; i < n; ++i) dx[i] = xb[i] - cx[i];
    double du = cd * 0.5; for (int i = 0; i < ne; ++i) du += cK[i] * dx[i];
    ub[0] = cu + du;
    if (STORES) EL(pUn, k) = ub[0];
    if (COST) { double c = 0; for (int i = 0; i < n; ++i) c += xb[i] * Qs[i] * xb[i] + Qs[4 + i] * xb[i]; J += 0.5 * c + Qs[8] + 0.5 * ub[0] * Qs[9] * ub[0]; J += fabs(cd) / (fabs(ub[0]) + 1.0); }
    if (DYN) rk_step<M, double>(P, 0, xb, ub, h, xn); else for (int i = 0; i < n; ++i) xn[i] = xb[i] * 0.999 + ub[0] * 1e-3;
    for (int i = 0; i < n; ++i) { xb[i] = xn[i]; if (STORES) EL(pXn, (k + 1) * n + i) = xn[i]; }
  }
  out[(blockIdx.y * gridDim.x + tile) * 64 + lane] = J + xb[0] + xb[1];
}
template <int A, int B, int C, int D> float run(int tiles, int T, int N, const double* X, const double* U, const double* K, const double* d, double* Xn, double* Un, double* out, const double* Qs, const double* mp) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<A, B, C, D>), dim3(tiles, T), dim3(64), 0, 0, X, U, K, d, Xn, Un, out, Qs, N, 0.05, mp);
  hipEventRecord(e0); const int R = 20;
  for (int r = 0; r < R; ++r) hipLaunchKernelGGL((k<A, B, C, D>), dim3(tiles, T), dim3(64), 0, 0, X, U, K, d, Xn, Un, out, Qs, N, 0.05, mp);
  hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); return ms / R * 1000;
}
int main() {
  const int tiles = 16, N = 101, Tmax = 20; size_t sz = (size_t)tiles * N * 4 * 64;
  double *X, *U, *K, *d, *Xn, *Un, *out, *Qs, *mp;
  hipMalloc(&X, sz * 8); hipMalloc(&U, sz * 8); hipMalloc(&K, sz * 8); hipMalloc(&d, sz * 8); hipMalloc(&Xn, sz * 8 * Tmax); hipMalloc(&Un, sz * 8 * Tmax); hipMalloc(&out, 64 * tiles * Tmax * 8);
  hipMalloc(&Qs, 16 * 8); hipMalloc(&mp, 16 * 8);
  std::vector<double> hx(sz, 0.01); hipMemcpy(X, hx.data(), sz * 8, hipMemcpyHostToDevice); hipMemcpy(U, hx.data(), sz * 8, hipMemcpyHostToDevice);
  hipMemcpy(K, hx.data(), sz * 8, hipMemcpyHostToDevice); hipMemcpy(d, hx.data(), sz * 8, hipMemcpyHostToDevice);
  double q[16] = {0.01, 0.01, 0.01, 0.01, 0, -0.03, 0, 0, 0.05, 0.1}; hipMemcpy(Qs, q, sizeof(q), hipMemcpyHostToDevice);
  double p[16] = {1.0, 0.2, 0.5, 9.81}; hipMemcpy(mp, p, sizeof(p), hipMemcpyHostToDevice);
#define RUN(A, B, C, D, T) printf("loads=%d stores=%d cost=%d dyn=%d T=%2d : %8.1f us\n", A, B, C, D, T, run<A, B, C, D>(tiles, T, N, X, U, K, d, Xn, Un, out, Qs, mp));
  RUN(1, 1, 1, 1, 1) RUN(0, 1, 1, 1, 1) RUN(1, 0, 1, 1, 1) RUN(1, 1, 0, 1, 1) RUN(1, 1, 1, 0, 1) RUN(0, 0, 0, 1, 1) RUN(0, 0, 0, 0, 1)
  RUN(1, 1, 1, 1, 4) RUN(1, 1, 1, 1, 8) RUN(1, 1, 1, 1, 16) RUN(1, 1, 1, 1, 20)
  return 0;
}

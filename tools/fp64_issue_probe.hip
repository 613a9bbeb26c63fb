// Development probe: FP64 FMA issue rate per wave as a function of how many single-wave workgroups are in flight.
// Answers "do the 4 SIMDs of a CU run FP64 independently?" — build: hipcc --offload-arch=gfx950 -O3 -o tools/fp64_issue_probe.bin tools/fp64_issue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int ILP>
__global__ __launch_bounds__(64) void k_fma(double* out, int iters, double a, double b) {
  double x[ILP];
#pragma unroll
  for (int i = 0; i < ILP; ++i) x[i] = threadIdx.x * 1e-3 + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
      for (int i = 0; i < ILP; ++i) x[i] = __builtin_fma(x[i], a, b);
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < ILP; ++i) s += x[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int ILP>
void run(const char* name) {
  double* out; hipMalloc(&out, sizeof(double) * 64 * 8192);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000;
  for (int waves : {64, 128, 256, 512, 1024, 2048, 4096}) {
    k_fma<ILP><<<waves, 64>>>(out, 100, 0.999, 1e-3);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_fma<ILP><<<waves, 64>>>(out, iters, 0.999, 1e-3);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double nfma = (double)iters * 16 * ILP;
    printf("%s waves=%5d  %.3f ms  %.2f ns/FMA/wave  (%.1f cycles @2.4GHz)  %.2f TFLOP/s\n", name, waves, ms, ms * 1e6 / nfma,
           ms * 1e6 / nfma * 2.4, 2.0 * nfma * 64 * waves / (ms * 1e-3) / 1e12);
  }
  hipFree(out);
}

// partially filled waves: does the FP64 issue rate depend on the EXEC mask?
template <int ILP>
__global__ __launch_bounds__(64) void k_fma_masked(double* out, int iters, double a, double b, unsigned long long mask) {
  if (!((mask >> threadIdx.x) & 1ull)) return;
  double x[ILP];
#pragma unroll
  for (int i = 0; i < ILP; ++i) x[i] = threadIdx.x * 1e-3 + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
#pragma unroll
      for (int i = 0; i < ILP; ++i) x[i] = __builtin_fma(x[i], a, b);
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < ILP; ++i) s += x[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
void masked() {
  double* out; (void)hipMalloc(&out, sizeof(double) * 64 * 8192);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const unsigned long long masks[] = {~0ull, 0xffffffffull, 0xffffull, 0xfull, 0x1ull, 0x8000000000000001ull, 0x0001000100010001ull, 0x1111111111111111ull};
  for (unsigned long long mk : masks) {
    k_fma_masked<4><<<256, 64>>>(out, 100, 0.999, 1e-3, mk);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k_fma_masked<4><<<256, 64>>>(out, 4000, 0.999, 1e-3, mk);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("mask %016llx: %.3f ms  (%.2f cycles/FMA @2.4GHz)\n", mk, ms, ms * 1e6 / (4000.0 * 16 * 4) * 2.4);
  }
}

// sustained light load: does the clock governor slow a nearly idle chip?  (few live waves for many milliseconds)
void sustained(int waves) {
  double* out; hipMalloc(&out, sizeof(double) * 64 * 8192);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 8000;  // ~0.7 ms per launch
  for (int rep = 0; rep < 6; ++rep) {
    hipEventRecord(e0);
    for (int l = 0; l < 50; ++l) k_fma<4><<<waves, 64>>>(out, iters, 0.999, 1e-3);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("sustained waves=%4d rep %d: %.3f ms per launch\n", waves, rep, ms / 50);
  }
  hipFree(out);
}

int main(int argc, char** argv) {
  if (argc > 1 && argv[1][0] == 'm') { masked(); return 0; }
  if (argc > 1) { sustained(1024); sustained(16); sustained(1024); sustained(16); sustained(256); return 0; }
  run<1>("ILP1");
  run<4>("ILP4");
  run<8>("ILP8");
  return 0;
}

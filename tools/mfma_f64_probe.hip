// Development probe for the MFMA Riccati kernel: checks the operand / result lane maps of
// v_mfma_f64_16x16x4_f64 against a scalar product with ASYMMETRIC inputs, the "result register r is K-slice r" chaining
// property the backward pass relies on, and times dependent / independent MFMA chains, ds_bpermute and v_readlane.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/mfma_f64_probe.bin tools/mfma_f64_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

typedef double v4d __attribute__((ext_vector_type(4)));

// D = A(16x16) * B(16x16) through 4 K-slices.  Operands are given in "result layout": lane (g,c) register r holds X[g+4r][c].
// A is passed TRANSPOSED in result layout (At[k][i]), so that slice s of the A operand (A[i=c][k=4s+g]) is register s.
__global__ void k_layout(const double* At, const double* B, double* D) {
  const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
  v4d acc = {0, 0, 0, 0};
  double a[4], b[4];
  for (int r = 0; r < 4; ++r) { a[r] = At[(g + 4 * r) * 16 + c]; b[r] = B[(g + 4 * r) * 16 + c]; }
#pragma unroll
  for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[s], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[(g + 4 * r) * 16 + c] = acc[r];
}

// chained: E = (A*B)^T-free chain  E = Ct * (A*B) with the intermediate used straight from its result registers
__global__ void k_chain(const double* At, const double* B, const double* Ct, double* E) {
  const int lane = threadIdx.x, g = lane >> 4, c = lane & 15;
  v4d t = {0, 0, 0, 0}, e = {0, 0, 0, 0};
  double a[4], b[4], ct[4];
  for (int r = 0; r < 4; ++r) { a[r] = At[(g + 4 * r) * 16 + c]; b[r] = B[(g + 4 * r) * 16 + c]; ct[r] = Ct[(g + 4 * r) * 16 + c]; }
#pragma unroll
  for (int s = 0; s < 4; ++s) t = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b[s], t, 0, 0, 0);
#pragma unroll
  for (int s = 0; s < 4; ++s) e = __builtin_amdgcn_mfma_f64_16x16x4f64(ct[s], t[s], e, 0, 0, 0);  // e = C * T with C given transposed
  for (int r = 0; r < 4; ++r) E[(g + 4 * r) * 16 + c] = e[r];
}

template <int MODE>  // 0: dependent MFMA chain, 1: 4 independent accumulators, 2: bpermute chain, 3: readlane chain, 4: dependent FMA chain
__global__ __launch_bounds__(64) void k_time(double* out, int iters, double x0) {
  const int lane = threadIdx.x;
  double a = x0 + lane * 1e-3, b = 1.0 - lane * 1e-4;
  v4d acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  double v = a;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int r = 0; r < 8; ++r) acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[0], 0, 0, 0);
    } else if (MODE == 1) {
#pragma unroll
      for (int r = 0; r < 8; ++r) acc[r & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[r & 3], 0, 0, 0);
    } else if (MODE == 2) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        int lo = __builtin_amdgcn_ds_bpermute(((lane + 16) & 63) * 4, __double2loint(v));
        int hi = __builtin_amdgcn_ds_bpermute(((lane + 16) & 63) * 4, __double2hiint(v));
        v = __hiloint2double(hi, lo) + 1.0;
      }
    } else if (MODE == 3) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        int lo = __builtin_amdgcn_readlane(__double2loint(v), 13);
        int hi = __builtin_amdgcn_readlane(__double2hiint(v), 13);
        v = __hiloint2double(hi, lo) + a;
      }
    } else {
#pragma unroll
      for (int r = 0; r < 8; ++r) v = __builtin_fma(v, b, a);
    }
  }
  out[blockIdx.x * 64 + lane] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + v;
}

template <int MODE>
void timeit(const char* name, double* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int waves : {256, 1024, 2048, 4096}) {
    k_time<MODE><<<waves, 64>>>(out, 100, 0.5);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_time<MODE><<<waves, 64>>>(out, iters, 0.5);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s waves=%5d  %.1f cycles per op per wave-slot @2.4GHz (%.3f ms)\n", name, waves, ms * 1e-3 * 2.4e9 / (iters * 8.0), ms);
  }
}

int main() {
  std::vector<double> A(256), B(256), Cm(256), At(256), Ct(256), D(256), E(256), Dref(256), Eref(256);
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      A[i * 16 + j] = std::sin(1.0 + 3 * i + 0.7 * j); B[i * 16 + j] = std::cos(0.3 * i * i + 1.1 * j) + 0.01 * i; Cm[i * 16 + j] = 0.1 * i - 0.37 * j + 0.5;
    }
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) { At[j * 16 + i] = A[i * 16 + j]; Ct[j * 16 + i] = Cm[i * 16 + j]; }
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 16; ++k) s += A[i * 16 + k] * B[k * 16 + j]; Dref[i * 16 + j] = s; }
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 16; ++k) s += Cm[i * 16 + k] * Dref[k * 16 + j]; Eref[i * 16 + j] = s; }
  double *dA, *dB, *dC, *dD, *dE, *dout;
  hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dC, 2048); hipMalloc(&dD, 2048); hipMalloc(&dE, 2048); hipMalloc(&dout, 8 * 64 * 4096);
  hipMemcpy(dA, At.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dC, Ct.data(), 2048, hipMemcpyHostToDevice);
  k_layout<<<1, 64>>>(dA, dB, dD);
  k_chain<<<1, 64>>>(dA, dB, dC, dE);
  hipMemcpy(D.data(), dD, 2048, hipMemcpyDeviceToHost); hipMemcpy(E.data(), dE, 2048, hipMemcpyDeviceToHost);
  double e1 = 0, e2 = 0;
  for (int i = 0; i < 256; ++i) { e1 = std::fmax(e1, std::fabs(D[i] - Dref[i])); e2 = std::fmax(e2, std::fabs(E[i] - Eref[i])); }
  printf("layout check: max |D - A*B| = %.3e   chained max |E - C*(A*B)| = %.3e   (%s)\n", e1, e2, (e1 < 1e-12 && e2 < 1e-11) ? "LAYOUT OK" : "LAYOUT WRONG");
  timeit<0>("mfma f64 16x16x4 dependent", dout);
  timeit<1>("mfma f64 16x16x4 4 accs", dout);
  timeit<2>("bpermute pair + add chain", dout);
  timeit<3>("readlane pair + add chain", dout);
  timeit<4>("dependent fma f64", dout);
  return 0;
}

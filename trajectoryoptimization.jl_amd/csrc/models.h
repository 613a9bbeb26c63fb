// models.h — device-side model dynamics, integrators and error-state maps for gfx950.
//
// Every model is written ONCE, templated on the scalar type: `double` for rollouts and `Dual`
// (value + one directional derivative) for the expansion kernel, which evaluates one column of the
// exact RK Jacobian per thread by forward-mode differentiation through all RK stages — the same
// construction ForwardDiff applies to RobotDynamics' discretised dynamics.
//
// Restated (not ported) from: examples/quickstart.jl:11-23 (double integrator), docs/src/model.md:20-51
// (Cartpole), examples/Quadrotor.ipynb cells 4,8 + RobotDynamics RigidBody (Quadrotor).
#pragma once
#include <hip/hip_runtime.h>

namespace to {

struct Dual {
  double v, d;
  __host__ __device__ Dual() : v(0.0), d(0.0) {}
  __host__ __device__ Dual(double a) : v(a), d(0.0) {}
  __host__ __device__ Dual(double a, double b) : v(a), d(b) {}
};
__device__ __forceinline__ Dual operator+(Dual a, Dual b) { return Dual(a.v + b.v, a.d + b.d); }
__device__ __forceinline__ Dual operator-(Dual a, Dual b) { return Dual(a.v - b.v, a.d - b.d); }
__device__ __forceinline__ Dual operator-(Dual a) { return Dual(-a.v, -a.d); }
__device__ __forceinline__ Dual operator*(Dual a, Dual b) { return Dual(a.v * b.v, a.d * b.v + a.v * b.d); }
__device__ __forceinline__ Dual operator/(Dual a, Dual b) {
  double q = a.v / b.v;
  return Dual(q, (a.d - q * b.d) / b.v);
}
__device__ __forceinline__ Dual operator+(Dual a, double b) { return Dual(a.v + b, a.d); }
__device__ __forceinline__ Dual operator+(double a, Dual b) { return Dual(a + b.v, b.d); }
__device__ __forceinline__ Dual operator-(Dual a, double b) { return Dual(a.v - b, a.d); }
__device__ __forceinline__ Dual operator-(double a, Dual b) { return Dual(a - b.v, -b.d); }
__device__ __forceinline__ Dual operator*(Dual a, double b) { return Dual(a.v * b, a.d * b); }
__device__ __forceinline__ Dual operator*(double a, Dual b) { return Dual(a * b.v, a * b.d); }
__device__ __forceinline__ Dual operator/(Dual a, double b) { return Dual(a.v / b, a.d / b); }

__device__ __forceinline__ void sincos_t(double x, double* s, double* c) { sincos(x, s, c); }
__device__ __forceinline__ void sincos_t(Dual x, Dual* s, Dual* c) {
  double sv, cv;
  sincos(x.v, &sv, &cv);
  *s = Dual(sv, cv * x.d);
  *c = Dual(cv, -sv * x.d);
}
// max(0, x): derivative is the indicator x > 0 (the rotor-force clamp of the Quadrotor)
__device__ __forceinline__ double relu_t(double x) { return fmax(0.0, x); }
__device__ __forceinline__ Dual relu_t(Dual x) { return x.v > 0.0 ? x : Dual(0.0, 0.0); }
__device__ __forceinline__ double val(double x) { return x; }
__device__ __forceinline__ double val(Dual x) { return x.v; }

// ------------------------------------------------------------------------------------------------
// Models.  P = model_params of the descriptor (wave-uniform, lives in SGPRs).
// ------------------------------------------------------------------------------------------------
template <int D>
struct DoubleIntegratorModel {  // examples/quickstart.jl:15-20
  static constexpr int n = 2 * D, m = D, ne = 2 * D;
  static constexpr bool lie = false;
  template <class T>
  __device__ __forceinline__ static void f(const double* P, const T* x, const T* u, T* xd) {
    const double mass = P[0];
#pragma unroll
    for (int i = 0; i < D; ++i) {
      xd[i] = x[D + i];
      xd[D + i] = u[i] / mass;
    }
  }
};

struct CartpoleModel {  // docs/src/model.md:34-50
  static constexpr int n = 4, m = 1, ne = 4;
  static constexpr bool lie = false;
  template <class T>
  __device__ __forceinline__ static void f(const double* P, const T* x, const T* u, T* xd) {
    const double mc = P[0], mp = P[1], l = P[2], g = P[3];
    T qd1 = x[2], qd2 = x[3];
    T s, c;
    sincos_t(x[1], &s, &c);
    const double h11 = mc + mp, h22 = mp * l * l;
    T h12 = (mp * l) * c;
    T c12 = -((mp * qd2) * l) * s;
    // b = C*qd + G - B*u
    T b1 = c12 * qd2 - u[0];
    T b2 = ((mp * g) * l) * s;
    // 2x2 solve as StaticArrays does it: ((a22 b1 - a12 b2)/d, (a11 b2 - a21 b1)/d)
    T d = h11 * h22 - h12 * h12;
    T s1 = (h22 * b1 - h12 * b2) / d;
    T s2 = (h11 * b2 - h12 * b1) / d;
    xd[0] = qd1;
    xd[1] = qd2;
    xd[2] = -s1;
    xd[3] = -s2;
  }
};

struct QuadrotorModel {  // RigidBody dynamics, world-frame velocity; state [r(3) q(w,x,y,z) v(3) ω(3)]
  static constexpr int n = 13, m = 4, ne = 12;
  static constexpr bool lie = true;
  template <class T>
  __device__ __forceinline__ static void f(const double* P, const T* x, const T* u, T* xd) {
    const double mass = P[0], J1 = P[1], J2 = P[2], J3 = P[3];
    const double g1 = P[4], g2 = P[5], g3 = P[6], L = P[7], kf = P[8], km = P[9];
    T qw = x[3], qx = x[4], qy = x[5], qz = x[6];
    T w1 = x[10], w2 = x[11], w3 = x[12];
    T F1 = relu_t(kf * u[0]), F2 = relu_t(kf * u[1]), F3 = relu_t(kf * u[2]), F4 = relu_t(kf * u[3]);
    T Fz = F1 + F2 + F3 + F4;
    // q*F with r=(0,0,Fz), NOT normalised: (w² − v·v) r + 2 v (v·r) + 2 w (v × r)   
    T vv = qx * qx + qy * qy + qz * qz;
    T sc = qw * qw - vv;
    T vr = qz * Fz;
    T qF1 = (2.0 * qx) * vr + (2.0 * qw) * (qy * Fz);
    T qF2 = (2.0 * qy) * vr - (2.0 * qw) * (qx * Fz);
    T qF3 = sc * Fz + (2.0 * qz) * vr;
    T t1 = L * (F2 - F4), t2 = L * (F3 - F1);
    T t3 = km * u[0] - km * u[1] + km * u[2] - km * u[3];
    xd[0] = x[7];
    xd[1] = x[8];
    xd[2] = x[9];
    xd[3] = 0.5 * (-(qx * w1) - qy * w2 - qz * w3);
    xd[4] = 0.5 * (qw * w1 + qy * w3 - qz * w2);
    xd[5] = 0.5 * (qw * w2 - qx * w3 + qz * w1);
    xd[6] = 0.5 * (qw * w3 + qx * w2 - qy * w1);
    xd[7] = (mass * g1 + qF1) / mass;
    xd[8] = (mass * g2 + qF2) / mass;
    xd[9] = (mass * g3 + qF3) / mass;
    T Jw1 = J1 * w1, Jw2 = J2 * w2, Jw3 = J3 * w3;
    xd[10] = (1.0 / J1) * (t1 - (w2 * Jw3 - w3 * Jw2));
    xd[11] = (1.0 / J2) * (t2 - (w3 * Jw1 - w1 * Jw3));
    xd[12] = (1.0 / J3) * (t3 - (w1 * Jw2 - w2 * Jw1));
  }
};

// ------------------------------------------------------------------------------------------------
// Integrators: h multiplied into each stage, then combined.
// ------------------------------------------------------------------------------------------------
enum { INTEG_RK4 = 0, INTEG_RK3 = 1, INTEG_EULER = 2 };

template <class M, class T>
__device__ __forceinline__ void rk_step(const double* P, int integrator, const T* x, const T* u, double h, T* xn) {
  constexpr int n = M::n;
  T k1[n], k2[n], k3[n], k4[n], xt[n];
  M::f(P, x, u, k1);
  if (integrator == INTEG_EULER) {
#pragma unroll
    for (int i = 0; i < n; ++i) xn[i] = x[i] + k1[i] * h;
    return;
  }
#pragma unroll
  for (int i = 0; i < n; ++i) { k1[i] = k1[i] * h; xt[i] = x[i] + k1[i] / 2.0; }
  M::f(P, xt, u, k2);
  if (integrator == INTEG_RK3) {
#pragma unroll
    for (int i = 0; i < n; ++i) { k2[i] = k2[i] * h; xt[i] = x[i] - k1[i] + 2.0 * k2[i]; }
    M::f(P, xt, u, k3);
#pragma unroll
    for (int i = 0; i < n; ++i) { k3[i] = k3[i] * h; xn[i] = x[i] + (k1[i] + 4.0 * k2[i] + k3[i]) / 6.0; }
    return;
  }
#pragma unroll
  for (int i = 0; i < n; ++i) { k2[i] = k2[i] * h; xt[i] = x[i] + k2[i] / 2.0; }
  M::f(P, xt, u, k3);
#pragma unroll
  for (int i = 0; i < n; ++i) { k3[i] = k3[i] * h; xt[i] = x[i] + k3[i]; }
  M::f(P, xt, u, k4);
#pragma unroll
  for (int i = 0; i < n; ++i) { k4[i] = k4[i] * h; xn[i] = x[i] + (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]) / 6.0; }
}

// ------------------------------------------------------------------------------------------------
// Error-state maps.  Identity for vector-space models.
// ------------------------------------------------------------------------------------------------
// v (n) = column j of G(x)   (n x ne attitude Jacobian blkdiag(I3, L(q)H, I3, I3); no 1/2: Cayley map)
template <class M>
__device__ __forceinline__ void errstate_col(const double* x, int j, double* v) {
  constexpr int n = M::n;
#pragma unroll
  for (int i = 0; i < n; ++i) v[i] = 0.0;
  if constexpr (!M::lie) {
#pragma unroll
    for (int i = 0; i < n; ++i) v[i] = (i == j) ? 1.0 : 0.0;
  } else {
    const double w = x[3], a = x[4], b = x[5], c = x[6];
    if (j < 3) {
#pragma unroll
      for (int i = 0; i < 3; ++i) v[i] = (i == j) ? 1.0 : 0.0;
    } else if (j == 3) { v[3] = -a; v[4] = w;  v[5] = c;  v[6] = -b; }
    else if (j == 4)   { v[3] = -b; v[4] = -c; v[5] = w;  v[6] = a; }
    else if (j == 5)   { v[3] = -c; v[4] = b;  v[5] = -a; v[6] = w; }
    else {
#pragma unroll
      for (int i = 7; i < 13; ++i) v[i] = (i == j + 1) ? 1.0 : 0.0;
    }
  }
}

// out (ne) = G(x)' y (n)
template <class M>
__device__ __forceinline__ void errstate_tmul(const double* x, const double* y, double* out) {
  if constexpr (!M::lie) {
#pragma unroll
    for (int i = 0; i < M::n; ++i) out[i] = y[i];
  } else {
    const double w = x[3], a = x[4], b = x[5], c = x[6];
    out[0] = y[0]; out[1] = y[1]; out[2] = y[2];
    out[3] = -a * y[3] + w * y[4] + c * y[5] - b * y[6];
    out[4] = -b * y[3] - c * y[4] + w * y[5] + a * y[6];
    out[5] = -c * y[3] + b * y[4] - a * y[5] + w * y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) out[6 + i] = y[7 + i];
  }
}

// dx (ne) = x (-) x0   (RD.state_diff, Cayley map)
template <class M>
__device__ __forceinline__ void state_diff(const double* x, const double* x0, double* dx) {
  if constexpr (!M::lie) {
#pragma unroll
    for (int i = 0; i < M::n; ++i) dx[i] = x[i] - x0[i];
  } else {
#pragma unroll
    for (int i = 0; i < 3; ++i) dx[i] = x[i] - x0[i];
    const double w0 = x0[3], a0 = x0[4], b0 = x0[5], c0 = x0[6];
    const double w = x[3], a = x[4], b = x[5], c = x[6];
    const double s = w0 * w + a0 * a + b0 * b + c0 * c;
    const double v1 = w0 * a - a0 * w - (b0 * c - c0 * b);
    const double v2 = w0 * b - b0 * w - (c0 * a - a0 * c);
    const double v3 = w0 * c - c0 * w - (a0 * b - b0 * a);
    dx[3] = v1 / s; dx[4] = v2 / s; dx[5] = v3 / s;
#pragma unroll
    for (int i = 0; i < 6; ++i) dx[6 + i] = x[7 + i] - x0[7 + i];
  }
}

}  // namespace to

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

#include <type_traits>

namespace to {

// 1/x to ~1 ulp: hardware reciprocal estimate + two Newton steps (5 VALU ops instead of the ~27 of an IEEE
// double division — on gfx950 the division sequence was the single largest cost of every hot kernel).
__device__ __forceinline__ double rcp_fast(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}

// 1/sqrt(x) to ~1 ulp: hardware estimate + two Newton steps.  With l = x * rsqrt(x) a Cholesky pivot costs 10 VALU
// operations instead of the ~30 of an IEEE sqrt followed by a reciprocal.
__device__ __forceinline__ double rsqrt_fast(double x) {
  double y = __builtin_amdgcn_rsq(x);
  double e = fma(-(x * y), y, 1.0);
  y = fma(0.5 * y, e, y);
  e = fma(-(x * y), y, 1.0);
  y = fma(0.5 * y, e, y);
  return y;
}

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
  const double rb = rcp_fast(b.v);
  const double q = a.v * rb;
  return Dual(q, (a.d - q * b.d) * rb);
}
__device__ __forceinline__ Dual operator+(Dual a, double b) { return Dual(a.v + b, a.d); }
__device__ __forceinline__ Dual operator+(double a, Dual b) { return Dual(a + b.v, b.d); }
__device__ __forceinline__ Dual operator-(Dual a, double b) { return Dual(a.v - b, a.d); }
__device__ __forceinline__ Dual operator-(double a, Dual b) { return Dual(a - b.v, -b.d); }
__device__ __forceinline__ Dual operator*(Dual a, double b) { return Dual(a.v * b, a.d * b); }
__device__ __forceinline__ Dual operator*(double a, Dual b) { return Dual(a * b.v, a * b.d); }
__device__ __forceinline__ Dual operator/(Dual a, double b) { const double rb = rcp_fast(b); return Dual(a.v * rb, a.d * rb); }
__device__ __forceinline__ double recip_t(double x) { return rcp_fast(x); }
__device__ __forceinline__ Dual recip_t(Dual x) { const double r = rcp_fast(x.v); return Dual(r, -(x.d * r) * r); }

// sin and cos together: 4-term Cody-Waite reduction by π/2 (exact products for |x| < 2^19·π/2, FMA-based) followed by
// the classic minimax kernels on [-π/4, π/4] (< 1 ulp).  ~45 FP64 instructions instead of libm's ~150 plus its
// Payne-Hanek path.  Beyond 8.2e5 rad the argument is first folded by whole turns in double precision (absolute error
// ≈ |x|·2e-16 in the angle): such states only occur in diverged line-search candidates, whose cost is astronomically
// large either way; NaN/Inf propagate to NaN and are caught by the state-limit check.
__device__ __forceinline__ void sincos_fast(double x, double* s, double* c) {
  if (!(fabs(x) < 8.2e5)) {  // (a wave-uniform skip of this fold measured slower: 100 -> 105 us per Cartpole forward pass)
    const double t = x * 1.59154943091895335769e-01;  // 1/(2π)
    x = (t - rint(t)) * 6.28318530717958647693e+00;
  }
  const double fn = rint(x * 6.36619772367581382433e-01);
  double r = fma(-fn, 1.57079632673412561417e+00, x);   // pio2_1  (33 bits)
  r = fma(-fn, 6.07710050630396597660e-11, r);          // pio2_2  (33 bits)
  r = fma(-fn, 2.02226624871116645580e-21, r);          // pio2_3  (33 bits); the tail beyond it is < 5e-26 for |fn| < 5.3e5
  const double z = r * r;
  // sin(r) = r + r^3 (S1 + z (S2 + ...)),  cos(r) = 1 - z/2 + z^2 (C1 + z (C2 + ...))
  double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  ps = fma(z, ps, 2.75573137070700676789e-06);
  ps = fma(z, ps, -1.98412698298579493134e-04);
  ps = fma(z, ps, 8.33333333332248946124e-03);
  ps = fma(z, ps, -1.66666666666666324348e-01);
  const double sr = fma(z * r, ps, r);
  double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  pc = fma(z, pc, -2.75573143513906633035e-07);
  pc = fma(z, pc, 2.48015872894767294178e-05);
  pc = fma(z, pc, -1.38888888888741095749e-03);
  pc = fma(z, pc, 4.16666666666666019037e-02);
  const double cr = fma(z * z, pc, fma(z, -0.5, 1.0));
  const int q = (int)fn & 3;
  const double s0 = (q & 1) ? cr : sr, c0 = (q & 1) ? sr : cr;
  *s = (q & 2) ? -s0 : s0;
  *c = ((q + 1) & 2) ? -c0 : c0;
}
__device__ __forceinline__ void sincos_t(double x, double* s, double* c) { sincos_fast(x, s, c); }
__device__ __forceinline__ void sincos_t(Dual x, Dual* s, Dual* c) {
  double sv, cv;
  sincos_fast(x.v, &sv, &cv);
  *s = Dual(sv, cv * x.d);
  *c = Dual(cv, -sv * x.d);
}
// sin and cos of a SMALL angle, |r| <= π/4: the polynomial kernels of sincos_fast without its range reduction and quadrant
// select (12 FMAs instead of ~45 instructions).  Used by the RK stages of the models that declare `trig_index`: stages 2-4 evaluate
// the dynamics at θ + δ with δ = O(h θ̇), and sin / cos of θ + δ follow from stage 1's by the angle-addition formulas.
__device__ __forceinline__ void sincos_small(double r, double* s, double* c) {
  const double z = r * r;
  double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  ps = fma(z, ps, 2.75573137070700676789e-06);
  ps = fma(z, ps, -1.98412698298579493134e-04);
  ps = fma(z, ps, 8.33333333332248946124e-03);
  ps = fma(z, ps, -1.66666666666666324348e-01);
  *s = fma(z * r, ps, r);
  double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  pc = fma(z, pc, -2.75573143513906633035e-07);
  pc = fma(z, pc, 2.48015872894767294178e-05);
  pc = fma(z, pc, -1.38888888888741095749e-03);
  pc = fma(z, pc, 4.16666666666666019037e-02);
  *c = fma(z * z, pc, fma(z, -0.5, 1.0));
}
constexpr double TRIG_SMALL_MAX = 0.75;  // beyond: the full evaluation (per lane: a lane's result never depends on its neighbours)
__device__ __forceinline__ bool any_lane(bool p) {
#ifdef __HIP_DEVICE_COMPILE__
  return __ballot(p) != 0;
#else
  return p;
#endif
}
__device__ __forceinline__ double value_of(double x) { return x; }
__device__ __forceinline__ double value_of(Dual x) { return x.v; }
__device__ __forceinline__ void sincos_small_t(double x, double* s, double* c) { sincos_small(x, s, c); }
__device__ __forceinline__ void sincos_small_t(Dual x, Dual* s, Dual* c) {
  double sv, cv;
  sincos_small(x.v, &sv, &cv);
  *s = Dual(sv, cv * x.d);
  *c = Dual(cv, -sv * x.d);
}
// max(0, x): derivative is the indicator x > 0 (the rotor-force clamp of the Quadrotor)
__device__ __forceinline__ double relu_t(double x) { return fmax(0.0, x); }
// (component-wise selects: a select between two Dual objects was lowered through scratch memory in the larger kernels)
__device__ __forceinline__ Dual relu_t(Dual x) { const bool on = x.v > 0.0; return Dual(on ? x.v : 0.0, on ? x.d : 0.0); }

// Value + K directional derivatives at once (ForwardDiff's chunk mode): the lane-per-trajectory expansion of the small
// models (k_expand.h, k_expand_lane) pushes ALL nc = ne + m input directions through the RK stages in one pass, so the value
// part of the dynamics — half of a single-direction Dual evaluation — is computed once per knot instead of once per column
// lane.  Every derivative component goes through exactly the operations of Dual's (same expressions, same order).
template <int K>
struct MDual {
  double v, d[K];
  __host__ __device__ MDual() : v(0.0) {
#pragma unroll
    for (int i = 0; i < K; ++i) d[i] = 0.0;
  }
  __host__ __device__ MDual(double a) : v(a) {
#pragma unroll
    for (int i = 0; i < K; ++i) d[i] = 0.0;
  }
};
#define TO_MD_LOOP _Pragma("unroll") for (int i_ = 0; i_ < K; ++i_)
template <int K> __device__ __forceinline__ MDual<K> operator+(MDual<K> a, MDual<K> b) { MDual<K> r; r.v = a.v + b.v; TO_MD_LOOP r.d[i_] = a.d[i_] + b.d[i_]; return r; }
template <int K> __device__ __forceinline__ MDual<K> operator-(MDual<K> a, MDual<K> b) { MDual<K> r; r.v = a.v - b.v; TO_MD_LOOP r.d[i_] = a.d[i_] - b.d[i_]; return r; }
template <int K> __device__ __forceinline__ MDual<K> operator-(MDual<K> a) { MDual<K> r; r.v = -a.v; TO_MD_LOOP r.d[i_] = -a.d[i_]; return r; }
template <int K> __device__ __forceinline__ MDual<K> operator*(MDual<K> a, MDual<K> b) { MDual<K> r; r.v = a.v * b.v; TO_MD_LOOP r.d[i_] = a.d[i_] * b.v + a.v * b.d[i_]; return r; }
template <int K> __device__ __forceinline__ MDual<K> operator/(MDual<K> a, MDual<K> b) {
  const double rb = rcp_fast(b.v);
  const double q = a.v * rb;
  MDual<K> r; r.v = q;
  TO_MD_LOOP r.d[i_] = (a.d[i_] - q * b.d[i_]) * rb;
  return r;
}
template <int K> __device__ __forceinline__ MDual<K> operator+(MDual<K> a, double b) { MDual<K> r = a; r.v = a.v + b; return r; }
template <int K> __device__ __forceinline__ MDual<K> operator+(double a, MDual<K> b) { MDual<K> r = b; r.v = a + b.v; return r; }
template <int K> __device__ __forceinline__ MDual<K> operator-(MDual<K> a, double b) { MDual<K> r = a; r.v = a.v - b; return r; }
template <int K> __device__ __forceinline__ MDual<K> operator-(double a, MDual<K> b) { MDual<K> r; r.v = a - b.v; TO_MD_LOOP r.d[i_] = -b.d[i_]; return r; }
template <int K> __device__ __forceinline__ MDual<K> operator*(MDual<K> a, double b) { MDual<K> r; r.v = a.v * b; TO_MD_LOOP r.d[i_] = a.d[i_] * b; return r; }
template <int K> __device__ __forceinline__ MDual<K> operator*(double a, MDual<K> b) { MDual<K> r; r.v = a * b.v; TO_MD_LOOP r.d[i_] = a * b.d[i_]; return r; }
template <int K> __device__ __forceinline__ MDual<K> operator/(MDual<K> a, double b) { const double rb = rcp_fast(b); MDual<K> r; r.v = a.v * rb; TO_MD_LOOP r.d[i_] = a.d[i_] * rb; return r; }
template <int K> __device__ __forceinline__ MDual<K> recip_t(MDual<K> x) { const double r0 = rcp_fast(x.v); MDual<K> r; r.v = r0; TO_MD_LOOP r.d[i_] = -(x.d[i_] * r0) * r0; return r; }
template <int K> __device__ __forceinline__ void sincos_t(MDual<K> x, MDual<K>* s, MDual<K>* c) {
  double sv, cv;
  sincos_fast(x.v, &sv, &cv);
  s->v = sv; c->v = cv;
  TO_MD_LOOP { s->d[i_] = cv * x.d[i_]; c->d[i_] = -sv * x.d[i_]; }
}
template <int K> __device__ __forceinline__ double value_of(MDual<K> x) { return x.v; }
template <int K> __device__ __forceinline__ void sincos_small_t(MDual<K> x, MDual<K>* s, MDual<K>* c) {
  double sv, cv;
  sincos_small(x.v, &sv, &cv);
  s->v = sv; c->v = cv;
  TO_MD_LOOP { s->d[i_] = cv * x.d[i_]; c->d[i_] = -sv * x.d[i_]; }
}
template <int K> __device__ __forceinline__ MDual<K> relu_t(MDual<K> x) {
  const bool on = x.v > 0.0;
  MDual<K> r; r.v = on ? x.v : 0.0;
  TO_MD_LOOP r.d[i_] = on ? x.d[i_] : 0.0;
  return r;
}
#undef TO_MD_LOOP

// component-wise select (a select between two dual-number objects went through scratch memory in the larger kernels)
__device__ __forceinline__ double select_t(bool c, double a, double b) { return c ? a : b; }
__device__ __forceinline__ Dual select_t(bool c, Dual a, Dual b) { return Dual(c ? a.v : b.v, c ? a.d : b.d); }
template <int K> __device__ __forceinline__ MDual<K> select_t(bool c, MDual<K> a, MDual<K> b) {
  MDual<K> r; r.v = c ? a.v : b.v;
#pragma unroll
  for (int i = 0; i < K; ++i) r.d[i] = c ? a.d[i] : b.d[i];
  return r;
}

// ------------------------------------------------------------------------------------------------
// Models.  P = model_params of the descriptor (wave-uniform, lives in SGPRs).
// ------------------------------------------------------------------------------------------------
// attitude representation carried by a model's state (RobotDynamics RigidBody{R}: R = QuatRotation, MRP, RodriguesParam —
// the three src/lie_costs.jl:1-3 names); ATT_NONE: vector-space model
enum { ATT_NONE = 0, ATT_QUAT = 1, ATT_MRP = 2, ATT_RP = 3 };

template <int D>
struct DoubleIntegratorModel {  // examples/quickstart.jl:15-20
  static constexpr int n = 2 * D, m = D, ne = 2 * D;
  static constexpr bool lie = false;
  static constexpr int att = ATT_NONE;
  static constexpr bool pin_rk4 = true;   // kernels instantiate a compile-time RK4 variant (rk_step<.., FIXED>)
  static constexpr int expand_knots = 1;
  static constexpr bool accept_write_through = true;   // the accepted step is written through to slot 0 by the next expansion (k_expand.h)
  static constexpr bool lds_gains = false;             // forward pass: the gains row of a knot is a handful of doubles, loaded directly
  static constexpr int ls_first_round = 4;             // step sizes tried concurrently (results do not depend on it): these models accept
                                                       // within the first 4 in 99.9 % of the iterations (tools/ls_hist.py); more only adds candidate traffic
  static constexpr bool mfma_backward = false, coop_backward = true;  // backward-pass kernels instantiated (k_backward.h)
  static constexpr bool lane_backward = (3 * D <= 6);  // one lane per trajectory while the blocks fit a lane's registers
  template <class T>
  __device__ __forceinline__ static void f(const double* P, const T* x, const T* u, T* xd) {
    const double inv_mass = rcp_fast(P[0]);
#pragma unroll
    for (int i = 0; i < D; ++i) {
      xd[i] = x[D + i];
      xd[D + i] = u[i] * inv_mass;
    }
  }
};

struct CartpoleModel {  // docs/src/model.md:34-50
  static constexpr int n = 4, m = 1, ne = 4;
  static constexpr bool lie = false;
  static constexpr int att = ATT_NONE;
  static constexpr bool pin_rk4 = true;   // kernels instantiate a compile-time RK4 variant (rk_step<.., FIXED>)
  static constexpr int expand_knots = 1;
  static constexpr bool accept_write_through = true;   // the accepted step is written through to slot 0 by the next expansion (k_expand.h)
  static constexpr bool lds_gains = false;             // forward pass: the gains row of a knot is a handful of doubles, loaded directly
  static constexpr int ls_first_round = 4;             // step sizes tried concurrently (results do not depend on it): these models accept
                                                       // within the first 4 in 99.9 % of the iterations (tools/ls_hist.py); more only adds candidate traffic
  static constexpr bool mfma_backward = true, coop_backward = true;  // MFMA and cooperative backward passes stay built for A/B runs (TRAJOPT_BACKWARD)
  static constexpr bool lane_backward = true;  // default: one lane per trajectory
  static constexpr int trig_index = 1;  // the state entry whose sine / cosine the dynamics need (rk_step carries them from stage to stage)
#ifndef TO_NO_STAGE_JAC  // (A/B builds: -DTO_NO_STAGE_JAC keeps the dual-number expansion everywhere)
  static constexpr bool stage_jac = true;  // cartpole_rk4_jac: the RK4 Jacobian by the chain rule over hand-derived stage partials (lane expansion)
#endif
  template <class T>
  __device__ __forceinline__ static void f(const double* P, const T* x, const T* u, T* xd) {
    T s, c;
    sincos_t(x[1], &s, &c);
    f_sc(P, x, u, s, c, xd);
  }
  // the dynamics with sin(x[1]), cos(x[1]) given
  template <class T>
  __device__ __forceinline__ static void f_sc(const double* P, const T* x, const T* u, T s, T c, T* xd) {
    const double mc = P[0], mp = P[1], l = P[2], g = P[3];
    T qd1 = x[2], qd2 = x[3];
    const double h11 = mc + mp, h22 = mp * l * l;
    T h12 = (mp * l) * c;
    T c12 = -((mp * qd2) * l) * s;
    // b = C*qd + G - B*u
    T b1 = c12 * qd2 - u[0];
    T b2 = ((mp * g) * l) * s;
    // 2x2 solve as StaticArrays does it: ((a22 b1 - a12 b2)/d, (a11 b2 - a21 b1)/d)
    T d = h11 * h22 - h12 * h12;
    T rd = recip_t(d);  // one reciprocal, two products (the reference divides twice; a few ulp apart)
    T s1 = (h22 * b1 - h12 * b2) * rd;
    T s2 = (h11 * b2 - h12 * b1) * rd;
    xd[0] = qd1;
    xd[1] = qd2;
    xd[2] = -s1;
    xd[3] = -s2;
  }
};

struct QuadrotorModel {  // RigidBody dynamics, world-frame velocity; state [r(3) q(w,x,y,z) v(3) ω(3)]
  static constexpr int n = 13, m = 4, ne = 12;
  static constexpr bool lie = true;
  static constexpr int att = ATT_QUAT;
  static constexpr bool pin_rk4 = false;  // compile-time RK4 costs registers here: measured slower than the runtime switch
#ifndef TO_QUAD_EXPAND_KNOTS
#define TO_QUAD_EXPAND_KNOTS 2
#endif
  // knots one expansion wave walks (software-pipelined loads).  Round 6: 2 for the unconstrained variant, 4 (k_expand.h TO_EXPAND_KC_CONS) for
  // the constrained ones — interleaved A/B of library builds, depth-4 pipelined / alone, M it/s (profiles/r06_ab/ab_expand_occupancy.jsonl):
  // C3 1.73-1.77 / 1.12 with 4, 1.83 / 1.12 with 2, 1.68 / 1.08 with 1, 1.80 / 1.08 with 1 knot at two waves per SIMD;
  // C5 1.42-1.43 / 1.14 with 4, 1.42 / 1.12 with 2, 1.37 / 1.10 with 1, 1.34 / 1.07 with 1 knot at two waves per SIMD
  static constexpr int expand_knots = TO_QUAD_EXPAND_KNOTS;
  static constexpr bool accept_write_through = false;  // accepted steps are copied onto slot 0 by k_accept after every forward pass
  static constexpr bool lds_gains = true;  // forward pass: the 52-double gains row of a knot comes through LDS (DMA), not prefetch VGPRs
  static constexpr int ls_first_round = 16;  // accepted step sizes sit at 2^-5 .. 2^-12 late in these solves (tools/ls_hist.py): a deep first round
  static constexpr bool mfma_backward = true, coop_backward = false;  // MFMA backward pass only (the cooperative kernel needed 256 VGPR + 236 AGPR here)
  static constexpr bool lane_backward = false;
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
    const double inv_mass = rcp_fast(mass);
    xd[7] = (mass * g1 + qF1) * inv_mass;
    xd[8] = (mass * g2 + qF2) * inv_mass;
    xd[9] = (mass * g3 + qF3) * inv_mass;
    T Jw1 = J1 * w1, Jw2 = J2 * w2, Jw3 = J3 * w3;
    xd[10] = rcp_fast(J1) * (t1 - (w2 * Jw3 - w3 * Jw2));
    xd[11] = rcp_fast(J2) * (t2 - (w3 * Jw1 - w1 * Jw3));
    xd[12] = rcp_fast(J3) * (t3 - (w1 * Jw2 - w2 * Jw1));
  }
};

// The same rigid body with a THREE-parameter attitude (RigidBody{MRP} / RigidBody{RodriguesParam}; src/lie_costs.jl:1-3,
// examples/Quadrotor.ipynb cell 5): state [r(3) p(3) v(3) ω(3)], n = ne = 12.  Kinematics (Rotations.kinematics):
//   MRP  ṗ = ¼[(1 − |p|²) ω + 2 p×ω + 2 p (p·ω)]        RodriguesParam  ġ = ½[ω + g×ω + g (g·ω)]
// and the body force is rotated by the unit quaternion of the attitude, q = [(1−|p|²), 2p]/(1+|p|²) resp. [1, g]/√(1+|g|²),
// written in the rational form  q*r = M² [(w̃² − ṽ·ṽ) r + 2 ṽ (ṽ·r) + 2 w̃ (ṽ × r)],  q̃ = [w̃, ṽ] the unnormalised quaternion,
// M² = 1/|q̃|².
template <int ATT>
struct QuadrotorAttModel {
  static_assert(ATT == ATT_MRP || ATT == ATT_RP, "three-parameter attitudes");
  static constexpr int n = 12, m = 4, ne = 12;
  static constexpr bool lie = true;
  static constexpr int att = ATT;
  static constexpr bool pin_rk4 = false;
  static constexpr int expand_knots = 4;
  static constexpr bool accept_write_through = false;
  static constexpr bool lds_gains = true;
  static constexpr int ls_first_round = 16;
  static constexpr bool mfma_backward = true, coop_backward = false;
  static constexpr bool lane_backward = false;
  template <class T>
  __device__ __forceinline__ static void f(const double* P, const T* x, const T* u, T* xd) {
    const double mass = P[0], J1 = P[1], J2 = P[2], J3 = P[3];
    const double g1 = P[4], g2 = P[5], g3 = P[6], L = P[7], kf = P[8], km = P[9];
    T p1 = x[3], p2 = x[4], p3 = x[5];
    T w1 = x[9], w2 = x[10], w3 = x[11];
    T F1 = relu_t(kf * u[0]), F2 = relu_t(kf * u[1]), F3 = relu_t(kf * u[2]), F4 = relu_t(kf * u[3]);
    T Fz = F1 + F2 + F3 + F4;
    T n2 = p1 * p1 + p2 * p2 + p3 * p3;
    T pw = p1 * w1 + p2 * w2 + p3 * w3;
    T c1 = p2 * w3 - p3 * w2, c2 = p3 * w1 - p1 * w3, c3 = p1 * w2 - p2 * w1;  // p × ω
    // unnormalised quaternion of the attitude and 1/|q̃|²
    T qw, qx, qy, qz, M2;
    if constexpr (ATT == ATT_MRP) {
      T s = 1.0 - n2;
      xd[3] = 0.25 * (s * w1 + 2.0 * c1 + (2.0 * p1) * pw);
      xd[4] = 0.25 * (s * w2 + 2.0 * c2 + (2.0 * p2) * pw);
      xd[5] = 0.25 * (s * w3 + 2.0 * c3 + (2.0 * p3) * pw);
      qw = s; qx = 2.0 * p1; qy = 2.0 * p2; qz = 2.0 * p3;
      T d = 1.0 + n2;
      M2 = recip_t(d * d);
    } else {
      xd[3] = 0.5 * (w1 + c1 + p1 * pw);
      xd[4] = 0.5 * (w2 + c2 + p2 * pw);
      xd[5] = 0.5 * (w3 + c3 + p3 * pw);
      qw = T(1.0); qx = p1; qy = p2; qz = p3;
      M2 = recip_t(1.0 + n2);
    }
    T vv = qx * qx + qy * qy + qz * qz;
    T sc = qw * qw - vv;
    T vr = qz * Fz;
    T qF1 = ((2.0 * qx) * vr + (2.0 * qw) * (qy * Fz)) * M2;
    T qF2 = ((2.0 * qy) * vr - (2.0 * qw) * (qx * Fz)) * M2;
    T qF3 = (sc * Fz + (2.0 * qz) * vr) * M2;
    T t1 = L * (F2 - F4), t2 = L * (F3 - F1);
    T t3 = km * u[0] - km * u[1] + km * u[2] - km * u[3];
    xd[0] = x[6];
    xd[1] = x[7];
    xd[2] = x[8];
    const double inv_mass = rcp_fast(mass);
    xd[6] = (mass * g1 + qF1) * inv_mass;
    xd[7] = (mass * g2 + qF2) * inv_mass;
    xd[8] = (mass * g3 + qF3) * inv_mass;
    T Jw1 = J1 * w1, Jw2 = J2 * w2, Jw3 = J3 * w3;
    xd[9] = rcp_fast(J1) * (t1 - (w2 * Jw3 - w3 * Jw2));
    xd[10] = rcp_fast(J2) * (t2 - (w3 * Jw1 - w1 * Jw3));
    xd[11] = rcp_fast(J3) * (t3 - (w1 * Jw2 - w2 * Jw1));
  }
};

// 3 x 3 attitude block D(p) of the error-state Jacobian of a three-parameter attitude (Rotations ∇differential with the Cayley
// error map: d(p ⊕ φ)/dφ at φ = 0; ṗ = D(p) ω/2):  MRP  ½[(1−|p|²) I + 2[p]× + 2pp']   RodriguesParam  I + [g]× + gg'.
// Row-major D[3*i + j].
template <int ATT>
__device__ __forceinline__ void att_differential(const double* p, double* D) {
  const double a = p[0], b = p[1], c = p[2];
  if constexpr (ATT == ATT_MRP) {
    const double h = 0.5 * (1.0 - (a * a + b * b + c * c));
    D[0] = h + a * a; D[1] = -c + a * b; D[2] = b + a * c;
    D[3] = c + b * a; D[4] = h + b * b;  D[5] = -a + b * c;
    D[6] = -b + c * a; D[7] = a + c * b; D[8] = h + c * c;
  } else {
    D[0] = 1.0 + a * a; D[1] = -c + a * b;   D[2] = b + a * c;
    D[3] = c + b * a;   D[4] = 1.0 + b * b;  D[5] = -a + b * c;
    D[6] = -b + c * a;  D[7] = a + c * b;    D[8] = 1.0 + c * c;
  }
}
// Hessian of φ -> b·(p ⊕ φ) at φ = 0 (Rotations ∇²differential; the second-order term of the error-state cost Hessian):
//   MRP  ½[a p' + p a'] − (b·p)[(1+|p|²)/2 I − 2 pp'],  a = (1−|p|²) b + 2 b×p
//   RP   c g' + g c' + 2 (b·g) g g',                    c = b + b×g                       (symmetric; row-major H[3*i + j])
template <int ATT>
__device__ __forceinline__ void att_differential2(const double* p, const double* b, double* H) {
  const double bp = b[0] * p[0] + b[1] * p[1] + b[2] * p[2];
  const double cx[3] = {b[1] * p[2] - b[2] * p[1], b[2] * p[0] - b[0] * p[2], b[0] * p[1] - b[1] * p[0]};  // b × p
  if constexpr (ATT == ATT_MRP) {
    const double n2 = p[0] * p[0] + p[1] * p[1] + p[2] * p[2];
    double a[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) a[i] = (1.0 - n2) * b[i] + 2.0 * cx[i];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) H[3 * i + j] = 0.5 * (a[i] * p[j] + p[i] * a[j]) - bp * (((i == j) ? 0.5 * (1.0 + n2) : 0.0) - 2.0 * p[i] * p[j]);
  } else {
    double c[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) c[i] = b[i] + cx[i];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) H[3 * i + j] = c[i] * p[j] + p[i] * c[j] + 2.0 * bp * p[i] * p[j];
  }
}

// ------------------------------------------------------------------------------------------------
// Integrators: h multiplied into each stage, then combined.
// ------------------------------------------------------------------------------------------------
enum { INTEG_RK4 = 0, INTEG_RK3 = 1, INTEG_EULER = 2 };

// FIXED >= 0 pins the scheme at compile time: with a runtime switch the compiler merges the three schemes into one loop
// full of selects (the Cartpole rollout loop was 857 instructions per knot, 300 of them FP64).
template <class M, class = void>
struct has_trig : std::false_type {};
template <class M>
struct has_trig<M, decltype((void)M::trig_index)> : std::true_type {};

// One stage evaluation at xt = x + (increment with xt[trig_index] - x[trig_index] = delta) for a model whose dynamics need sin / cos
// of ONE state entry: sin(θ + δ) = s1 cos δ + c1 sin δ, cos(θ + δ) = c1 cos δ - s1 sin δ with the small-angle kernels — three
// Cody-Waite reductions and quadrant selects per RK4 step saved (the Cartpole roller's knot: 358 -> ~285 instructions).  A lane
// whose |δ| exceeds TRIG_SMALL_MAX takes the full evaluation of xt (computed for the whole wave only when some lane needs it,
// selected per lane).
template <class M, class T>
__device__ __forceinline__ void trig_stage(const double* P, const T* xt, const T* u, T s1, T c1, T delta, T* k) {
  T sd, cd;
  sincos_small_t(delta, &sd, &cd);
  T s = s1 * cd + c1 * sd, c = c1 * cd - s1 * sd;
  const bool big = !(fabs(value_of(delta)) <= TRIG_SMALL_MAX);
  if (any_lane(big)) {
    T sf, cf;
    sincos_t(xt[M::trig_index], &sf, &cf);
    s = select_t(big, sf, s); c = select_t(big, cf, c);
  }
  M::f_sc(P, xt, u, s, c, k);
}

template <class M, class T, int FIXED = -1>
__device__ __forceinline__ void rk_step(const double* P, int integrator_rt, const T* x, const T* u, double h, T* xn) {
  const int integrator = FIXED >= 0 ? FIXED : integrator_rt;
  if constexpr (has_trig<M>::value) {  // same scheme, same operation order; only sin / cos of the later stages come by angle addition
    if (integrator != INTEG_EULER) {
      constexpr int n = M::n, ti = M::trig_index;
      T k[n], acc[n], xt[n], s1, c1;
      sincos_t(x[ti], &s1, &c1);
      M::f_sc(P, x, u, s1, c1, k);
#pragma unroll
      for (int i = 0; i < n; ++i) { k[i] = k[i] * h; acc[i] = k[i]; xt[i] = x[i] + k[i] * 0.5; }
      if (integrator == INTEG_RK3) {
        T k1[n];
#pragma unroll
        for (int i = 0; i < n; ++i) k1[i] = k[i];
        trig_stage<M, T>(P, xt, u, s1, c1, k1[ti] * 0.5, k);
#pragma unroll
        for (int i = 0; i < n; ++i) { k[i] = k[i] * h; acc[i] = acc[i] + 4.0 * k[i]; xt[i] = x[i] - k1[i] + 2.0 * k[i]; }
        trig_stage<M, T>(P, xt, u, s1, c1, 2.0 * k[ti] - k1[ti], k);
#pragma unroll
        for (int i = 0; i < n; ++i) { k[i] = k[i] * h; xn[i] = x[i] + (acc[i] + k[i]) * (1.0 / 6.0); }
        return;
      }
      trig_stage<M, T>(P, xt, u, s1, c1, k[ti] * 0.5, k);
#pragma unroll
      for (int i = 0; i < n; ++i) { k[i] = k[i] * h; acc[i] = acc[i] + 2.0 * k[i]; xt[i] = x[i] + k[i] * 0.5; }
      trig_stage<M, T>(P, xt, u, s1, c1, k[ti] * 0.5, k);
#pragma unroll
      for (int i = 0; i < n; ++i) { k[i] = k[i] * h; acc[i] = acc[i] + 2.0 * k[i]; xt[i] = x[i] + k[i]; }
      trig_stage<M, T>(P, xt, u, s1, c1, k[ti], k);
#pragma unroll
      for (int i = 0; i < n; ++i) { k[i] = k[i] * h; xn[i] = x[i] + (acc[i] + k[i]) * (1.0 / 6.0); }
      return;
    }
  }
  // Stage slopes are folded into a running sum as they are produced (same left-to-right order as
  // x + (k1 + 2k2 + 2k3 + k4)/6), so only {x, xt, k, acc} are live instead of {x, xt, k1..k4}: for the Quadrotor in
  // dual numbers that is the difference between fitting the register file and spilling.
  constexpr int n = M::n;
  T k[n], acc[n], xt[n];
  M::f(P, x, u, k);
  if (integrator == INTEG_EULER) {
#pragma unroll
    for (int i = 0; i < n; ++i) xn[i] = x[i] + k[i] * h;
    return;
  }
#pragma unroll
  for (int i = 0; i < n; ++i) { k[i] = k[i] * h; acc[i] = k[i]; xt[i] = x[i] + k[i] * 0.5; }
  if (integrator == INTEG_RK3) {
    T k1[n];
#pragma unroll
    for (int i = 0; i < n; ++i) k1[i] = k[i];
    M::f(P, xt, u, k);
#pragma unroll
    for (int i = 0; i < n; ++i) { k[i] = k[i] * h; acc[i] = acc[i] + 4.0 * k[i]; xt[i] = x[i] - k1[i] + 2.0 * k[i]; }
    M::f(P, xt, u, k);
#pragma unroll
    for (int i = 0; i < n; ++i) { k[i] = k[i] * h; xn[i] = x[i] + (acc[i] + k[i]) * (1.0 / 6.0); }
    return;
  }
  M::f(P, xt, u, k);
#pragma unroll
  for (int i = 0; i < n; ++i) { k[i] = k[i] * h; acc[i] = acc[i] + 2.0 * k[i]; xt[i] = x[i] + k[i] * 0.5; }
  M::f(P, xt, u, k);
#pragma unroll
  for (int i = 0; i < n; ++i) { k[i] = k[i] * h; acc[i] = acc[i] + 2.0 * k[i]; xt[i] = x[i] + k[i]; }
  M::f(P, xt, u, k);
#pragma unroll
  for (int i = 0; i < n; ++i) { k[i] = k[i] * h; xn[i] = x[i] + (acc[i] + k[i]) * (1.0 / 6.0); }
}

// ------------------------------------------------------------------------------------------------
// Stage Jacobians by hand where a model provides them (has_stage_jac): the RK4 Jacobian [A B] of the Cartpole by the chain rule
// through the four stages, for the one-lane-per-trajectory expansion of large batches (k_expand.h expand_lane_knot).  Chunk-mode
// dual numbers push 5 derivative components through EVERY operation of a stage (sin, cos, the 2 x 2 solve: ~700 FP64 instructions per
// knot even with their structural zeros folded); here the accelerations' partials with respect to the three inputs they depend on
// (theta, thetadot, u) are formed once per stage (~30 operations) and applied to the incoming 4 x 4 sensitivity block:
//   k_i = h f(xt_i, u),  D_i = dk_i/dz = h [ Y_i[pdot,:]; Y_i[thetadot,:]; a_theta Y_i[theta,:] + a_omega Y_i[thetadot,:] + a_u e_u ],
//   Y_1 = E (the inputs themselves), Y_2 = E + D_1/2, Y_3 = E + D_2/2, Y_4 = E + D_3,   d x+/dz = E + (D_1 + 2 D_2 + 2 D_3 + D_4)/6
// with z = (theta, pdot, thetadot, u); nothing depends on the cart position p: d x+/dp = e_p exactly.  The same exact derivative of the same
// RK4 map as the dual numbers give (tests/test_device_math_on_host.py: both against each other to rounding and against central
// differences); every other kernel keeps the dual-number path, so the two still check each other on the GPU.
template <class M, class = void>
struct has_stage_jac : std::false_type {};
template <class M>
struct has_stage_jac<M, decltype((void)M::stage_jac)> : std::true_type {};

// Cartpole accelerations a = (pddot, thetaddot) at (theta with sin s / cos c, thetadot w, u) and their partials Ja[2][3] w.r.t. (theta, w, u)
__device__ __forceinline__ void cartpole_accel_jac(const double* P, double s, double c, double w, double u, double* a, double (*Ja)[3]) {
  const double mc = P[0], mp = P[1], l = P[2], g = P[3];
  const double h11 = mc + mp, h22 = mp * l * l, mpl = mp * l;
  const double h12 = mpl * c;
  const double c12 = -(mpl * w) * s;
  const double b1 = c12 * w - u, b2 = (mpl * g) * s;
  const double d = h11 * h22 - h12 * h12;
  const double rd = rcp_fast(d);
  const double s1 = (h22 * b1 - h12 * b2) * rd, s2 = (h11 * b2 - h12 * b1) * rd;
  a[0] = -s1; a[1] = -s2;
  // d/dtheta
  const double dh12 = -mpl * s, db1 = -(mpl * c) * (w * w), db2 = (mpl * g) * c;
  const double dd = -2.0 * h12 * dh12;
  const double dN1 = h22 * db1 - dh12 * b2 - h12 * db2, dN2 = h11 * db2 - dh12 * b1 - h12 * db1;
  Ja[0][0] = -((dN1 - s1 * dd) * rd); Ja[1][0] = -((dN2 - s2 * dd) * rd);
  // d/dthetadot: b1 = -mp l s w^2 - u
  const double wb1 = 2.0 * c12;
  Ja[0][1] = -(h22 * wb1 * rd); Ja[1][1] = (h12 * wb1) * rd;
  // d/du: b1' = -1
  Ja[0][2] = h22 * rd; Ja[1][2] = -(h12 * rd);
}
// Mk[i*5 + j] = d x+_i / d [p, theta, pdot, thetadot, u]_j of one RK4 step of the Cartpole
__device__ __forceinline__ void cartpole_rk4_jac(const double* P, const double* x, const double* u, double h, double* Mk) {
  double s1, c1;
  sincos_fast(x[1], &s1, &c1);
  // rows: p, theta, pdot, thetadot; columns: theta, pdot, thetadot, u
  double Y[4][4], acc[4][4], kprev[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) { Y[r][cc] = (r >= 1 && cc == r - 1) ? 1.0 : 0.0; acc[r][cc] = 0.0; }
#pragma unroll
  for (int st = 0; st < 4; ++st) {
    const double cf = (st == 0) ? 0.0 : (st == 3 ? 1.0 : 0.5);   // xt = x + cf * k_prev
    const double th = x[1] + cf * kprev[1], w = x[3] + cf * kprev[3];
    double s = s1, c = c1;
    if (st > 0) {
      const double delta = cf * kprev[1];
      double sd, cd;
      sincos_small(delta, &sd, &cd);
      s = s1 * cd + c1 * sd; c = c1 * cd - s1 * sd;
      const bool big = !(fabs(delta) <= TRIG_SMALL_MAX);
      if (any_lane(big)) {
        double sf, cf2;
        sincos_fast(th, &sf, &cf2);
        s = big ? sf : s; c = big ? cf2 : c;
      }
    }
    double a[2], Ja[2][3];
    cartpole_accel_jac(P, s, c, w, u[0], a, Ja);
    const double k[4] = {h * (x[2] + cf * kprev[2]), h * w, h * a[0], h * a[1]};
    double D[4][4];
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      D[0][cc] = h * Y[2][cc];
      D[1][cc] = h * Y[3][cc];
      D[2][cc] = h * (Ja[0][0] * Y[1][cc] + Ja[0][1] * Y[3][cc] + (cc == 3 ? Ja[0][2] : 0.0));
      D[3][cc] = h * (Ja[1][0] * Y[1][cc] + Ja[1][1] * Y[3][cc] + (cc == 3 ? Ja[1][2] : 0.0));
    }
    const double wgt = (st == 0 || st == 3) ? 1.0 : 2.0, nf = (st == 2) ? 1.0 : 0.5;   // weight in the sum; factor of the NEXT stage's point
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
        acc[r][cc] += wgt * D[r][cc];
        Y[r][cc] = ((r >= 1 && cc == r - 1) ? 1.0 : 0.0) + nf * D[r][cc];
      }
      kprev[r] = k[r];
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    Mk[r * 5 + 0] = (r == 0) ? 1.0 : 0.0;
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) Mk[r * 5 + 1 + cc] = ((r >= 1 && cc == r - 1) ? 1.0 : 0.0) + acc[r][cc] * (1.0 / 6.0);
  }
}

// ------------------------------------------------------------------------------------------------
// Hybrid model vectors (src/dynamics.jl:15-31, test/hybrid_dynamics_model.jl): the model — and with it the LIVE state / control
// dimension — changes along the horizon.  A hybrid model is compiled in like any other, at the largest (n, m) of its phases;
// states and controls of the narrower phases are stored zero-padded, and the model itself maps time step k to a phase:
//   static constexpr bool hybrid = true;
//   step<T, FIXED>(P, integrator, k, x, u, h, xn)   one time step of whatever model governs step k (padded in, padded out)
//   knot_dims(P, N, k, &nx, &nu)                      live dimensions at knot k (host side: to_knot_dims, validation)
// Every kernel takes its time step through model_step, which hands the knot index to such a model and is rk_step otherwise.
template <class M, class = void>
struct is_hybrid { static constexpr bool value = false; };
template <class M>
struct is_hybrid<M, decltype((void)M::hybrid)> { static constexpr bool value = M::hybrid; };

template <class M, class T, int FIXED = -1>
__device__ __forceinline__ void model_step(const double* P, int integrator_rt, int k, const T* x, const T* u, double h, T* xn) {
  if constexpr (is_hybrid<M>::value) M::template step<T, FIXED>(P, integrator_rt, k, x, u, h, xn);
  else rk_step<M, T, FIXED>(P, integrator_rt, x, u, h, xn);
}


// The model vector of test/hybrid_dynamics_model.jl:14-52: a 2-D double integrator (4, 2) for the first S = P[1] time steps, a
// jump map (4, 2) -> 2, x+ = [(x3 + x4)/2, (u1 + u2)/2] (test :31-33; a discrete map here), then a 1-D double integrator (2, 1).
// Stored at (4, 2); P[0] = mass.  Branch-free: all three maps are evaluated and selected by the (per-lane) knot index, so the
// kernels whose lanes sit on different knots keep their EXEC mask full.
struct HybridDoubleIntegratorModel {
  static constexpr int n = 4, m = 2, ne = 4;
  static constexpr bool lie = false;
  static constexpr int att = ATT_NONE;
  static constexpr bool hybrid = true;
  static constexpr bool pin_rk4 = true;
  static constexpr int expand_knots = 1;
  static constexpr bool accept_write_through = true;
  static constexpr bool lds_gains = false;
  static constexpr int ls_first_round = 4;
  static constexpr bool mfma_backward = false, coop_backward = true;
  static constexpr bool lane_backward = true;
  template <class T>
  __device__ __forceinline__ static void f(const double* P, const T* x, const T* u, T* xd) { DoubleIntegratorModel<2>::f(P, x, u, xd); }
  template <class T, int FIXED>
  __device__ __forceinline__ static void step(const double* P, int integrator_rt, int k, const T* x, const T* u, double h, T* xn) {
    const int S = (int)P[1];
    T ya[4], yc[2];
    rk_step<DoubleIntegratorModel<2>, T, FIXED>(P, integrator_rt, x, u, h, ya);
    rk_step<DoubleIntegratorModel<1>, T, FIXED>(P, integrator_rt, x, u, h, yc);  // reads x[0..1], u[0]
    const T j0 = (x[2] + x[3]) * 0.5, j1 = (u[0] + u[1]) * 0.5;
    const T zero(0.0);
    const bool first = k < S, jump = k == S;
    xn[0] = select_t(first, ya[0], select_t(jump, j0, yc[0]));
    xn[1] = select_t(first, ya[1], select_t(jump, j1, yc[1]));
    xn[2] = select_t(first, ya[2], zero);
    xn[3] = select_t(first, ya[3], zero);
  }
  __host__ __device__ static void knot_dims(const double* P, int N, int k, int* nx, int* nu) {  // knot k = 0 .. N-1
    const int S = (int)P[1];
    *nx = k <= S ? 4 : 2;          // knots 0..S carry the 2-D state (the jump map's input included)
    *nu = k <= S ? 2 : 1;          // the terminal knot reports the control dimension of the last model (RD.dims)
  }
};

// Problem(models::Vector{<:DiscreteDynamics}, ...) in general (src/problem.jl:36-73, src/dynamics.jl:15-31; TO_MODEL_VECTOR): one
// model per time step out of a per-step TABLE in device memory — any mix of double integrators (D = 1, 2, 3), Cartpoles and
// linear discrete maps x+ = A x + B u whose dimensions chain.  Stored at (6, 3) with the narrower knots zero-padded, like the
// hybrid double integrator above.  One record of 64 doubles per step: [0] kind (0 double integrator, 1 Cartpole, 2 linear map),
// [1] n, [2] m, [3] n_out, [4..] the sub-model's parameters as that model reads them ([4] mass, [5] D / [4..7] mc, mp, l, g),
// [8..44) A row-major 6 x 6, [44..62) B row-major 6 x 3 (zero-padded).  P[0] carries the table's ADDRESS (bit pattern: the
// kernels move model parameters around as doubles, never compute with them).  Branch-free like the hybrid model: every kind is
// evaluated, the record's kind selects (the lanes of the knot-parallel kernels sit on different steps).
struct ModelVectorModel {
  static constexpr int n = 6, m = 3, ne = 6;
  static constexpr bool lie = false;
  static constexpr int att = ATT_NONE;
  static constexpr bool hybrid = true;
  static constexpr bool pin_rk4 = true;
  static constexpr int expand_knots = 1;
  static constexpr bool accept_write_through = true;
  static constexpr bool lds_gains = false;
  static constexpr int ls_first_round = 4;
  static constexpr bool mfma_backward = false, coop_backward = true;
  static constexpr bool lane_backward = false;
  static constexpr int REC = 64;
  __host__ __device__ static const double* table(const double* P) {
    unsigned long long bits;
    __builtin_memcpy(&bits, &P[0], sizeof(bits));
    return reinterpret_cast<const double*>(bits);
  }
  template <class T>
  __device__ __forceinline__ static void f(const double* P, const T* x, const T* u, T* xd) { DoubleIntegratorModel<3>::f(P, x, u, xd); }  // (unused: step dispatches)
  template <class T, int FIXED>
  __device__ __forceinline__ static void step(const double* P, int integrator_rt, int k, const T* x, const T* u, double h, T* xn) {
    const double* rec = table(P) + (size_t)REC * k;
    const int kind = (int)rec[0], D = (int)rec[1] / 2;
    T y1[2], y2[4], y3[6], yc[4], yl[6];
    rk_step<DoubleIntegratorModel<1>, T, FIXED>(rec + 4, integrator_rt, x, u, h, y1);
    rk_step<DoubleIntegratorModel<2>, T, FIXED>(rec + 4, integrator_rt, x, u, h, y2);
    rk_step<DoubleIntegratorModel<3>, T, FIXED>(rec + 4, integrator_rt, x, u, h, y3);
    rk_step<CartpoleModel, T, FIXED>(rec + 4, integrator_rt, x, u, h, yc);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      T s(0.0);
#pragma unroll
      for (int j = 0; j < 6; ++j) s = s + rec[8 + 6 * i + j] * x[j];
#pragma unroll
      for (int j = 0; j < 3; ++j) s = s + rec[44 + 3 * i + j] * u[j];
      yl[i] = s;
    }
    const T zero(0.0);
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const T di = select_t(D == 1, i < 2 ? y1[i < 2 ? i : 0] : zero, select_t(D == 2, i < 4 ? y2[i < 4 ? i : 0] : zero, y3[i]));
      xn[i] = select_t(kind == 2, yl[i], select_t(kind == 1, i < 4 ? yc[i < 4 ? i : 0] : zero, di));
    }
  }
  __host__ __device__ static void knot_dims(const double* P, int N, int k, int* nx, int* nu) {  // knot k = 0 .. N-1 (host: table = host copy)
    const double* tab = table(P);
    if (k < N - 1) { *nx = (int)tab[(size_t)REC * k + 1]; *nu = (int)tab[(size_t)REC * k + 2]; }
    else { *nx = (int)tab[(size_t)REC * (N - 2) + 3]; *nu = (int)tab[(size_t)REC * (N - 2) + 2]; }
  }
};

// Altro's InfeasibleModel (the state augmentation behind ALTRO's infeasible start; SURVEY §8(f)4, the reason the reference carries
// the change_dimension family: src/constraints.jl:820-936, src/constraint_list.jl:208-217, src/cost_functions.jl:391-401):
//     x+ = f_d(x, u[0 .. m0)) + u[m0 .. m0 + n)
// — the discretised base model plus one slack control per state, so that ANY state trajectory (initial_states!, src/problem.jl:242-253)
// is dynamically feasible with the slacks w_k = X_{k+1} - f_d(X_k, U_k) (to_infeasible_controls).  The host mirrors compose the rest
// exactly as Altro does: costs and constraints lifted with change_dimension, R_inf on the slacks, the equality w = 0 on every stage
// knot.  A hybrid-style model (step hook): every kernel takes its time step through model_step; dual numbers flow through the sum, so
// the expansions see [A  B  I].  Vector-space bases only.
template <class Base>
struct InfeasibleModel {
  static_assert(!Base::lie, "infeasible start: vector-space base models only");
  static constexpr int n = Base::n, m = Base::m + Base::n, ne = Base::ne, m0 = Base::m;
  static constexpr bool lie = false;
  static constexpr int att = ATT_NONE;
  static constexpr bool hybrid = true;
  static constexpr bool pin_rk4 = true;
  static constexpr int expand_knots = 1;
  static constexpr bool accept_write_through = true;
  static constexpr bool lds_gains = false;
  static constexpr int ls_first_round = 4;
  static constexpr bool mfma_backward = false, coop_backward = true;
  static constexpr bool lane_backward = false;
  template <class T>
  __device__ __forceinline__ static void f(const double* P, const T* x, const T* u, T* xd) { Base::f(P, x, u, xd); }  // (unused: step dispatches)
  template <class T, int FIXED>
  __device__ __forceinline__ static void step(const double* P, int integrator_rt, int k, const T* x, const T* u, double h, T* xn) {
    T y[n];
    rk_step<Base, T, FIXED>(P, integrator_rt, x, u, h, y);  // reads u[0 .. m0)
#pragma unroll
    for (int i = 0; i < n; ++i) xn[i] = y[i] + u[m0 + i];
  }
  __host__ __device__ static void knot_dims(const double*, int, int, int* nx, int* nu) { *nx = n; *nu = m; }
};

// ------------------------------------------------------------------------------------------------
// Error-state maps.  Identity for vector-space models.
// ------------------------------------------------------------------------------------------------
// v (n) = column j of G(x)   (n x ne attitude Jacobian blkdiag(I3, L(q)H, I3, I3); no 1/2: Cayley map; three-parameter
// attitudes: blkdiag(I3, D(p), I3, I3), att_differential)
template <class M>
__device__ __forceinline__ void errstate_col(const double* x, int j, double* v) {
  constexpr int n = M::n;
#pragma unroll
  for (int i = 0; i < n; ++i) v[i] = 0.0;
  if constexpr (!M::lie) {
#pragma unroll
    for (int i = 0; i < n; ++i) v[i] = (i == j) ? 1.0 : 0.0;
  } else if constexpr (M::att == ATT_QUAT) {
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
  } else {
    double D[9];
    att_differential<M::att>(x + 3, D);
#pragma unroll
    for (int i = 0; i < n; ++i) v[i] = (i == j) ? 1.0 : 0.0;
    if (j >= 3 && j < 6) {
#pragma unroll
      for (int i = 0; i < 3; ++i) v[3 + i] = (j == 3) ? D[3 * i] : (j == 4) ? D[3 * i + 1] : D[3 * i + 2];
    }
  }
}

// out (ne) = G(x)' y (n)
template <class M>
__device__ __forceinline__ void errstate_tmul(const double* x, const double* y, double* out) {
  if constexpr (!M::lie) {
#pragma unroll
    for (int i = 0; i < M::n; ++i) out[i] = y[i];
  } else if constexpr (M::att == ATT_QUAT) {
    const double w = x[3], a = x[4], b = x[5], c = x[6];
    out[0] = y[0]; out[1] = y[1]; out[2] = y[2];
    out[3] = -a * y[3] + w * y[4] + c * y[5] - b * y[6];
    out[4] = -b * y[3] - c * y[4] + w * y[5] + a * y[6];
    out[5] = -c * y[3] + b * y[4] - a * y[5] + w * y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) out[6 + i] = y[7 + i];
  } else {
    double D[9];
    att_differential<M::att>(x + 3, D);
#pragma unroll
    for (int i = 0; i < M::n; ++i) out[i] = y[i];
#pragma unroll
    for (int j = 0; j < 3; ++j) out[3 + j] = D[j] * y[3] + D[3 + j] * y[4] + D[6 + j] * y[5];
  }
}

// out (ne) = E(x) y (n),  E(x) = d(y' ⊖ x)/dy' at y' = x: the left inverse of G(x) that maps a change of x_{k+1} to a change of
// its error state.  Unit quaternions: G(x)' (orthonormal columns), as the reference stack has it (A_err = G(x⁺)'AG(x)); for a
// three-parameter attitude G'G ≠ I, so the consistent Jacobian of state_diff is used: D(p)⁻¹ in closed form,
// MRP 4 D'/(1+|p|²)², RodriguesParam (I − [g]×)/(1+|g|²).
template <class M>
__device__ __forceinline__ void errstate_invmul(const double* x, const double* y, double* out) {
  if constexpr (M::att == ATT_MRP || M::att == ATT_RP) {
#pragma unroll
    for (int i = 0; i < M::n; ++i) out[i] = y[i];
    const double a = x[3], b = x[4], c = x[5];
    const double rn = rcp_fast(1.0 + (a * a + b * b + c * c));
    if constexpr (M::att == ATT_MRP) {
      double D[9];
      att_differential<ATT_MRP>(x + 3, D);
      const double f = 4.0 * rn * rn;
#pragma unroll
      for (int i = 0; i < 3; ++i) out[3 + i] = f * (D[i] * y[3] + D[3 + i] * y[4] + D[6 + i] * y[5]);  // 4 D'/(1+n)² y  (row i of D' = column i of D)
    } else {
      out[3] = rn * (y[3] + c * y[4] - b * y[5]);   // (I − [g]×) y / (1+n)
      out[4] = rn * (-c * y[3] + y[4] + a * y[5]);
      out[5] = rn * (b * y[3] - a * y[4] + y[5]);
    }
  } else errstate_tmul<M>(x, y, out);
}

// dx (ne) = x (-) x0   (RD.state_diff, Cayley map: the Rodrigues vector of the relative rotation, vec(q0⁻¹ ⊗ q)/scalar(q0⁻¹ ⊗ q);
// scale-invariant, so three-parameter attitudes use their unnormalised quaternions q̃ = [1−|p|², 2p] resp. [1, g])
template <class M>
__device__ __forceinline__ void state_diff(const double* x, const double* x0, double* dx) {
  if constexpr (!M::lie) {
#pragma unroll
    for (int i = 0; i < M::n; ++i) dx[i] = x[i] - x0[i];
  } else {
#pragma unroll
    for (int i = 0; i < 3; ++i) dx[i] = x[i] - x0[i];
    double w0, a0, b0, c0, w, a, b, c;
    if constexpr (M::att == ATT_QUAT) {
      w0 = x0[3]; a0 = x0[4]; b0 = x0[5]; c0 = x0[6];
      w = x[3]; a = x[4]; b = x[5]; c = x[6];
    } else if constexpr (M::att == ATT_MRP) {
      w0 = 1.0 - (x0[3] * x0[3] + x0[4] * x0[4] + x0[5] * x0[5]); a0 = 2.0 * x0[3]; b0 = 2.0 * x0[4]; c0 = 2.0 * x0[5];
      w = 1.0 - (x[3] * x[3] + x[4] * x[4] + x[5] * x[5]); a = 2.0 * x[3]; b = 2.0 * x[4]; c = 2.0 * x[5];
    } else {
      w0 = 1.0; a0 = x0[3]; b0 = x0[4]; c0 = x0[5];
      w = 1.0; a = x[3]; b = x[4]; c = x[5];
    }
    const double s = w0 * w + a0 * a + b0 * b + c0 * c;
    const double v1 = w0 * a - a0 * w - (b0 * c - c0 * b);
    const double v2 = w0 * b - b0 * w - (c0 * a - a0 * c);
    const double v3 = w0 * c - c0 * w - (a0 * b - b0 * a);
    const double rs = rcp_fast(s);
    dx[3] = v1 * rs; dx[4] = v2 * rs; dx[5] = v3 * rs;
    constexpr int o = (M::att == ATT_QUAT) ? 7 : 6;  // first velocity entry of the state
#pragma unroll
    for (int i = 0; i < 6; ++i) dx[6 + i] = x[o + i] - x0[o + i];
  }
}

// xo (n) = x (+) dx (ne): the inverse of state_diff (state_diff(x (+) dx, x) = dx with the Cayley map) — the retraction the
// projected-Newton polish (k_pn.h) moves states along.  Attitude: q (x) [1, phi] / sqrt(1 + |phi|^2) keeps |q|; three-parameter
// attitudes compose through their unnormalised quaternion and map back (MRP p = v / (|q| + w), RodriguesParam g = v / w).
template <class M>
__device__ __forceinline__ void state_add(const double* x, const double* dx, double* xo) {
  if constexpr (!M::lie) {
#pragma unroll
    for (int i = 0; i < M::n; ++i) xo[i] = x[i] + dx[i];
  } else {
#pragma unroll
    for (int i = 0; i < 3; ++i) xo[i] = x[i] + dx[i];
    constexpr int o = (M::att == ATT_QUAT) ? 7 : 6;
#pragma unroll
    for (int i = 0; i < 6; ++i) xo[o + i] = x[o + i] + dx[6 + i];
    double w, a, b, c;
    if constexpr (M::att == ATT_QUAT) { w = x[3]; a = x[4]; b = x[5]; c = x[6]; }
    else if constexpr (M::att == ATT_MRP) { w = 1.0 - (x[3] * x[3] + x[4] * x[4] + x[5] * x[5]); a = 2.0 * x[3]; b = 2.0 * x[4]; c = 2.0 * x[5]; }
    else { w = 1.0; a = x[3]; b = x[4]; c = x[5]; }
    const double f1 = dx[3], f2 = dx[4], f3 = dx[5];
    const double r0 = w - a * f1 - b * f2 - c * f3;
    const double r1 = a + w * f1 + b * f3 - c * f2;
    const double r2 = b + w * f2 + c * f1 - a * f3;
    const double r3 = c + w * f3 + a * f2 - b * f1;
    if constexpr (M::att == ATT_QUAT) {
      const double s = 1.0 / sqrt(1.0 + (f1 * f1 + f2 * f2 + f3 * f3));
      xo[3] = r0 * s; xo[4] = r1 * s; xo[5] = r2 * s; xo[6] = r3 * s;
    } else if constexpr (M::att == ATT_MRP) {
      const double s = 1.0 / (sqrt(r0 * r0 + r1 * r1 + r2 * r2 + r3 * r3) + r0);
      xo[3] = r1 * s; xo[4] = r2 * s; xo[5] = r3 * s;
    } else {
      const double s = 1.0 / r0;
      xo[3] = r1 * s; xo[4] = r2 * s; xo[5] = r3 * s;
    }
  }
}

}  // namespace to

/*
 * oracle_math.h — TEST INFRASTRUCTURE ONLY (CPU oracle).  Not part of the product; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may build/load anything under oracle/.
 *
 * Plain scalar restatement of the arithmetic on the hot path:
 *   - models:      double integrator (examples/quickstart.jl:11-23), Cartpole (docs/src/model.md:20-51),
 *                  Quadrotor / RigidBody (examples/Quadrotor.ipynb cells 4,8)
 *   - integrators: RK4 / RK3 / Euler (default RK4 src/problem.jl:120)
 *   - analytic continuous Jacobians + exact RK Jacobian by the chain rule
 *   - error-state maps for the quaternion model
 *   - cones (src/cones.jl), costs (src/cost_functions.jl, src/lie_costs.jl), constraints (src/constraints.jl)
 * Everything is runtime-dimensioned, dense, loop-based: deliberately the opposite of the GPU code's
 * compile-time-specialised forward-mode implementation, so the two check each other.
 */
#ifndef ORACLE_MATH_H
#define ORACLE_MATH_H

#include <cmath>
#include <cstring>
#include <vector>

#include "../include/trajopt_hip.h"

namespace oracle {

constexpr int MAXN = TO_MAX_N;
constexpr int MAXM = TO_MAX_M;
constexpr int MAXZ = TO_MAX_N + TO_MAX_M;

struct Model {
  int id = 0;
  int n = 0, m = 0, ne = 0; /* ne = error-state dim (RD.errstate_dim): n, or 12 for the quadrotor */
  double p[16];
  const to_step_model* steps = nullptr; /* TO_MODEL_VECTOR: one model per time step (the descriptor's table, owned by the Problem) */
  /* attitude representation of a rigid-body state (to_rotation; -1: vector-space model) */
  int rot() const { return id == TO_MODEL_QUADROTOR ? (int)p[10] : -1; }
  bool lie() const { return id == TO_MODEL_QUADROTOR; }
};

inline int model_dims(int id, const double* params, int* n, int* m, int* ne) {
  switch (id) {
    case TO_MODEL_DOUBLE_INTEGRATOR: {
      int D = (int)params[1];
      if (D < 1 || D > 3) return -1;
      *n = 2 * D; *m = D; *ne = 2 * D; return 0;
    }
    case TO_MODEL_CARTPOLE: *n = 4; *m = 1; *ne = 4; return 0;
    case TO_MODEL_QUADROTOR: { /* params[10] = to_rotation: RigidBody{QuatRotation} n = 13, {MRP} / {RodriguesParam} n = 12 */
      int rot = (int)params[10];
      if (rot < TO_ROT_QUATERNION || rot > TO_ROT_RODRIGUES) return -1;
      *n = rot == TO_ROT_QUATERNION ? 13 : 12; *m = 4; *ne = 12; return 0;
    }
    case TO_MODEL_HYBRID_DOUBLE_INTEGRATOR: *n = 4; *m = 2; *ne = 4; return 0; /* stored at the largest (n, m) of its phases */
    case TO_MODEL_VECTOR: *n = TO_VECTOR_N; *m = TO_VECTOR_M; *ne = TO_VECTOR_N; return 0; /* stored at (6, 3), narrower knots zero-padded */
    case TO_MODEL_INFEASIBLE: { /* Altro's InfeasibleModel over a vector-space base (params[15] = its to_model_id): one slack control per state */
      int base = (int)params[15], n0, m0, ne0;
      if (base != TO_MODEL_DOUBLE_INTEGRATOR && base != TO_MODEL_CARTPOLE) return -1;
      if (model_dims(base, params, &n0, &m0, &ne0) != 0 || m0 + n0 > TO_MAX_M) return -1;
      *n = n0; *m = m0 + n0; *ne = ne0; return 0;
    }
  }
  return -1;
}

/* ---- three-parameter attitudes (RigidBody{MRP}, RigidBody{RodriguesParam}; src/lie_costs.jl:1-3) -------------------------
 * Rotations.kinematics:  MRP  pdot = 1/4 [(1-|p|^2) w + 2 p x w + 2 p (p.w)]     RodriguesParam  gdot = 1/2 [w + g x w + g (g.w)]
 * D(p) = d(p (+) phi)/dphi at 0 with the Cayley error map (so pdot = D(p) w / 2):
 *   MRP  1/2 [(1-|p|^2) I + 2[p]x + 2pp']        RodriguesParam  I + [g]x + gg'                      (row-major 3x3) */
inline void att_differential(int rot, const double* p, double* D) {
  const double a = p[0], b = p[1], c = p[2];
  const double h = rot == TO_ROT_MRP ? 0.5 * (1.0 - (a * a + b * b + c * c)) : 1.0;
  D[0] = h + a * a;  D[1] = -c + a * b; D[2] = b + a * c;
  D[3] = c + b * a;  D[4] = h + b * b;  D[5] = -a + b * c;
  D[6] = -b + c * a; D[7] = a + c * b;  D[8] = h + c * c;
}
/* Hessian of phi -> b.(p (+) phi) at 0 (Rotations ∇²differential): the second-order term of the error-state cost Hessian.
 *   MRP  1/2 [a p' + p a'] - (b.p) [(1+|p|^2)/2 I - 2 pp'],  a = (1-|p|^2) b + 2 b x p
 *   RP   c g' + g c' + 2 (b.g) g g',                         c = b + b x g */
inline void att_differential2(int rot, const double* p, const double* b, double* H) {
  const double bp = b[0] * p[0] + b[1] * p[1] + b[2] * p[2];
  const double cx[3] = {b[1] * p[2] - b[2] * p[1], b[2] * p[0] - b[0] * p[2], b[0] * p[1] - b[1] * p[0]};
  if (rot == TO_ROT_MRP) {
    const double n2 = p[0] * p[0] + p[1] * p[1] + p[2] * p[2];
    double a[3];
    for (int i = 0; i < 3; ++i) a[i] = (1.0 - n2) * b[i] + 2.0 * cx[i];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j)
      H[3 * i + j] = 0.5 * (a[i] * p[j] + p[i] * a[j]) - bp * ((i == j ? 0.5 * (1.0 + n2) : 0.0) - 2.0 * p[i] * p[j]);
  } else {
    double c[3];
    for (int i = 0; i < 3; ++i) c[i] = b[i] + cx[i];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) H[3 * i + j] = c[i] * p[j] + p[i] * c[j] + 2.0 * bp * p[i] * p[j];
  }
}
/* unnormalised quaternion of a three-parameter attitude: MRP [1-|p|^2, 2p] (norm 1+|p|^2), RodriguesParam [1, g] */
inline void att_quat(int rot, const double* p, double* q) {
  if (rot == TO_ROT_MRP) { q[0] = 1.0 - (p[0] * p[0] + p[1] * p[1] + p[2] * p[2]); q[1] = 2 * p[0]; q[2] = 2 * p[1]; q[3] = 2 * p[2]; }
  else { q[0] = 1.0; q[1] = p[0]; q[2] = p[1]; q[3] = p[2]; }
}
/* a(p) = R(p) e3 (third column of the rotation matrix) and its Jacobian da/dp (row-major 3x3), in closed form:
 *   MRP  a = [c0 e3 + 8 p3 p + 4 s (p x e3)] / (1+n)^2,  s = 1-n, c0 = s^2 - 4n, n = |p|^2
 *   RP   a = [(1-n) e3 + 2 g3 g + 2 (g x e3)] / (1+n) */
inline void att_thrust_axis(int rot, const double* p, double* a, double* da) {
  const double n = p[0] * p[0] + p[1] * p[1] + p[2] * p[2];
  const double pxe[3] = {p[1], -p[0], 0.0}; /* p x e3 */
  double N[3], dN[9], W, dW[3];
  if (rot == TO_ROT_MRP) {
    const double s = 1.0 - n, c0 = s * s - 4.0 * n;
    for (int i = 0; i < 3; ++i) N[i] = (i == 2 ? c0 : 0.0) + 8.0 * p[2] * p[i] + 4.0 * s * pxe[i];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
      double d = 0.0;
      if (i == 2) d += -4.0 * p[j] * (3.0 - n);                 /* d c0 / d p_j */
      d += 8.0 * ((j == 2 ? p[i] : 0.0) + (i == j ? p[2] : 0.0)); /* d (8 p3 p_i) */
      d += 4.0 * (-2.0 * p[j]) * pxe[i];                         /* d (4 s) (p x e3)_i */
      if (i == 0 && j == 1) d += 4.0 * s;
      if (i == 1 && j == 0) d += -4.0 * s;
      dN[3 * i + j] = d;
    }
    W = 1.0 / ((1.0 + n) * (1.0 + n));
    for (int j = 0; j < 3; ++j) dW[j] = -4.0 * W * p[j] / (1.0 + n);
  } else {
    for (int i = 0; i < 3; ++i) N[i] = (i == 2 ? 1.0 - n : 0.0) + 2.0 * p[2] * p[i] + 2.0 * pxe[i];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
      double d = 0.0;
      if (i == 2) d += -2.0 * p[j];
      d += 2.0 * ((j == 2 ? p[i] : 0.0) + (i == j ? p[2] : 0.0));
      if (i == 0 && j == 1) d += 2.0;
      if (i == 1 && j == 0) d += -2.0;
      dN[3 * i + j] = d;
    }
    W = 1.0 / (1.0 + n);
    for (int j = 0; j < 3; ++j) dW[j] = -2.0 * W * W * p[j];
  }
  for (int i = 0; i < 3; ++i) {
    a[i] = N[i] * W;
    if (da) for (int j = 0; j < 3; ++j) da[3 * i + j] = dN[3 * i + j] * W + N[i] * dW[j];
  }
}

/* ---------------------------------------------------------------- continuous dynamics xdot = f(x,u) */
inline void dynamics(const Model& M, const double* x, const double* u, double* xd) {
  switch (M.id) {
    case TO_MODEL_DOUBLE_INTEGRATOR: { /* examples/quickstart.jl:15-20 */
      int D = M.m; double mass = M.p[0];
      for (int i = 0; i < D; ++i) { xd[i] = x[D + i]; xd[D + i] = u[i] / mass; }
      return;
    }
    case TO_MODEL_CARTPOLE: { /* docs/src/model.md:34-50 */
      double mc = M.p[0], mp = M.p[1], l = M.p[2], g = M.p[3];
      double qd1 = x[2], qd2 = x[3];
      double s = std::sin(x[1]), c = std::cos(x[1]);
      double h11 = mc + mp, h12 = mp * l * c, h21 = mp * l * c, h22 = mp * l * l;
      double c12 = -mp * qd2 * l * s;
      /* b = C*qd + G - B*u */
      double b1 = (0.0 * qd1 + c12 * qd2) + 0.0 - 1.0 * u[0];
      double b2 = (0.0 * qd1 + 0.0 * qd2) + mp * g * l * s - 0.0 * u[0];
      /* StaticArrays 2x2 solve: x = ((a22 b1 - a12 b2)/d, (a11 b2 - a21 b1)/d) */
      double d = h11 * h22 - h12 * h21;
      double s1 = (h22 * b1 - h12 * b2) / d;
      double s2 = (h11 * b2 - h21 * b1) / d;
      xd[0] = qd1; xd[1] = qd2; xd[2] = -s1; xd[3] = -s2;
      return;
    }
    case TO_MODEL_QUADROTOR: { /* RigidBody dynamics, world-frame velocity (bodyframe=false) */
      if (M.rot() != TO_ROT_QUATERNION) { /* x = [r, p, v, w]: three-parameter attitude */
        const int rot = M.rot();
        const double mass = M.p[0], J1 = M.p[1], J2 = M.p[2], J3 = M.p[3], L = M.p[7], kf = M.p[8], km = M.p[9];
        const double* p = x + 3; const double* w = x + 9;
        double F[4], Fz = 0.0;
        for (int i = 0; i < 4; ++i) { F[i] = std::fmax(0.0, kf * u[i]); Fz += F[i]; }
        const double n2 = p[0] * p[0] + p[1] * p[1] + p[2] * p[2], pw = p[0] * w[0] + p[1] * w[1] + p[2] * w[2];
        const double c[3] = {p[1] * w[2] - p[2] * w[1], p[2] * w[0] - p[0] * w[2], p[0] * w[1] - p[1] * w[0]};
        for (int i = 0; i < 3; ++i) {
          xd[i] = x[6 + i];
          xd[3 + i] = rot == TO_ROT_MRP ? 0.25 * ((1.0 - n2) * w[i] + 2.0 * c[i] + 2.0 * p[i] * pw) : 0.5 * (w[i] + c[i] + p[i] * pw);
        }
        double a[3]; att_thrust_axis(rot, p, a, nullptr);
        for (int i = 0; i < 3; ++i) xd[6 + i] = (mass * M.p[4 + i] + a[i] * Fz) / mass;
        const double t1 = L * (F[1] - F[3]), t2 = L * (F[2] - F[0]), t3 = km * u[0] - km * u[1] + km * u[2] - km * u[3];
        const double Jw1 = J1 * w[0], Jw2 = J2 * w[1], Jw3 = J3 * w[2];
        xd[9] = (1.0 / J1) * (t1 - (w[1] * Jw3 - w[2] * Jw2));
        xd[10] = (1.0 / J2) * (t2 - (w[2] * Jw1 - w[0] * Jw3));
        xd[11] = (1.0 / J3) * (t3 - (w[0] * Jw2 - w[1] * Jw1));
        return;
      }
      double mass = M.p[0], J1 = M.p[1], J2 = M.p[2], J3 = M.p[3];
      double g1 = M.p[4], g2 = M.p[5], g3 = M.p[6];
      double L = M.p[7], kf = M.p[8], km = M.p[9];
      double qw = x[3], qx = x[4], qy = x[5], qz = x[6];
      double w1 = x[10], w2 = x[11], w3 = x[12];
      double F1 = std::fmax(0.0, kf * u[0]), F2 = std::fmax(0.0, kf * u[1]);
      double F3 = std::fmax(0.0, kf * u[2]), F4 = std::fmax(0.0, kf * u[3]);
      double Fz = F1 + F2 + F3 + F4; /* body-frame thrust [0,0,Fz] */
      /* q*F, NOT normalised: (w^2 - v'v) r + 2 v (v'r) + 2 w (v x r) with r=(0,0,Fz) */
      double vv = qx * qx + qy * qy + qz * qz;
      double sc = qw * qw - vv;
      double vr = qz * Fz;
      double qF1 = sc * 0.0 + 2.0 * qx * vr + 2.0 * qw * (qy * Fz - qz * 0.0);
      double qF2 = sc * 0.0 + 2.0 * qy * vr + 2.0 * qw * (qz * 0.0 - qx * Fz);
      double qF3 = sc * Fz + 2.0 * qz * vr + 2.0 * qw * (qx * 0.0 - qy * 0.0);
      double Fw1 = mass * g1 + qF1, Fw2 = mass * g2 + qF2, Fw3 = mass * g3 + qF3;
      double t1 = L * (F2 - F4), t2 = L * (F3 - F1);
      double t3 = km * u[0] - km * u[1] + km * u[2] - km * u[3];
      /* rdot = v */
      xd[0] = x[7]; xd[1] = x[8]; xd[2] = x[9];
      /* qdot = 0.5 * q (x) (0, omega) */
      xd[3] = 0.5 * (-qx * w1 - qy * w2 - qz * w3);
      xd[4] = 0.5 * (qw * w1 + qy * w3 - qz * w2);
      xd[5] = 0.5 * (qw * w2 - qx * w3 + qz * w1);
      xd[6] = 0.5 * (qw * w3 + qx * w2 - qy * w1);
      /* vdot = F/m */
      xd[7] = Fw1 / mass; xd[8] = Fw2 / mass; xd[9] = Fw3 / mass;
      /* omegadot = Jinv (tau - omega x J omega) */
      double Jw1 = J1 * w1, Jw2 = J2 * w2, Jw3 = J3 * w3;
      xd[10] = (1.0 / J1) * (t1 - (w2 * Jw3 - w3 * Jw2));
      xd[11] = (1.0 / J2) * (t2 - (w3 * Jw1 - w1 * Jw3));
      xd[12] = (1.0 / J3) * (t3 - (w1 * Jw2 - w2 * Jw1));
      return;
    }
  }
}

/* analytic continuous Jacobians fx (n x n), fu (n x m), row-major [i*n + j] / [i*m + j] */
inline void dynamics_jacobian(const Model& M, const double* x, const double* u, double* fx, double* fu) {
  const int n = M.n, m = M.m;
  std::memset(fx, 0, sizeof(double) * n * n);
  std::memset(fu, 0, sizeof(double) * n * m);
  switch (M.id) {
    case TO_MODEL_DOUBLE_INTEGRATOR: {
      int D = M.m; double mass = M.p[0];
      for (int i = 0; i < D; ++i) { fx[i * n + D + i] = 1.0; fu[(D + i) * m + i] = 1.0 / mass; }
      return;
    }
    case TO_MODEL_CARTPOLE: {
      double mc = M.p[0], mp = M.p[1], l = M.p[2], g = M.p[3];
      double om = x[3];
      double s = std::sin(x[1]), c = std::cos(x[1]);
      double a11 = mc + mp, a12 = mp * l * c, a22 = mp * l * l;
      double b1 = -mp * l * s * om * om - u[0];
      double b2 = mp * g * l * s;
      double d = a11 * a22 - a12 * a12;
      double x1 = (a22 * b1 - a12 * b2) / d;
      double x2 = (a11 * b2 - a12 * b1) / d;
      /* d/dtheta */
      double db1 = -mp * l * c * om * om, db2 = mp * g * l * c, da12 = -mp * l * s;
      double dd = -2.0 * a12 * da12;
      double dx1_t = (a22 * db1 - da12 * b2 - a12 * db2) / d - x1 * dd / d;
      double dx2_t = (a11 * db2 - da12 * b1 - a12 * db1) / d - x2 * dd / d;
      /* d/domega */
      double db1_o = -2.0 * mp * l * s * om;
      double dx1_o = a22 * db1_o / d, dx2_o = -a12 * db1_o / d;
      fx[0 * 4 + 2] = 1.0; fx[1 * 4 + 3] = 1.0;
      fx[2 * 4 + 1] = -dx1_t; fx[2 * 4 + 3] = -dx1_o;
      fx[3 * 4 + 1] = -dx2_t; fx[3 * 4 + 3] = -dx2_o;
      /* d/du: db1/du = -1 */
      fu[2] = a22 / d; fu[3] = -a12 / d;
      return;
    }
    case TO_MODEL_QUADROTOR: {
      if (M.rot() != TO_ROT_QUATERNION) { /* hand-derived, like the quaternion case below; x = [r, p, v, w], n = 12 */
        const int rot = M.rot();
        const double mass = M.p[0], J1 = M.p[1], J2 = M.p[2], J3 = M.p[3], L = M.p[7], kf = M.p[8], km = M.p[9];
        const double* p = x + 3; const double* w = x + 9;
        double dF[4], Fz = 0.0;
        for (int i = 0; i < 4; ++i) { dF[i] = (kf * u[i] > 0.0) ? kf : 0.0; Fz += std::fmax(0.0, kf * u[i]); }
        for (int i = 0; i < 3; ++i) fx[i * 12 + 6 + i] = 1.0;
        const double n2 = p[0] * p[0] + p[1] * p[1] + p[2] * p[2], pw = p[0] * w[0] + p[1] * w[1] + p[2] * w[2];
        /* [w]x */
        const double Wx[9] = {0, -w[2], w[1], w[2], 0, -w[0], -w[1], w[0], 0};
        const double Px[9] = {0, -p[2], p[1], p[2], 0, -p[0], -p[1], p[0], 0};
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
          const double I = (i == j) ? 1.0 : 0.0;
          if (rot == TO_ROT_MRP) {
            /* d/dp 1/4[(1-n2) w + 2 p x w + 2 p (p.w)] = 1/4[-2 w p' - 2 [w]x + 2 (p.w) I + 2 p w'] */
            fx[(3 + i) * 12 + 3 + j] = 0.25 * (-2.0 * w[i] * p[j] - 2.0 * Wx[3 * i + j] + 2.0 * pw * I + 2.0 * p[i] * w[j]);
            fx[(3 + i) * 12 + 9 + j] = 0.25 * ((1.0 - n2) * I + 2.0 * Px[3 * i + j] + 2.0 * p[i] * p[j]);
          } else {
            /* d/dg 1/2[w + g x w + g (g.w)] = 1/2[-[w]x + (g.w) I + g w'] */
            fx[(3 + i) * 12 + 3 + j] = 0.5 * (-Wx[3 * i + j] + pw * I + p[i] * w[j]);
            fx[(3 + i) * 12 + 9 + j] = 0.5 * (I + Px[3 * i + j] + p[i] * p[j]);
          }
        }
        double a[3], da[9]; att_thrust_axis(rot, p, a, da);
        for (int i = 0; i < 3; ++i) {
          for (int j = 0; j < 3; ++j) fx[(6 + i) * 12 + 3 + j] = da[3 * i + j] * Fz / mass;
          for (int j = 0; j < 4; ++j) fu[(6 + i) * 4 + j] = a[i] * dF[j] / mass;
        }
        fx[9 * 12 + 10] = -(J3 - J2) * w[2] / J1;  fx[9 * 12 + 11] = -(J3 - J2) * w[1] / J1;
        fx[10 * 12 + 9] = -(J1 - J3) * w[2] / J2;  fx[10 * 12 + 11] = -(J1 - J3) * w[0] / J2;
        fx[11 * 12 + 9] = -(J2 - J1) * w[1] / J3;  fx[11 * 12 + 10] = -(J2 - J1) * w[0] / J3;
        fu[9 * 4 + 1] = L * dF[1] / J1;  fu[9 * 4 + 3] = -L * dF[3] / J1;
        fu[10 * 4 + 2] = L * dF[2] / J2; fu[10 * 4 + 0] = -L * dF[0] / J2;
        fu[11 * 4 + 0] = km / J3; fu[11 * 4 + 1] = -km / J3; fu[11 * 4 + 2] = km / J3; fu[11 * 4 + 3] = -km / J3;
        return;
      }
      double mass = M.p[0], J1 = M.p[1], J2 = M.p[2], J3 = M.p[3];
      double L = M.p[7], kf = M.p[8], km = M.p[9];
      double qw = x[3], qx = x[4], qy = x[5], qz = x[6];
      double w1 = x[10], w2 = x[11], w3 = x[12];
      double dF[4], Fz = 0.0;
      for (int i = 0; i < 4; ++i) { dF[i] = (kf * u[i] > 0.0) ? kf : 0.0; Fz += std::fmax(0.0, kf * u[i]); }
      /* rdot = v */
      fx[0 * 13 + 7] = 1.0; fx[1 * 13 + 8] = 1.0; fx[2 * 13 + 9] = 1.0;
      /* qdot wrt q */
      fx[3 * 13 + 4] = -0.5 * w1; fx[3 * 13 + 5] = -0.5 * w2; fx[3 * 13 + 6] = -0.5 * w3;
      fx[4 * 13 + 3] = 0.5 * w1;  fx[4 * 13 + 5] = 0.5 * w3;  fx[4 * 13 + 6] = -0.5 * w2;
      fx[5 * 13 + 3] = 0.5 * w2;  fx[5 * 13 + 4] = -0.5 * w3; fx[5 * 13 + 6] = 0.5 * w1;
      fx[6 * 13 + 3] = 0.5 * w3;  fx[6 * 13 + 4] = 0.5 * w2;  fx[6 * 13 + 5] = -0.5 * w1;
      /* qdot wrt omega */
      fx[3 * 13 + 10] = -0.5 * qx; fx[3 * 13 + 11] = -0.5 * qy; fx[3 * 13 + 12] = -0.5 * qz;
      fx[4 * 13 + 10] = 0.5 * qw;  fx[4 * 13 + 11] = -0.5 * qz; fx[4 * 13 + 12] = 0.5 * qy;
      fx[5 * 13 + 10] = 0.5 * qz;  fx[5 * 13 + 11] = 0.5 * qw;  fx[5 * 13 + 12] = -0.5 * qx;
      fx[6 * 13 + 10] = -0.5 * qy; fx[6 * 13 + 11] = 0.5 * qx;  fx[6 * 13 + 12] = 0.5 * qw;
      /* vdot = g + a(q) Fz/m, a = q*(0,0,1) = (2(xz+wy), 2(yz-wx), w^2-x^2-y^2+z^2) */
      double k = Fz / mass;
      fx[7 * 13 + 3] = k * 2.0 * qy;  fx[7 * 13 + 4] = k * 2.0 * qz;  fx[7 * 13 + 5] = k * 2.0 * qw;  fx[7 * 13 + 6] = k * 2.0 * qx;
      fx[8 * 13 + 3] = -k * 2.0 * qx; fx[8 * 13 + 4] = -k * 2.0 * qw; fx[8 * 13 + 5] = k * 2.0 * qz;  fx[8 * 13 + 6] = k * 2.0 * qy;
      fx[9 * 13 + 3] = k * 2.0 * qw;  fx[9 * 13 + 4] = -k * 2.0 * qx; fx[9 * 13 + 5] = -k * 2.0 * qy; fx[9 * 13 + 6] = k * 2.0 * qz;
      double a1 = 2.0 * (qx * qz + qw * qy), a2 = 2.0 * (qy * qz - qw * qx);
      double a3 = qw * qw - qx * qx - qy * qy + qz * qz;
      for (int i = 0; i < 4; ++i) {
        fu[7 * 4 + i] = a1 * dF[i] / mass; fu[8 * 4 + i] = a2 * dF[i] / mass; fu[9 * 4 + i] = a3 * dF[i] / mass;
      }
      /* omegadot */
      fx[10 * 13 + 11] = -(J3 - J2) * w3 / J1; fx[10 * 13 + 12] = -(J3 - J2) * w2 / J1;
      fx[11 * 13 + 10] = -(J1 - J3) * w3 / J2; fx[11 * 13 + 12] = -(J1 - J3) * w1 / J2;
      fx[12 * 13 + 10] = -(J2 - J1) * w2 / J3; fx[12 * 13 + 11] = -(J2 - J1) * w1 / J3;
      /* tau = [L(F2-F4), L(F3-F1), km(u1-u2+u3-u4)] */
      fu[10 * 4 + 1] = L * dF[1] / J1; fu[10 * 4 + 3] = -L * dF[3] / J1;
      fu[11 * 4 + 2] = L * dF[2] / J2; fu[11 * 4 + 0] = -L * dF[0] / J2;
      fu[12 * 4 + 0] = km / J3; fu[12 * 4 + 1] = -km / J3; fu[12 * 4 + 2] = km / J3; fu[12 * 4 + 3] = -km / J3;
      return;
    }
  }
}

/* ---------------------------------------------------------------- discrete dynamics */
inline void discrete_dynamics(const Model& M, int integrator, const double* x, const double* u, double h, double* xn) {
  const int n = M.n;
  double k1[MAXN], k2[MAXN], k3[MAXN], k4[MAXN], xt[MAXN];
  if (integrator == TO_EULER) {
    dynamics(M, x, u, k1);
    for (int i = 0; i < n; ++i) xn[i] = x[i] + k1[i] * h;
    return;
  }
  if (integrator == TO_RK3) {
    dynamics(M, x, u, k1); for (int i = 0; i < n; ++i) k1[i] *= h;
    for (int i = 0; i < n; ++i) xt[i] = x[i] + k1[i] / 2;
    dynamics(M, xt, u, k2); for (int i = 0; i < n; ++i) k2[i] *= h;
    for (int i = 0; i < n; ++i) xt[i] = x[i] - k1[i] + 2 * k2[i];
    dynamics(M, xt, u, k3); for (int i = 0; i < n; ++i) k3[i] *= h;
    for (int i = 0; i < n; ++i) xn[i] = x[i] + (k1[i] + 4 * k2[i] + k3[i]) / 6;
    return;
  }
  dynamics(M, x, u, k1); for (int i = 0; i < n; ++i) k1[i] *= h;
  for (int i = 0; i < n; ++i) xt[i] = x[i] + k1[i] / 2;
  dynamics(M, xt, u, k2); for (int i = 0; i < n; ++i) k2[i] *= h;
  for (int i = 0; i < n; ++i) xt[i] = x[i] + k2[i] / 2;
  dynamics(M, xt, u, k3); for (int i = 0; i < n; ++i) k3[i] *= h;
  for (int i = 0; i < n; ++i) xt[i] = x[i] + k3[i];
  dynamics(M, xt, u, k4); for (int i = 0; i < n; ++i) k4[i] *= h;
  for (int i = 0; i < n; ++i) xn[i] = x[i] + (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]) / 6;
}

/* C (r x c) = A (r x k) * B (k x c), all row-major */
inline void matmul(const double* A, const double* B, double* C, int r, int k, int c) {
  for (int i = 0; i < r; ++i)
    for (int j = 0; j < c; ++j) {
      double s = 0.0;
      for (int t = 0; t < k; ++t) s += A[i * k + t] * B[t * c + j];
      C[i * c + j] = s;
    }
}

/* exact Jacobian of the RK step: A (n x n), Bm (n x m), row-major */
inline void discrete_jacobian(const Model& M, int integrator, const double* x, const double* u, double h,
                              double* A, double* Bm) {
  const int n = M.n, m = M.m;
  /* scratch lives in thread-local storage: the batched driver calls this ~1e7 times from many OpenMP threads */
  static thread_local std::vector<double> fx, fu, T, Tu, D1x, D1u, D2x, D2u, D3x, D3u, D4x, D4u, Du;
  for (auto* v : {&fx, &T, &D1x, &D2x, &D3x, &D4x}) v->resize(n * n);
  for (auto* v : {&fu, &Tu, &D1u, &D2u, &D3u, &D4u, &Du}) v->resize(n * m);
  double k1[MAXN], k2[MAXN], k3[MAXN], xt[MAXN];
  auto eye_plus = [&](const double* D, double scale, double* out) { /* out = I + scale*D */
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) out[i * n + j] = (i == j ? 1.0 : 0.0) + scale * D[i * n + j];
  };
  /* stage 1 */
  dynamics(M, x, u, k1);
  dynamics_jacobian(M, x, u, fx.data(), fu.data());
  for (int i = 0; i < n * n; ++i) D1x[i] = h * fx[i];
  for (int i = 0; i < n * m; ++i) D1u[i] = h * fu[i];
  if (integrator == TO_EULER) {
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) A[i * n + j] = (i == j ? 1.0 : 0.0) + D1x[i * n + j];
    for (int i = 0; i < n * m; ++i) Bm[i] = D1u[i];
    return;
  }
  for (int i = 0; i < n; ++i) k1[i] *= h;
  /* stage 2 at x + k1/2 */
  for (int i = 0; i < n; ++i) xt[i] = x[i] + k1[i] / 2;
  dynamics(M, xt, u, k2); for (int i = 0; i < n; ++i) k2[i] *= h;
  dynamics_jacobian(M, xt, u, fx.data(), fu.data());
  eye_plus(D1x.data(), 0.5, T.data());
  matmul(fx.data(), T.data(), D2x.data(), n, n, n);
  matmul(fx.data(), D1u.data(), Tu.data(), n, n, m);
  for (int i = 0; i < n * n; ++i) D2x[i] *= h;
  for (int i = 0; i < n * m; ++i) D2u[i] = h * (0.5 * Tu[i] + fu[i]);
  if (integrator == TO_RK3) {
    /* stage 3 at x - k1 + 2 k2 */
    for (int i = 0; i < n; ++i) xt[i] = x[i] - k1[i] + 2 * k2[i];
    dynamics_jacobian(M, xt, u, fx.data(), fu.data());
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j)
      T[i * n + j] = (i == j ? 1.0 : 0.0) - D1x[i * n + j] + 2.0 * D2x[i * n + j];
    matmul(fx.data(), T.data(), D3x.data(), n, n, n);
    for (int i = 0; i < n * m; ++i) Du[i] = -D1u[i] + 2.0 * D2u[i];
    matmul(fx.data(), Du.data(), Tu.data(), n, n, m);
    for (int i = 0; i < n * n; ++i) D3x[i] *= h;
    for (int i = 0; i < n * m; ++i) D3u[i] = h * (Tu[i] + fu[i]);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j)
      A[i * n + j] = (i == j ? 1.0 : 0.0) + (D1x[i * n + j] + 4.0 * D2x[i * n + j] + D3x[i * n + j]) / 6.0;
    for (int i = 0; i < n * m; ++i) Bm[i] = (D1u[i] + 4.0 * D2u[i] + D3u[i]) / 6.0;
    return;
  }
  /* stage 3 at x + k2/2 */
  for (int i = 0; i < n; ++i) xt[i] = x[i] + k2[i] / 2;
  dynamics(M, xt, u, k3); for (int i = 0; i < n; ++i) k3[i] *= h;
  dynamics_jacobian(M, xt, u, fx.data(), fu.data());
  eye_plus(D2x.data(), 0.5, T.data());
  matmul(fx.data(), T.data(), D3x.data(), n, n, n);
  matmul(fx.data(), D2u.data(), Tu.data(), n, n, m);
  for (int i = 0; i < n * n; ++i) D3x[i] *= h;
  for (int i = 0; i < n * m; ++i) D3u[i] = h * (0.5 * Tu[i] + fu[i]);
  /* stage 4 at x + k3 */
  for (int i = 0; i < n; ++i) xt[i] = x[i] + k3[i];
  dynamics_jacobian(M, xt, u, fx.data(), fu.data());
  eye_plus(D3x.data(), 1.0, T.data());
  matmul(fx.data(), T.data(), D4x.data(), n, n, n);
  matmul(fx.data(), D3u.data(), Tu.data(), n, n, m);
  for (int i = 0; i < n * n; ++i) D4x[i] *= h;
  for (int i = 0; i < n * m; ++i) D4u[i] = h * (Tu[i] + fu[i]);
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j)
    A[i * n + j] = (i == j ? 1.0 : 0.0) + (D1x[i * n + j] + 2.0 * D2x[i * n + j] + 2.0 * D3x[i * n + j] + D4x[i * n + j]) / 6.0;
  for (int i = 0; i < n * m; ++i) Bm[i] = (D1u[i] + 2.0 * D2u[i] + 2.0 * D3u[i] + D4u[i]) / 6.0;
}

/* ---------------------------------------------------------------- model vectors (src/dynamics.jl:15-31)
 * One time step of whatever model governs step k, and its Jacobian.  For every compiled-in model but the hybrid one this is the
 * RK step above.  TO_MODEL_HYBRID_DOUBLE_INTEGRATOR (test/hybrid_dynamics_model.jl:14-52) stores its states / controls zero-padded
 * at (4, 2): steps 0 .. S-1 (S = p[1]) a 2-D double integrator, step S the jump map x+ = [(x3 + x4)/2, (u1 + u2)/2] (:31-33),
 * later steps a 1-D double integrator on (x1, x2, u1). */
inline Model hybrid_phase(const Model& M, int D) {
  Model S; S.id = TO_MODEL_DOUBLE_INTEGRATOR; S.n = 2 * D; S.m = D; S.ne = 2 * D;
  std::memset(S.p, 0, sizeof(S.p)); S.p[0] = M.p[0]; S.p[1] = D;
  return S;
}
/* the model of one step of a model vector (TO_MODEL_VECTOR) as a stand-alone Model */
inline Model step_model(const to_step_model& s) {
  Model S; std::memset(S.p, 0, sizeof(S.p));
  S.n = s.n; S.m = s.m; S.ne = s.n;
  if (s.kind == TO_STEP_CARTPOLE) { S.id = TO_MODEL_CARTPOLE; for (int i = 0; i < 4; ++i) S.p[i] = s.params[i]; }
  else { S.id = TO_MODEL_DOUBLE_INTEGRATOR; S.p[0] = s.params[0]; S.p[1] = s.m; }
  return S;
}
inline void knot_dims(const Model& M, int k, int* nx, int* nu, int N = 0) { /* knot k = 0 .. N-1; RD.dims(models): the terminal knot carries the last model's control dimension */
  *nx = M.n; *nu = M.m;
  if (M.id == TO_MODEL_VECTOR) {
    if (k < N - 1) { *nx = M.steps[k].n; *nu = M.steps[k].m; } else { *nx = M.steps[N - 2].n_out; *nu = M.steps[N - 2].m; }
  }
  if (M.id == TO_MODEL_HYBRID_DOUBLE_INTEGRATOR) { const int S = (int)M.p[1]; *nx = k <= S ? 4 : 2; *nu = k <= S ? 2 : 1; }
}
/* Altro's InfeasibleModel (the state augmentation of ALTRO's infeasible start; SURVEY §8(f)4 — what the reference's change_dimension
 * family exists for, src/constraints.jl:820-936, src/constraint_list.jl:208-217, src/cost_functions.jl:391-401):
 *   x+ = f_d(x, u[0 .. m0)) + u[m0 .. m0 + n)        Jacobian [A  B  I] */
inline Model infeasible_base(const Model& M) {
  Model S; S.id = (int)M.p[15]; std::memcpy(S.p, M.p, sizeof(S.p)); S.p[15] = 0.0;
  S.n = M.n; S.m = M.m - M.n; S.ne = M.ne;
  return S;
}
inline void knot_step(const Model& M, int integrator, int k, const double* x, const double* u, double h, double* xn) {
  if (M.id == TO_MODEL_INFEASIBLE) {
    const Model S = infeasible_base(M);
    discrete_dynamics(S, integrator, x, u, h, xn);
    for (int i = 0; i < M.n; ++i) xn[i] = xn[i] + u[S.m + i];
    return;
  }
  if (M.id == TO_MODEL_VECTOR) { /* per-step model on the live prefix of the padded vectors; the padding of the result is zero */
    const to_step_model& s = M.steps[k];
    for (int i = 0; i < M.n; ++i) xn[i] = 0.0;
    if (s.kind == TO_STEP_LINEAR_MAP) {
      for (int i = 0; i < s.n_out; ++i) {
        double v = 0.0;
        for (int j = 0; j < s.n; ++j) v += s.params[i + s.n_out * j] * x[j];
        for (int j = 0; j < s.m; ++j) v += s.params[s.n_out * s.n + i + s.n_out * j] * u[j];
        xn[i] = v;
      }
    } else discrete_dynamics(step_model(s), integrator, x, u, h, xn);
    return;
  }
  if (M.id != TO_MODEL_HYBRID_DOUBLE_INTEGRATOR) { discrete_dynamics(M, integrator, x, u, h, xn); return; }
  const int S = (int)M.p[1];
  if (k < S) { discrete_dynamics(hybrid_phase(M, 2), integrator, x, u, h, xn); return; }
  xn[2] = 0.0; xn[3] = 0.0;
  if (k == S) { xn[0] = (x[2] + x[3]) * 0.5; xn[1] = (u[0] + u[1]) * 0.5; return; }
  discrete_dynamics(hybrid_phase(M, 1), integrator, x, u, h, xn); /* reads x[0..1], u[0]; writes xn[0..1] */
}
inline void knot_step_jacobian(const Model& M, int integrator, int k, const double* x, const double* u, double h, double* A, double* Bm) {
  if (M.id == TO_MODEL_INFEASIBLE) {
    const Model S = infeasible_base(M);
    const int n = M.n, m = M.m, m0 = S.m;
    double b0[MAXN * MAXM];
    discrete_jacobian(S, integrator, x, u, h, A, b0);
    for (int i = 0; i < n; ++i) {
      for (int j = 0; j < m0; ++j) Bm[i * m + j] = b0[i * m0 + j];
      for (int j = 0; j < n; ++j) Bm[i * m + m0 + j] = (i == j) ? 1.0 : 0.0;
    }
    return;
  }
  if (M.id == TO_MODEL_VECTOR) { /* Jacobian of the live block, embedded in the padded (n x n), (n x m) */
    const to_step_model& s = M.steps[k];
    const int n = M.n, m = M.m;
    std::memset(A, 0, sizeof(double) * n * n); std::memset(Bm, 0, sizeof(double) * n * m);
    if (s.kind == TO_STEP_LINEAR_MAP) {
      for (int i = 0; i < s.n_out; ++i) {
        for (int j = 0; j < s.n; ++j) A[i * n + j] = s.params[i + s.n_out * j];
        for (int j = 0; j < s.m; ++j) Bm[i * m + j] = s.params[s.n_out * s.n + i + s.n_out * j];
      }
    } else {
      double a[MAXN * MAXN], b[MAXN * MAXM];
      discrete_jacobian(step_model(s), integrator, x, u, h, a, b);
      for (int i = 0; i < s.n; ++i) { for (int j = 0; j < s.n; ++j) A[i * n + j] = a[i * s.n + j]; for (int j = 0; j < s.m; ++j) Bm[i * m + j] = b[i * s.m + j]; }
    }
    return;
  }
  if (M.id != TO_MODEL_HYBRID_DOUBLE_INTEGRATOR) { discrete_jacobian(M, integrator, x, u, h, A, Bm); return; }
  const int S = (int)M.p[1];
  if (k < S) { discrete_jacobian(hybrid_phase(M, 2), integrator, x, u, h, A, Bm); return; }
  std::memset(A, 0, sizeof(double) * 16); std::memset(Bm, 0, sizeof(double) * 8);
  if (k == S) { A[0 * 4 + 2] = 0.5; A[0 * 4 + 3] = 0.5; Bm[1 * 2 + 0] = 0.5; Bm[1 * 2 + 1] = 0.5; return; }
  double a[4], b[2];
  discrete_jacobian(hybrid_phase(M, 1), integrator, x, u, h, a, b);
  A[0] = a[0]; A[1] = a[1]; A[4] = a[2]; A[5] = a[3];
  Bm[0] = b[0]; Bm[2] = b[1];
}

/* ---------------------------------------------------------------- error state */
/* G(x): n x ne row-major.  Identity for vector-space models. */
inline void errstate_jacobian(const Model& M, const double* x, double* G) {
  const int n = M.n, ne = M.ne;
  std::memset(G, 0, sizeof(double) * n * ne);
  if (M.id != TO_MODEL_QUADROTOR) { for (int i = 0; i < n; ++i) G[i * ne + i] = 1.0; return; }
  if (M.rot() != TO_ROT_QUATERNION) { /* blkdiag(I3, D(p), I3, I3) */
    for (int i = 0; i < n; ++i) G[i * ne + i] = 1.0;
    double D[9]; att_differential(M.rot(), x + 3, D);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) G[(3 + i) * ne + 3 + j] = D[3 * i + j];
    return;
  }
  for (int i = 0; i < 3; ++i) G[i * ne + i] = 1.0;
  double w = x[3], a = x[4], b = x[5], c = x[6];
  /* L(q) H : 4x3, no 1/2 factor (Cayley map) */
  double LH[12] = {-a, -b, -c,  w, -c, b,  c, w, -a,  -b, a, w};
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 3; ++j) G[(3 + i) * ne + 3 + j] = LH[i * 3 + j];
  for (int i = 0; i < 6; ++i) G[(7 + i) * ne + 6 + i] = 1.0;
}

/* E(x) = d(y (-) x)/dy at y = x: ne x n row-major, the left inverse of G(x).  Unit quaternions: G' (its columns are
 * orthonormal), which is what the reference stack multiplies by (Altro error_expansion!: A_err = G(x+)' A G(x)).  For a
 * three-parameter attitude G' is NOT an inverse of G (G'G != I), so the consistent Jacobian of state_diff is used instead —
 * D(p)^-1 in closed form (MRP: 4 D'/(1+|p|^2)^2, RodriguesParam: (I - [g]x)/(1+|g|^2)) — so that the expansion linearises
 * exactly the map the forward pass applies (checked against finite differences of x+(x (+) d) (-) x+(x) in the tests). */
inline void errstate_left_inverse(const Model& M, const double* x, double* E) {
  const int n = M.n, ne = M.ne;
  static thread_local std::vector<double> G;
  G.resize((size_t)n * ne);
  errstate_jacobian(M, x, G.data());
  for (int i = 0; i < ne; ++i) for (int r = 0; r < n; ++r) E[i * n + r] = G[r * ne + i];
  if (M.id == TO_MODEL_QUADROTOR && M.rot() != TO_ROT_QUATERNION) {
    const double* p = x + 3; const double n2 = p[0] * p[0] + p[1] * p[1] + p[2] * p[2];
    double D[9]; att_differential(M.rot(), p, D);
    const double Px[9] = {0, -p[2], p[1], p[2], 0, -p[0], -p[1], p[0], 0};
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j)
      E[(3 + i) * n + 3 + j] = M.rot() == TO_ROT_MRP ? 4.0 * D[3 * j + i] / ((1.0 + n2) * (1.0 + n2)) : ((i == j ? 1.0 : 0.0) - Px[3 * i + j]) / (1.0 + n2);
  }
}

/* dx (ne) = x (-) x0  (RD.state_diff with the Cayley map) */
inline void state_diff(const Model& M, const double* x, const double* x0, double* dx) {
  if (M.id != TO_MODEL_QUADROTOR) { for (int i = 0; i < M.n; ++i) dx[i] = x[i] - x0[i]; return; }
  for (int i = 0; i < 3; ++i) dx[i] = x[i] - x0[i];
  if (M.rot() != TO_ROT_QUATERNION) { /* Rodrigues vector of the relative rotation from the unnormalised quaternions */
    double q0[4], q[4]; att_quat(M.rot(), x0 + 3, q0); att_quat(M.rot(), x + 3, q);
    double s = q0[0] * q[0] + q0[1] * q[1] + q0[2] * q[2] + q0[3] * q[3];
    dx[3] = (q0[0] * q[1] - q0[1] * q[0] - (q0[2] * q[3] - q0[3] * q[2])) / s;
    dx[4] = (q0[0] * q[2] - q0[2] * q[0] - (q0[3] * q[1] - q0[1] * q[3])) / s;
    dx[5] = (q0[0] * q[3] - q0[3] * q[0] - (q0[1] * q[2] - q0[2] * q[1])) / s;
    for (int i = 0; i < 6; ++i) dx[6 + i] = x[6 + i] - x0[6 + i];
    return;
  }
  double w0 = x0[3], a0 = x0[4], b0 = x0[5], c0 = x0[6];
  double w = x[3], a = x[4], b = x[5], c = x[6];
  /* dq = conj(q0) (x) q */
  double s = w0 * w + a0 * a + b0 * b + c0 * c;
  double v1 = w0 * a - a0 * w - (b0 * c - c0 * b);
  double v2 = w0 * b - b0 * w - (c0 * a - a0 * c);
  double v3 = w0 * c - c0 * w - (a0 * b - b0 * a);
  dx[3] = v1 / s; dx[4] = v2 / s; dx[5] = v3 / s;
  for (int i = 0; i < 6; ++i) dx[6 + i] = x[7 + i] - x0[7 + i];
}

/* xo = x (+) dx: the inverse of state_diff (RD's state_diff with the Cayley map: state_diff(x (+) dx, x) = dx).  The projected-Newton
 * polish (Altro ProjectedNewtonSolver; out of tree) moves states along error-state steps.  Attitude: q (x) [1, phi] / sqrt(1 + |phi|^2)
 * keeps |q|; three-parameter attitudes compose through their unnormalised quaternion and map back (MRP p = v / (|q| + w),
 * RodriguesParam g = v / w). */
inline void state_add(const Model& M, const double* x, const double* dx, double* xo) {
  if (M.id != TO_MODEL_QUADROTOR) { for (int i = 0; i < M.n; ++i) xo[i] = x[i] + dx[i]; return; }
  for (int i = 0; i < 3; ++i) xo[i] = x[i] + dx[i];
  const int rot = M.rot();
  const int o = rot == TO_ROT_QUATERNION ? 7 : 6;
  for (int i = 0; i < 6; ++i) xo[o + i] = x[o + i] + dx[6 + i];
  double q[4];
  if (rot == TO_ROT_QUATERNION) { for (int i = 0; i < 4; ++i) q[i] = x[3 + i]; } else att_quat(rot, x + 3, q);
  const double f1 = dx[3], f2 = dx[4], f3 = dx[5];
  /* q (x) [1, phi] */
  double r[4] = {q[0] - q[1] * f1 - q[2] * f2 - q[3] * f3,
                 q[1] + q[0] * f1 + q[2] * f3 - q[3] * f2,
                 q[2] + q[0] * f2 + q[3] * f1 - q[1] * f3,
                 q[3] + q[0] * f3 + q[1] * f2 - q[2] * f1};
  if (rot == TO_ROT_QUATERNION) {
    const double s = 1.0 / std::sqrt(1.0 + (f1 * f1 + f2 * f2 + f3 * f3));
    for (int i = 0; i < 4; ++i) xo[3 + i] = r[i] * s;
  } else if (rot == TO_ROT_MRP) {
    const double nr = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2] + r[3] * r[3]);
    const double s = 1.0 / (nr + r[0]);
    for (int i = 0; i < 3; ++i) xo[3 + i] = r[1 + i] * s;
  } else {
    const double s = 1.0 / r[0];
    for (int i = 0; i < 3; ++i) xo[3 + i] = r[1 + i] * s;
  }
}

/* ---------------------------------------------------------------- cones (src/cones.jl) */
/* returns SOC branch: 0 below, 1 in, 2 outside, -1 invalid (NaN) */
inline int soc_status(const double* x, int dim) {
  double s = x[dim - 1], a2 = 0.0;
  for (int i = 0; i < dim - 1; ++i) a2 += x[i] * x[i];
  double a = std::sqrt(a2);
  if (a <= -s) return 0;
  if (a <= s) return 1;
  if (a >= std::fabs(s)) return 2;
  return -1;
}

inline int cone_projection(int cone, const double* x, double* px, int dim) { /* src/cones.jl:96-127 */
  switch (cone) {
    case TO_CONE_IDENTITY: for (int i = 0; i < dim; ++i) px[i] = x[i]; return 0;
    case TO_CONE_ZERO: for (int i = 0; i < dim; ++i) px[i] = 0.0; return 0;
    case TO_CONE_NEGATIVE_ORTHANT: for (int i = 0; i < dim; ++i) px[i] = std::fmin(0.0, x[i]); return 0;
    case TO_CONE_POSITIVE_ORTHANT: for (int i = 0; i < dim; ++i) px[i] = std::fmax(0.0, x[i]); return 0;
    case TO_CONE_SECOND_ORDER: {
      double s = x[dim - 1], a2 = 0.0;
      for (int i = 0; i < dim - 1; ++i) a2 += x[i] * x[i];
      double a = std::sqrt(a2);
      if (a <= -s) { for (int i = 0; i < dim; ++i) px[i] = 0.0; return 0; }
      if (a <= s) { for (int i = 0; i < dim; ++i) px[i] = x[i]; return 1; }
      if (a >= std::fabs(s)) {
        double c = 0.5 * (1 + s / a);
        for (int i = 0; i < dim - 1; ++i) px[i] = x[i] * c;
        px[dim - 1] = a * c;
        return 2;
      }
      return -1;
    }
  }
  return -1;
}

/* J (dim x dim, row-major), fully written. src/cones.jl:129-188.  NOTE the reference's NegativeOrthant
 * method only writes the diagonal; the oracle writes the full (zero off-diagonal) matrix. */
inline int cone_projection_jacobian(int cone, const double* x, double* J, int dim) {
  std::memset(J, 0, sizeof(double) * dim * dim);
  switch (cone) {
    case TO_CONE_IDENTITY: for (int i = 0; i < dim; ++i) J[i * dim + i] = 1.0; return 0;
    case TO_CONE_ZERO: return 0;
    case TO_CONE_NEGATIVE_ORTHANT: for (int i = 0; i < dim; ++i) J[i * dim + i] = x[i] <= 0 ? 1.0 : 0.0; return 0;
    case TO_CONE_POSITIVE_ORTHANT: for (int i = 0; i < dim; ++i) J[i * dim + i] = x[i] >= 0 ? 1.0 : 0.0; return 0;
    case TO_CONE_SECOND_ORDER: {
      int n = dim;
      double s = x[n - 1], a2 = 0.0;
      for (int i = 0; i < n - 1; ++i) a2 += x[i] * x[i];
      double a = std::sqrt(a2);
      if (a <= -s) return 0;
      if (a <= s) { for (int i = 0; i < n; ++i) J[i * n + i] = 1.0; return 1; }
      if (a >= std::fabs(s)) {
        double c = 0.5 * (1 + s / a);
        for (int i = 0; i < n - 1; ++i)
          for (int j = 0; j < n - 1; ++j) {
            J[i * n + j] = -0.5 * s / (a * a * a) * x[i] * x[j];
            if (i == j) J[i * n + j] += c;
          }
        for (int i = 0; i < n - 1; ++i) J[i * n + (n - 1)] = 0.5 * x[i] / a;
        for (int i = 0; i < n - 1; ++i) J[(n - 1) * n + i] = ((-0.5 * s / (a * a)) + c / a) * x[i];
        J[(n - 1) * n + (n - 1)] = 0.5;
        return 2;
      }
      return -1;
    }
  }
  return -1;
}

/* hess (dim x dim) = Hessian of b' Pi(x). src/cones.jl:201-276 */
inline int cone_projection_hessian(int cone, const double* x, const double* b, double* H, int dim) {
  std::memset(H, 0, sizeof(double) * dim * dim);
  if (cone != TO_CONE_SECOND_ORDER) return 0;
  int n = dim - 1;
  double s = x[n], bs = b[n], a2 = 0.0, vbv = 0.0;
  for (int i = 0; i < n; ++i) { a2 += x[i] * x[i]; vbv += x[i] * b[i]; }
  double a = std::sqrt(a2);
  if (a <= -s) return 0;
  if (a <= s) return 1;
  if (a > std::fabs(s)) {
    for (int i = 0; i < n; ++i) {
      double hi = 0.0;
      for (int j = 0; j < n; ++j) {
        double Hij = -x[i] * x[j] / (a * a);
        if (i == j) Hij += 1;
        hi += Hij * b[j];
      }
      H[i * dim + n] = hi / (2 * a);
      H[n * dim + i] = H[i * dim + n];
      for (int j = 0; j <= i; ++j) {
        double vij = x[i] * x[j];
        double H1 = hi * x[j] * (-s / (a * a * a));
        double H2 = vij * (2 * vbv) / (a * a * a * a) - x[i] * b[j] / (a * a);
        double H3 = -vij / (a * a);
        if (i == j) { H2 -= vbv / (a * a); H3 += 1; }
        H2 *= s / a;
        H3 *= bs / a;
        H[i * dim + j] = (H1 + H2 + H3) / 2;
        H[j * dim + i] = H[i * dim + j];
      }
    }
    H[n * dim + n] = 0.0;
    return 2;
  }
  return -1;
}

/* ---------------------------------------------------------------- ErrorQuadratic (src/lie_costs.jl:178-241) */
/* dx = x (-) x_ref on a rigid-body state: dphi = vec(dq)/scalar(dq), dq = conj(q_ref) (x) q = [q0'q; V q], with q, q0 the
 * (unnormalised) quaternions of the two attitudes: the state's own 4 entries (ErrorQuadratic{QuatRotation}, n = 13), or
 * att_quat of its 3 attitude parameters (ErrorQuadratic{MRP} / {RodriguesParam}, n = 12; rot = to_rotation, carried in C.w). */
struct ErrQuadGeom {
  double s, phi[3], V[3][4], q0[4], q[4];
  int rot, ov; /* attitude representation; index of the first velocity entry of the state */
};
inline void errquad_geom(int rot, const double* x, const double* xr, ErrQuadGeom& G) {
  G.rot = rot; G.ov = rot == TO_ROT_QUATERNION ? 7 : 6;
  if (rot == TO_ROT_QUATERNION) { for (int t = 0; t < 4; ++t) { G.q0[t] = xr[3 + t]; G.q[t] = x[3 + t]; } }
  else { att_quat(rot, xr + 3, G.q0); att_quat(rot, x + 3, G.q); }
  const double w0 = G.q0[0], a0 = G.q0[1], b0 = G.q0[2], c0 = G.q0[3];
  const double V[3][4] = {{-a0, w0, c0, -b0}, {-b0, -c0, w0, a0}, {-c0, b0, -a0, w0}};
  G.s = 0.0;
  for (int t = 0; t < 4; ++t) G.s += G.q0[t] * G.q[t];
  for (int i = 0; i < 3; ++i) {
    double v = 0.0;
    for (int t = 0; t < 4; ++t) { G.V[i][t] = V[i][t]; v += V[i][t] * G.q[t]; }
    G.phi[i] = v / G.s;
  }
}
inline void errquad_dx(int rot, const double* x, const double* xr, double* dx) {
  ErrQuadGeom G; errquad_geom(rot, x, xr, G);
  for (int i = 0; i < 3; ++i) { dx[i] = x[i] - xr[i]; dx[3 + i] = G.phi[i]; }
  for (int i = 0; i < 6; ++i) dx[6 + i] = x[G.ov + i] - xr[G.ov + i];
}

/* ---------------------------------------------------------------- costs */
/* J = 0.5 x'Qx + q'x + c (+ 0.5 u'Ru + r'u) (+ u'Hx) (+ w min(1+dq,1-dq)); src/cost_functions.jl:89-104, src/lie_costs.jl:68-76.
 * `with_u`: the reference adds the control terms whenever u is non-empty. */
inline double cost_evaluate(const to_cost_desc& C, int n, int m, const double* x, const double* u) {
  double J = 0.0;
  if (C.kind == TO_COST_ERROR_QUADRATIC) { /* src/lie_costs.jl:237-240 */
    double dx[12], e = 0.0;
    errquad_dx((int)C.w, x, C.q, dx);
    for (int i = 0; i < 12; ++i) e += dx[i] * C.Q[i] * dx[i];
    J = 0.5 * e + C.c;
    if (u) {
      double uRu = 0.0, ru = 0.0;
      for (int i = 0; i < m; ++i) { uRu += u[i] * C.R[i] * u[i]; ru += C.r[i] * u[i]; }
      J += 0.5 * uRu + ru;
    }
    return J;
  }
  if (C.kind == TO_COST_QUADRATIC) {
    double xQx = 0.0;
    for (int j = 0; j < n; ++j) { double t = 0.0; for (int i = 0; i < n; ++i) t += x[i] * C.Q[i + n * j]; xQx += t * x[j]; }
    J = 0.5 * xQx;
  } else {
    double xQx = 0.0;
    for (int i = 0; i < n; ++i) xQx += x[i] * C.Q[i] * x[i];
    J = 0.5 * xQx;
  }
  double qx = 0.0; for (int i = 0; i < n; ++i) qx += C.q[i] * x[i];
  J = J + qx + C.c;
  if (u) {
    double uRu = 0.0, ru = 0.0;
    if (C.kind == TO_COST_QUADRATIC) {
      for (int j = 0; j < m; ++j) { double t = 0.0; for (int i = 0; i < m; ++i) t += u[i] * C.R[i + m * j]; uRu += t * u[j]; }
    } else {
      for (int i = 0; i < m; ++i) uRu += u[i] * C.R[i] * u[i];
    }
    for (int i = 0; i < m; ++i) ru += C.r[i] * u[i];
    J += 0.5 * uRu + ru;
    if (C.kind == TO_COST_QUADRATIC) {
      double uHx = 0.0;
      for (int j = 0; j < n; ++j) for (int i = 0; i < m; ++i) uHx += u[i] * C.H[i + m * j] * x[j];
      J += uHx;
    }
  }
  if (C.kind == TO_COST_DIAGONAL_QUAT) {
    double dq = 0.0;
    for (int i = 0; i < 4; ++i) dq += C.q_ref[i] * x[C.q_ind[i] - 1];
    J += C.w * std::fmin(1 + dq, 1 - dq);
  }
  return J;
}

/* grad (n+m) and hess ((n+m)^2 row-major) wrt z=[x;u]; u-parts are zero when terminal.
 * src/cost_functions.jl:137-233; quaternion term src/lie_costs.jl:82-90 (the intended gradient). */
inline void cost_expansion(const to_cost_desc& C, int n, int m, const double* x, const double* u, bool terminal,
                           double* grad, double* hess) {
  const int nz = n + m;
  std::memset(grad, 0, sizeof(double) * nz);
  std::memset(hess, 0, sizeof(double) * nz * nz);
  if (C.kind == TO_COST_ERROR_QUADRATIC) {
    /* exact derivatives of 0.5 dx'Q dx (the reference takes them with ForwardDiff): with D_i = (V_i - phi_i q0')/s,
     * grad_q = sum_i Q_i phi_i D_i',  H_qq = sum_i Q_i [D_i'D_i - phi_i (V_i'q0' + q0 V_i)/s^2 + 2 phi_i^2 q0 q0'/s^2] */
    ErrQuadGeom G; errquad_geom((int)C.w, x, C.q, G);
    for (int i = 0; i < 3; ++i) { grad[i] = C.Q[i] * (x[i] - C.q[i]); hess[i * nz + i] = C.Q[i]; }
    for (int i = 0; i < 6; ++i) { grad[G.ov + i] = C.Q[6 + i] * (x[G.ov + i] - C.q[G.ov + i]); hess[(G.ov + i) * nz + G.ov + i] = C.Q[6 + i]; }
    double gq[4] = {0, 0, 0, 0}, Hq[16] = {0}; /* derivatives with respect to the (unnormalised) quaternion */
    for (int i = 0; i < 3; ++i) {
      const double Qi = C.Q[3 + i], ph = G.phi[i];
      double D[4];
      for (int t = 0; t < 4; ++t) { D[t] = (G.V[i][t] - ph * G.q0[t]) / G.s; gq[t] += Qi * ph * D[t]; }
      for (int t = 0; t < 4; ++t) for (int r = 0; r < 4; ++r)
        Hq[t * 4 + r] += Qi * (D[t] * D[r] - ph * (G.V[i][t] * G.q0[r] + G.q0[t] * G.V[i][r]) / (G.s * G.s)
                               + 2 * ph * ph * G.q0[t] * G.q0[r] / (G.s * G.s));
    }
    if (G.rot == TO_ROT_QUATERNION) {
      for (int t = 0; t < 4; ++t) { grad[3 + t] = gq[t]; for (int r = 0; r < 4; ++r) hess[(3 + t) * nz + 3 + r] = Hq[t * 4 + r]; }
    } else { /* chain rule through q(p): dq/dp = [-2p'; 2I] and d2q_0/dp2 = -2I (MRP), dq/dg = [0; I] (RodriguesParam) */
      double Jq[4][3];
      for (int j = 0; j < 3; ++j) {
        Jq[0][j] = G.rot == TO_ROT_MRP ? -2.0 * x[3 + j] : 0.0;
        for (int t = 0; t < 3; ++t) Jq[1 + t][j] = (t == j) ? (G.rot == TO_ROT_MRP ? 2.0 : 1.0) : 0.0;
      }
      for (int j = 0; j < 3; ++j) { double g = 0.0; for (int t = 0; t < 4; ++t) g += Jq[t][j] * gq[t]; grad[3 + j] = g; }
      for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        double h = 0.0;
        for (int t = 0; t < 4; ++t) for (int r = 0; r < 4; ++r) h += Jq[t][i] * Hq[t * 4 + r] * Jq[r][j];
        if (G.rot == TO_ROT_MRP && i == j) h += -2.0 * gq[0];
        hess[(3 + i) * nz + 3 + j] = h;
      }
    }
    if (!terminal) for (int i = 0; i < m; ++i) { grad[n + i] = C.R[i] * u[i] + C.r[i]; hess[(n + i) * nz + n + i] = C.R[i]; }
    return;
  }
  if (C.kind == TO_COST_QUADRATIC) {
    for (int i = 0; i < n; ++i) { double t = C.q[i]; for (int j = 0; j < n; ++j) t += C.Q[i + n * j] * x[j]; grad[i] = t; }
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) hess[i * nz + j] = C.Q[i + n * j];
  } else {
    for (int i = 0; i < n; ++i) { grad[i] = C.Q[i] * x[i] + C.q[i]; hess[i * nz + i] = C.Q[i]; }
  }
  if (C.kind == TO_COST_DIAGONAL_QUAT) {
    double dq = 0.0;
    for (int i = 0; i < 4; ++i) dq += C.q_ref[i] * x[C.q_ind[i] - 1];
    for (int i = 0; i < 4; ++i) {
      if (dq < 0) grad[C.q_ind[i] - 1] += C.w * C.q_ref[i];
      else grad[C.q_ind[i] - 1] -= C.w * C.q_ref[i];
    }
  }
  if (!terminal) {
    if (C.kind == TO_COST_QUADRATIC) {
      for (int i = 0; i < m; ++i) { double t = C.r[i]; for (int j = 0; j < m; ++j) t += C.R[i + m * j] * u[j]; grad[n + i] = t; }
      for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j) hess[(n + i) * nz + n + j] = C.R[i + m * j];
      /* cross term u'Hx: grad_x += H'u, grad_u += Hx; hess[iu,ix] = H (and its transpose for the symmetric full Hessian) */
      for (int j = 0; j < n; ++j) for (int i = 0; i < m; ++i) {
        grad[j] += C.H[i + m * j] * u[i];
        grad[n + i] += C.H[i + m * j] * x[j];
        hess[(n + i) * nz + j] = C.H[i + m * j];
        hess[j * nz + n + i] = C.H[i + m * j];
      }
    } else {
      for (int i = 0; i < m; ++i) { grad[n + i] = C.R[i] * u[i] + C.r[i]; hess[(n + i) * nz + n + i] = C.R[i]; }
    }
  }
}

/* ---------------------------------------------------------------- constraints (src/constraints.jl) */
/* z = [x; u] (u = zeros at the terminal knot).  c (p), jac (p x (n+m) row-major, fully written). */
inline int constraint_output_dim(const to_constraint_desc& K, int n, int m) {
  switch (K.kind) {
    case TO_CON_GOAL: return K.n_inds;
    case TO_CON_BOUND: {
      int p = 0;
      for (int i = 0; i < n + m; ++i) if (std::isfinite(K.params[i])) ++p;
      for (int i = 0; i < n + m; ++i) if (std::isfinite(K.params[n + m + i])) ++p;
      return p;
    }
    case TO_CON_NORM: return K.sense == TO_CONE_SECOND_ORDER ? K.n_inds + 1 : 1;
    case TO_CON_CIRCLE: return K.n_params / 3;
    case TO_CON_SPHERE: return K.n_params / 4;
    case TO_CON_LINEAR: return K.n_params / (K.n_inds + 1);
    case TO_CON_COLLISION: return 1;
    case TO_CON_QUATVEC: return 3;
  }
  return -1;
}

/* H (w x w, column-major, w = n for state constraints else n+m) += sum_r lambda_r * Hessian of c_r — what the reference's
 * ∇jacobian! accumulates (src/abstract_constraint.jl:255-280; zero for Goal :70-73 and Bound :767-770, ForwardDiff default
 * elsewhere).  Closed forms: rows that are affine in z contribute nothing; ‖z_I‖² − a² gives 2 I on I; circle / sphere
 * −2 I on the centre coordinates; collision −2 [I −I; −I I]; QuatVecEq the second derivative of q/‖q‖. */
inline void constraint_hessian_add(const to_constraint_desc& K, int n, int m, const double* z, const double* lambda, double* H, int w) {
  const int p = constraint_output_dim(K, n, m);
  auto at = [&](int i, int j) -> double& { return H[i + (size_t)w * j]; };
  switch (K.kind) {
    case TO_CON_NORM:
      if (K.sense != TO_CONE_SECOND_ORDER) for (int t = 0; t < K.n_inds; ++t) at(K.inds[t] - 1, K.inds[t] - 1) += 2.0 * lambda[0];
      return;
    case TO_CON_CIRCLE:
      for (int i = 0; i < p; ++i) { at(K.inds[0] - 1, K.inds[0] - 1) -= 2.0 * lambda[i]; at(K.inds[1] - 1, K.inds[1] - 1) -= 2.0 * lambda[i]; }
      return;
    case TO_CON_SPHERE:
      for (int i = 0; i < p; ++i) for (int t = 0; t < 3; ++t) at(K.inds[t] - 1, K.inds[t] - 1) -= 2.0 * lambda[i];
      return;
    case TO_CON_COLLISION: {
      const int D = K.n_inds / 2;
      for (int t = 0; t < D; ++t) {
        const int a = K.inds[t] - 1, b = K.inds[D + t] - 1;
        at(a, a) -= 2.0 * lambda[0]; at(b, b) -= 2.0 * lambda[0]; at(a, b) += 2.0 * lambda[0]; at(b, a) += 2.0 * lambda[0];
      }
      return;
    }
    case TO_CON_QUATVEC: {
      double q[4], s2 = 0.0;
      for (int t = 0; t < 4; ++t) { q[t] = z[K.inds[t] - 1]; s2 += q[t] * q[t]; }
      const double s = std::sqrt(s2), s3 = s2 * s, s5 = s3 * s2;
      for (int r = 0; r < 3; ++r) {  /* c_r = q_{r+1}/|q| + const:  d2/dq_j dq_k = -(d_ij q_k + d_ik q_j + d_jk q_i)/s^3 + 3 q_i q_j q_k/s^5 */
        const int i = r + 1;
        for (int j = 0; j < 4; ++j) for (int k = 0; k < 4; ++k) {
          const double v = -((i == j ? q[k] : 0.0) + (i == k ? q[j] : 0.0) + (j == k ? q[i] : 0.0)) / s3 + 3.0 * q[i] * q[j] * q[k] / s5;
          at(K.inds[j] - 1, K.inds[k] - 1) += lambda[r] * v;
        }
      }
      return;
    }
    default: return;  /* GOAL, BOUND, LINEAR, NORM (SOC form): affine rows */
  }
}

inline void constraint_evaluate(const to_constraint_desc& K, int n, int m, const double* z, double* c, double* jac) {
  const int nz = n + m;
  const int p = constraint_output_dim(K, n, m);
  if (jac) std::memset(jac, 0, sizeof(double) * p * nz);
  switch (K.kind) {
    case TO_CON_GOAL: /* :55-68 */
      for (int i = 0; i < p; ++i) {
        int j = K.inds[i] - 1;
        c[i] = z[j] - K.params[i];
        if (jac) jac[i * nz + j] = 1.0;
      }
      return;
    case TO_CON_BOUND: { /* :738-765: [(z - z_max); (z_min - z)][finite] */
      int r = 0;
      for (int j = 0; j < nz; ++j) if (std::isfinite(K.params[j])) { c[r] = z[j] - K.params[j]; if (jac) jac[r * nz + j] = 1.0; ++r; }
      for (int j = 0; j < nz; ++j) if (std::isfinite(K.params[nz + j])) { c[r] = K.params[nz + j] - z[j]; if (jac) jac[r * nz + j] = -1.0; ++r; }
      return;
    }
    case TO_CON_NORM: { /* :462-517 */
      double val = K.params[0];
      if (K.sense == TO_CONE_SECOND_ORDER) {
        for (int i = 0; i < K.n_inds; ++i) { int j = K.inds[i] - 1; c[i] = z[j]; if (jac) jac[i * nz + j] = 1.0; }
        c[K.n_inds] = val;
      } else {
        double s = 0.0;
        for (int i = 0; i < K.n_inds; ++i) { int j = K.inds[i] - 1; s += z[j] * z[j]; if (jac) jac[j] = 2 * z[j]; }
        c[0] = s - val * val;
      }
      return;
    }
    case TO_CON_CIRCLE: { /* :199-228 */
      int P = p; int xi = K.inds[0] - 1, yi = K.inds[1] - 1;
      for (int i = 0; i < P; ++i) {
        double xc = K.params[i], yc = K.params[P + i], r = K.params[2 * P + i];
        double dx = z[xi] - xc, dy = z[yi] - yc;
        c[i] = -(dx * dx) - dy * dy + r * r;
        if (jac) { jac[i * nz + xi] = -2 * dx; jac[i * nz + yi] = -2 * dy; }
      }
      return;
    }
    case TO_CON_SPHERE: { /* :283-321 */
      int P = p; int xi = K.inds[0] - 1, yi = K.inds[1] - 1, zi = K.inds[2] - 1;
      for (int i = 0; i < P; ++i) {
        double xc = K.params[i], yc = K.params[P + i], zc = K.params[2 * P + i], r = K.params[3 * P + i];
        double dx = z[xi] - xc, dy = z[yi] - yc, dz = z[zi] - zc;
        c[i] = -(dx * dx) - dy * dy - dz * dz + r * r;
        if (jac) { jac[i * nz + xi] = -2 * dx; jac[i * nz + yi] = -2 * dy; jac[i * nz + zi] = -2 * dz; }
      }
      return;
    }
    case TO_CON_LINEAR: { /* :134-144: A z[inds] - b */
      int D = K.n_inds;
      for (int i = 0; i < p; ++i) {
        double s = 0.0;
        for (int t = 0; t < D; ++t) { int j = K.inds[t] - 1; s += K.params[i + p * t] * z[j]; if (jac) jac[i * nz + j] = K.params[i + p * t]; }
        c[i] = s - K.params[p * D + i];
      }
      return;
    }
    case TO_CON_QUATVEC: { /* :947-955: q = normalize(x[qind]); qf flipped onto q's hemisphere; c = vec(q) - vec(qf).
                              Jacobian (ForwardDiff in the reference) = rows 2..4 of (I - q q')/|x[qind]| */
      double q[4], nrm = 0.0, dq = 0.0;
      for (int t = 0; t < 4; ++t) { q[t] = z[K.inds[t] - 1]; nrm += q[t] * q[t]; }
      nrm = std::sqrt(nrm);
      for (int t = 0; t < 4; ++t) { q[t] /= nrm; dq += K.params[t] * q[t]; }
      const double sg = dq < 0 ? -1.0 : 1.0;
      for (int r = 0; r < 3; ++r) {
        c[r] = -(sg * K.params[r + 1] - q[r + 1]);
        if (jac) for (int t = 0; t < 4; ++t) jac[r * nz + K.inds[t] - 1] = ((t == r + 1 ? 1.0 : 0.0) - q[r + 1] * q[t]) / nrm;
      }
      return;
    }
    case TO_CON_COLLISION: { /* :362-387: r^2 - d'd with d = x[x1] - x[x2]; Jacobian entries are ASSIGNED (x1 first) */
      const int D = K.n_inds / 2;
      c[0] = K.params[0] * K.params[0];
      for (int i = 0; i < D; ++i) {
        const int j1 = K.inds[i] - 1, j2 = K.inds[D + i] - 1;
        const double d = z[j1] - z[j2];
        c[0] -= d * d;
        if (jac) { jac[j1] = -2 * d; jac[j2] = 2 * d; }
      }
      return;
    }
  }
}

inline bool constraint_is_state_only(int kind) {
  return kind == TO_CON_GOAL || kind == TO_CON_CIRCLE || kind == TO_CON_SPHERE || kind == TO_CON_COLLISION || kind == TO_CON_QUATVEC;
}

}  // namespace oracle
#endif

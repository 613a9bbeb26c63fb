// problem_dev.h — device-side problem description and the per-knot cost / constraint / cone
// arithmetic.  All descriptor data is wave-uniform (scalar loads); the
// per-trajectory data (x, u, duals) lives in registers, one trajectory per lane.
//
// Restated from src/cost_functions.jl:89-233, src/lie_costs.jl:68-95, src/constraints.jl (Goal :55-68,
// Bound :738-765, Norm :462-517, Circle :199-228, Sphere :283-321, Linear :134-144), src/cones.jl:96-276.
#pragma once
#include <hip/hip_runtime.h>

#include <utility>

#include "../../include/trajopt_hip.h"

namespace to {

struct DevCon {
  to_constraint_desc d;
  int p, width, k1, k2;     // k1,k2 0-based inclusive
  long long dual_off;       // row offset of this constraint's duals in the per-trajectory dual array
  int selector;             // 1: every row is s*(z[idx]-off) or a constant (GOAL, BOUND, NORM-SOC)
  int fast;                 // selector rows r map to consecutive entries with sign +1: 1 = z[r] (state prefix), 2 = z[n+r] (controls)
  int sidx[TO_MAX_P];       // selector rows: 0-based index into z, or -1 for a constant row (value = soff)
  double ssgn[TO_MAX_P];
  double soff[TO_MAX_P];
  int cp_off;               // >= 0: this constraint sees z = [x; u] shifted by the per-trajectory block DevProblem::cp[cp_off .. cp_off + n + m)
                            // (to_set_constraint_params_batch: one GoalConstraint target per trajectory); -1: shared parameters only
};

// Read-only descriptor tables are addressed through the CONSTANT address space: with a wave-uniform address the
// compiler then emits scalar loads (s_load, scalar cache) instead of per-lane vector loads that would sit on the
// critical path of every knot (they may otherwise alias the kernels' double stores).
#ifndef TO_CONST_AS  // (the host build of the projected-Newton kernel — tests/host_shim — defines it empty)
#define TO_CONST_AS __attribute__((address_space(4)))
#endif
typedef const to_cost_desc TO_CONST_AS CostC;
typedef const DevCon TO_CONST_AS ConC;
typedef const double TO_CONST_AS DoubleC;
typedef const int TO_CONST_AS IntC;

struct DevProblem {
  int n, m, ne, N, B, Bp, integrator, n_costs, n_cons;
  int expand_variant;  // bit0: a QuadraticCost / ErrorQuadratic exists; bit1: constraints exist; bit2: a non-selector constraint exists
  int unit_soc;      // 1: constraints exist and every control-block selector constraint (fast == 2) is a unit SOC (unit_soc_desc)
  int simple_stage;  // 1: every stage knot (k < N-1) uses the same diagonal-kind cost and the same dt (the common LQR-style objective)
  long long n_duals;
  double mp[16];
  to_solver_opts opts;
  DoubleC* dt;         // [N-1]
  IntC* cost_index;    // [N]
  CostC* costs;        // [n_costs]
  ConC* cons;          // [n_cons]
  // Per-trajectory linear cost terms (to_set_cost_linear_batch: set_LQR_goal! / set_goal_state! with one goal per trajectory,
  // src/cost_functions.jl:249-258, src/problem.jl:294-310): tiled array, L = n_costs * (n + m); entry ci*(n+m) + i of trajectory b is
  // what its q_i (i < n) / r_{i-n} differs by from cost ci's descriptor.  NULL (the default): every trajectory shares the descriptors.
  const double* gl;
  // Per-trajectory constraint parameters (to_set_constraint_params_batch: set_goal_state!(prob, Xf; constraint = true) with one goal per
  // trajectory, src/problem.jl:303-309): tiled array, L = n_cp = (n + m) * n_cons; block ci holds what trajectory b's GoalConstraint target
  // differs by from the descriptor's, scattered onto the state indices — the constraint with target xf + d IS the shared constraint
  // evaluated at x - d (same value, same Jacobian), so every evaluation site shifts the z it hands to a flagged constraint
  // (con_shift).  A LinearConstraint A z = b + db is the shared one seen from z - A^+ db (the host forms the minimum-norm A^+ db).  NULL (the default): every trajectory shares the descriptors.  Read by the GENERAL kernel variants only.
  const double* cp;
  int n_cp;
};

// z = [x; u] as constraint K sees it for this lane's trajectory (cp0: the lane's pointer to entry 0 of DevProblem::cp; blocks of nz = n + m)
template <int nz>
__device__ __forceinline__ void con_shift(const DevProblem& P, ConC& K, const double* cp0, double* z) {
  if (P.cp != nullptr && K.cp_off >= 0) {  // wave-uniform
#pragma unroll
    for (int i = 0; i < nz; ++i) z[i] -= cp0[(size_t)(K.cp_off + i) * 64];
  }
}

// cost of the per-trajectory linear terms of cost ci at (x, u), and their gradient (added to g); gl0 = this lane's pointer to entry 0
template <int n, int m>
__device__ __forceinline__ double goal_lin_cost(const double* gl0, int ci, const double* x, const double* u) {
  const double* g = gl0 + (size_t)ci * (n + m) * 64;
  double J = 0.0;
#pragma unroll
  for (int i = 0; i < n; ++i) J += g[(size_t)i * 64] * x[i];
#pragma unroll
  for (int j = 0; j < m; ++j) J += g[(size_t)(n + j) * 64] * u[j];
  return J;
}
template <int n, int m>
__device__ __forceinline__ void goal_lin_grad(const double* gl0, int ci, bool terminal, double* gr) {
  const double* g = gl0 + (size_t)ci * (n + m) * 64;
#pragma unroll
  for (int i = 0; i < n; ++i) gr[i] += g[(size_t)i * 64];
  if (!terminal) {
#pragma unroll
    for (int j = 0; j < m; ++j) gr[n + j] += g[(size_t)(n + j) * 64];
  }
}

// Pin a wave-uniform value into a VGPR.  Loop-invariant uniform operands otherwise live in SGPRs; the hot loops
// carry ~40 of them (model parameters, stage cost), which overflows the 102-SGPR file into spills and, with the
// one-SGPR-per-VALU-instruction constant-bus limit, into extra moves.
__device__ __forceinline__ double in_vgpr(double x) { asm volatile("" : "+v"(x)); return x; }

// register-array helpers with a wave-uniform runtime index (no scratch: unrolled selects)
template <int N_>
__device__ __forceinline__ double pick(const double* a, int idx) {
  double r = 0.0;
#pragma unroll
  for (int i = 0; i < N_; ++i) r = (i == idx) ? a[i] : r;
  return r;
}
template <int N_>
__device__ __forceinline__ void add_at(double* a, int idx, double v) {
#pragma unroll
  for (int i = 0; i < N_; ++i) a[i] += (i == idx) ? v : 0.0;
}

// q_ref' x[q_ind]: the quaternion sits at the default state indices 4:7 (src/lie_costs.jl:134) in every rigid-body
// model; that case indexes registers directly, anything else falls back to the select chain.
template <int n>
__device__ __forceinline__ double quat_dot(const double* qref, const int* qind, const double* x) {
  if constexpr (n >= 7) {
    if (qind[0] == 4 && qind[1] == 5 && qind[2] == 6 && qind[3] == 7)
      return qref[0] * x[3] + qref[1] * x[4] + qref[2] * x[5] + qref[3] * x[6];
  }
  double dq = 0.0;
#pragma unroll
  for (int i = 0; i < 4; ++i) dq += qref[i] * pick<n>(x, qind[i] - 1);
  return dq;
}

// ------------------------------------------------------------------------------------------------ ErrorQuadratic
// ½ dx'Q dx with dx = x ⊖ x_ref on a rigid body (src/lie_costs.jl:178-241).  Templated on the scalar: the value path runs it
// in double, the expansion evaluates the GRADIENT in dual numbers to get exact Hessian-vector products (the reference
// differentiates the value twice with ForwardDiff).  xr = x_ref, Qe = 12 error-state weights.  The attitude of the state is a
// unit quaternion (n = 13; ErrorQuadratic{QuatRotation}) or three parameters (n = 12): C.w names them — 0 quaternion, 1 MRP,
// 2 RodriguesParam (ErrorQuadratic{MRP} / {RodriguesParam}) — and the error is the Rodrigues vector of the relative rotation
// in every case (RD.state_diff with CayleyMap, hard-coded at src/lie_costs.jl:238), computed from the (unnormalised)
// quaternion q̃ of the attitude: q itself, [1 − |p|², 2p], or [1, g].
template <int n> struct ErrQuadLay { static constexpr int na = (n == 13) ? 4 : 3, ov = 3 + na; };  // attitude entries, first velocity entry
template <int n, class T>
__device__ __forceinline__ void errquad_quat(int rot, const T* xa, T* q /*4*/) {
  if constexpr (n == 13) { q[0] = xa[0]; q[1] = xa[1]; q[2] = xa[2]; q[3] = xa[3]; }
  else if (rot == 1) { q[0] = 1.0 - (xa[0] * xa[0] + xa[1] * xa[1] + xa[2] * xa[2]); q[1] = 2.0 * xa[0]; q[2] = 2.0 * xa[1]; q[3] = 2.0 * xa[2]; }
  else { q[0] = T(1.0); q[1] = xa[0]; q[2] = xa[1]; q[3] = xa[2]; }
}
template <int n, class T>
__device__ __forceinline__ void errquad_phi(CostC& C, const T* x, T* phi /*3*/, T* rs_out, T* q /*4*/, double* q0 /*4*/) {
  const int rot = (int)C.w;
  double xr[4];
#pragma unroll
  for (int i = 0; i < ErrQuadLay<n>::na; ++i) xr[i] = C.q[3 + i];
  errquad_quat<n, double>(rot, xr, q0);
  errquad_quat<n, T>(rot, x + 3, q);
  const double w0 = q0[0], a0 = q0[1], b0 = q0[2], c0 = q0[3];
  const T w = q[0], a = q[1], b = q[2], c = q[3];
  const T s = w0 * w + a0 * a + b0 * b + c0 * c;
  const T rs = recip_t(s);
  phi[0] = (w0 * a - a0 * w + c0 * b - b0 * c) * rs;
  phi[1] = (w0 * b - b0 * w - c0 * a + a0 * c) * rs;
  phi[2] = (w0 * c - c0 * w + b0 * a - a0 * b) * rs;
  *rs_out = rs;
}
template <int n, class T>
__device__ __forceinline__ T errquad_value(CostC& C, const T* x) {
  constexpr int ov = ErrQuadLay<n>::ov;
  T phi[3], rs, q[4];
  double q0[4];
  errquad_phi<n, T>(C, x, phi, &rs, q, q0);
  T e = T(0.0);
#pragma unroll
  for (int i = 0; i < 3; ++i) { const T d = x[i] - C.q[i]; e = e + d * C.Q[i] * d; }
#pragma unroll
  for (int i = 0; i < 3; ++i) e = e + phi[i] * C.Q[3 + i] * phi[i];
#pragma unroll
  for (int i = 0; i < 6; ++i) { const T d = x[ov + i] - C.q[ov + i]; e = e + d * C.Q[6 + i] * d; }
  return 0.5 * e;
}
template <int n, class T>
__device__ __forceinline__ void errquad_grad(CostC& C, const T* x, T* g /*n*/) {
  constexpr int ov = ErrQuadLay<n>::ov;
  T phi[3], rs, q[4];
  double q0[4];
  errquad_phi<n, T>(C, x, phi, &rs, q, q0);
  const double V[3][4] = {{-q0[1], q0[0], q0[3], -q0[2]}, {-q0[2], -q0[3], q0[0], q0[1]}, {-q0[3], q0[2], -q0[1], q0[0]}};
#pragma unroll
  for (int i = 0; i < 3; ++i) g[i] = C.Q[i] * (x[i] - C.q[i]);
#pragma unroll
  for (int i = 0; i < 6; ++i) g[ov + i] = C.Q[6 + i] * (x[ov + i] - C.q[ov + i]);
  T gq[4];  // gradient with respect to the (unnormalised) quaternion
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    T acc = T(0.0);
#pragma unroll
    for (int i = 0; i < 3; ++i) acc = acc + (C.Q[3 + i] * phi[i]) * (V[i][t] - phi[i] * q0[t]);
    gq[t] = acc * rs;
  }
  if constexpr (n == 13) {
#pragma unroll
    for (int t = 0; t < 4; ++t) g[3 + t] = gq[t];
  } else {  // chain rule through q̃(p) = [1 − |p|², 2p] (MRP) or [1, g] (RodriguesParam)
    const bool mrp = ((int)C.w == 1);
#pragma unroll
    for (int t = 0; t < 3; ++t) g[3 + t] = mrp ? (2.0 * gq[1 + t] - (2.0 * x[3 + t]) * gq[0]) : gq[1 + t];
  }
}

// ------------------------------------------------------------------------------------------------ costs
// J = ½x'Qx + q'x + c (+ ½u'Ru + r'u whenever u is given) (+ u'Hx) (+ w·min(1±dq))
template <int n, int m, bool DENSE = true>
__device__ __forceinline__ double cost_eval(CostC& C, const double* x, const double* u) {
  double J;
  if constexpr (DENSE && (n == 13 || n == 12)) {
    if (C.kind == TO_COST_ERROR_QUADRATIC) {
      double uRu = 0.0, ru = 0.0;
#pragma unroll
      for (int i = 0; i < m; ++i) { uRu += u[i] * C.R[i] * u[i]; ru += C.r[i] * u[i]; }
      return errquad_value<n, double>(C, x) + C.c + (0.5 * uRu + ru);
    }
  }
  if (DENSE && C.kind == TO_COST_QUADRATIC) {
    double xQx = 0.0;
#pragma unroll
    for (int j = 0; j < n; ++j) {
      double t = 0.0;
#pragma unroll
      for (int i = 0; i < n; ++i) t += x[i] * C.Q[i + n * j];
      xQx += t * x[j];
    }
    J = 0.5 * xQx;
  } else {
    double xQx = 0.0;
#pragma unroll
    for (int i = 0; i < n; ++i) xQx += x[i] * C.Q[i] * x[i];
    J = 0.5 * xQx;
  }
  double qx = 0.0;
#pragma unroll
  for (int i = 0; i < n; ++i) qx += C.q[i] * x[i];
  J = J + qx + C.c;
  {
    double uRu = 0.0, ru = 0.0;
    if (C.kind == TO_COST_QUADRATIC) {
#pragma unroll
      for (int j = 0; j < m; ++j) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < m; ++i) t += u[i] * C.R[i + m * j];
        uRu += t * u[j];
      }
    } else {
#pragma unroll
      for (int i = 0; i < m; ++i) uRu += u[i] * C.R[i] * u[i];
    }
#pragma unroll
    for (int i = 0; i < m; ++i) ru += C.r[i] * u[i];
    J += 0.5 * uRu + ru;
    if (C.kind == TO_COST_QUADRATIC) {
      double uHx = 0.0;
#pragma unroll
      for (int j = 0; j < n; ++j)
#pragma unroll
        for (int i = 0; i < m; ++i) uHx += u[i] * C.H[i + m * j] * x[j];
      J += uHx;
    }
  }
  if (C.kind == TO_COST_DIAGONAL_QUAT) {
    const double qr[4] = {C.q_ref[0], C.q_ref[1], C.q_ref[2], C.q_ref[3]};
    const int qi[4] = {C.q_ind[0], C.q_ind[1], C.q_ind[2], C.q_ind[3]};
    const double dq = quat_dot<n>(qr, qi, x);
    J += C.w * fmin(1 + dq, 1 - dq);
  }
  return J;
}

// Stage cost of a diagonal kind preloaded into registers once per kernel (hot rollout loops): identical arithmetic to
// cost_eval's diagonal branch, without the per-knot descriptor fetches.
template <int n, int m>
struct StageCostDiag {
  double Q[n], R[m], q[n], r[m], c, w, qref[4];
  int qind[4], kind;
  __device__ __forceinline__ void load(CostC& C) {
#pragma unroll
    for (int i = 0; i < n; ++i) { Q[i] = in_vgpr(C.Q[i]); q[i] = in_vgpr(C.q[i]); }
#pragma unroll
    for (int i = 0; i < m; ++i) { R[i] = in_vgpr(C.R[i]); r[i] = in_vgpr(C.r[i]); }
    c = in_vgpr(C.c); w = in_vgpr(C.w); kind = C.kind;
#pragma unroll
    for (int i = 0; i < 4; ++i) { qref[i] = in_vgpr(C.q_ref[i]); qind[i] = C.q_ind[i]; }
  }
  __device__ __forceinline__ double eval(const double* x, const double* u) const {
    double xQx = 0.0;
#pragma unroll
    for (int i = 0; i < n; ++i) xQx += x[i] * Q[i] * x[i];
    double J = 0.5 * xQx;
    double qx = 0.0;
#pragma unroll
    for (int i = 0; i < n; ++i) qx += q[i] * x[i];
    J = J + qx + c;
    double uRu = 0.0, ru = 0.0;
#pragma unroll
    for (int i = 0; i < m; ++i) uRu += u[i] * R[i] * u[i];
#pragma unroll
    for (int i = 0; i < m; ++i) ru += r[i] * u[i];
    J += 0.5 * uRu + ru;
    if (kind == TO_COST_DIAGONAL_QUAT) {
      const double dq = quat_dot<n>(qref, qind, x);
      J += w * fmin(1 + dq, 1 - dq);
    }
    return J;
  }
};

// The same stage cost kept in LDS instead of 80 VGPRs (n = 13, m = 4): the rollout loops of the large models run at one wave
// per SIMD with every VGPR taken; 2(n+m)+6 wave-uniform doubles read back with broadcast LDS loads cost ~20 ds_read_b128
// per knot and free the registers the gains row needs.  Same arithmetic, same order as StageCostDiag::eval.
template <int n, int m>
struct StageCostLds {
  static constexpr int oQ = 0, oR = n, oq = n + m, orr = 2 * n + m, oc = 2 * n + 2 * m, ow = oc + 1, oref = oc + 2, used = oc + 6;
  static constexpr int size = 64;  // one entry per lane: the fill below runs without any lane masked off
  static_assert(used <= 64, "stage-cost table is filled by one wave-wide store");
  const double* t;
  int qind[4], kind;
  // Call from every lane of the wave.  Branch-free on purpose: every lane stores one entry (the tail entries are
  // padding).  A lane-divergent region here made hipcc (ROCm 7.2) place a VGPR->AGPR spill of a live value in the join
  // block BEFORE exec was restored, which lost that value in the lanes outside the region (the line search's step size,
  // seen as wrong step sizes from the second round on); the forward kernels keep their EXEC mask full for that reason.
  __device__ __forceinline__ void load(CostC& C, double* tab, int hw) {
    t = tab;
    DoubleC* src = C.Q;
    int idx = hw;
    src = (hw >= oR) ? C.R : src;       idx = (hw >= oR) ? hw - oR : idx;
    src = (hw >= oq) ? C.q : src;       idx = (hw >= oq) ? hw - oq : idx;
    src = (hw >= orr) ? C.r : src;      idx = (hw >= orr) ? hw - orr : idx;
    src = (hw >= oc) ? &C.c : src;      idx = (hw >= oc) ? 0 : idx;
    src = (hw >= ow) ? &C.w : src;
    src = (hw >= oref) ? C.q_ref : src; idx = (hw >= oref) ? hw - oref : idx;
    src = (hw >= used) ? C.Q : src;     idx = (hw >= used) ? 0 : idx;
    tab[hw] = src[idx];
    kind = C.kind;
#pragma unroll
    for (int i = 0; i < 4; ++i) qind[i] = C.q_ind[i];
  }
  __device__ __forceinline__ double eval(const double* x, const double* u) const {
    double xQx = 0.0;
#pragma unroll
    for (int i = 0; i < n; ++i) xQx += x[i] * t[oQ + i] * x[i];
    double J = 0.5 * xQx;
    double qx = 0.0;
#pragma unroll
    for (int i = 0; i < n; ++i) qx += t[oq + i] * x[i];
    J = J + qx + t[oc];
    double uRu = 0.0, ru = 0.0;
#pragma unroll
    for (int i = 0; i < m; ++i) uRu += u[i] * t[oR + i] * u[i];
#pragma unroll
    for (int i = 0; i < m; ++i) ru += t[orr + i] * u[i];
    J += 0.5 * uRu + ru;
    if (kind == TO_COST_DIAGONAL_QUAT) {
      double qref[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) qref[i] = t[oref + i];
      const double dq = quat_dot<n>(qref, qind, x);
      J += t[ow] * fmin(1 + dq, 1 - dq);
    }
    return J;
  }
};

// gradient g (n+m) and Hessian-vector product y = H v (n+m) of the cost at (x,u); u-parts zero if terminal
// DENSE = false compiles the QuadraticCost (full Q, R, H) branches out.
template <int n, int m, bool DENSE = true>
__device__ __forceinline__ void cost_grad_hvp(CostC& C, const double* x, const double* u, bool terminal,
                                              const double* v, double* g, double* y) {
  if constexpr (DENSE && (n == 13 || n == 12)) {
    if (C.kind == TO_COST_ERROR_QUADRATIC) {
      Dual xd[n], gd[n];
#pragma unroll
      for (int i = 0; i < n; ++i) xd[i] = Dual(x[i], v[i]);
      errquad_grad<n, Dual>(C, xd, gd);
#pragma unroll
      for (int i = 0; i < n; ++i) { g[i] = gd[i].v; y[i] = gd[i].d; }
#pragma unroll
      for (int i = 0; i < m; ++i) {
        g[n + i] = terminal ? 0.0 : C.R[i] * u[i] + C.r[i];
        y[n + i] = terminal ? 0.0 : C.R[i] * v[n + i];
      }
      return;
    }
  }
  if (DENSE && C.kind == TO_COST_QUADRATIC) {
#pragma unroll
    for (int i = 0; i < n; ++i) {
      double t = C.q[i], h = 0.0;
#pragma unroll
      for (int j = 0; j < n; ++j) { t += C.Q[i + n * j] * x[j]; h += C.Q[i + n * j] * v[j]; }
      g[i] = t; y[i] = h;
    }
  } else {
#pragma unroll
    for (int i = 0; i < n; ++i) { g[i] = C.Q[i] * x[i] + C.q[i]; y[i] = C.Q[i] * v[i]; }
  }
  if (C.kind == TO_COST_DIAGONAL_QUAT) {  // the intended gradient src/lie_costs.jl:82-90
    const double qr[4] = {C.q_ref[0], C.q_ref[1], C.q_ref[2], C.q_ref[3]};
    const int qi[4] = {C.q_ind[0], C.q_ind[1], C.q_ind[2], C.q_ind[3]};
    const double dq = quat_dot<n>(qr, qi, x);
    bool fast = false;
    if constexpr (n >= 7) fast = (qi[0] == 4 && qi[1] == 5 && qi[2] == 6 && qi[3] == 7);
    if (fast) {
      if constexpr (n >= 7) {
#pragma unroll
        for (int i = 0; i < 4; ++i) g[3 + i] += dq < 0 ? C.w * qr[i] : -(C.w * qr[i]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) add_at<n>(g, qi[i] - 1, dq < 0 ? C.w * qr[i] : -(C.w * qr[i]));
    }
  }
#pragma unroll
  for (int i = 0; i < m; ++i) { g[n + i] = 0.0; y[n + i] = 0.0; }
  if (!terminal) {
    if (DENSE && C.kind == TO_COST_QUADRATIC) {
#pragma unroll
      for (int i = 0; i < m; ++i) {
        double t = C.r[i], h = 0.0;
#pragma unroll
        for (int j = 0; j < m; ++j) { t += C.R[i + m * j] * u[j]; h += C.R[i + m * j] * v[n + j]; }
        g[n + i] = t; y[n + i] = h;
      }
#pragma unroll
      for (int j = 0; j < n; ++j)
#pragma unroll
        for (int i = 0; i < m; ++i) {
          const double Hij = C.H[i + m * j];
          g[j] += Hij * u[i]; g[n + i] += Hij * x[j];
          y[j] += Hij * v[n + i]; y[n + i] += Hij * v[j];
        }
    } else {
#pragma unroll
      for (int i = 0; i < m; ++i) { g[n + i] = C.R[i] * u[i] + C.r[i]; y[n + i] = C.R[i] * v[n + i]; }
    }
  }
}

// ------------------------------------------------------------------------------------------------ constraints
// One row of a non-selector constraint: value c and the gradient coefficients on z[inds[t]], t < n_inds.
template <int nz>
__device__ __forceinline__ double con_row(ConC& K, const double* z, int r, double* coef /* [nz] */) {
  const to_constraint_desc TO_CONST_AS& d = K.d;
  double c = 0.0;
  switch (d.kind) {
    case TO_CON_NORM: {  // quadratic form: |z[inds]|² − val²
#pragma unroll
      for (int t = 0; t < nz; ++t)
        if (t < d.n_inds) { double zj = pick<nz>(z, d.inds[t] - 1); c += zj * zj; coef[t] = 2 * zj; }
      c -= d.params[0] * d.params[0];
      break;
    }
    case TO_CON_CIRCLE: {
      const int P = K.p;
      const double dx = pick<nz>(z, d.inds[0] - 1) - d.params[r], dy = pick<nz>(z, d.inds[1] - 1) - d.params[P + r];
      const double rad = d.params[2 * P + r];
      c = -(dx * dx) - dy * dy + rad * rad;
      coef[0] = -2 * dx; coef[1] = -2 * dy;
      break;
    }
    case TO_CON_SPHERE: {
      const int P = K.p;
      const double dx = pick<nz>(z, d.inds[0] - 1) - d.params[r], dy = pick<nz>(z, d.inds[1] - 1) - d.params[P + r];
      const double dz = pick<nz>(z, d.inds[2] - 1) - d.params[2 * P + r], rad = d.params[3 * P + r];
      c = -(dx * dx) - dy * dy - dz * dz + rad * rad;
      coef[0] = -2 * dx; coef[1] = -2 * dy;
      if constexpr (nz > 2) coef[2] = -2 * dz;
      break;
    }
    case TO_CON_LINEAR: {
      const int p = K.p;
#pragma unroll
      for (int t = 0; t < nz; ++t)
        if (t < d.n_inds) { const double a = d.params[r + p * t]; c += a * pick<nz>(z, d.inds[t] - 1); coef[t] = a; }
      c -= d.params[p * d.n_inds + r];
      break;
    }
    case TO_CON_QUATVEC: {  // row r of vec(q/|q|) − sign(qf'q) vec(qf); coefficients = row r+1 of (I − q̂q̂ᵀ)/|q|
      double q[4], n2 = 0.0, dq = 0.0;
#pragma unroll
      for (int t = 0; t < 4; ++t) { q[t] = pick<nz>(z, d.inds[t] - 1); n2 += q[t] * q[t]; }
      const double inv = rcp_fast(sqrt(n2));
#pragma unroll
      for (int t = 0; t < 4; ++t) { q[t] *= inv; dq += d.params[t] * q[t]; }
      const double qr = (r == 0) ? q[1] : (r == 1) ? q[2] : q[3];
      const double qfr = (r == 0) ? d.params[1] : (r == 1) ? d.params[2] : d.params[3];
      c = -((dq < 0 ? -qfr : qfr) - qr);
      if constexpr (nz >= 4) {
#pragma unroll
        for (int t = 0; t < 4; ++t) coef[t] = (((t == r + 1) ? 1.0 : 0.0) - qr * q[t]) * inv;
      }
      break;
    }
    case TO_CON_COLLISION: {  // r² − |x[x1] − x[x2]|²; inds = [x1; x2]
      const int D = d.n_inds / 2;
      c = d.params[0] * d.params[0];
#pragma unroll
      for (int t = 0; t < nz; ++t) coef[t] = 0.0;
#pragma unroll
      for (int t = 0; t < nz / 2; ++t)
        if (t < D) {
          const double dd = pick<nz>(z, d.inds[t] - 1) - pick<nz>(z, d.inds[D + t] - 1);
          c -= dd * dd;
          add_at<nz>(coef, t, -2 * dd); add_at<nz>(coef, D + t, 2 * dd);
        }
      break;
    }
    default: break;
  }
  return c;
}

// ---- selector rows with compile-time register indices --------------------------------------------------------
// The z-index of a selector row is wave-uniform but only known at run time; indexing registers with it costs an
// nz-long select chain per access.  The two layouts every rigid-body problem uses (goal on a state prefix, norm / SOC
// on the control block) are visited with compile-time indices instead; anything else takes the generic chain.
template <int I> struct SIdx { static constexpr int value = I; };
template <int N_, int I> __device__ __forceinline__ double zget(const double* a, SIdx<I>) { return a[I]; }
template <int N_> __device__ __forceinline__ double zget(const double* a, int idx) { return pick<N_>(a, idx); }
template <int N_, int I> __device__ __forceinline__ void zadd(double* a, SIdx<I>, double v) { a[I] += v; }
template <int N_> __device__ __forceinline__ void zadd(double* a, int idx, double v) { add_at<N_>(a, idx, v); }
template <int OFF, class F, int... Is>
__device__ __forceinline__ void visit_static(int D, F&& f, std::integer_sequence<int, Is...>) {
  ((Is < D ? (f(Is, SIdx<OFF + Is>{}), 0) : 0), ...);
}
// calls f(row, idx) for the first D selector rows of K; idx is an SIdx<> on the fast layouts, an int otherwise
// CTRL_ONLY: the caller knows K.fast == 2 (register-cached control-block constraints): the other paths, which would index
// the caller's register arrays with a run-time row and so push them into scratch memory, are not even compiled.
template <int n, int m, bool CTRL_ONLY = false, class F>
__device__ __forceinline__ void visit_rows(ConC& K, int D, F&& f) {
  if constexpr (CTRL_ONLY) { visit_static<n>(D, f, std::make_integer_sequence<int, m>{}); return; }
  if (K.fast == 1) visit_static<0>(D, f, std::make_integer_sequence<int, n>{});
  else if (K.fast == 2) visit_static<n>(D, f, std::make_integer_sequence<int, m>{});
  else for (int r = 0; r < D; ++r) f(r, (int)K.sidx[r]);
}

// value of row r of a selector constraint
template <int nz>
__device__ __forceinline__ double sel_row(ConC& K, const double* z, int r) {
  const int j = K.sidx[r];
  return j < 0 ? K.soff[r] : K.ssgn[r] * (pick<nz>(z, j) - K.soff[r]);
}

// AL penalty of one constraint at one knot.  lam: pointer to row 0 of this knot's duals
// (batch-fastest: row r at lam[r*stride]).
template <int n, int m, bool GENERIC = true>
__device__ __forceinline__ double al_term(ConC& K, const double* z, const double* lam, size_t stride, double mu) {
  constexpr int nz = n + m;
  const int p = K.p;
  double J = 0.0;
  if (K.d.sense == TO_CONE_SECOND_ORDER) {
    // psi = (|Π(lb)|² − |lam|²)/(2mu);  |Π|² = 0 | |lb|² | 2·coef²·a²
    double a2 = 0.0, l2 = 0.0;
    visit_rows<n, m>(K, p - 1, [&](int r, auto idx) {
      const double l = lam[r * stride];
      const double lb = l - mu * (K.ssgn[r] * (zget<nz>(z, idx) - K.soff[r]));
      l2 += l * l;
      a2 += lb * lb;
    });
    const double ls = lam[(p - 1) * stride];
    const double s = ls - mu * K.soff[p - 1];
    l2 += ls * ls;
    const double a = sqrt(a2);
    double pn;
    if (a <= -s) pn = 0.0;
    else if (a <= s) pn = a2 + s * s;
    else { const double cf = 0.5 * (1 + s * rcp_fast(a)); pn = (cf * cf) * a2 + (a * cf) * (a * cf); }
    J = (pn - l2) * (0.5 * rcp_fast(mu));
  } else if (K.selector) {
    const bool eq = (K.d.sense == TO_CONE_ZERO);
    visit_rows<n, m>(K, p, [&](int r, auto idx) {
      const double l = lam[r * stride], c = K.ssgn[r] * (zget<nz>(z, idx) - K.soff[r]);
      const bool active = eq || (c >= 0.0) || (l > 0.0);
      J += l * c + (active ? 0.5 * mu * c * c : 0.0);
    });
  } else if constexpr (GENERIC) {
    double coef[nz];
    for (int r = 0; r < p; ++r) {
      const double l = lam[r * stride], c = con_row<nz>(K, z, r, coef);
      const bool active = (K.d.sense == TO_CONE_ZERO) || (c >= 0.0) || (l > 0.0);
      J += l * c + (active ? 0.5 * mu * c * c : 0.0);
    }
  }
  return J;
}

// NormConstraint(SECOND_ORDER) over the whole control vector as the reference builds it: rows r < m are u_r itself (sign 1,
// offset 0), row m is the constant bound.  When every register-cacheable control constraint of a problem has this form
// (DevProblem::unit_soc) the forward pass runs variants compiled with UNIT = true, whose hot loops hold neither ssgn/soff (20
// SGPRs per constraint, spilled to VGPR lanes and reloaded one by one every knot) nor the per-row predicates: C5 forward
// pass 2325 -> 2021 us per batch step.  (A run-time branch to the same code does not help: the other path keeps the
// registers live.)  The expressions are those of the general path with ssgn = 1, soff = 0 substituted: bit-identical.
template <int m>
__host__ __device__ inline bool unit_soc_desc(int sense, int fast, int p, const double* ssgn, const double* soff) {
  bool unit = sense == TO_CONE_SECOND_ORDER && fast == 2 && p == m + 1;
  for (int r = 0; r < m && unit; ++r) unit = ssgn[r] == 1.0 && soff[r] == 0.0;
  return unit;
}
// A constraint whose selector rows are the control block (fast == 2) and that applies to every stage knot, cached in
// registers for the rollout loops: al_term re-reads k1/k2/p/sense/ssgn/soff from the descriptor table on every knot —
// ~80 dependent scalar loads per Quadrotor knot with the C5 constraint set.  Same expressions as al_term.
template <int n, int m, bool UNIT = false>
struct ConStage {
  int ci, p, sense;
  double ssgn[UNIT ? 1 : m + 1], soff[UNIT ? 1 : m + 1], mu;  // UNIT: soff[0] holds the bound (the general soff[m])
  double lcur[m + 1], lnxt[m + 1];  // duals of the current knot / fetched one knot ahead (a load at the point of use
                                    // costs a full memory round trip per knot: measured 2.2x the waits of the unconstrained loop)
  const double* lam;  // this lane's dual row 0 at knot 0; row r of knot k at lam[(k*p + r)*64]
  __device__ __forceinline__ void prefetch(int k) {
    const double* lk = lam + (size_t)k * p * 64;
#pragma unroll
    for (int r = 0; r < m + 1; ++r) lnxt[r] = (UNIT || r < p) ? lk[r * 64] : 0.0;
  }
  __device__ __forceinline__ void advance() {
#pragma unroll
    for (int r = 0; r < m + 1; ++r) lcur[r] = lnxt[r];
  }
  __device__ __forceinline__ void load(ConC& K, int ci_, const double* lam0, const double* mu0) {
    ci = ci_; p = UNIT ? m + 1 : K.p; sense = K.d.sense;
    if constexpr (UNIT) { ssgn[0] = 1.0; soff[0] = K.soff[m]; }
    else {
#pragma unroll
      for (int r = 0; r < m + 1; ++r) { ssgn[r] = (r < p) ? K.ssgn[r] : 0.0; soff[r] = (r < p) ? K.soff[r] : 0.0; }
    }
    mu = mu0[(size_t)ci_ * 64];
    lam = lam0 + (size_t)K.dual_off * 64;
  }
  // uses the duals in lcur (prefetch(k) + advance() by the caller)
  __device__ __forceinline__ double term(const double* u) const {
    double J = 0.0;
    if (UNIT || sense == TO_CONE_SECOND_ORDER) {
      double a2 = 0.0, l2 = 0.0, ls = 0.0, so = 0.0;
      if constexpr (UNIT) {
#pragma unroll
        for (int r = 0; r < m; ++r) { const double l = lcur[r]; const double lb = l - mu * u[r]; l2 += l * l; a2 += lb * lb; }
        ls = lcur[m]; so = soff[0];
      } else {
#pragma unroll
        for (int r = 0; r < m; ++r)
          if (r < p - 1) {
            const double l = lcur[r];
            const double lb = l - mu * (ssgn[r] * (u[r] - soff[r]));
            l2 += l * l;
            a2 += lb * lb;
          }
#pragma unroll
        for (int r = 0; r < m + 1; ++r)
          if (r == p - 1) { ls = lcur[r]; so = soff[r]; }
      }
      const double s = ls - mu * so;
      l2 += ls * ls;
      const double a = sqrt(a2);
      double pn;
      if (a <= -s) pn = 0.0;
      else if (a <= s) pn = a2 + s * s;
      else { const double cf = 0.5 * (1 + s * rcp_fast(a)); pn = (cf * cf) * a2 + (a * cf) * (a * cf); }
      J = (pn - l2) * (0.5 * rcp_fast(mu));
    } else {
      const bool eq = (sense == TO_CONE_ZERO);
#pragma unroll
      for (int r = 0; r < m; ++r)
        if (r < p) {
          const double l = lcur[r], c = ssgn[r] * (u[r] - soff[r]);
          const bool active = eq || (c >= 0.0) || (l > 0.0);
          J += l * c + (active ? 0.5 * mu * c * c : 0.0);
        }
    }
    return J;
  }
};

// adds the AL gradient (g += ∇c' y) and Gauss-Newton Hessian-vector product (y += ∇c' W ∇c v) of one constraint.
// For the SOC the reference composes ∇Π'∇Π + ∇²Π[Π] (src/cones.jl); since ∇(½|Π(x)|²) = Π(x) this equals ∇Π(x), and
// ∇Π(x)'Π(x) = Π(x); the closed forms below are those identities (checked against the explicit composition in tests).
// GENERIC = false compiles the non-selector constraint kinds (circle, sphere, linear, quadratic-form norm) out.
// REGROWS > 0: lam points to a register copy of the first REGROWS dual rows (stride 1, fetched early by the caller).
// y += yr * (d2 c_r / dz2) v for row r of a non-selector constraint: the closed forms of to_constraint_hessians
// (k_constraint_hessian; src/abstract_constraint.jl:255-280).  Affine rows (LINEAR) contribute nothing.
template <int nz>
__device__ __forceinline__ void con_curvature_v(ConC& K, int r, const double* z, const double* v, double yr, double* y) {
  const int kind = K.d.kind;
  if (kind == TO_CON_NORM) {  // |z_I|^2 - a^2 (the SOC form is a selector constraint and never gets here): 2 I on I
    for (int t = 0; t < K.d.n_inds; ++t) { const int i = K.d.inds[t] - 1; add_at<nz>(y, i, 2.0 * yr * pick<nz>(v, i)); }
  } else if (kind == TO_CON_CIRCLE || kind == TO_CON_SPHERE) {  // r^2 - |x - c|^2: -2 I on the centre coordinates
    const int D = kind == TO_CON_CIRCLE ? 2 : 3;
    for (int t = 0; t < D; ++t) { const int i = K.d.inds[t] - 1; add_at<nz>(y, i, -2.0 * yr * pick<nz>(v, i)); }
  } else if (kind == TO_CON_COLLISION) {  // r^2 - |x_a - x_b|^2: -2 [I -I; -I I]
    const int D = K.d.n_inds / 2;
    for (int t = 0; t < D; ++t) {
      const int ia = K.d.inds[t] - 1, ib = K.d.inds[D + t] - 1;
      const double d = pick<nz>(v, ia) - pick<nz>(v, ib);
      add_at<nz>(y, ia, -2.0 * yr * d);
      add_at<nz>(y, ib, 2.0 * yr * d);
    }
  } else if (kind == TO_CON_QUATVEC) {  // second derivative of q_{r+1} / |q|
    double q[4], vq[4], s2 = 0.0;
    for (int t = 0; t < 4; ++t) { q[t] = pick<nz>(z, K.d.inds[t] - 1); vq[t] = pick<nz>(v, K.d.inds[t] - 1); s2 += q[t] * q[t]; }
    const double s = sqrt(s2), s3 = s2 * s, s5 = s3 * s2;
    const int i = r + 1;
    for (int j = 0; j < 4; ++j) {
      double t = 0.0;
      for (int kq = 0; kq < 4; ++kq)
        t += (-((i == j ? q[kq] : 0.0) + (i == kq ? q[j] : 0.0) + (j == kq ? q[i] : 0.0)) / s3 + 3.0 * q[i] * q[j] * q[kq] / s5) * vq[kq];
      add_at<nz>(y, K.d.inds[j] - 1, yr * t);
    }
  }
}

// full_newton (to_solver_opts::al_full_newton): the Hessian-vector product also carries the constraint curvature
// sum_r ybar_r d2c_r/dz2 v (non-selector constraints; selector rows are affine and the SOC closed forms are already exact).
template <int n, int m, bool GENERIC = true, int REGROWS = 0>
__device__ __forceinline__ void al_grad_hvp(ConC& K, const double* z, const double* lam, size_t stride_rt, double mu,
                                            const double* v, double* g, double* y, bool full_newton = false) {
  constexpr int nz = n + m;
  const size_t stride = REGROWS > 0 ? (size_t)1 : stride_rt;
  const int p = K.p;
  if (K.d.sense == TO_CONE_SECOND_ORDER) {
    double a2 = 0.0, lw = 0.0;  // lw = lb_v · w_v with w = ∇c v
    visit_rows<n, m, (REGROWS > 0)>(K, p - 1, [&](int r, auto idx) {
      const double lb = lam[r * stride] - mu * (K.ssgn[r] * (zget<nz>(z, idx) - K.soff[r]));
      a2 += lb * lb;
      lw += lb * (K.ssgn[r] * zget<nz>(v, idx));
    });
    const double llast = REGROWS > 0 ? pick<(REGROWS > 0 ? REGROWS : 1)>(lam, p - 1) : lam[(p - 1) * stride];
    const double s = llast - mu * K.soff[p - 1];
    const double a = sqrt(a2);
    if (a <= -s) return;  // Π = 0, ∇Π = 0
    const bool inside = (a <= s);
    const double ra = rcp_fast(a);  // reciprocal-multiplies: an IEEE double division is ~27 VALU instructions on gfx950
    const double cf = inside ? 1.0 : 0.5 * (1 + s * ra);
    const double k3 = inside ? 0.0 : (0.5 * s) * (ra * ra * ra);
    visit_rows<n, m, (REGROWS > 0)>(K, p - 1, [&](int r, auto idx) {
      const double sg = K.ssgn[r];
      const double lb = lam[r * stride] - mu * (sg * (zget<nz>(z, idx) - K.soff[r]));
      zadd<nz>(g, idx, -sg * (cf * lb));                       // −∇c'Π(lb)
      const double w = sg * zget<nz>(v, idx);
      zadd<nz>(y, idx, mu * sg * (cf * w - k3 * lb * lw));     // µ ∇c' ∇Π(lb) ∇c v  (the s-row of ∇c is zero)
    });
  } else if (K.selector) {
    const bool eq = (K.d.sense == TO_CONE_ZERO);
    visit_rows<n, m, (REGROWS > 0)>(K, p, [&](int r, auto idx) {
      const double sg = K.ssgn[r];
      const double l = lam[r * stride], c = sg * (zget<nz>(z, idx) - K.soff[r]);
      const bool active = eq || (c >= 0.0) || (l > 0.0);
      zadd<nz>(g, idx, sg * (l + (active ? mu * c : 0.0)));
      zadd<nz>(y, idx, active ? mu * zget<nz>(v, idx) : 0.0);
    });
  } else if constexpr (GENERIC && REGROWS == 0) {
    double coef[nz];
    for (int r = 0; r < p; ++r) {
      const double l = lam[r * stride], c = con_row<nz>(K, z, r, coef);
      const bool active = (K.d.sense == TO_CONE_ZERO) || (c >= 0.0) || (l > 0.0);
      const double yr = l + (active ? mu * c : 0.0);
      double cv = 0.0;
#pragma unroll
      for (int t = 0; t < nz; ++t) if (t < K.d.n_inds) cv += coef[t] * pick<nz>(v, K.d.inds[t] - 1);
      const double wv = active ? mu * cv : 0.0;
#pragma unroll
      for (int t = 0; t < nz; ++t)
        if (t < K.d.n_inds) { add_at<nz>(g, K.d.inds[t] - 1, coef[t] * yr); add_at<nz>(y, K.d.inds[t] - 1, coef[t] * wv); }
      if (full_newton) con_curvature_v<nz>(K, r, z, v, yr, y);
    }
  }
}

// A control-block selector constraint (fast == 2, p <= m + 1) with its descriptor fields held in registers (SGPRs: they are
// wave-uniform).  The expansion walks several knots per wave; read through the descriptor table, every knot of the C5
// constraint set paid ~85 dependent scalar-load round trips (388 `s_waitcnt lgkmcnt(0)` in the 4-knot kernel against 48
// without constraints) that one wave per SIMD cannot hide.  Same expressions as al_grad_hvp<..., REGROWS>.
template <int m>
struct ConExp {
  int ci = -1, p = 0, sense = 0, k1 = 0, k2 = -1;
  long long dual_off = 0;
  double ssgn[m + 1], soff[m + 1];
  __device__ __forceinline__ void load(ConC& K, int ci_) {
    ci = ci_; p = K.p; sense = K.d.sense; k1 = K.k1; k2 = K.k2; dual_off = K.dual_off;
#pragma unroll
    for (int r = 0; r < m + 1; ++r) { ssgn[r] = (r < p) ? K.ssgn[r] : 0.0; soff[r] = (r < p) ? K.soff[r] : 0.0; }
  }
  __device__ __forceinline__ bool at(int k) const { return ci >= 0 && k >= k1 && k <= k2; }
};
// l: this knot's duals of the constraint (m + 1 registers), z = [x; u], v the direction; g += grad, y += Hessian-vector
template <int n, int m>
__device__ __forceinline__ void al_grad_hvp_ctrl(const ConExp<m>& C, const double* z, const double* l, double mu, const double* v,
                                                 double* g, double* y) {
  // (a variant specialised for unit SOCs, as the forward pass has, measured SLOWER here: C5 expansion 1580 -> 1675 us)
  const int p = C.p;
  if (C.sense == TO_CONE_SECOND_ORDER) {
    double a2 = 0.0, lw = 0.0, llast = 0.0, solast = 0.0;
#pragma unroll
    for (int r = 0; r < m; ++r)
      if (r < p - 1) {
        const double lb = l[r] - mu * (C.ssgn[r] * (z[n + r] - C.soff[r]));
        a2 += lb * lb;
        lw += lb * (C.ssgn[r] * v[n + r]);
      }
#pragma unroll
    for (int r = 0; r < m + 1; ++r)
      if (r == p - 1) { llast = l[r]; solast = C.soff[r]; }
    const double s = llast - mu * solast;
    const double a = sqrt(a2);
    if (a <= -s) return;  // Π = 0, ∇Π = 0
    const bool inside = (a <= s);
    const double ra = rcp_fast(a);
    const double cf = inside ? 1.0 : 0.5 * (1 + s * ra);
    const double k3 = inside ? 0.0 : (0.5 * s) * (ra * ra * ra);
#pragma unroll
    for (int r = 0; r < m; ++r)
      if (r < p - 1) {
        const double sg = C.ssgn[r];
        const double lb = l[r] - mu * (sg * (z[n + r] - C.soff[r]));
        g[n + r] += -sg * (cf * lb);
        const double w = sg * v[n + r];
        y[n + r] += mu * sg * (cf * w - k3 * lb * lw);
      }
  } else {
    const bool eq = (C.sense == TO_CONE_ZERO);
#pragma unroll
    for (int r = 0; r < m; ++r)
      if (r < p) {
        const double sg = C.ssgn[r];
        const double c = sg * (z[n + r] - C.soff[r]);
        const bool active = eq || (c >= 0.0) || (l[r] > 0.0);
        g[n + r] += sg * (l[r] + (active ? mu * c : 0.0));
        y[n + r] += active ? mu * v[n + r] : 0.0;
      }
  }
}

// max violation of one constraint at one knot
template <int nz>
__device__ __forceinline__ double con_violation(ConC& K, const double* z) {
  const int p = K.p;
  double vmax = 0.0;
  if (K.d.sense == TO_CONE_SECOND_ORDER) {
    double a2 = 0.0, s = 0.0;
    for (int r = 0; r < p; ++r) { const double c = sel_row<nz>(K, z, r); if (r < p - 1) a2 += c * c; else s = c; }
    const double a = sqrt(a2);
    if (a <= -s) { for (int r = 0; r < p; ++r) { const double v = fabs(sel_row<nz>(K, z, r)); if (!(v <= vmax)) vmax = v; } }
    else if (a <= s) vmax = 0.0;
    else {
      const double cf = 0.5 * (1 + s / a);
      for (int r = 0; r < p; ++r) {
        const double c = sel_row<nz>(K, z, r);
        const double pc = (r < p - 1) ? c * cf : a * cf;
        const double v = fabs(c - pc);
        if (!(v <= vmax)) vmax = v;
      }
    }
    return vmax;
  }
  double coef[nz];
  for (int r = 0; r < p; ++r) {
    const double c = K.selector ? sel_row<nz>(K, z, r) : con_row<nz>(K, z, r, coef);
    const double v = (K.d.sense == TO_CONE_ZERO) ? fabs(c) : fmax(0.0, c);
    if (!(v <= vmax)) vmax = v;
  }
  return vmax;
}

// dual update of one constraint at one knot (lam in place)
template <int nz>
__device__ __forceinline__ void con_dual_update(ConC& K, const double* z, double* lam, size_t stride, double mu, double dual_max) {
  const int p = K.p;
  if (K.d.sense == TO_CONE_SECOND_ORDER) {
    double a2 = 0.0, s = 0.0;
    for (int r = 0; r < p; ++r) { const double lb = lam[r * stride] - mu * sel_row<nz>(K, z, r); if (r < p - 1) a2 += lb * lb; else s = lb; }
    const double a = sqrt(a2);
    for (int r = 0; r < p; ++r) {
      const double lb = lam[r * stride] - mu * sel_row<nz>(K, z, r);
      double out;
      if (a <= -s) out = 0.0;
      else if (a <= s) out = lb;
      else { const double cf = 0.5 * (1 + s / a); out = (r < p - 1) ? lb * cf : a * cf; }
      lam[r * stride] = out;
    }
    return;
  }
  double coef[nz];
  for (int r = 0; r < p; ++r) {
    const double c = K.selector ? sel_row<nz>(K, z, r) : con_row<nz>(K, z, r, coef);
    const double l = lam[r * stride] + mu * c;
    lam[r * stride] = (K.d.sense == TO_CONE_ZERO) ? fmax(-dual_max, fmin(dual_max, l)) : fmin(dual_max, fmax(0.0, l));
  }
}

}  // namespace to

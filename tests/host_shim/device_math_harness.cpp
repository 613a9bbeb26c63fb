// Compiles csrc/models.h for the host and checks the device-side model math without a GPU (test infrastructure):
//   value of f against the numbers passed in (the oracle's), dual-number RK Jacobians against central differences,
//   chunk-mode duals against single-direction duals, and the error-state maps' algebraic identities.
#include "models.h"

#include <cstdio>
#include <cstdlib>
using namespace to;

static double rnd() { return 2.0 * rand() / RAND_MAX - 1.0; }

template <class M>
void run(const char* name, const double* P) {
  constexpr int n = M::n, m = M::m, ne = M::ne, nc = n + m;
  double worst_fd = 0, worst_md = 0, worst_inv = 0, worst_sym = 0;
  for (int trial = 0; trial < 40; ++trial) {
    double x[n], u[m];
    for (int i = 0; i < n; ++i) x[i] = 0.4 * rnd();
    if (M::att == ATT_QUAT) { double s = 0; x[3] += 1.0; for (int i = 3; i < 7; ++i) s += x[i] * x[i]; for (int i = 3; i < 7; ++i) x[i] /= std::sqrt(s); }
    for (int i = 0; i < m; ++i) u[i] = 1.2 + 0.8 * rnd();
    if (trial == 0) {  // one sample for the comparison with the oracle's dynamics
      double xd[n];
      M::template f<double>(P, x, u, xd);
      printf("%s x", name); for (int i = 0; i < n; ++i) printf(" %.17g", x[i]);
      printf("\n%s u", name); for (int i = 0; i < m; ++i) printf(" %.17g", u[i]);
      printf("\n%s f", name); for (int i = 0; i < n; ++i) printf(" %.17g", xd[i]);
      for (int integ = 0; integ < 3; ++integ) {  // one discrete step per integrator (RK4, RK3, Euler), h = 0.05
        double xs[n];
        rk_step<M, double>(P, integ, x, u, 0.05, xs);
        printf("\n%s step%d", name, integ); for (int i = 0; i < n; ++i) printf(" %.17g", xs[i]);
      }
      printf("\n");
    }
    for (int integ = 0; integ < 3; ++integ) {
      MDual<nc> xm[n], um[m], xnm[n];
      for (int i = 0; i < n; ++i) { xm[i].v = x[i]; xm[i].d[i] = 1.0; }
      for (int i = 0; i < m; ++i) { um[i].v = u[i]; um[i].d[n + i] = 1.0; }
      rk_step<M, MDual<nc>>(P, integ, xm, um, 0.05, xnm);
      for (int j = 0; j < nc; ++j) {
        Dual xd[n], ud[m], xn[n];
        for (int i = 0; i < n; ++i) xd[i] = Dual(x[i], i == j ? 1.0 : 0.0);
        for (int i = 0; i < m; ++i) ud[i] = Dual(u[i], n + i == j ? 1.0 : 0.0);
        rk_step<M, Dual>(P, integ, xd, ud, 0.05, xn);
        double xp[n], xq[n], up[m], uq[m], fp[n], fq[n];
        const double e = 1e-6;
        for (int i = 0; i < n; ++i) { xp[i] = x[i] + (i == j ? e : 0); xq[i] = x[i] - (i == j ? e : 0); }
        for (int i = 0; i < m; ++i) { up[i] = u[i] + (n + i == j ? e : 0); uq[i] = u[i] - (n + i == j ? e : 0); }
        rk_step<M, double>(P, integ, xp, up, 0.05, fp);
        rk_step<M, double>(P, integ, xq, uq, 0.05, fq);
        for (int i = 0; i < n; ++i) {
          const double fd = (fp[i] - fq[i]) / (2 * e);
          worst_fd = std::fmax(worst_fd, std::fabs(fd - xn[i].d));
          worst_md = std::fmax(worst_md, std::fabs(xnm[i].d[j] - xn[i].d));
        }
      }
    }
    // E(x) G(x) = I: errstate_invmul is a left inverse of the columns errstate_col produces
    for (int j = 0; j < ne; ++j) {
      double v[n], out[ne];
      errstate_col<M>(x, j, v);
      errstate_invmul<M>(x, v, out);
      for (int i = 0; i < ne; ++i) worst_inv = std::fmax(worst_inv, std::fabs(out[i] - (i == j ? 1.0 : 0.0)));
    }
    if constexpr (M::att == ATT_MRP || M::att == ATT_RP) {
      double b[3] = {rnd(), rnd(), rnd()}, H[9];
      att_differential2<M::att>(x + 3, b, H);
      for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) worst_sym = std::fmax(worst_sym, std::fabs(H[3 * i + j] - H[3 * j + i]));
    }
  }
  printf("%s dual_vs_fd %.3e mdual_vs_dual %.3e left_inverse %.3e hess_sym %.3e\n", name, worst_fd, worst_md, worst_inv, worst_sym);
}

// cartpole_rk4_jac (the chain rule over hand-derived stage partials, models.h) against the chunk-mode dual-number RK4 Jacobian and
// against central differences — small angles, large angles (the full sin / cos path of the later stages) and fast rotation
static void cartpole_stage_jac(const double* P) {
  double worst_dual = 0, worst_fd = 0, scale = 0;
  for (int trial = 0; trial < 200; ++trial) {
    double x[4], u[1], Mk[20];
    const double amp = trial < 100 ? 0.4 : 6.0, wamp = trial % 3 == 0 ? 25.0 : 3.0;
    x[0] = amp * rnd(); x[1] = amp * rnd(); x[2] = 3.0 * rnd(); x[3] = wamp * rnd(); u[0] = 4.0 * rnd();
    const double h = trial % 2 ? 0.05 : 0.02;
    cartpole_rk4_jac(P, x, u, h, Mk);
    MDual<5> xm[4], um[1], xn[4];
    for (int i = 0; i < 4; ++i) { xm[i].v = x[i]; xm[i].d[i] = 1.0; }
    um[0].v = u[0]; um[0].d[4] = 1.0;
    rk_step<CartpoleModel, MDual<5>, INTEG_RK4>(P, INTEG_RK4, xm, um, h, xn);
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 5; ++j) {
        worst_dual = std::fmax(worst_dual, std::fabs(Mk[i * 5 + j] - xn[i].d[j]));
        scale = std::fmax(scale, std::fabs(xn[i].d[j]));
      }
    for (int j = 0; j < 5; ++j) {
      double xp[4], xq[4], up[1] = {u[0]}, uq[1] = {u[0]}, fp[4], fq[4];
      const double e = 1e-6;
      for (int i = 0; i < 4; ++i) { xp[i] = x[i] + (i == j ? e : 0); xq[i] = x[i] - (i == j ? e : 0); }
      if (j == 4) { up[0] += e; uq[0] -= e; }
      rk_step<CartpoleModel, double, INTEG_RK4>(P, INTEG_RK4, xp, up, h, fp);
      rk_step<CartpoleModel, double, INTEG_RK4>(P, INTEG_RK4, xq, uq, h, fq);
      for (int i = 0; i < 4; ++i) worst_fd = std::fmax(worst_fd, std::fabs((fp[i] - fq[i]) / (2 * e) - Mk[i * 5 + j]));
    }
  }
  printf("cartpole_stage_jac vs_dual %.3e vs_fd %.3e scale %.3e\n", worst_dual, worst_fd, scale);
}

int main() {
  srand(7);
  const double Pq[16] = {0.5, 0.0023, 0.0023, 0.004, 0, 0, -9.81, 0.175, 1.0, 0.0245, 0};
  const double Pc[16] = {1.0, 0.2, 0.5, 9.81};
  const double Pd[16] = {1.3, 2};
  run<QuadrotorModel>("quat", Pq);
  run<QuadrotorAttModel<ATT_MRP>>("mrp", Pq);
  run<QuadrotorAttModel<ATT_RP>>("rp", Pq);
  run<CartpoleModel>("cartpole", Pc);
  run<DoubleIntegratorModel<2>>("di2", Pd);
  cartpole_stage_jac(Pc);
  return 0;
}

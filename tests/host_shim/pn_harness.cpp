// Compiles the projected-Newton KERNEL source (csrc/k_pn.h) for the host (TO_PN_HOST: every PN_FOR phase becomes a plain loop)
// together with the library's own descriptor lowering (csrc/desc_lower.h) and runs it on trajectories handed over in the
// C-ABI's host layout.  tests/test_pn_host.py compares the result with the oracle's polish: the logic of the GPU kernel —
// active set, block-tridiagonal factorisation, sweeps, refinement, line search — is checked on the CPU before any GPU run.
// Test infrastructure only; nothing in the product builds or loads this.
#define TO_PN_HOST 1
#include "k_pn.h"
#include "desc_lower.h"

#include <cstdint>
#include <string>
#include <vector>

namespace to {
static std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }
}  // namespace to
using namespace to;

template <class M>
static int run(const DevProblem& P, KArgs& a, std::vector<DevCon>& cons) {
  constexpr int ne = M::ne;
  const int N = P.N;
  std::vector<int> pak(N, 0);
  for (const DevCon& c : cons)
    for (int k = c.k1; k <= c.k2; ++k) pak[k] += (c.d.sense == TO_CONE_SECOND_ORDER) ? 1 : c.p;
  std::vector<long long> koff(N + 1, 0);
  koff[0] = PN_HEADER;
  int nbmax = 0;
  for (int k = 0; k < N; ++k) {
    if (pak[k] > PN_MAX_ROWS || ne + pak[k] > PN_NB_LIMIT) return fail(TO_ERR_UNSUPPORTED, "too many rows on one knot");
    nbmax = std::max(nbmax, ne + pak[k]);
    koff[k + 1] = koff[k] + pn_rec_size<M>(pak[k], k > 0 ? pak[k - 1] : 0, k == 0);
  }
  std::vector<double> ws(koff[N], 0.0), lds(pn_lds_doubles<M>(nbmax), 0.0);
  std::vector<int> list(P.B);
  for (int b = 0; b < P.B; ++b) list[b] = b;
  PnArgs q;
  q.a = a; q.pak = pak.data(); q.koff = koff.data(); q.ws = ws.data(); q.list = list.data(); q.base = 0; q.nbmax = nbmax;
  q.it_pn = a.it_pn; q.cmax_out = a.pn_cmax;
  constexpr int nc = M::ne + M::m;
  for (int b = 0; b < P.B; ++b)  // the launch sequence of csrc/ops_pn.hip, one trajectory at a time
    for (int round = 0; round <= P.opts.n_steps + 1; ++round) {
      pn_begin<M>(q, b, ws.data(), lds.data(), round);
      if (round == P.opts.n_steps + 1) break;
      if (ws[PN_H_STATE] == (double)PN_ACTIVE) {
        for (int it = 0; it < (N - 1) * nc; ++it) pn_lin_column<M>(q, ws.data(), it);
        for (int k = 0; k < N; ++k) pn_lin_knot<M>(q, ws.data(), k);
      }
      pn_project<M>(q, b, ws.data(), lds.data());
    }
  return TO_OK;
}

extern "C" const char* pn_host_last_error() { return g_err.c_str(); }

extern "C" int pn_host_solve(const to_problem_desc* desc, const to_solver_opts* opts, const double* x0, double* X, double* U,
                             int32_t* status, int32_t* it_pn, double* cmax) {
  int n, m, ne, key;
  if (model_dims(desc->model, desc->model_params, &n, &m, &ne, &key)) return fail(TO_ERR_UNSUPPORTED, "unknown model");
  const int N = desc->N, B = desc->B, Bp = (B + 63) / 64 * 64;
  DevProblem P;
  std::memset(&P, 0, sizeof(P));
  P.n = n; P.m = m; P.ne = ne; P.N = N; P.B = B; P.Bp = Bp; P.integrator = desc->integrator;
  std::memcpy(P.mp, desc->model_params, sizeof(P.mp));
  std::vector<double> step_table;
  if (desc->model == TO_MODEL_VECTOR) {
    int r = lower_step_models(desc->step_models, N, &step_table);
    if (r) return r;
    const unsigned long long bits = (unsigned long long)reinterpret_cast<uintptr_t>(step_table.data());
    std::memset(P.mp, 0, sizeof(P.mp));
    std::memcpy(&P.mp[0], &bits, sizeof(bits));
  }
  if (opts) { int r = validate_opts(*opts); if (r) return r; P.opts = *opts; } else default_opts(&P.opts);
  std::vector<double> dt(N - 1);
  for (int k = 0; k < N - 1; ++k) dt[k] = desc->dt ? desc->dt[k] : (desc->tf - desc->t0) / (N - 1);
  std::vector<int> cost_index(N);
  for (int k = 0; k < N; ++k) cost_index[k] = desc->cost_index ? desc->cost_index[k] : (k == N - 1 ? 1 : 0);
  std::vector<to_cost_desc> costs(desc->costs, desc->costs + desc->n_costs);
  std::vector<DevCon> cons(desc->n_constraints);
  long long duals = 0;
  for (int i = 0; i < desc->n_constraints; ++i) {
    int r = validate_constraint(n, m, N, desc->constraints[i], &cons[i]);
    if (r) return r;
    cons[i].dual_off = duals;
    duals += (long long)cons[i].p * (cons[i].k2 - cons[i].k1 + 1);
  }
  P.n_costs = desc->n_costs; P.n_cons = desc->n_constraints; P.n_duals = duals;
  P.dt = dt.data(); P.cost_index = cost_index.data(); P.costs = costs.data(); P.cons = cons.data();
  // tiled arrays (csrc/common.h): element e of trajectory b at base[((b/64)*L + e)*64 + b%64]
  const int Lx = N * n, Lu = (N - 1) * m;
  std::vector<double> Xs((size_t)Bp * Lx, 0.0), Us((size_t)Bp * Lu, 0.0), x0s((size_t)Bp * n, 0.0), pc(Bp, 0.0);
  std::vector<int> st(Bp, 0), ip(Bp, 0);
  auto at = [](std::vector<double>& v, int L, int b, int e) -> double& { return v[((size_t)(b / 64) * L + e) * 64 + b % 64]; };
  for (int b = 0; b < B; ++b) {
    for (int e = 0; e < Lx; ++e) at(Xs, Lx, b, e) = X[(size_t)b * Lx + e];
    for (int e = 0; e < Lu; ++e) at(Us, Lu, b, e) = U[(size_t)b * Lu + e];
    for (int e = 0; e < n; ++e) at(x0s, n, b, e) = x0[(size_t)b * n + e];
  }
  KArgs a;
  std::memset(&a, 0, sizeof(a));
  a.P = P; a.Xs = Xs.data(); a.Us = Us.data(); a.x0 = x0s.data(); a.status = st.data(); a.it_pn = ip.data(); a.pn_cmax = pc.data();
  int rc = TO_ERR_UNSUPPORTED;
  switch (key) {
    case 0: rc = run<DoubleIntegratorModel<1>>(P, a, cons); break;
    case 1: rc = run<DoubleIntegratorModel<2>>(P, a, cons); break;
    case 2: rc = run<DoubleIntegratorModel<3>>(P, a, cons); break;
    case 3: rc = run<CartpoleModel>(P, a, cons); break;
    case 4: rc = run<QuadrotorModel>(P, a, cons); break;
    case 5: rc = run<QuadrotorAttModel<ATT_MRP>>(P, a, cons); break;
    case 6: rc = run<QuadrotorAttModel<ATT_RP>>(P, a, cons); break;
    case 7: rc = run<HybridDoubleIntegratorModel>(P, a, cons); break;
    case 8: rc = run<ModelVectorModel>(P, a, cons); break;
  }
  if (rc) return rc;
  for (int b = 0; b < B; ++b) {
    for (int e = 0; e < Lx; ++e) X[(size_t)b * Lx + e] = at(Xs, Lx, b, e);
    for (int e = 0; e < Lu; ++e) U[(size_t)b * Lu + e] = at(Us, Lu, b, e);
    status[b] = st[b]; it_pn[b] = ip[b]; cmax[b] = pc[b];
  }
  return TO_OK;
}

// The integer helpers of the factorisation (k_pn.h): pn_div against /, pn_tri_row against the definition, over their whole domains.
// Returns 0, or the first failing case encoded as 1000000 * which + value.
extern "C" int pn_host_index_selftest() {
  for (int d = 2; d <= 64; ++d) {
    const unsigned r = pn_recip(d);
    for (int e = 0; e < 4096; ++e)
      if (pn_div(e, r) != e / d) return 1000000 + e;
  }
  for (int e = 0; e < 4096; ++e) {
    const int i = pn_tri_row(e);
    if (!(i * (i + 1) / 2 <= e && e < (i + 1) * (i + 2) / 2)) return 2000000 + e;
  }
  return 0;
}

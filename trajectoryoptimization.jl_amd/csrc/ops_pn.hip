// ops_pn.hip — projected-Newton polish (k_pn.h) and the dynamics-defect kernel, every model.
#include <algorithm>
#include <vector>

#include "handle.h"
#include "k_pn.h"

namespace to {

// The polish of a batch of trajectories in two steps, so that launches can go to any stream without touching the host tables:
//   op_pn_prepare: per-knot tables (pak[k] = candidate rows of knot k, record offsets) on the device and a workspace of one record
//                  per knot for up to `want` trajectories, capped by the budget (TRAJOPT_PN_WS_GB, default 16 GB of the 288):
//                  h->pn_cap slots.  Synchronises h->stream when it has to (re)allocate.
//   op_pn_launch:  polish the `count` trajectories h->pn_list[slot0 ...] (device array, filled by the caller) from their current
//                  nominal (X, U), in workspace slots slot0 ..., on `stream`, with the solver options `opts`.
template <class M>
int op_pn_prepare(to_handle* h, int want) {
  constexpr int ne = M::ne;
  const DevProblem& P = h->a.P;
  const int N = P.N;
  std::vector<int> pak(N, 0);
  for (const DevCon& c : h->cons)
    for (int k = c.k1; k <= c.k2; ++k) pak[k] += (c.d.sense == TO_CONE_SECOND_ORDER) ? 1 : c.p;
  int nbmax = 0;
  std::vector<long long> koff(N + 1, 0);
  koff[0] = PN_HEADER;
  for (int k = 0; k < N; ++k) {
    if (pak[k] > PN_MAX_ROWS || ne + pak[k] > PN_NB_LIMIT)
      return fail(TO_ERR_UNSUPPORTED, "projected Newton: more than " + std::to_string(std::min(PN_MAX_ROWS, PN_NB_LIMIT - ne)) + " constraint rows on one knot");
    nbmax = std::max(nbmax, ne + pak[k]);
    koff[k + 1] = koff[k] + pn_rec_size<M>(pak[k], k > 0 ? pak[k - 1] : 0, k == 0);
  }
  h->pn_nbmax = nbmax;
  double gb = 16.0;
  if (const char* env = std::getenv("TRAJOPT_PN_WS_GB")) gb = std::max(0.01, std::atof(env));
  const size_t per = (size_t)koff[N] * sizeof(double);
  want = std::max(1, want);
  const int cap = (int)std::min<size_t>((size_t)want, std::max<size_t>(1, (size_t)(gb * 1073741824.0) / per));
  if (h->pn_ws_bytes < per * cap) {
    HIPCHECK(hipDeviceSynchronize());  // the polish may still be running on its own stream
    if (h->pn_ws) { g_free(h, h->pn_ws); h->pn_ws = nullptr; h->pn_ws_bytes = 0; }
    TRY(g_malloc(h, (void**)&h->pn_ws, per * cap, "pn_ws"));
    h->pn_ws_bytes = per * cap;
  }
  h->pn_cap = (int)(h->pn_ws_bytes / per);
  h->pn_per = (long long)koff[N];
  if (h->pn_tab_len < N + 1) {
    HIPCHECK(hipDeviceSynchronize());
    if (h->pn_pak) g_free(h, h->pn_pak);
    if (h->pn_koff) g_free(h, h->pn_koff);
    if (h->pn_list) g_free(h, h->pn_list);
    if (h->pn_list_host) HIPCHECK(hipHostFree(h->pn_list_host));
    TRY(g_malloc(h, (void**)&h->pn_pak, sizeof(int) * (N + 1), "pn_pak"));
    TRY(g_malloc(h, (void**)&h->pn_koff, sizeof(long long) * (N + 1), "pn_koff"));
    TRY(g_malloc(h, (void**)&h->pn_list, sizeof(int) * P.Bp, "pn_list"));
    HIPCHECK(hipHostMalloc((void**)&h->pn_list_host, sizeof(int) * P.Bp));
    h->pn_tab_len = N + 1;
  }
  HIPCHECK(hipMemcpyAsync(h->pn_pak, pak.data(), sizeof(int) * N, hipMemcpyHostToDevice, h->stream));
  HIPCHECK(hipMemcpyAsync(h->pn_koff, koff.data(), sizeof(long long) * (N + 1), hipMemcpyHostToDevice, h->stream));
  HIPCHECK(hipStreamSynchronize(h->stream));  // the host vectors go out of scope
  return TO_OK;
}

template <class M>
int op_pn_launch(to_handle* h, int slot0, int count, hipStream_t stream, const to_solver_opts* opts) {
  if (count <= 0) return TO_OK;
  const int N = h->a.P.N;
  PnArgs q;
  q.a = h->a;
  q.a.P.opts = *opts;
  q.pak = h->pn_pak; q.koff = h->pn_koff; q.list = h->pn_list; q.nbmax = h->pn_nbmax;
  q.ws = h->pn_ws + (size_t)slot0 * (size_t)h->pn_per;
  q.base = slot0;
  q.it_pn = h->a.it_pn; q.cmax_out = h->a.pn_cmax;
  const size_t lds = sizeof(double) * (size_t)pn_lds_doubles<M>(q.nbmax);
  constexpr int nc = M::ne + M::m;
  const int col_blocks = ((N - 1) * nc + 63) / 64, knot_blocks = N;
  for (int round = 0; round <= opts->n_steps + 1; ++round) {
    hipLaunchKernelGGL(k_pn_begin<M>, dim3(count), dim3(64), lds, stream, q, round);
    if (round == opts->n_steps + 1) break;
    hipLaunchKernelGGL(k_pn_lin_col<M>, dim3(count, col_blocks), dim3(64), 0, stream, q);
    hipLaunchKernelGGL(k_pn_lin_knot<M>, dim3(count, knot_blocks), dim3(64), 0, stream, q);
    hipLaunchKernelGGL(k_pn_project<M>, dim3(count), dim3(64), lds, stream, q);
  }
  HIPCHECK(hipGetLastError());
  return TO_OK;
}

template <class M>
int op_defect(to_handle* h, double* out) {
  hipLaunchKernelGGL(k_defect<M>, grid_b(h), dim3(BLOCK), 0, h->stream, h->a, out);
  HIPCHECK(hipGetLastError());
  return TO_OK;
}

template <class M>
static void fill_one(ModelOps& o) { o.pn_prepare = op_pn_prepare<M>; o.pn_launch = op_pn_launch<M>; o.defect = op_defect<M>; }

void fill_ops_pn(ModelOps* t) {
  fill_one<DoubleIntegratorModel<1>>(t[0]);
  fill_one<DoubleIntegratorModel<2>>(t[1]);
  fill_one<DoubleIntegratorModel<3>>(t[2]);
  fill_one<CartpoleModel>(t[3]);
  fill_one<QuadrotorModel>(t[4]);
  fill_one<QuadrotorAttModel<ATT_MRP>>(t[5]);
  fill_one<QuadrotorAttModel<ATT_RP>>(t[6]);
  fill_one<HybridDoubleIntegratorModel>(t[7]);
  fill_one<ModelVectorModel>(t[8]);
  fill_one<InfeasibleModel<DoubleIntegratorModel<1>>>(t[9]);
  fill_one<InfeasibleModel<DoubleIntegratorModel<2>>>(t[10]);
  fill_one<InfeasibleModel<CartpoleModel>>(t[11]);
}
}  // namespace to

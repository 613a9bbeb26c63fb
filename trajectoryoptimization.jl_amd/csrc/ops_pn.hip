// ops_pn.hip — projected-Newton polish (k_pn.h) and the dynamics-defect kernel, every model.
#include <algorithm>
#include <vector>

#include "handle.h"
#include "k_pn.h"

namespace to {

// Polish the trajectories listed in `list` (host array of `count` indices into the batch) from their current nominal (X, U).
// Workspace: one record per knot, sized from the constraint list (pak[k] = candidate rows of knot k); the trajectories are
// processed in chunks that fit the workspace budget (TRAJOPT_PN_WS_GB, default 16 GB of the 288).
template <class M>
int op_pn(to_handle* h, const int* list, int count) {
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
  if (count <= 0) return TO_OK;
  double gb = 16.0;
  if (const char* env = std::getenv("TRAJOPT_PN_WS_GB")) gb = std::max(0.01, std::atof(env));
  const size_t per = (size_t)koff[N] * sizeof(double);
  int chunk = (int)std::min<size_t>((size_t)count, std::max<size_t>(1, (size_t)(gb * 1073741824.0) / per));
  if (h->pn_ws_bytes < per * chunk) {
    if (h->pn_ws) { HIPCHECK(hipStreamSynchronize(h->stream)); HIPCHECK(hipFree(h->pn_ws)); h->pn_ws = nullptr; h->pn_ws_bytes = 0; }
    HIPCHECK(hipMalloc((void**)&h->pn_ws, per * chunk));
    h->pn_ws_bytes = per * chunk;
  }
  if (h->pn_tab_len < N + 1 || h->pn_list_len < count) {
    HIPCHECK(hipStreamSynchronize(h->stream));
    if (h->pn_pak) HIPCHECK(hipFree(h->pn_pak));
    if (h->pn_koff) HIPCHECK(hipFree(h->pn_koff));
    if (h->pn_list) HIPCHECK(hipFree(h->pn_list));
    HIPCHECK(hipMalloc((void**)&h->pn_pak, sizeof(int) * (N + 1)));
    HIPCHECK(hipMalloc((void**)&h->pn_koff, sizeof(long long) * (N + 1)));
    h->pn_list_len = std::max(count, P.Bp);
    HIPCHECK(hipMalloc((void**)&h->pn_list, sizeof(int) * h->pn_list_len));
    h->pn_tab_len = N + 1;
  }
  HIPCHECK(hipMemcpyAsync(h->pn_pak, pak.data(), sizeof(int) * N, hipMemcpyHostToDevice, h->stream));
  HIPCHECK(hipMemcpyAsync(h->pn_koff, koff.data(), sizeof(long long) * (N + 1), hipMemcpyHostToDevice, h->stream));
  HIPCHECK(hipMemcpyAsync(h->pn_list, list, sizeof(int) * count, hipMemcpyHostToDevice, h->stream));
  HIPCHECK(hipStreamSynchronize(h->stream));  // the host vectors go out of scope
  PnArgs q;
  q.a = h->a; q.pak = h->pn_pak; q.koff = h->pn_koff; q.ws = h->pn_ws; q.list = h->pn_list; q.nbmax = nbmax;
  q.it_pn = h->a.it_pn; q.cmax_out = h->a.pn_cmax;
  const size_t lds = sizeof(double) * (size_t)pn_lds_doubles<M>(nbmax);
  constexpr int nc = M::ne + M::m;
  const int col_blocks = ((N - 1) * nc + 63) / 64, knot_blocks = N;
  for (int base = 0; base < count; base += chunk) {
    q.base = base;
    const int cnt = std::min(chunk, count - base);
    for (int round = 0; round <= P.opts.n_steps + 1; ++round) {
      hipLaunchKernelGGL(k_pn_begin<M>, dim3(cnt), dim3(64), lds, h->stream, q, round);
      if (round == P.opts.n_steps + 1) break;
      hipLaunchKernelGGL(k_pn_lin_col<M>, dim3(cnt, col_blocks), dim3(64), 0, h->stream, q);
      hipLaunchKernelGGL(k_pn_lin_knot<M>, dim3(cnt, knot_blocks), dim3(64), 0, h->stream, q);
      hipLaunchKernelGGL(k_pn_project<M>, dim3(cnt), dim3(64), lds, h->stream, q);
    }
    HIPCHECK(hipGetLastError());
  }
  return TO_OK;
}

template <class M>
int op_defect(to_handle* h, double* out) {
  hipLaunchKernelGGL(k_defect<M>, grid_b(h), dim3(BLOCK), 0, h->stream, h->a, out);
  HIPCHECK(hipGetLastError());
  return TO_OK;
}

template <class M>
static void fill_one(ModelOps& o) { o.pn = op_pn<M>; o.defect = op_defect<M>; }

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
}
}  // namespace to

// trajopt_hip.hip — host side of libtrajopt_hip.so: descriptor validation, device storage, kernel
// launches and the C-ABI declared in include/trajopt_hip.h.  gfx950 only; no CPU fallback.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <dlfcn.h>
#include <time.h>

#include <mutex>
#include <thread>

#include "desc_lower.h"
#include "handle.h"
#include "k_generic.h"

using namespace to;

namespace {
thread_local std::string g_err;
}
namespace to {
int fail(int code, const std::string& msg) { g_err = msg; return code; }
}

namespace to {
// ---- guard mode (handle.h): red zones around every handle-owned device array ---------------------------------------------------------
constexpr size_t GUARD_BYTES = 4096;                          // per zone: half a tile row of the batch-fastest layout and then some
constexpr unsigned long long GUARD_WORD = 0xA5C3A5C3DEADBEEFull;
struct GuardZones { unsigned long long* front; unsigned long long* back; };
__global__ void k_guard_fill(unsigned long long* z) { z[blockIdx.x * 64 + threadIdx.x] = GUARD_WORD; }
__global__ void k_guard_check(const GuardZones* tab, int* bad) {
  const GuardZones g = tab[blockIdx.x];
  constexpr int words = (int)(GUARD_BYTES / 8);
  bool hit = false;
  for (int i = threadIdx.x; i < words; i += 64) hit = hit || g.front[i] != GUARD_WORD || g.back[i] != GUARD_WORD;
  if (hit) atomicMin(bad, (int)blockIdx.x);
}
__global__ void k_guard_poke(double* p, long long off) { p[off] = 1.0; }  // TRAJOPT_GUARD_SELFTEST: the overrun the check must find

int g_malloc(to_handle* h, void** p, size_t bytes, const char* name) {
  if (!h->guard) {
    HIPCHECK(hipMalloc(p, bytes));
    return TO_OK;
  }
  const size_t payload = (bytes + 255) / 256 * 256;  // the back zone starts right behind the (rounded-up) payload
  void* base = nullptr;
  HIPCHECK(hipMalloc(&base, payload + 2 * GUARD_BYTES));
  char* q = (char*)base + GUARD_BYTES;
  hipLaunchKernelGGL(k_guard_fill, dim3(GUARD_BYTES / 8 / 64), dim3(64), 0, h->stream, (unsigned long long*)base);
  hipLaunchKernelGGL(k_guard_fill, dim3(GUARD_BYTES / 8 / 64), dim3(64), 0, h->stream, (unsigned long long*)(q + payload));
  if (payload > bytes) HIPCHECK(hipMemsetAsync(q + bytes, 0, payload - bytes, h->stream));
  HIPCHECK(hipGetLastError());
  h->guards.push_back({base, q, payload, name ? name : ""});
  h->guard_dirty = true;
  *p = q;
  return TO_OK;
}
void g_free(to_handle* h, void* p) {
  if (!p) return;
  if (h->guard)
    for (size_t i = 0; i < h->guards.size(); ++i)
      if (h->guards[i].payload == p) {
        hipFree(h->guards[i].base);
        h->guards.erase(h->guards.begin() + i);
        h->guard_dirty = true;
        return;
      }
  hipFree(p);
}
// every red zone of the handle intact?  Synchronises the stream (a debugging mode).
int check_guards(to_handle* h, const char* where) {
  if (!h->guard || h->guards.empty()) return TO_OK;
  if (!h->guard_bad) HIPCHECK(hipMalloc((void**)&h->guard_bad, sizeof(int)));
  if (h->guard_dirty) {
    std::vector<GuardZones> tab;
    for (const auto& g : h->guards) tab.push_back({(unsigned long long*)g.base, (unsigned long long*)((char*)g.payload + g.bytes)});
    HIPCHECK(hipStreamSynchronize(h->stream));
    if (h->guard_tab) HIPCHECK(hipFree(h->guard_tab));
    HIPCHECK(hipMalloc(&h->guard_tab, tab.size() * sizeof(GuardZones)));
    HIPCHECK(hipMemcpy(h->guard_tab, tab.data(), tab.size() * sizeof(GuardZones), hipMemcpyHostToDevice));
    h->guard_dirty = false;
  }
  int bad = 0x7fffffff;
  HIPCHECK(hipMemcpyAsync(h->guard_bad, &bad, sizeof(int), hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(k_guard_check, dim3((unsigned)h->guards.size()), dim3(64), 0, h->stream, (const GuardZones*)h->guard_tab, h->guard_bad);
  HIPCHECK(hipGetLastError());
  HIPCHECK(hipMemcpyAsync(&bad, h->guard_bad, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(hipStreamSynchronize(h->stream));
  if (bad != 0x7fffffff) {
    const auto& g = h->guards[bad];
    return fail(TO_ERR_HIP, "guard: a red zone of device array '" + g.name + "' (" + std::to_string(g.bytes) + " bytes) was overwritten — first seen after " + where);
  }
  return TO_OK;
}
}  // namespace to

namespace {

#define CHECK_H(h) do { if (!(h)) return fail(TO_ERR_NULL, "null handle"); } while (0)
#define CHECK_P(p) do { if (!(p)) return fail(TO_ERR_NULL, "null pointer"); } while (0)
#define CHECK_IDLE(h) do { if ((h)->inflight) return fail(TO_ERR_ARGUMENT, "an asynchronous solve is in flight on this handle (call to_solve_wait first)"); } while (0)

// launch table, one entry per model key; filled once by the ops_*.hip translation units
ModelOps g_ops[N_MODEL_KEYS];
std::once_flag g_ops_once;
const ModelOps* model_ops(int key) {
  std::call_once(g_ops_once, [] {
    fill_ops_small(g_ops); fill_ops_small_forward(g_ops); fill_ops_small_lane(g_ops);
    fill_ops_quad_misc(g_ops); fill_ops_quad_expand(g_ops); fill_ops_quad_backward(g_ops);
    fill_ops_quad_forward_a(g_ops); fill_ops_quad_forward_b(g_ops); fill_ops_quad_forward_c(g_ops);
    fill_ops_quad_forward2_a(g_ops); fill_ops_quad_forward2_b(g_ops); fill_ops_quad_forward2_c(g_ops);
    fill_ops_quadatt_misc(g_ops); fill_ops_quadmrp_expand(g_ops); fill_ops_quadrp_expand(g_ops);
    fill_ops_quadmrp_forward(g_ops); fill_ops_quadrp_forward(g_ops);
    fill_ops_hybrid(g_ops); fill_ops_small_forward2(g_ops); fill_ops_small_scan(g_ops); fill_ops_pn(g_ops); fill_ops_vector(g_ops); fill_ops_infeasible_a(g_ops); fill_ops_infeasible_b(g_ops);
  });
  return (key >= 0 && key < N_MODEL_KEYS) ? &g_ops[key] : nullptr;
}

template <class T>
int dev_alloc_named(to_handle* h, const char* name, T** p, size_t count, bool zero = true) {
  void* q = nullptr;
  TRY(g_malloc(h, &q, std::max<size_t>(count, 1) * sizeof(T), name));
  h->allocs.push_back(q);
  if (zero) HIPCHECK(hipMemsetAsync(q, 0, std::max<size_t>(count, 1) * sizeof(T), h->stream));
  *p = (T*)q;
  return TO_OK;
}
#define dev_alloc(h, p, ...) dev_alloc_named(h, #p, p, __VA_ARGS__)

int ensure_stage(to_handle* h, size_t bytes) {
  if (h->stage_bytes >= bytes) return TO_OK;
  if (h->stage) { HIPCHECK(hipStreamSynchronize(h->stream)); g_free(h, h->stage); h->stage = nullptr; h->stage_bytes = 0; }
  TRY(g_malloc(h, (void**)&h->stage, bytes, "stage"));
  h->stage_bytes = bytes;
  return TO_OK;
}

int use_device(to_handle* h) { HIPCHECK(hipSetDevice(h->device)); return TO_OK; }

// host (cnt per trajectory, column-major (dim, K, B)) -> tiled device array (elements e0..e0+cnt-1 of L) via the staging buffer
int upload_vec(to_handle* h, const double* host, double* d, int cnt, int L = -1, int e0 = 0) {
  if (L < 0) L = cnt;
  const size_t total = (size_t)cnt * h->a.P.B;
  TRY(ensure_stage(h, total * sizeof(double)));
  HIPCHECK(hipMemcpyAsync(h->stage, host, total * sizeof(double), hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(k_to_device, grid_b(h, cnt), dim3(BLOCK), 0, h->stream, h->stage, d, L, e0, cnt, h->a.P.B);
  HIPCHECK(hipGetLastError());
  HIPCHECK(hipStreamSynchronize(h->stream));
  return TO_OK;
}
int download_vec(to_handle* h, double* host, const double* d, int cnt, int L = -1, int e0 = 0) {
  if (L < 0) L = cnt;
  const size_t total = (size_t)cnt * h->a.P.B;
  TRY(ensure_stage(h, total * sizeof(double)));
  hipLaunchKernelGGL(k_to_host, grid_b(h, cnt), dim3(BLOCK), 0, h->stream, d, h->stage, L, e0, cnt, h->a.P.B);
  HIPCHECK(hipGetLastError());
  HIPCHECK(hipMemcpyAsync(host, h->stage, total * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(hipStreamSynchronize(h->stream));
  return TO_OK;
}
// the nominal trajectory (slot 0 of a slotted X/U array) -> host layout or a caller-owned device buffer
int download_nominal(to_handle* h, double* host, const double* slot0, int cnt, void* dev_dst = nullptr) {
  const size_t total = (size_t)cnt * h->a.P.B;
  double* dst = (double*)dev_dst;
  if (!dst) { TRY(ensure_stage(h, total * sizeof(double))); dst = h->stage; }
  hipLaunchKernelGGL(k_to_host, grid_b(h, cnt), dim3(BLOCK), 0, h->stream, slot0, dst, cnt, 0, cnt, h->a.P.B);
  HIPCHECK(hipGetLastError());
  if (host) HIPCHECK(hipMemcpyAsync(host, dst, total * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(hipStreamSynchronize(h->stream));
  return TO_OK;
}
int download_scalar(to_handle* h, double* host, const double* d) {
  if (!host) return TO_OK;
  HIPCHECK(hipMemcpyAsync(host, d, sizeof(double) * h->a.P.B, hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(hipStreamSynchronize(h->stream));
  return TO_OK;
}
int download_int(to_handle* h, int32_t* host, const int* d) {
  if (!host) return TO_OK;
  HIPCHECK(hipMemcpyAsync(host, d, sizeof(int) * h->a.P.B, hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(hipStreamSynchronize(h->stream));
  return TO_OK;
}

// Is the Q-function cost block of every knot block-diagonal (diagonal Qxx + the attitude block of Lie-group models, no
// Qux, any Quu)?  Then the tangent-matrix expansion stores ONE row per knot instead of NR (k_backward.h, compact_row).
bool compact_cost_blocks(const to_handle* h) {
  const int n = h->a.P.n;
  for (const auto& c : h->costs) if (c.kind != TO_COST_DIAGONAL && c.kind != TO_COST_DIAGONAL_QUAT) return false;
  for (const DevCon& c : h->cons) {
    if (c.d.kind == TO_CON_GOAL || c.d.kind == TO_CON_BOUND) continue;  // rows pick single entries of [x;u]: diagonal
    if (c.d.kind == TO_CON_NORM) {  // |z[inds]|: dense over inds — fine when they all sit in the control block
      bool ctrl = true;
      for (int i = 0; i < c.d.n_inds; ++i) ctrl = ctrl && c.d.inds[i] > n;
      if (ctrl) continue;
    }
    return false;
  }
  return true;
}

// Column layout (cooperative backward pass; vector-space models only): is the cost block of every knot DIAGONAL — diagonal
// costs, and constraints whose rows pick single entries of [x; u] (goal, bounds)?  Then lane j keeps one entry per knot.
bool diagonal_cost_blocks(const to_handle* h) {
  for (const auto& c : h->costs) if (c.kind != TO_COST_DIAGONAL) return false;
  for (const DevCon& c : h->cons) if (c.d.kind != TO_CON_GOAL && c.d.kind != TO_CON_BOUND) return false;
  return h->a.P.ne == h->a.P.n;
}

int upload_tables(to_handle* h) {
  {  // kernel-variant flags derived from the tables (refreshed whenever a cost or constraint is replaced)
    DevProblem& P = h->a.P;
    const int N = P.N;
    // simple_stage: one diagonal-kind cost on every stage knot and a uniform dt
    const int k0 = h->costs[h->cost_index[0]].kind;
    bool simple = k0 != TO_COST_QUADRATIC && k0 != TO_COST_ERROR_QUADRATIC;
    for (int k = 1; k < N - 1; ++k) simple = simple && h->cost_index[k] == h->cost_index[0] && h->dt[k] == h->dt[0];
    P.simple_stage = simple ? 1 : 0;
    bool dense = false, generic = false;
    for (const auto& c : h->costs) dense = dense || c.kind == TO_COST_QUADRATIC || c.kind == TO_COST_ERROR_QUADRATIC;
    for (const auto& c : h->cons) generic = generic || !c.selector;
    for (const auto& c : h->cons) generic = generic || c.cp_off >= 0;  // per-trajectory parameters are read by the general variants only
    P.expand_variant = (dense ? 1 : 0) | (h->cons.empty() ? 0 : 2) | (generic ? 4 : 0);
    // unit-SOC forward-pass variants (problem_dev.h unit_soc_desc): at least one control-block constraint, and all of them unit
    bool any_ctrl = false, all_unit = true;
    for (const auto& c : h->cons)
      if (c.fast == 2) {
        any_ctrl = true;
        const bool u = P.m == 1 ? unit_soc_desc<1>(c.d.sense, c.fast, c.p, c.ssgn, c.soff) : P.m == 2 ? unit_soc_desc<2>(c.d.sense, c.fast, c.p, c.ssgn, c.soff)
                     : P.m == 3 ? unit_soc_desc<3>(c.d.sense, c.fast, c.p, c.ssgn, c.soff) : unit_soc_desc<4>(c.d.sense, c.fast, c.p, c.ssgn, c.soff);
        all_unit = all_unit && u;
      }
    P.unit_soc = (any_ctrl && all_unit) ? 1 : 0;
    if (const char* env = std::getenv("TRAJOPT_UNIT_SOC")) if (!std::atoi(env)) P.unit_soc = 0;  // A/B knob
    // (the compact cost block exists for the variants 0 and 2 of the tangent-matrix expansion; the general variant 7 — dense costs, generic
    // constraints, per-trajectory constraint parameters — writes the full block)
    h->a.h_compact = (h->a.bwd_mfma && compact_cost_blocks(h) && (P.expand_variant == 0 || P.expand_variant == 2)) ? 1 : 0;
    h->a.h_diag = (!h->a.bwd_mfma && !h->a.bwd_lane && diagonal_cost_blocks(h)) ? 1 : 0;
    if (const char* env = std::getenv("TRAJOPT_FULL_COST_BLOCKS")) if (std::atoi(env)) { h->a.h_compact = 0; h->a.h_diag = 0; }  // testing knob
  }
  // packed expansion (k_expand.h): whenever the compact cost block is (back) in use, the constant columns of [A B] are in place —
  // another variant of the expansion (general: full cost block) writes them from its dual numbers, equal to rounding only
  if (h->a.Mt && h->a.h_compact && h->expand_pack && h->ops->expand_const) TRY(h->ops->expand_const(h));
  HIPCHECK(hipMemcpyAsync(h->d_costs, h->costs.data(), h->costs.size() * sizeof(to_cost_desc), hipMemcpyHostToDevice, h->stream));
  if (!h->cons.empty()) HIPCHECK(hipMemcpyAsync(h->d_cons, h->cons.data(), h->cons.size() * sizeof(DevCon), hipMemcpyHostToDevice, h->stream));
  HIPCHECK(hipStreamSynchronize(h->stream));
  return TO_OK;
}

int launch_set_active(to_handle* h, int v, int clear_bpfail = 1) {
  hipLaunchKernelGGL(k_set_active, grid_b(h), dim3(BLOCK), 0, h->stream, h->a, v, clear_bpfail);
  HIPCHECK(hipGetLastError());
  return TO_OK;
}
int launch_rollout(to_handle* h) { return h->ops->rollout(h); }
int launch_cost(to_handle* h, int with_al, double* out, double* Jk) { return h->ops->cost(h, with_al, out, Jk); }
int launch_expand(to_handle* h) { return h->ops->expand(h); }
int launch_backward(to_handle* h) { return h->ops->backward(h); }
int launch_accept(to_handle* h) {  // materialise accepted candidate slots on slot 0, then forget them
  if (!h->a.store_x) return h->ops->accept_roll(h);  // their states were not stored: rolled out again from the stored controls (clears acc itself)
  hipLaunchKernelGGL(k_accept, grid_b(h, 1, h->accept_chunks), dim3(BLOCK), 0, h->stream, h->a);
  hipLaunchKernelGGL(k_clear_acc, grid_b(h), dim3(BLOCK), 0, h->stream, h->a);
  HIPCHECK(hipGetLastError());
  return TO_OK;
}
// forward pass: ONE launch runs the whole line search (CW step sizes per round, concurrently, inside each wave) and the
// per-trajectory state machine (k_forward.h).  Kernel variants: bit0 simple stage cost, bit1 constraints, bit2
// compile-time RK4 (models that pin it), bit3 dense costs / generic constraints.
int launch_forward(to_handle* h, bool accept = true, bool two_wave = false) {
  const KArgs& a = h->a;
  // (bit3 also with per-trajectory linear cost terms, DevProblem::gl: only the general variants read them)
  int mode = (a.P.simple_stage ? 1 : 0) | (a.P.n_cons > 0 ? 2 : 0) | (a.P.integrator == INTEG_RK4 ? 4 : 0) | (((a.P.expand_variant & 5) || a.P.gl || a.P.cp) ? 8 : 0);
  if (!h->ops->forward[mode]) mode &= ~4;  // the model does not pin RK4
  if (a.P.unit_soc && h->ops->forward[mode | 16]) mode |= 16;
  if (!h->ops->forward[mode]) mode = (mode | 8) & ~1 & ~16;  // the general variant (any cost kind, stage cost read per knot): a superset
  if (!h->ops->forward[mode]) return fail(TO_ERR_UNSUPPORTED, "forward-pass variant not compiled for this model");
  if ((two_wave || h->fwd2 == 1) && h->ops->forward2[mode]) TRY(h->ops->forward2[mode](h));
  else TRY(h->ops->forward[mode](h));
  if (accept) TRY(launch_accept(h));  // inside a solve the next expansion writes the accepted step through instead
  return TO_OK;
}
int launch_outer(to_handle* h) { return h->ops->outer(h); }
int launch_violation(to_handle* h, double* out) { return h->ops->violation(h, out); }

// stats of the last solve; with_defect: c_max also counts the dynamics / initial-condition defects (a polished trajectory is
// not an exact rollout)
int fill_stats(to_handle* h, to_solve_stats* st, bool with_defect) {
  KArgs& a = h->a;
  const DevProblem& P = a.P;
  const int B = P.B;
  std::vector<int32_t> its(B);
  TRY(download_int(h, its.data(), a.iterations));
  int64_t tot = 0;
  for (int b = 0; b < B; ++b) tot += its[b];
  if (st->iterations) std::memcpy(st->iterations, its.data(), sizeof(int32_t) * B);
  TRY(download_int(h, st->iterations_outer, a.outer));
  TRY(download_int(h, st->status, a.status));
  TRY(download_int(h, st->iterations_pn, a.it_pn));
  if (st->cost) { TRY(launch_cost(h, 0, h->d_tmp, nullptr)); TRY(download_scalar(h, st->cost, h->d_tmp)); }
  TRY(download_scalar(h, st->dJ, a.dJ));
  TRY(download_scalar(h, st->gradient, a.grad));
  if (st->c_max) {
    if (P.n_cons > 0) { TRY(launch_violation(h, h->d_tmp)); TRY(download_scalar(h, st->c_max, h->d_tmp)); }
    else std::memset(st->c_max, 0, sizeof(double) * B);
    if (with_defect) {
      std::vector<double> df(B);
      TRY(h->ops->defect(h, h->d_tmp));
      TRY(download_scalar(h, df.data(), h->d_tmp));
      for (int b = 0; b < B; ++b) if (df[b] > st->c_max[b] || df[b] != df[b]) st->c_max[b] = df[b];
    }
  }
  if (st->penalty_max) {
    hipLaunchKernelGGL(k_penalty_max, grid_b(h), dim3(BLOCK), 0, h->stream, a, h->d_tmp);
    HIPCHECK(hipGetLastError());
    TRY(download_scalar(h, st->penalty_max, h->d_tmp));
  }
  st->total_iterations = tot;
  st->batch_steps = h->last_steps;
  st->solve_ms = h->last_ms;
  return TO_OK;
}

// ---- repacked working set (k_generic.h k_repack_*) ------------------------------------------------------------------------
void*& rp_field(to_handle* h, size_t off) { return *reinterpret_cast<void**>(reinterpret_cast<char*>(&h->a) + off); }
int rp_setup(to_handle* h) {
  if (!h->rp_arr.empty()) return TO_OK;
  KArgs& a = h->a;
  const DevProblem& P = a.P;
  auto add = [&](void* field_addr, int kind, int L) {
    h->rp_arr.push_back({(size_t)(reinterpret_cast<char*>(field_addr) - reinterpret_cast<char*>(&a)), kind, L});
  };
  add(&a.Xs, 0, P.N * P.n); add(&a.Us, 0, (P.N - 1) * P.m); add(&a.x0, 0, P.n);
  if (a.P.gl) add(&a.P.gl, 0, P.n_costs * (P.n + P.m));
  // duals and penalties: an iLQR solve (the only kind that repacks) never writes them, but its expansion, forward pass and cost read
  // them through the tile of the WORKING position whenever the problem has constraints (a hand-built AL loop, to_set_duals, an iLQR
  // re-solve behind an AL solve: per-trajectory values) — moved along, never copied home
  if (P.n_cons > 0) { add(&a.lam, 3, (int)P.n_duals); add(&a.mu, 3, P.n_cons); }
  if (a.P.cp) add(&a.P.cp, 3, P.n_cp);
  for (double** f : {&a.J, &a.dJ, &a.grad, &a.rho, &a.drho, &a.cmax}) add(f, 1, 1);
  for (int** f : {&a.status, &a.iterations, &a.it_inner, &a.outer, &a.dJzero, &a.ls_index, &a.active, &a.budget, &a.bpfail, &a.acc, &a.accp}) add(f, 2, 1);
  if ((int)h->rp_arr.size() > RP_MAX) return fail(TO_ERR_UNSUPPORTED, "repack table too long");
  return TO_OK;
}
size_t rp_bytes(const to_handle::RpArr& r, int Bp) { return (r.kind == 0 || r.kind == 3) ? sizeof(double) * (size_t)r.L * Bp : (r.kind == 1 ? sizeof(double) : sizeof(int)) * (size_t)Bp; }
// move the `count` active trajectories (list of the NEXT step, built by k_compact) into the other working set and go on there
int rp_move(to_handle* h, int count) {
  KArgs& a = h->a;
  TRY(rp_setup(h));
  const int lvl = h->rp_level, w = lvl & 1;          // target working set: 0, 1, 0, ...
  const int Bp_new = (count + 63) / 64 * 64;
  if (lvl == 0) {
    h->rp_home.clear();
    for (const auto& r : h->rp_arr) h->rp_home.push_back(rp_field(h, r.off));
    h->rp_B = a.P.B; h->rp_Bp = a.P.Bp;
  }
  if (h->rp_cap[w] < Bp_new || h->rp_work[w].size() != h->rp_arr.size()) {  // (sized once per handle: the first move of a solve is the largest for this working set)
    for (void* q : h->rp_work[w]) if (q) g_free(h, q);
    h->rp_work[w].assign(h->rp_arr.size(), nullptr);
    if (h->rp_map[w]) g_free(h, h->rp_map[w]);
    h->rp_map[w] = nullptr; h->rp_cap[w] = 0;
    // (+ one spare tile, like the home arrays: k_accept_roll's lanes without an accepted step store into the tile behind the batch)
    bool ok = true;
    for (size_t i = 0; ok && i < h->rp_arr.size(); ++i) ok = g_malloc(h, &h->rp_work[w][i], rp_bytes(h->rp_arr[i], Bp_new + 64), "repacked working set") == TO_OK;
    ok = ok && g_malloc(h, (void**)&h->rp_map[w], sizeof(int) * Bp_new, "repack map") == TO_OK;
    if (!ok) {  // no memory for a working set: the repack is an optimisation — the solve goes on where it is (return value 1: declined)
      (void)hipGetLastError();
      for (void*& q : h->rp_work[w]) { if (q) g_free(h, q); q = nullptr; }
      if (h->rp_map[w]) { g_free(h, h->rp_map[w]); h->rp_map[w] = nullptr; }
      return 1;
    }
    h->rp_cap[w] = Bp_new;
  }
  RpArgs mv, hm;
  mv.n = hm.n = (int)h->rp_arr.size();
  for (int i = 0; i < mv.n; ++i) {
    mv.kind[i] = hm.kind[i] = h->rp_arr[i].kind; mv.L[i] = hm.L[i] = h->rp_arr[i].L;
    mv.src[i] = rp_field(h, h->rp_arr[i].off); mv.dst[i] = h->rp_work[w][i];
    hm.src[i] = rp_field(h, h->rp_arr[i].off); hm.dst[i] = h->rp_home[i];
    HIPCHECK(hipMemsetAsync(h->rp_work[w][i], 0, rp_bytes(h->rp_arr[i], Bp_new), h->stream));  // the padding lanes hold valid (zero) data, active = 0
  }
  const int* omap_old = lvl == 0 ? nullptr : h->rp_map[(lvl - 1) & 1];
  const int next = (a.step + 1) & 1;
  const int* list = a.alist + (size_t)next * a.P.Bp;
  const bool dbg = std::getenv("TRAJOPT_SYNC_DEBUG") != nullptr;
  auto dbg_sync = [&](const char* what) {
    if (!dbg) return;
    const hipError_t e = hipStreamSynchronize(h->stream);
    std::fprintf(stderr, "[sync-debug] rp_move level %d -> %d count %d (B %d Bp %d, cap %d / %d) after %s: %s\n", lvl, lvl + 1, count, a.P.B, a.P.Bp, h->rp_cap[0], h->rp_cap[1], what, hipGetErrorString(e));
  };
  dbg_sync("memsets");
  if (lvl > 0)  // what has finished in the set we leave goes home first (level 0 IS home)
    hipLaunchKernelGGL(k_repack_home, dim3((a.P.B + 63) / 64, hm.n), dim3(64), 0, h->stream, hm, a.active, a.P.B, omap_old, 0);
  dbg_sync("k_repack_home");
  hipLaunchKernelGGL(k_repack_move, dim3((count + 63) / 64, mv.n), dim3(64), 0, h->stream, mv, list, count, omap_old, h->rp_map[w]);
  HIPCHECK(hipGetLastError());
  dbg_sync("k_repack_move");
  for (int i = 0; i < mv.n; ++i) rp_field(h, h->rp_arr[i].off) = h->rp_work[w][i];
  a.P.B = count; a.P.Bp = Bp_new;
  hipLaunchKernelGGL(k_repack_list, dim3((count + 255) / 256), dim3(256), 0, h->stream, a.alist + (size_t)next * a.P.Bp, a.acount + next, count);
  HIPCHECK(hipGetLastError());
  h->rp_level = lvl + 1;
  return TO_OK;
}
// end of a solve: everything in the working set goes home; the handle works on its home arrays again
int rp_finish(to_handle* h, bool copy) {
  if (h->rp_level == 0) return TO_OK;
  KArgs& a = h->a;
  const int w = (h->rp_level - 1) & 1;
  int rc = TO_OK;
  if (copy) {
    RpArgs hm;
    hm.n = (int)h->rp_arr.size();
    for (int i = 0; i < hm.n; ++i) { hm.kind[i] = h->rp_arr[i].kind; hm.L[i] = h->rp_arr[i].L; hm.src[i] = rp_field(h, h->rp_arr[i].off); hm.dst[i] = h->rp_home[i]; }
    hipLaunchKernelGGL(k_repack_home, dim3((a.P.B + 63) / 64, hm.n), dim3(64), 0, h->stream, hm, a.active, a.P.B, h->rp_map[w], 1);
    if (hipGetLastError() != hipSuccess) rc = fail(TO_ERR_HIP, "k_repack_home launch failed");
  }
  for (size_t i = 0; i < h->rp_arr.size(); ++i) rp_field(h, h->rp_arr[i].off) = h->rp_home[i];
  a.P.B = h->rp_B; a.P.Bp = h->rp_Bp;
  h->rp_level = 0;
  return rc;
}

// what the solve loop knows about its batch, published for to_solve_progress / to_solve_wait_below
// TRAJOPT_TRACE=<file>: every progress report of every solve loop as one line "<handle> <seconds> <batch steps> <active> <tag>" (host clock,
// CLOCK_MONOTONIC) — the timeline of pipelined solves over several handles (tools/ab/pipeline_timeline.py)
void trace_line(to_handle* h, int active, int steps, const char* tag) {
  static const char* path = std::getenv("TRAJOPT_TRACE");
  if (!path) return;
  static std::mutex mu;
  static FILE* f = std::fopen(path, "a");
  if (!f) return;
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  std::lock_guard<std::mutex> lk(mu);
  std::fprintf(f, "%p %.6f %d %d %s\n", (void*)h, ts.tv_sec + 1e-9 * ts.tv_nsec, steps, active, tag);
  std::fflush(f);
}
void publish_progress(to_handle* h, int active, int steps) {
  trace_line(h, active, steps, "step");
  const int before = h->prog_active.exchange(active);
  h->prog_steps = steps;
  if (active < before) { std::lock_guard<std::mutex> lk(h->prog_mu); h->prog_cv.notify_all(); }
}
int solve_impl(to_handle* h, to_solve_stats* st, int al_mode);
int solve(to_handle* h, to_solve_stats* st, int al_mode) {
  int rc = solve_impl(h, st, al_mode);
  if (rc == TO_OK && h->guard) rc = check_guards(h, "the end of a solve (final accept, working set home, statistics)");
  publish_progress(h, 0, h->prog_steps);  // on every exit path: nobody may wait for a count that will not come
  rp_finish(h, false);   // (an error path may leave the handle on a working set: back to the home arrays, without the copy)
  h->a.control = 0;  // on every exit path: the phase API must never find the state machine armed
  h->a.compact = 0;  // ... nor take its trajectories from a solve's active list
  h->a.CW = h->cw_base; h->a.TW = h->tw_base;
  h->a.store_x = 1;      // ... and stores whole candidates
  return rc;
}
// Altro solve!(::ProjectedNewtonSolver) on the trajectories of `list` (k_pn.h), with the options `opts`, on the handle's stream;
// device time is added to h->last_ms.  The list goes through the workspace in chunks of h->pn_cap trajectories.
int pn_run(to_handle* h, const std::vector<int>& list, const to_solver_opts& opts) {
  if (list.empty()) return TO_OK;
  if (!h->ops->pn_launch) return fail(TO_ERR_UNSUPPORTED, "projected Newton not compiled for this model");
  const int count = (int)list.size();
  TRY(h->ops->pn_prepare(h, count));
  hipEvent_t e0 = h->sev[0], e1 = h->sev[1];
  HIPCHECK(hipEventRecord(e0, h->stream));
  std::memcpy(h->pn_list_host, list.data(), sizeof(int) * count);
  for (int base = 0; base < count; base += h->pn_cap) {
    const int cnt = std::min(h->pn_cap, count - base);
    HIPCHECK(hipMemcpyAsync(h->pn_list, h->pn_list_host + base, sizeof(int) * cnt, hipMemcpyHostToDevice, h->stream));
    TRY(h->ops->pn_launch(h, 0, cnt, h->stream, &opts));
  }
  HIPCHECK(hipEventRecord(e1, h->stream));
  HIPCHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
  h->last_ms += ms;
  if (h->profile) { h->prof_ms[3] += ms; h->prof_launches[3] += 1; }  // slot 3: the polish (all its launches of one solve)
  if (h->guard) TRY(check_guards(h, "the projected-Newton polish"));
  return TO_OK;
}
int ensure_solve_events(to_handle* h) {
  if (!h->sev[0]) {
    HIPCHECK(hipEventCreate(&h->sev[0]));
    HIPCHECK(hipEventCreate(&h->sev[1]));
    HIPCHECK(hipEventCreateWithFlags(&h->sev[2], hipEventDisableTiming));
    HIPCHECK(hipEventCreateWithFlags(&h->sev[3], hipEventDisableTiming));
  }
  return TO_OK;
}

// ---- early polish: inside an ALTRO solve, trajectories whose AL stage has ended are polished on a second stream while the
// rest of the batch is still iterating.  The last third of an AL solve's batch steps works on a drained batch (C5: 85 of 275
// steps hold under a quarter of the trajectories: latency-bound kernels on a mostly idle chip), and the polish of a trajectory
// depends on nothing but its own (X, U): the same result, partly hidden behind that tail.  When the active count has fallen
// to B / 4 the AL loop takes a snapshot of the per-trajectory state, the host lists the trajectories that are finished,
// converged and above constraint_tolerance, and their polish goes to a low-priority stream, into workspace slots of its own; the
// rest is polished after the AL stage.  Finished trajectories are never touched again by the AL kernels; the polish writes
// (X, U), status and its own statistics of ITS trajectories only — results are equal to the one-polish path
// (tests/test_gpu_pn.py::test_early_polish_matches_polish_after_the_al_stage).
// Measured on C5 (B = 8192, 3 solves each, M trajectory-iterations/s): no hand-over 1.060; one at B/2 1.072, at B/4 1.088-1.094, at
// B/8 1.078, at B/16 1.067, at B/32 1.074; two from B/8 1.043, three from B/2 1.032, five 0.976 — the polish waves live for
// milliseconds and k_expand needs whole SIMDs (256 VGPR + AGPR), so a polish that starts while the AL stage still fills
// the chip costs it more than it hides; restricting the polish stream to half or a quarter of the compute units
// (hipExtStreamCreateWithCUMask) changed nothing (1.092-1.100).  TRAJOPT_PN_EARLY=0 switches the hand-over off, =n allows n of them
// (halving the threshold each time), TRAJOPT_PN_EARLY_AT=d starts at B / d.
int early_polish_setup(to_handle* h) {
  if (!h->pn_stream) {
    int lo = 0, hi = 0;
    HIPCHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));  // lo: the numerically largest = lowest priority
    HIPCHECK(hipStreamCreateWithPriority(&h->pn_stream, hipStreamNonBlocking, lo));
    HIPCHECK(hipEventCreateWithFlags(&h->pn_ev[0], hipEventDisableTiming));
    HIPCHECK(hipEventCreate(&h->pn_ev[1]));
    HIPCHECK(hipEventCreate(&h->pn_ev[2]));
    const size_t Bp = h->a.P.Bp;
    HIPCHECK(hipHostMalloc((void**)&h->snap_status, sizeof(int32_t) * Bp));
    HIPCHECK(hipHostMalloc((void**)&h->snap_active, sizeof(int32_t) * Bp));
    HIPCHECK(hipHostMalloc((void**)&h->snap_cmax, sizeof(double) * Bp));
  }
  return TO_OK;
}
// AL loop: the state of every trajectory as of this point of the handle's stream -> pinned host arrays; pn_ev[0] marks it
int early_polish_snapshot(to_handle* h) {
  const KArgs& a = h->a;
  const size_t B = a.P.B;
  HIPCHECK(hipMemcpyAsync(h->snap_status, a.status, sizeof(int32_t) * B, hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(hipMemcpyAsync(h->snap_active, a.active, sizeof(int32_t) * B, hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(hipMemcpyAsync(h->snap_cmax, a.cmax, sizeof(double) * B, hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(hipEventRecord(h->pn_ev[0], h->stream));
  return TO_OK;
}
// ... and, once it has arrived, the polish of what it shows finished (enqueued on pn_stream behind the snapshot)
int early_polish_launch(to_handle* h) {
  HIPCHECK(hipEventSynchronize(h->pn_ev[0]));
  const int B = h->a.P.B, s0 = h->pn_early_slots;
  int cnt = 0;
  for (int b = 0; b < B; ++b)
    if (!h->pn_done_early[b] && h->snap_active[b] == 0 && h->snap_status[b] == TO_SOLVE_SUCCEEDED && h->snap_cmax[b] > h->pn_opts.constraint_tolerance) {
      h->pn_list_host[s0 + cnt++] = b;
      h->pn_done_early[b] = 1;
    }
  if (cnt == 0) return TO_OK;
  HIPCHECK(hipStreamWaitEvent(h->pn_stream, h->pn_ev[0], 0));
  if (s0 == 0) HIPCHECK(hipEventRecord(h->pn_ev[2], h->pn_stream));
  HIPCHECK(hipMemcpyAsync(h->pn_list + s0, h->pn_list_host + s0, sizeof(int) * cnt, hipMemcpyHostToDevice, h->pn_stream));
  TRY(h->ops->pn_launch(h, s0, cnt, h->pn_stream, &h->pn_opts));
  HIPCHECK(hipEventRecord(h->pn_ev[1], h->pn_stream));
  h->pn_early_slots += cnt;
  return TO_OK;
}
int pn_solve(to_handle* h, to_solve_stats* st) {
  TRY(use_device(h));
  TRY(ensure_solve_events(h));
  KArgs& a = h->a;
  const int Bp = a.P.Bp, B = a.P.B;
  HIPCHECK(hipMemsetAsync(a.iterations, 0, sizeof(int) * Bp, h->stream));
  HIPCHECK(hipMemsetAsync(a.outer, 0, sizeof(int) * Bp, h->stream));
  HIPCHECK(hipMemsetAsync(a.it_pn, 0, sizeof(int) * Bp, h->stream));
  std::vector<int> list(B);
  for (int b = 0; b < B; ++b) list[b] = b;
  h->last_steps = 0; h->last_ms = 0.0;
  TRY(pn_run(h, list, a.P.opts));
  if (st) TRY(fill_stats(h, st, true));
  return TO_OK;
}
// Altro solve!(::ALTROSolver): AL stage down to projected_newton_tolerance, polish of what it left SOLVE_SUCCEEDED above
// constraint_tolerance — as the trajectories finish (early polish, above) and, for the rest, after the AL stage
int altro_solve(to_handle* h, to_solve_stats* st) {
  const to_solver_opts user = h->a.P.opts;
  const bool pn = user.projected_newton && h->a.P.n_cons > 0;
  if (!pn) return solve(h, st, 1);
  TRY(use_device(h));
  const int B = h->a.P.B;
  int early = 1;
  if (const char* env = std::getenv("TRAJOPT_PN_EARLY")) early = std::max(0, std::min(8, std::atoi(env)));
  h->pn_early = 0; h->pn_early_slots = 0;
  if (early > 0 && h->ops->pn_launch && !h->ops->write_through) {  // (write-through models keep accepted steps in candidate slots until the solve ends)
    // (a problem the polish cannot take — too many rows on a knot, no memory for the workspace — still gets its AL stage)
    if (h->ops->pn_prepare(h, B) == TO_OK && h->pn_cap >= B) {  // every trajectory has a workspace slot of its own
      TRY(early_polish_setup(h));
      h->pn_done_early.assign(B, 0);
      h->pn_opts = user;
      h->pn_early = early;
    }
  }
  const bool had_early = h->pn_early > 0;
  h->a.P.opts.constraint_tolerance = user.projected_newton_tolerance;
  const int rc = solve(h, nullptr, 1);
  h->a.P.opts = user;
  h->pn_early = 0;
  if (had_early && h->pn_early_slots > 0) {  // on every path: nothing of this solve may still be running when it returns
    const hipError_t e = hipEventSynchronize(h->pn_ev[1]);
    if (rc == TO_OK) HIPCHECK(e);
  }
  if (rc != TO_OK) return rc;
  if (had_early && h->pn_early_slots > 0) {
    float ms = 0.f, tail = 0.f;
    HIPCHECK(hipEventElapsedTime(&ms, h->pn_ev[2], h->pn_ev[1]));  // first launch .. end of the last: the span on the second stream
    if (h->profile) { h->prof_ms[3] += ms; h->prof_launches[3] += 1; }
    HIPCHECK(hipEventElapsedTime(&tail, h->sev[1], h->pn_ev[1]));   // what of it outlasted the AL stage
    if (tail > 0.f) h->last_ms += tail;
  }
  std::vector<int32_t> status(B);
  std::vector<double> cmax(B);
  TRY(download_int(h, status.data(), h->a.status));
  TRY(download_scalar(h, cmax.data(), h->a.cmax));
  std::vector<int> list;
  for (int b = 0; b < B; ++b)
    if (status[b] == TO_SOLVE_SUCCEEDED && cmax[b] > user.constraint_tolerance && !(had_early && h->pn_done_early[b])) list.push_back(b);
  // A problem outside the polish's limits (PN_NB_LIMIT rows on one knot) keeps the result of its AL stage: the call succeeds, the
  // trajectories keep their AL status / violation and to_last_error() says why nothing was polished
  trace_line(h, (int)list.size(), h->last_steps, "polish");
  const int prc = pn_run(h, list, user);
  trace_line(h, 0, h->last_steps, "polish_done");
  if (prc != TO_OK && prc != TO_ERR_UNSUPPORTED) return prc;
  if (st) { const std::string note = prc == TO_OK ? std::string() : g_err; TRY(fill_stats(h, st, true)); if (!note.empty()) g_err = "polish skipped: " + note; }
  return TO_OK;
}

int solve_impl(to_handle* h, to_solve_stats* st, int al_mode) {
  TRY(use_device(h));
  KArgs& a = h->a;
  const DevProblem& P = a.P;
  a.al_mode = al_mode;
  a.control = 1;
  h->rp_arr.clear();  // the table of carried arrays is rebuilt per solve (per-trajectory cost terms may have appeared)
  a.compact = h->compact;
  const int max_steps = (al_mode ? P.opts.iterations_total : P.opts.iterations) + 1;
  if (h->counter_len < max_steps) {
    int* c = nullptr;
    TRY(g_malloc(h, (void**)&c, sizeof(int) * max_steps, "step counters"));
    h->allocs.push_back(c);
    a.counter = c;
    if (h->counter_host) HIPCHECK(hipHostFree(h->counter_host));
    HIPCHECK(hipHostMalloc((void**)&h->counter_host, sizeof(int) * max_steps));
    h->counter_len = max_steps;
  }
  HIPCHECK(hipMemsetAsync(a.counter, 0, sizeof(int) * max_steps, h->stream));
  TRY(ensure_solve_events(h));
  const hipEvent_t e0 = h->sev[0], e1 = h->sev[1];
  HIPCHECK(hipEventRecord(e0, h->stream));
  HIPCHECK(hipMemsetAsync(a.it_pn, 0, sizeof(int) * P.Bp, h->stream));
  hipLaunchKernelGGL(k_solve_init, grid_b(h), dim3(BLOCK), 0, h->stream, a, al_mode);
  HIPCHECK(hipGetLastError());
  TRY(launch_rollout(h));
  TRY(launch_cost(h, 1, a.J, nullptr));
  if (a.compact) {  // the initial rollout may have ended trajectories (TO_STATE_LIMIT / TO_CONTROL_LIMIT): the list of step 0 without them
    a.step = -1;
    if (P.Bp <= 16384) hipLaunchKernelGGL(k_compact, dim3(1), dim3(1024), 0, h->stream, a);
    else {
      int per = ((P.Bp + 255) / 256 + 1023) / 1024 * 1024;
      if (per > 65536) per = 65536;
      const int nb = (P.Bp + per - 1) / per;
      if (nb > 256) return fail(TO_ERR_UNSUPPORTED, "batch too large for the compaction kernels (16 777 216 trajectories)");
      hipLaunchKernelGGL(k_compact_count, dim3(nb), dim3(1024), 0, h->stream, a, per);
      hipLaunchKernelGGL(k_compact_write, dim3(nb), dim3(1024), 0, h->stream, a, per);
    }
    HIPCHECK(hipGetLastError());
  }
  int steps = 0;
  if (h->profile) {
    while ((int)h->ev.size() < 4 * max_steps) { hipEvent_t e; HIPCHECK(hipEventCreate(&e)); h->ev.push_back(e); }
  }
  // The host only needs to know WHEN every trajectory has finished: it enqueues CHECK_EVERY batch steps back to back
  // and reads the per-step "still active" counters once per chunk.  The read-back is pipelined: chunk g+1 is already
  // enqueued when the host waits for the counters of chunk g, so the GPU never idles on the round trip; the price is at
  // most one chunk of launches whose kernels find nothing to do (kernels of finished trajectories exit at once).
  constexpr int CHECK_EVERY = 4;
  const hipEvent_t cev[2] = {h->sev[2], h->sev[3]};
  int launched = 0, checked = 0, nchunks = 0, last_active = P.B;
  bool done = false;
  h->prog_active = P.B; h->prog_steps = 0;
  const bool dbg_sync = std::getenv("TRAJOPT_SYNC_DEBUG") != nullptr;
  auto enqueue_chunk = [&]() -> int {
    const int chunk = std::min(CHECK_EVERY, max_steps - launched);
    for (int c = 0; c < chunk; ++c) {
      const int step = launched + c;
      a.step = step;
      if (h->profile) HIPCHECK(hipEventRecord(h->ev[4 * step + 0], h->stream));
      const bool fcoop = h->fused_coop && a.h_diag && (P.expand_variant == 0 || P.expand_variant == 2);
      if (!h->fused_lane && !fcoop) TRY(launch_expand(h));
      if (h->profile) HIPCHECK(hipEventRecord(h->ev[4 * step + 1], h->stream));
      if (h->fused_lane) TRY(h->ops->expand_backward(h));  // expansion in the registers of the lane that runs the recursion
      else if (fcoop && h->scan && P.expand_variant == 0 && last_active <= h->scan_max_active)
        TRY(h->ops->expand_backward_scan(h));  // unconstrained, diagonal cost blocks: the recursion as a scan over the horizon (k_scan.h)
      else if (fcoop) TRY(h->ops->expand_backward_coop(h));  // expansion by a second wave of the workgroup, through an LDS ring
      else TRY(launch_backward(h));
      if (h->profile) HIPCHECK(hipEventRecord(h->ev[4 * step + 2], h->stream));
      // forward-wave shape of this step, from the last active count the host has seen (results do not depend on it)
      const bool deep = h->cw_deep && last_active <= h->deep_max_active;
      a.CW = deep ? h->cw_deep : h->cw_base; a.TW = deep ? h->tw_deep : h->tw_base;
      // ... and its workgroup shape: two waves per candidate group (roller + accountant, k_forward2) shorten the rollout's latency
      // chain by a third, but need twice the wave slots — taken once both waves of every workgroup get a SIMD of their own
      // (C3: 610 vs 812 us per step with the chip full, 480 vs 320 us once the batch has drained)
      const bool two = h->fwd2 == 2 && (long long)2 * ((last_active + a.TW - 1) / a.TW) <= (long long)h->simds;
      // ... and what it stores per candidate: with the chip full the pass is bound by its stores, 3/4 of them candidate states
      // that are read once (the accepted one) or never — from roll_min active trajectories on only the controls go out and the accepted
      // candidates are rolled out again (k_accept_roll: bit-identical states, one more latency chain of N-1 steps)
      // (measured, always vs never, whole solve: C5 +0.9 / +4.6 / +7.7 / +7.3 % at B = 2048 / 4096 / 8192 / 16384, C3 -3 / -1 / +1.9 / +4.4 %:
      // the copy by k_accept grows with the accepted trajectories — 93 us at 4096, 227 us at 8192 — the second rollout does not,
      // and an AL line search goes through more rounds, each of which stores its candidates, than an unconstrained one)
      // Small models (write-through; 4 step sizes x 16 trajectories per wave): at the large-batch plateau the forward pass wrote 19 KB per
      // active trajectory for 4 KB of result and the next expansion gathered the accepted candidate through 4x-amplified sectors;
      // with the controls only and the re-roll both kernels stream the nominal (TRAJOPT_ACCEPT_ROLL_MIN overrides every default)
      // (round 6: 2048 for iLQR solves too — alone, C3 at B = 4096 runs 1.12 M it/s with either threshold, and next to other solves on
      // the device (pipelined handles) the candidate-state stores and k_accept's copy cost the others bandwidth: 1.57 -> 1.70 M it/s over three
      // handles; the forward phase's counter traffic drops with it)
      const int roll_min = h->roll_min_active >= 0 ? h->roll_min_active : h->ops->write_through ? h->roll_min_small : 2048;
      // ... and, for those models, only while the batch is still DENSE: the active list is in index order, so once half of the batch has
      // converged a wave's 64 trajectories sit in several tiles and every store of the re-roll becomes scattered 8-byte writes (r05 trace,
      // Cartpole at B = 1 048 576: the re-roll takes 0.9 ms with every trajectory active and 1.8 ms with a quarter of them); the
      // write-through of the next expansion makes the same scattered stores, but behind 2 000 instructions per knot
      const bool dense = !h->ops->write_through || (double)last_active >= h->roll_min_frac * (double)P.B;
      a.store_x = (roll_min > 0 && h->ops->accept_roll && !two && last_active >= roll_min && dense) ? 0 : 1;
      if (!a.store_x && h->ls2_cwa && a.compact && h->fwd2 != 1) {
        // two-launch line search (common.h ls_phase): launch A — one round for everybody; flags -> list; launch B — the rest of the
        // search for the flagged trajectories only; then the accept.  Same candidates, same first accepted step size: bit-identical.
        const int cw0 = a.CW, tw0 = a.TW, dump0 = a.dump_wave;
        a.dump_wave = h->ls2_dump;
        a.CW = h->ls2_cwa; a.TW = 64 / h->ls2_cwa; a.ls_phase = 1; a.blk0 = 0;
        TRY(launch_forward(h, false, false));
        {
          int per = ((P.Bp + 255) / 256 + 1023) / 1024 * 1024;
          if (per > 65536) per = 65536;
          const int nb = (P.Bp + per - 1) / per;
          if (nb > 256) return fail(TO_ERR_UNSUPPORTED, "batch too large for the compaction kernels (16 777 216 trajectories)");
          hipLaunchKernelGGL(k_flags_count, dim3(nb), dim3(1024), 0, h->stream, a.pending, P.Bp, per, a.ccount);
          hipLaunchKernelGGL(k_flags_write, dim3(nb), dim3(1024), 0, h->stream, a.pending, P.Bp, per, a.ccount, a.plist, a.pcount);
          HIPCHECK(hipGetLastError());
        }
        a.CW = h->ls2_cwb; a.TW = 64 / h->ls2_cwb; a.ls_phase = 2; a.ls_c0 = h->ls2_cwa; a.blk0 = h->ls2_blkA;
        TRY(launch_forward(h, false, false));
        a.ls_phase = 0; a.blk0 = 0; a.CW = cw0; a.TW = tw0; a.dump_wave = dump0;
        TRY(launch_accept(h));
      } else
      TRY(launch_forward(h, !h->ops->write_through || !a.store_x, two));
      a.store_x = 1;
      if (al_mode) TRY(launch_outer(h));
      if (a.compact) {  // the list of the trajectories that go on, for the next step's kernels
        if (P.Bp <= 16384) hipLaunchKernelGGL(k_compact, dim3(1), dim3(1024), 0, h->stream, a);
        else {  // two launches of up to 256 workgroups, each owning `per` flags (a multiple of 1024, at most 64 slices of 1024)
          int per = ((P.Bp + 255) / 256 + 1023) / 1024 * 1024;
          if (per > 65536) per = 65536;
          const int nb = (P.Bp + per - 1) / per;
          if (nb > 256) return fail(TO_ERR_UNSUPPORTED, "batch too large for the compaction kernels (16 777 216 trajectories)");
          hipLaunchKernelGGL(k_compact_count, dim3(nb), dim3(1024), 0, h->stream, a, per);
          hipLaunchKernelGGL(k_compact_write, dim3(nb), dim3(1024), 0, h->stream, a, per);
        }
        HIPCHECK(hipGetLastError());
      }
      if (h->profile) HIPCHECK(hipEventRecord(h->ev[4 * step + 3], h->stream));
      if (h->guard) {
        if (step == 0) if (const char* env = std::getenv("TRAJOPT_GUARD_SELFTEST")) if (std::atoi(env))  // one double behind the nominal states
          hipLaunchKernelGGL(k_guard_poke, dim3(1), dim3(1), 0, h->stream, a.Xs, (long long)P.N * P.n * (P.Bp + 64) + 3);
        TRY(check_guards(h, ("batch step " + std::to_string(step) + " of a solve").c_str()));
      }
      if (dbg_sync) {  // TRAJOPT_SYNC_DEBUG=1: localise a device fault to a batch step
        const hipError_t e = hipStreamSynchronize(h->stream);
        std::fprintf(stderr, "[sync-debug] step %d B=%d Bp=%d level=%d store_x-path=%d: %s\n", step, a.P.B, a.P.Bp, h->rp_level, (int)(last_active), hipGetErrorString(e));
        if (e != hipSuccess) return fail(TO_ERR_HIP, "device fault (sync debug)");
      }
    }
    HIPCHECK(hipMemcpyAsync(&h->counter_host[launched], &a.counter[launched], sizeof(int) * chunk, hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(hipEventRecord(cev[nchunks & 1], h->stream));
    launched += chunk;
    ++nchunks;
    return TO_OK;
  };
  int waited = 0;  // chunks whose counters have been inspected
  // early polish (ALTRO solves): the next active count at which the finished trajectories are handed over
  int early_div = 4;  // hand-over when a quarter of the batch is left (TRAJOPT_PN_EARLY_AT: the divisor)
  if (const char* env = std::getenv("TRAJOPT_PN_EARLY_AT")) early_div = std::max(1, std::atoi(env));
  int early_thr = (al_mode && h->pn_early > 0) ? P.B / early_div : -1, early_left = h->pn_early;
  bool snapshot_pending = false;
  // repacked working set (above): iLQR solves of the small models on the fused lane path with compaction
  bool repack = !al_mode && h->fused_lane && a.compact && h->ops->write_through && h->rp_min > 0;
  if (max_steps > 0) TRY(enqueue_chunk());
  while (!done && waited < nchunks) {
    if (repack && last_active >= 1 && (double)last_active <= h->rp_at * (double)a.P.B && a.P.B >= h->rp_min) {
      // the exact count is needed on the host (B of every later launch): drain the queue once — a handful of times per solve
      HIPCHECK(hipStreamSynchronize(h->stream));
      for (; checked < launched; ++checked) {
        ++steps;
        last_active = h->counter_host[checked];
        if (last_active == 0) { done = true; break; }
      }
      waited = nchunks;
      publish_progress(h, last_active, steps);
      if (done) break;
      a.step = launched - 1;   // (k_compact of that step built the list the move reads)
      {
        const int mrc = rp_move(h, last_active);
        if (mrc > 0) repack = false;  // declined (out of memory): no further attempts in this solve
        else TRY(mrc);
        if (h->guard) TRY(check_guards(h, "a move into a repacked working set"));
      }
      if (launched < max_steps) TRY(enqueue_chunk());
      continue;
    }
    if (launched < max_steps) TRY(enqueue_chunk());  // keep the queue one chunk ahead
    if (snapshot_pending) {  // taken before that chunk: the device keeps working while the host lists and launches
      TRY(early_polish_launch(h));
      snapshot_pending = false;
    }
    HIPCHECK(hipEventSynchronize(cev[waited & 1]));
    const int upto = std::min(launched, (waited + 1) * CHECK_EVERY);
    for (; checked < upto; ++checked) {
      ++steps;
      last_active = h->counter_host[checked];
      if (h->counter_host[checked] == 0) { done = true; break; }
    }
    ++waited;
    publish_progress(h, last_active, steps);
    if (!done && early_left > 0 && last_active <= early_thr) {
      TRY(early_polish_snapshot(h));
      snapshot_pending = true;
      --early_left;
      while (early_thr > 0 && last_active <= early_thr) early_thr /= 2;
    }
  }
  TRY(launch_accept(h));  // trajectories keep the slot of their last accepted step until here
  a.CW = h->cw_base; a.TW = h->tw_base;
  TRY(rp_finish(h, true));  // a repacked working set goes home
  HIPCHECK(hipEventRecord(e1, h->stream));
  HIPCHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  HIPCHECK(hipEventElapsedTime(&ms, e0, e1));
  if (h->profile) {
    for (int step = 0; step < steps; ++step)
      for (int s = 0; s < 3; ++s) {
        float t = 0.f;
        HIPCHECK(hipEventElapsedTime(&t, h->ev[4 * step + s], h->ev[4 * step + s + 1]));
        h->prof_ms[s] += t;
        h->prof_launches[s] += 1;
      }
  }
  h->last_steps = steps; h->last_ms = ms;
  if (st) TRY(fill_stats(h, st, false));
  return TO_OK;
}

}  // namespace

namespace {
struct Rccl {
  struct Id { char b[128]; };  // ncclUniqueId (NCCL_UNIQUE_ID_BYTES), passed by value
  void* lib = nullptr;
  int (*GetUniqueId)(void*) = nullptr;
  int (*CommInitRank)(void**, int, Id, int) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;
std::once_flag g_rccl_once;
int load_rccl() {
  std::call_once(g_rccl_once, [] {
    // TRAJOPT_RCCL_LIB names the collective library outright (any library exporting these eight nccl* entry points: a site build of
    // RCCL, or the shared-memory stand-in of tests/rccl_stub that lets several ranks share one GPU); nothing else is tried then
    if (const char* env = std::getenv("TRAJOPT_RCCL_LIB")) g_rccl.lib = dlopen(env, RTLD_NOW | RTLD_GLOBAL);
    else
      for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        g_rccl.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (g_rccl.lib) break;
      }
    if (!g_rccl.lib) return;
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(g_rccl.lib, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(g_rccl.lib, "ncclCommInitRank");
    g_rccl.AllGather = (decltype(g_rccl.AllGather))dlsym(g_rccl.lib, "ncclAllGather");
    g_rccl.Broadcast = (decltype(g_rccl.Broadcast))dlsym(g_rccl.lib, "ncclBroadcast");
    g_rccl.GroupStart = (decltype(g_rccl.GroupStart))dlsym(g_rccl.lib, "ncclGroupStart");
    g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))dlsym(g_rccl.lib, "ncclGroupEnd");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(g_rccl.lib, "ncclCommDestroy");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(g_rccl.lib, "ncclGetErrorString");
  });
  if (!g_rccl.lib || !g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllGather || !g_rccl.CommDestroy || !g_rccl.Broadcast ||
      !g_rccl.GroupStart || !g_rccl.GroupEnd)
    return fail(TO_ERR_HIP, "librccl.so not found (or incomplete): the multi-GPU entry points need RCCL");
  return TO_OK;
}
#define RCCLCHECK(expr)                                                                                          \
  do {                                                                                                           \
    int r_ = (expr);                                                                                             \
    if (r_ != 0) return fail(TO_ERR_HIP, std::string(#expr) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "rccl error")); \
  } while (0)
constexpr int kNcclDouble = 8;  // ncclFloat64 (rccl.h)
constexpr int kNcclInt32 = 2;   // ncclInt32
}  // namespace


extern "C" {

int to_abi_version(void) { return TO_ABI_VERSION; }
#ifndef TO_BUILD_ID
#define TO_BUILD_ID "unstamped"
#endif
static const char g_build_stamp[] = "TO_BUILD_ID=" TO_BUILD_ID;  // also found by scanning the file (build.py)
const char* to_build_id(void) { return g_build_stamp + 12; }
const char* to_last_error(void) { return g_err.c_str(); }
int to_device_count(int* count) {
  CHECK_P(count);
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { *count = 0; return fail(TO_ERR_HIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e)); }
  *count = n;
  return TO_OK;
}
int to_default_options(to_solver_opts* o) { CHECK_P(o); default_opts(o); return TO_OK; }

int to_create(const to_problem_desc* desc, const to_solver_opts* opts, int device, to_handle** out) {
  CHECK_P(out);
  if (!desc) return fail(TO_ERR_NULL, "null descriptor");
  if (desc->abi_version != TO_ABI_VERSION) return fail(TO_ERR_ARGUMENT, "ABI version mismatch");
  if (opts) TRY(validate_opts(*opts));
  int n, m, ne, key;
  if (model_dims(desc->model, desc->model_params, &n, &m, &ne, &key)) return fail(TO_ERR_UNSUPPORTED, "unknown model");
  if (desc->n != n || desc->m != m) return fail(TO_ERR_DIMENSION_MISMATCH, "Objective state/control dimensions don't match model.");  // src/problem.jl:65-68
  if (desc->N < 2) return fail(TO_ERR_ASSERTION, "length(models) == N-1 requires N >= 2");                                         // src/problem.jl:49
  if (desc->B < 1) return fail(TO_ERR_ARGUMENT, "batch must be positive");
  if (!(desc->tf > desc->t0)) return fail(TO_ERR_ASSERTION, "tf > t0");                                                             // src/problem.jl:50
  if (desc->integrator < TO_RK4 || desc->integrator > TO_EULER) return fail(TO_ERR_UNSUPPORTED, "unknown integrator");
  const int N = desc->N, B = desc->B;
  std::vector<double> dt(N - 1);
  if (desc->dt) {
    double s = 0;
    for (int k = 0; k < N - 1; ++k) { dt[k] = desc->dt[k]; s += dt[k]; if (!(dt[k] > 0)) return fail(TO_ERR_ASSERTION, "dt must be positive"); }
    if (std::fabs(s - (desc->tf - desc->t0)) > 1e-8 * std::fmax(1.0, std::fabs(desc->tf - desc->t0)))
      return fail(TO_ERR_ASSERTION, "time(Z[end]) ≈ tf: time steps are inconsistent with the final time");                          // src/problem.jl:52
  } else {
    for (int k = 0; k < N - 1; ++k) dt[k] = (desc->tf - desc->t0) / (N - 1);
  }
  if (desc->n_costs < 1 || !desc->costs) return fail(TO_ERR_ARGUMENT, "objective needs at least one cost function");
  const int rot = desc->model == TO_MODEL_QUADROTOR ? (int)desc->model_params[10] : -1;
  for (int i = 0; i < desc->n_costs; ++i) TRY(validate_cost(n, rot, desc->costs[i]));
  if (desc->model == TO_MODEL_HYBRID_DOUBLE_INTEGRATOR) {
    const double S = desc->model_params[1];
    if (!(S >= 1.0) || !(S <= (double)(N - 2)) || S != std::floor(S))
      return fail(TO_ERR_ARGUMENT, "hybrid double integrator: params[1] (time steps of the first model) must be an integer in 1 .. N-2");
  }
  std::vector<double> step_table;  // TO_MODEL_VECTOR: the per-step model table (models.h ModelVectorModel), validated like RD.dims(models)
  if (desc->model == TO_MODEL_VECTOR) TRY(lower_step_models(desc->step_models, N, &step_table));
  std::vector<int> cost_index(N);
  if (desc->cost_index) {
    for (int k = 0; k < N; ++k) {
      if (desc->cost_index[k] < 0 || desc->cost_index[k] >= desc->n_costs) return fail(TO_ERR_DIMENSION_MISMATCH, "cost_index outside the cost list");
      cost_index[k] = desc->cost_index[k];
    }
  } else {
    if (desc->n_costs < 2) return fail(TO_ERR_ARGUMENT, "Objective(stage, terminal, N) needs two cost functions");
    for (int k = 0; k < N; ++k) cost_index[k] = (k == N - 1) ? 1 : 0;
  }
  if (desc->n_constraints < 0 || (desc->n_constraints > 0 && !desc->constraints)) return fail(TO_ERR_ARGUMENT, "bad constraint list");
  std::vector<DevCon> cons;
  long long n_duals = 0;
  for (int i = 0; i < desc->n_constraints; ++i) {
    DevCon ci;
    TRY(validate_constraint(n, m, N, desc->constraints[i], &ci));
    ci.dual_off = n_duals;
    n_duals += (long long)ci.p * (ci.k2 - ci.k1 + 1);
    cons.push_back(ci);
  }
  int ndev = 0;
  {
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev < 1) return fail(TO_ERR_HIP, "no usable HIP device (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(TO_ERR_ARGUMENT, "device ordinal out of range");
  }
  to_handle* h = new to_handle();
  h->device = device;
  h->model_key = key;
  if (const char* env = std::getenv("TRAJOPT_GUARD")) h->guard = std::atoi(env) != 0;
  h->R = (ne + m) <= 4 ? 4 : (ne + m) <= 8 ? 8 : 16;
  h->G = 64 / h->R;
  h->ops = model_ops(key);
  if (!h->ops || !h->ops->rollout || !h->ops->expand || !h->ops->backward) { delete h; return fail(TO_ERR_UNSUPPORTED, "model kernels not linked"); }
  h->costs.assign(desc->costs, desc->costs + desc->n_costs);
  h->cons = cons; h->dt = dt; h->cost_index = cost_index; h->step_table = step_table;
  auto bail = [&](int rc) { std::string e = g_err; to_destroy(h); g_err = e; return rc; };
#define TRYB(expr) do { int r_ = (expr); if (r_ != TO_OK) return bail(r_); } while (0)
#define HIPB(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return bail(fail(TO_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_))); } while (0)
  HIPB(hipSetDevice(device));
  HIPB(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
  KArgs& a = h->a;
  std::memset(&a, 0, sizeof(a));
  DevProblem& P = a.P;
  P.n = n; P.m = m; P.ne = ne; P.N = N; P.B = B; P.Bp = ((B + BLOCK - 1) / BLOCK) * BLOCK;
  P.integrator = desc->integrator; P.n_costs = desc->n_costs; P.n_cons = (int)cons.size(); P.n_duals = n_duals;
  std::memcpy(P.mp, desc->model_params, sizeof(P.mp));
  if (opts) P.opts = *opts; else default_opts(&P.opts);
  const size_t Bp = P.Bp;
  TRYB(dev_alloc(h, &h->d_costs, h->costs.size()));
  TRYB(dev_alloc(h, &h->d_cons, cons.size()));
  TRYB(dev_alloc(h, &h->d_dt, (size_t)N - 1));
  TRYB(dev_alloc(h, &h->d_cost_index, (size_t)N));
  HIPB(hipMemcpyAsync(h->d_dt, dt.data(), sizeof(double) * (N - 1), hipMemcpyHostToDevice, h->stream));
  HIPB(hipMemcpyAsync(h->d_cost_index, cost_index.data(), sizeof(int) * N, hipMemcpyHostToDevice, h->stream));
  if (!h->step_table.empty()) {  // model vector: the table's device address travels in model_params[0] (bit pattern; models.h)
    double* d_tab = nullptr;
    TRYB(dev_alloc(h, &d_tab, h->step_table.size()));
    HIPB(hipMemcpyAsync(d_tab, h->step_table.data(), sizeof(double) * h->step_table.size(), hipMemcpyHostToDevice, h->stream));
    const unsigned long long bits = (unsigned long long)reinterpret_cast<uintptr_t>(d_tab);
    std::memset(P.mp, 0, sizeof(P.mp));
    std::memcpy(&P.mp[0], &bits, sizeof(bits));
  }
  TRYB(upload_tables(h));
  P.dt = (DoubleC*)h->d_dt; P.cost_index = (IntC*)h->d_cost_index; P.costs = (CostC*)h->d_costs; P.cons = (ConC*)h->d_cons;
  // Line-search candidates evaluated concurrently per trajectory, CW (a power of two): a forward wave holds CW
  // candidates x 64/CW trajectories, so the launch has Bp*CW/64 waves — enough to cover the 1024 SIMDs of the chip for
  // small batches, at most the model's ls_first_round (4 for the small models, 16 for the Quadrotor; the default search
  // depth is 20: further in-kernel rounds cover the rest).
  {
    int cw = std::max(1, std::min(h->ops->ls_first_round, 1024 / (P.Bp / BLOCK)));  // one forward wave per SIMD (measured: C5 0.71 M it/s with 8, 0.68 M with 16)
    // ... the models that search narrowly anyway (the small ones: 4 step sizes) keep their full first round at every batch size:
    // a trajectory that rejects everything offered costs its wave a second full pass, which is worse than the extra lanes —
    // measured at B = 131 072 (fused lane path): 25.4 / 29.6 / 37.4 M trajectory-iterations/s with 1 / 2 / 4 step sizes per
    // round (forward pass 959 -> 582 us per batch step from 2 to 4), and 18.7 -> 21.8 M at B = 32 768
    if (h->ops->ls_first_round <= 4) cw = h->ops->ls_first_round;
    if (const char* env = std::getenv("TRAJOPT_LS_CANDIDATES")) cw = std::max(1, std::min(16, std::atoi(env)));  // tuning knob
    int lg = 0;
    while ((2 << lg) <= cw) ++lg;
    h->cw_base = 1 << lg; h->tw_base = 64 / h->cw_base;
    // The small (write-through) models take any width: their lane map is the static one in every round (k_forward.h) and nothing in it
    // needs a power of two — lanes CW*TW .. 63 ride along without a candidate.  THREE step sizes x 21 trajectories per wave: the C2-shaped
    // Cartpole solves accept within the first three step sizes in 99.5 % of their line searches (alpha = 1 / 0.5 / 0.25: 18 / 40 / 42 %,
    // measured on the oracle), so a wave serves 21 trajectories instead of 16 per pass for one extra pass in ~10 % of the waves.
    if (h->ops->write_through && cw >= 1 && cw <= 16) { h->cw_base = cw; h->tw_base = 64 / cw; }
    a.CW = h->cw_base; a.TW = h->tw_base;
    // Deep shape: the WHOLE search depth in one round (20 step sizes x 3 trajectories per wave by default).  A trajectory
    // that rejects the first CW step sizes otherwise costs the batch a second full rollout pass (the Quadrotor solves do so
    // in half of their steps: forward 1.0 ms instead of 0.55 ms).  It needs 64/TW = 21 waves per 64 trajectories instead of
    // 16, so the solve loop switches to it once the active trajectories fit the chip that way (one wave per SIMD).
    const int total = P.opts.iterations_linesearch;
    const char* deep_env = std::getenv("TRAJOPT_LS_DEEP");  // 0: never switch to the deep shape (tests of the round logic)
    // Only while a wave still holds >= 2 trajectories that way (total <= 32): with 33..64 step sizes a deep wave would carry
    // ONE trajectory and the candidate arrays 64x the nominal storage (several GB on C5) — those depths run as further
    // rounds of the base shape instead.
    if (!h->ops->write_through && total > h->cw_base && total <= 32 && !(deep_env && std::atoi(deep_env) == 0)) {
      hipDeviceProp_t prop;
      HIPB(hipGetDeviceProperties(&prop, device));
      h->cw_deep = total; h->tw_deep = 64 / total;
      h->deep_max_active = prop.multiProcessorCount * 4 * h->tw_deep;
    }
  }
  // backward-pass flavour: one wave per trajectory on the matrix cores (tangent-matrix expansion) where the model has it,
  // else the cooperative LDS kernel on the column layout.  TRAJOPT_BACKWARD=coop|mfma overrides (A/B measurements).
  // Models that have the cooperative kernel as well (small ones: several trajectories per wave) default to it: measured on
  // the Cartpole at B = 1024, 78 us cooperative vs 90 us MFMA per backward pass (the 5x5 blocks fill 2 % of a 16x16 tile).
  a.bwd_mfma = (h->ops->mfma_backward && !h->ops->coop_backward) ? 1 : 0;
  // Small models: the cooperative kernel (R lanes per trajectory, LDS exchanges) has the shorter critical path — 78 vs 111 us
  // per pass on the Cartpole at B = 1024, a single lane issues every FMA of a knot itself — and the lane kernel the fewer
  // instructions: it takes over once the cooperative waves would stack three deep on every SIMD (measured at B = 32 768:
  // 12.0 vs 9.8 M trajectory-iterations/s).
  {
    int cus = 256;
    HIPB(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device));
    h->simds = 4 * cus;
    const long coop_waves = ((long)B + h->G - 1) / h->G;
    // (with the expansions fused into both kernels the crossover sits at ~12 000 Cartpole trajectories: measured fused lane vs
    // fused cooperative 10.8 vs 9.9 M it/s at B = 12 288, 7.9 vs 9.3 M at B = 8 192)
    // (and where the scan kernel runs ahead of the cooperative one — unconstrained problems with diagonal cost blocks — at
    // ~20 000: scan + cooperative vs fused lane 15.2 vs 11.2 M it/s at B = 12 288, 17.1 vs 14.4 at 16 384, 19.4 vs 20.4 at 24 576)
    const bool scan_path = h->ops->expand_backward_scan && h->cons.empty() && N <= 126 && diagonal_cost_blocks(h) &&
                           !(std::getenv("TRAJOPT_SCAN") && !std::atoi(std::getenv("TRAJOPT_SCAN")));
    const long lane_from = scan_path ? 10L : (h->ops->expand_backward ? 6L : 12L);
    a.bwd_lane = (h->ops->lane_backward && coop_waves >= lane_from * cus) ? 1 : 0;
  }
  if (const char* env = std::getenv("TRAJOPT_BACKWARD")) {
    if (!std::strcmp(env, "coop") && h->ops->coop_backward) { a.bwd_mfma = 0; a.bwd_lane = 0; }
    if (!std::strcmp(env, "mfma") && h->ops->mfma_backward) { a.bwd_mfma = 1; a.bwd_lane = 0; }
    if (!std::strcmp(env, "lane") && h->ops->lane_backward) { a.bwd_mfma = 0; a.bwd_lane = 1; }
  }
  if (const char* env = std::getenv("TRAJOPT_EXPAND_LANE")) h->expand_lane = std::atoi(env) != 0;
  h->fused_coop = (!a.bwd_lane && !a.bwd_mfma && h->ops->expand_backward_coop) ? 1 : 0;  // used while the cost blocks are diagonal (KArgs::h_diag)
  if (const char* env = std::getenv("TRAJOPT_FUSED_COOP")) if (!std::atoi(env)) h->fused_coop = 0;
  h->roll_min_active = -1;
  if (const char* env = std::getenv("TRAJOPT_ACCEPT_ROLL_MIN")) h->roll_min_active = std::atoi(env);
  if (const char* env = std::getenv("TRAJOPT_ACCEPT_ROLL_FRAC")) h->roll_min_frac = std::atof(env);
  if (const char* env = std::getenv("TRAJOPT_REPACK")) h->rp_min = std::atoi(env);  // 0: never; n: while the working set holds >= n trajectories
  if (const char* env = std::getenv("TRAJOPT_REPACK_AT")) h->rp_at = std::min(0.95, std::max(0.05, std::atof(env)));
  if (const char* env = std::getenv("TRAJOPT_EXPAND_PACK")) h->expand_pack = std::atoi(env) != 0;
  h->fwd2 = 2;  // 0: one-wave forward pass only; 1: two-wave always (phase API included); 2: per batch step, by the active count
  if (const char* env = std::getenv("TRAJOPT_FWD2")) h->fwd2 = std::atoi(env);
  if (h->fwd2 < 0 || h->fwd2 > 2) h->fwd2 = 2;
  a.coop_merge = 1;
  if (const char* env = std::getenv("TRAJOPT_COOP_MERGE")) a.coop_merge = std::atoi(env) != 0;
  // scan (parallel-in-time) backward pass ahead of the fused cooperative kernel, over the whole range of batches the cooperative
  // path serves (measured, Cartpole: 40.7 vs 95.2 us per step at B = 1024, 76 vs 103 at 4096, 119 vs 164 at 8192)
  h->scan = (h->fused_coop && h->ops->expand_backward_scan && N <= 126) ? 1 : 0;
  if (const char* env = std::getenv("TRAJOPT_SCAN")) { if (!std::atoi(env)) h->scan = 0; else if (h->scan && std::atoi(env) == 2) h->scan = 2; }
  h->scan_max_active = 1 << 30;
  if (const char* env = std::getenv("TRAJOPT_SCAN_MAX")) h->scan_max_active = std::atoi(env);
  h->fused_lane = (a.bwd_lane && h->ops->expand_backward) ? 1 : 0;
  if (const char* env = std::getenv("TRAJOPT_FUSED_LANE")) if (!std::atoi(env)) h->fused_lane = 0;
  TRYB(upload_tables(h));  // again: h_compact depends on bwd_mfma
  h->accept_chunks = std::max(1, std::min(128, (N * n + (N - 1) * P.m + 31) / 32));
  TRYB(dev_alloc(h, &a.Xs, (size_t)N * n * (Bp + 64)));  // (+ one spare tile: where k_accept_roll's lanes without an accepted step store)
  TRYB(dev_alloc(h, &a.Us, (size_t)(N - 1) * m * Bp));
  {  // candidates, forward-wave-major (common.h): 64 lanes per wave in either shape
    size_t waves = (Bp + h->tw_base - 1) / h->tw_base;
    if (h->cw_deep) waves = std::max(waves, (Bp + h->tw_deep - 1) / h->tw_deep);
    a.store_x = 1;
    a.dump_wave = (int)waves;  // one spare block: the store target of lanes that hold no candidate (k_forward.h)
    // ... and, for the models whose search goes through several rounds of the base shape, a second block per wave for
    // the repacked last round (k_forward.h LsRound; TRAJOPT_LS_REPACK=0 switches it off)
    size_t extra = 0;
    const char* rp_env = std::getenv("TRAJOPT_LS_REPACK");
    a.repack_block0 = 0;
    if (!h->ops->write_through && !(rp_env && std::atoi(rp_env) == 0)) {
      extra = waves;  // as many as either wave shape launches: a search deeper than the deep shape (options changed after creation) repacks there too
      a.repack_block0 = (int)waves + 1;
    }
    // two-launch line search (small models, dense large batches: the steps that store candidate controls only): launch A's blocks,
    // launch B's behind them, one dump block — control candidates only, so only Uc grows
    size_t ublocks = waves + 1 + extra;
    h->ls2_cwa = 0;
    if (h->ops->write_through && h->ops->accept_roll && Bp >= 32768) {
      int ca = 2, cb = 2;  // measured default (r05, Cartpole at B = 1 048 576: 74.1 M it/s with 2 + 2, 71.5 with 1 + 2, 70.9 with one launch; TRAJOPT_LS_TWO=a,b overrides, a = 0: off)
      if (const char* env = std::getenv("TRAJOPT_LS_TWO")) { ca = std::atoi(env); const char* c2 = std::strchr(env, ','); cb = c2 ? std::atoi(c2 + 1) : 2; }
      if (ca >= 1 && ca <= 16 && cb >= 1 && cb <= 16 && ca < P.opts.iterations_linesearch) {
        h->ls2_cwa = ca; h->ls2_cwb = cb;
        const size_t nA = (Bp + (64 / ca) - 1) / (64 / ca), nB = (Bp + (64 / cb) - 1) / (64 / cb);
        h->ls2_blkA = (int)nA; h->ls2_dump = (int)(nA + nB);
        ublocks = std::max(ublocks, nA + nB + 1);
      }
    }
    TRYB(dev_alloc(h, &a.Xc, (size_t)N * n * (waves + 1 + extra) * 64));
    TRYB(dev_alloc(h, &a.Uc, (size_t)(N - 1) * m * ublocks * 64));
  }
  if (h->ls2_cwa) { TRYB(dev_alloc(h, &a.pending, Bp)); TRYB(dev_alloc(h, &a.plist, Bp)); TRYB(dev_alloc(h, &a.pcount, 1)); }
  TRYB(dev_alloc(h, &a.x0, (size_t)n * Bp));
  TRYB(dev_alloc(h, &a.acc, Bp));
  TRYB(dev_alloc(h, &a.accp, Bp));
  TRYB(dev_alloc(h, &a.alist, 2 * (size_t)Bp)); TRYB(dev_alloc(h, &a.acount, 2)); TRYB(dev_alloc(h, &a.ccount, 256));
  // active-list compaction: the fused lane path (large batches of the small models) and the MFMA path (Quadrotor: its expansion
  // waves hold four trajectories each and the solves end with long straggler tails — 141 batch steps for a mean of 52
  // iterations on C3); the cooperative small-batch path is latency-bound and keeps its fixed mapping;
  h->compact = (h->fused_lane || a.bwd_mfma) ? 1 : 0;
  if (const char* env = std::getenv("TRAJOPT_COMPACT")) if (!std::atoi(env)) h->compact = 0;  // armed only inside a solve
  TRYB(dev_alloc(h, &a.oflag, Bp)); TRYB(dev_alloc(h, &a.ost, Bp));
  TRYB(dev_alloc(h, &a.olist, 2 * (size_t)Bp)); TRYB(dev_alloc(h, &a.ocount, 2));
  TRYB(dev_alloc(h, &a.knotbuf, (size_t)N * Bp));
  TRYB(dev_alloc(h, &a.mu_next, (size_t)std::max<size_t>(1, cons.size()) * Bp));
  if (a.bwd_mfma) {  // tangent-matrix layout (k_backward.h): RS rows of 64 per knot for [A B], up to RS+1 for the cost block
    const int rs = h->ops->rs;
    TRYB(dev_alloc(h, &a.Mt, (size_t)Bp * (N - 1) * rs * 64));
    TRYB(dev_alloc(h, &a.Ht, (size_t)Bp * N * (rs + 1) * 64));
    TRYB(dev_alloc(h, &a.gt, (size_t)Bp * N * 16));
    TRYB(dev_alloc(h, &h->d_crow, 64));
    HIPB(hipMemcpyAsync(h->d_crow, h->ops->crow, sizeof(int) * 64, hipMemcpyHostToDevice, h->stream));
  } else if (a.bwd_lane) {  // lane layout (k_backward.h LaneLay), in the column-layout pointers
    const size_t tiles = (size_t)Bp / 64, nc = (size_t)ne + m;
    TRYB(dev_alloc(h, &a.Mc, tiles * (size_t)(N - 1) * ne * nc * 64));
    TRYB(dev_alloc(h, &a.Hc, tiles * (size_t)N * (nc * (nc + 1) / 2) * 64));
    TRYB(dev_alloc(h, &a.gc, tiles * (size_t)N * nc * 64));
  } else {
    const size_t gtiles = ((size_t)B + h->G - 1) / h->G;  // waves of the column-layout kernels
    TRYB(dev_alloc(h, &a.Mc, gtiles * (size_t)(N - 1) * ne * 64));
    TRYB(dev_alloc(h, &a.Hc, gtiles * (size_t)N * (ne + m) * 64));
    TRYB(dev_alloc(h, &a.gc, gtiles * (size_t)N * 64));
  }
  TRYB(dev_alloc(h, &a.Kt, (size_t)Bp * (N - 1) * m * (ne + 1)));  // gains rows, trajectory-major
  TRYB(dev_alloc(h, &a.lam, (size_t)n_duals * Bp));
  TRYB(dev_alloc(h, &a.mu, cons.size() * Bp));
  TRYB(dev_alloc(h, &a.J, Bp)); TRYB(dev_alloc(h, &a.dJ, Bp)); TRYB(dev_alloc(h, &a.grad, Bp));
  TRYB(dev_alloc(h, &a.rho, Bp)); TRYB(dev_alloc(h, &a.drho, Bp)); TRYB(dev_alloc(h, &a.dV, 2 * Bp));
  TRYB(dev_alloc(h, &a.cmax, Bp)); TRYB(dev_alloc(h, &a.Jout, Bp));
  TRYB(dev_alloc(h, &a.it_pn, Bp)); TRYB(dev_alloc(h, &a.pn_cmax, Bp));
  TRYB(dev_alloc(h, &a.status, Bp)); TRYB(dev_alloc(h, &a.iterations, Bp)); TRYB(dev_alloc(h, &a.it_inner, Bp));
  TRYB(dev_alloc(h, &a.outer, Bp)); TRYB(dev_alloc(h, &a.dJzero, Bp)); TRYB(dev_alloc(h, &a.ls_index, Bp));
  TRYB(dev_alloc(h, &a.active, Bp)); TRYB(dev_alloc(h, &a.budget, Bp)); TRYB(dev_alloc(h, &a.bpfail, Bp));
  TRYB(dev_alloc(h, &h->d_tmp, Bp)); TRYB(dev_alloc(h, &h->d_tmp2, Bp));
  // reference defaults: X0 = NaN, U0 = 0 (src/problem.jl:83-84); duals 0, penalties penalty_initial
  {
    std::vector<double> nanv((size_t)N * n * Bp, std::nan(""));
    HIPB(hipMemcpyAsync(a.Xs, nanv.data(), nanv.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIPB(hipStreamSynchronize(h->stream));
    {  // the regularisation a phase-API backward pass starts from (a solve resets it the same way: k_solve_init)
      std::vector<double> rho0((size_t)Bp, P.opts.bp_reg_initial);
      HIPB(hipMemcpyAsync(a.rho, rho0.data(), rho0.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
      HIPB(hipStreamSynchronize(h->stream));
    }
    if (!cons.empty()) {
      std::vector<double> mu(cons.size() * Bp, P.opts.penalty_initial);
      HIPB(hipMemcpyAsync(a.mu, mu.data(), mu.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
      HIPB(hipStreamSynchronize(h->stream));
    }
  }
  if (a.Mt && a.h_compact && h->expand_pack && h->ops->expand_const) TRYB(h->ops->expand_const(h));
  HIPB(hipStreamSynchronize(h->stream));
#undef TRYB
#undef HIPB
  *out = h;
  return TO_OK;
}

int to_destroy(to_handle* h) {
  if (!h) return TO_OK;
  if (h->inflight) { h->worker.join(); h->inflight = false; }
  hipSetDevice(h->device);
  if (h->stream) hipStreamSynchronize(h->stream);
  if (h->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(h->comm);
  for (void* p : h->allocs) g_free(h, p);
  g_free(h, h->stage);
  g_free(h, h->pn_ws);
  g_free(h, h->pn_pak);
  g_free(h, h->pn_koff);
  g_free(h, h->pn_list);
  if (h->guard_tab) hipFree(h->guard_tab);
  if (h->guard_bad) hipFree(h->guard_bad);
  if (h->pn_list_host) hipHostFree(h->pn_list_host);
  if (h->snap_status) hipHostFree(h->snap_status);
  if (h->snap_active) hipHostFree(h->snap_active);
  if (h->snap_cmax) hipHostFree(h->snap_cmax);
  for (hipEvent_t e : h->pn_ev) if (e) hipEventDestroy(e);
  if (h->pn_stream) hipStreamDestroy(h->pn_stream);
  for (int w = 0; w < 2; ++w) { for (void* q : h->rp_work[w]) g_free(h, q); g_free(h, h->rp_map[w]); }
  if (h->counter_host) hipHostFree(h->counter_host);
  for (hipEvent_t e : h->ev) hipEventDestroy(e);
  for (hipEvent_t e : h->sev) if (e) hipEventDestroy(e);
  if (h->stream) hipStreamDestroy(h->stream);
  delete h;
  return TO_OK;
}

int to_set_options(to_handle* h, const to_solver_opts* o) { CHECK_H(h); CHECK_IDLE(h); CHECK_P(o); TRY(validate_opts(*o)); h->a.P.opts = *o; return TO_OK; }
int to_get_options(const to_handle* h, to_solver_opts* o) { CHECK_H(h); CHECK_P(o); *o = h->a.P.opts; return TO_OK; }
int to_sync(to_handle* h) { CHECK_H(h); CHECK_IDLE(h); TRY(use_device(h)); HIPCHECK(hipStreamSynchronize(h->stream)); return TO_OK; }
void* to_stream(to_handle* h) { return h ? (void*)h->stream : nullptr; }

int to_solver_path(const to_handle* h, int32_t* info) {
  CHECK_H(h); CHECK_P(info);
  const KArgs& a = h->a;
  info[0] = a.bwd_mfma ? 1 : a.bwd_lane ? 2 : 0;
  const DevProblem& P = a.P;
  const bool fcoop = h->fused_coop && a.h_diag && (P.expand_variant == 0 || P.expand_variant == 2);
  info[1] = (h->fused_lane || (!a.bwd_mfma && !a.bwd_lane && fcoop)) ? 1 : 0;
  info[2] = h->compact;
  info[3] = h->cw_base;
  bool has2 = false;
  for (int i = 0; i < 32; ++i) has2 = has2 || h->ops->forward2[i] != nullptr;
  info[4] = (h->fwd2 && has2) ? 2 : 1;  // (two-wave workgroups are used while the active trajectories leave room for them)
  info[5] = (h->scan && fcoop && P.expand_variant == 0 && !a.bwd_mfma && !a.bwd_lane) ? 1 : 0;
  info[6] = (h->ops->accept_roll && h->roll_min_active != 0) ? 1 : 0;  // full-chip batch steps store candidate controls only (k_accept_roll)
  info[7] = a.repack_block0 != 0 ? 1 : 0;                               // repacked last line-search round
  if (h->fused_lane && h->compact && h->ops->write_through && h->rp_min > 0 && P.B >= h->rp_min) info[7] |= 2;  // repacked working set (iLQR solves)
  return TO_OK;
}
int to_knot_dims(const to_handle* h, int32_t* nx, int32_t* nu) {
  CHECK_H(h); CHECK_P(nx); CHECK_P(nu);
  const DevProblem& P = h->a.P;
  for (int k = 0; k < P.N; ++k) {
    int a = P.n, b = P.m;
    if (h->model_key == 7) HybridDoubleIntegratorModel::knot_dims(P.mp, P.N, k, &a, &b);
    if (h->model_key == 8) {  // the host copy of the table (P.mp[0] holds the DEVICE address)
      const unsigned long long bits = (unsigned long long)reinterpret_cast<uintptr_t>(h->step_table.data());
      double hp[1];
      std::memcpy(&hp[0], &bits, sizeof(bits));
      ModelVectorModel::knot_dims(hp, P.N, k, &a, &b);
    }
    nx[k] = a; nu[k] = b;
  }
  return TO_OK;
}
int to_set_profiling(to_handle* h, int enable) { CHECK_H(h); CHECK_IDLE(h); h->profile = enable != 0; return TO_OK; }
int to_reset_profile(to_handle* h) {
  CHECK_H(h); CHECK_IDLE(h);
  for (int i = 0; i < TO_PROFILE_SLOTS; ++i) { h->prof_ms[i] = 0; h->prof_launches[i] = 0; }
  return TO_OK;
}
int to_get_profile(to_handle* h, double* ms, int64_t* launches) {
  CHECK_H(h);
  for (int i = 0; i < TO_PROFILE_SLOTS; ++i) { if (ms) ms[i] = h->prof_ms[i]; if (launches) launches[i] = h->prof_launches[i]; }
  return TO_OK;
}

int to_dims(const to_handle* h, int32_t* n, int32_t* m, int32_t* ne, int32_t* N, int32_t* B) {
  CHECK_H(h);
  const DevProblem& P = h->a.P;
  if (n) *n = P.n; if (m) *m = P.m; if (ne) *ne = P.ne; if (N) *N = P.N; if (B) *B = P.B;
  return TO_OK;
}
int to_num_constraints(const to_handle* h, int32_t* p) {
  CHECK_H(h); CHECK_P(p);
  for (int k = 0; k < h->a.P.N; ++k) p[k] = 0;
  for (const DevCon& ci : h->cons) for (int k = ci.k1; k <= ci.k2; ++k) p[k] += ci.p;
  return TO_OK;
}

int to_set_initial_state(to_handle* h, const double* x0) {
  CHECK_H(h); CHECK_IDLE(h); CHECK_P(x0); TRY(use_device(h));
  return upload_vec(h, x0, h->a.x0, h->a.P.n);
}
int to_get_initial_state(to_handle* h, double* x0) {
  CHECK_H(h); CHECK_IDLE(h); CHECK_P(x0); TRY(use_device(h));
  return download_vec(h, x0, h->a.x0, h->a.P.n);
}
int to_set_controls(to_handle* h, const double* U) {
  CHECK_H(h); CHECK_IDLE(h); CHECK_P(U); TRY(use_device(h));
  return upload_vec(h, U, h->a.Us, h->a.P.m * (h->a.P.N - 1));
}
int to_set_states(to_handle* h, const double* X) {
  CHECK_H(h); CHECK_IDLE(h); CHECK_P(X); TRY(use_device(h));
  return upload_vec(h, X, h->a.Xs, h->a.P.n * h->a.P.N);
}
int to_infeasible_controls(to_handle* h) {
  CHECK_H(h); CHECK_IDLE(h); TRY(use_device(h));
  if (!h->ops->infeasible_controls) return fail(TO_ERR_ARGUMENT, "to_infeasible_controls: the handle's model is not TO_MODEL_INFEASIBLE");
  TRY(h->ops->infeasible_controls(h));
  HIPCHECK(hipStreamSynchronize(h->stream));
  return TO_OK;
}
int to_set_controls_uniform(to_handle* h, const double* u) {
  CHECK_H(h); CHECK_IDLE(h); CHECK_P(u); TRY(use_device(h));
  const DevProblem& P = h->a.P;
  TRY(ensure_stage(h, sizeof(double) * P.m));
  HIPCHECK(hipMemcpyAsync(h->stage, u, sizeof(double) * P.m, hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(k_fill_uniform, grid_b(h, P.m * (P.N - 1)), dim3(BLOCK), 0, h->stream, h->a.Us, h->stage, P.m, P.m * (P.N - 1), P.B);
  HIPCHECK(hipGetLastError());
  HIPCHECK(hipStreamSynchronize(h->stream));
  return TO_OK;
}
int to_get_states(to_handle* h, double* X) {
  CHECK_H(h); CHECK_IDLE(h); CHECK_P(X); TRY(use_device(h));
  return download_nominal(h, X, h->a.Xs, h->a.P.n * h->a.P.N);
}
int to_get_controls(to_handle* h, double* U) {
  CHECK_H(h); CHECK_IDLE(h); CHECK_P(U); TRY(use_device(h));
  return download_nominal(h, U, h->a.Us, h->a.P.m * (h->a.P.N - 1));
}
int to_get_states_device(to_handle* h, void* dX) {
  CHECK_H(h); CHECK_IDLE(h); CHECK_P(dX); TRY(use_device(h));
  return download_nominal(h, nullptr, h->a.Xs, h->a.P.n * h->a.P.N, dX);
}
int to_get_controls_device(to_handle* h, void* dU) {
  CHECK_H(h); CHECK_IDLE(h); CHECK_P(dU); TRY(use_device(h));
  return download_nominal(h, nullptr, h->a.Us, h->a.P.m * (h->a.P.N - 1), dU);
}
int to_set_cost(to_handle* h, int32_t id, const to_cost_desc* c) {
  CHECK_H(h); CHECK_IDLE(h); CHECK_P(c); TRY(use_device(h));
  if (id < 0 || id >= (int)h->costs.size()) return fail(TO_ERR_ARGUMENT, "cost id out of range");
  TRY(validate_cost(h->a.P.n, (h->model_key >= 4 && h->model_key <= 6) ? (int)h->a.P.mp[10] : -1, *c));  // rigid bodies only (key 7: hybrid double integrator)
  h->costs[id] = *c;
  if (h->d_gl) {  // the cost's per-trajectory linear terms start over (they were relative to the old descriptor)
    const int nz = h->a.P.n + h->a.P.m, L = (int)h->costs.size() * nz;
    std::vector<double> zero((size_t)nz * h->a.P.B, 0.0);
    TRY(upload_vec(h, zero.data(), h->d_gl, nz, L, id * nz));
    h->gl_set[id] = 0;
    bool any = false;
    for (char f : h->gl_set) any = any || f;
    if (!any) h->a.P.gl = nullptr;  // every cost is back on its shared descriptor: the default forward variants again
  }
  return upload_tables(h);
}
int to_set_cost_linear_batch(to_handle* h, int32_t id, const double* q, const double* r) {
  CHECK_H(h); CHECK_IDLE(h); TRY(use_device(h));
  if (id < 0 || id >= (int)h->costs.size()) return fail(TO_ERR_ARGUMENT, "cost id out of range");
  if (!q && !r) return fail(TO_ERR_NULL, "null pointer");
  const to_cost_desc& c = h->costs[id];
  if (c.kind == TO_COST_ERROR_QUADRATIC) return fail(TO_ERR_UNSUPPORTED, "per-trajectory linear terms: not for ErrorQuadratic (its q slot carries x_ref)");
  DevProblem& P = h->a.P;
  const int n = P.n, m = P.m, nz = n + m, L = (int)h->costs.size() * nz, B = P.B;
  if (!h->d_gl) TRY(dev_alloc(h, &h->d_gl, (size_t)L * P.Bp));  // zero-filled: every cost starts with its descriptor's terms
  std::vector<double> delta;
  if (q) {  // stored as the difference from the descriptor's q: the kernels ADD it to what the shared code path computes
    delta.resize((size_t)n * B);
    for (int b = 0; b < B; ++b) for (int i = 0; i < n; ++i) delta[i + (size_t)n * b] = q[i + (size_t)n * b] - c.q[i];
    TRY(upload_vec(h, delta.data(), h->d_gl, n, L, id * nz));
  }
  if (r) {
    delta.resize((size_t)m * B);
    for (int b = 0; b < B; ++b) for (int j = 0; j < m; ++j) delta[j + (size_t)m * b] = r[j + (size_t)m * b] - c.r[j];
    TRY(upload_vec(h, delta.data(), h->d_gl, m, L, id * nz + n));
  }
  P.gl = h->d_gl;
  h->gl_set.resize(h->costs.size(), 0);
  h->gl_set[id] = 1;
  return TO_OK;
}
int to_clear_cost_linear_batch(to_handle* h) {
  CHECK_H(h); CHECK_IDLE(h); TRY(use_device(h));
  if (h->d_gl) {
    HIPCHECK(hipMemsetAsync(h->d_gl, 0, sizeof(double) * h->costs.size() * (h->a.P.n + h->a.P.m) * h->a.P.Bp, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
  }
  h->a.P.gl = nullptr;
  h->gl_set.assign(h->costs.size(), 0);
  return TO_OK;
}
int to_set_constraint(to_handle* h, int32_t id, const to_constraint_desc* c) {
  CHECK_H(h); CHECK_IDLE(h); CHECK_P(c); TRY(use_device(h));
  if (id < 0 || id >= (int)h->cons.size()) return fail(TO_ERR_ARGUMENT, "constraint id out of range");
  DevCon ci;
  TRY(validate_constraint(h->a.P.n, h->a.P.m, h->a.P.N, *c, &ci));
  const DevCon& old = h->cons[id];
  if (ci.p != old.p || ci.k1 != old.k1 || ci.k2 != old.k2) return fail(TO_ERR_DIMENSION_MISMATCH, "replacement constraint must keep p and the knot range");
  ci.dual_off = old.dual_off;
  h->cons[id] = ci;  // (cp_off = -1: the replaced constraint starts on shared parameters again)
  bool any = false;
  for (const DevCon& c : h->cons) any = any || c.cp_off >= 0;
  if (!any) h->a.P.cp = nullptr;
  return upload_tables(h);
}
// One parameter set per TRAJECTORY for constraint con_id.  GOAL: params[p, B] = xf_b[inds]; LINEAR: params[p, B] = b_b.  Stored as the shift of
// z = [x; u] the constraint sees (DevProblem::cp): the GoalConstraint with target xf + d is the shared one evaluated at x - d, the
// LinearConstraint with b + db the shared one at z - A^+ db.
int to_set_constraint_params_batch(to_handle* h, int32_t id, const double* params) {
  CHECK_H(h); CHECK_IDLE(h); CHECK_P(params); TRY(use_device(h));
  if (id < 0 || id >= (int)h->cons.size()) return fail(TO_ERR_ARGUMENT, "constraint id out of range");
  DevCon& ci = h->cons[id];
  if (ci.d.kind != TO_CON_GOAL && ci.d.kind != TO_CON_LINEAR)
    return fail(TO_ERR_UNSUPPORTED, "per-trajectory constraint parameters: GoalConstraint (its target) and LinearConstraint (its b) only");
  DevProblem& P = h->a.P;
  const int n = P.n, nz = P.n + P.m, B = P.B, p = ci.p, L = nz * (int)h->cons.size();
  std::vector<double> shift((size_t)nz * B, 0.0);
  if (ci.d.kind == TO_CON_GOAL) {
    for (int b = 0; b < B; ++b)
      for (int r = 0; r < p; ++r) shift[(size_t)(ci.d.inds[r] - 1) + (size_t)nz * b] = params[r + (size_t)p * b] - ci.d.params[r];
  } else {
    // A z[inds] - (b + db) = A (z[inds] - dz) - b with the minimum-norm dz = A'(A A')^-1 db: needs linearly independent rows
    const int D = ci.d.n_inds;
    const double* A = ci.d.params;          // p x D, column-major
    const double* b0 = ci.d.params + (size_t)p * D;
    std::vector<double> G((size_t)p * p, 0.0), Lc((size_t)p * p, 0.0), w(p), y(p);
    double gmax = 0.0;
    for (int i = 0; i < p; ++i)
      for (int j = 0; j < p; ++j) { double t = 0.0; for (int c = 0; c < D; ++c) t += A[i + (size_t)p * c] * A[j + (size_t)p * c]; G[i + (size_t)p * j] = t; if (i == j) gmax = std::fmax(gmax, t); }
    for (int j = 0; j < p; ++j) {  // Cholesky of A A'
      double d = G[j + (size_t)p * j];
      for (int k = 0; k < j; ++k) d -= Lc[j + (size_t)p * k] * Lc[j + (size_t)p * k];
      if (!(d > 1e-12 * gmax)) return fail(TO_ERR_UNSUPPORTED, "per-trajectory b of a LinearConstraint: the rows of A must be linearly independent");
      Lc[j + (size_t)p * j] = std::sqrt(d);
      for (int i = j + 1; i < p; ++i) {
        double t = G[i + (size_t)p * j];
        for (int k = 0; k < j; ++k) t -= Lc[i + (size_t)p * k] * Lc[j + (size_t)p * k];
        Lc[i + (size_t)p * j] = t / Lc[j + (size_t)p * j];
      }
    }
    for (int b = 0; b < B; ++b) {
      for (int i = 0; i < p; ++i) { double t = params[i + (size_t)p * b] - b0[i]; for (int k = 0; k < i; ++k) t -= Lc[i + (size_t)p * k] * y[k]; y[i] = t / Lc[i + (size_t)p * i]; }
      for (int i = p - 1; i >= 0; --i) { double t = y[i]; for (int k = i + 1; k < p; ++k) t -= Lc[k + (size_t)p * i] * w[k]; w[i] = t / Lc[i + (size_t)p * i]; }
      for (int c = 0; c < D; ++c) { double t = 0.0; for (int i = 0; i < p; ++i) t += A[i + (size_t)p * c] * w[i]; shift[(size_t)(ci.d.inds[c] - 1) + (size_t)nz * b] += t; }
    }
    (void)n;
  }
  if (!h->d_cp) TRY(dev_alloc(h, &h->d_cp, (size_t)L * P.Bp));  // zero-filled
  TRY(upload_vec(h, shift.data(), h->d_cp, nz, L, id * nz));
  ci.cp_off = id * nz;
  P.cp = h->d_cp; P.n_cp = L;
  return upload_tables(h);
}
int to_clear_constraint_params_batch(to_handle* h) {
  CHECK_H(h); CHECK_IDLE(h); TRY(use_device(h));
  for (DevCon& c : h->cons) c.cp_off = -1;
  h->a.P.cp = nullptr;
  return upload_tables(h);
}

int to_rollout(to_handle* h) { CHECK_H(h); CHECK_IDLE(h); TRY(use_device(h)); TRY(launch_rollout(h)); HIPCHECK(hipStreamSynchronize(h->stream)); return check_guards(h, "to_rollout"); }
int to_cost(to_handle* h, double* J) {
  CHECK_H(h); CHECK_IDLE(h); CHECK_P(J); TRY(use_device(h));
  TRY(launch_cost(h, 0, h->d_tmp, nullptr));
  return download_scalar(h, J, h->d_tmp);
}
int to_al_cost(to_handle* h, double* J) {
  CHECK_H(h); CHECK_IDLE(h); CHECK_P(J); TRY(use_device(h));
  TRY(launch_cost(h, 1, h->d_tmp, nullptr));
  return download_scalar(h, J, h->d_tmp);
}
int to_stage_costs(to_handle* h, double* Jk) {
  CHECK_H(h); CHECK_IDLE(h); CHECK_P(Jk); TRY(use_device(h));
  const DevProblem& P = h->a.P;
  // per-knot values land in a tiled array (L = N) in the upper half of the staging buffer, then transpose to host (N,B)
  const size_t cnt = (size_t)P.N * P.Bp;
  TRY(ensure_stage(h, 2 * cnt * sizeof(double)));
  double* dJk = h->stage + cnt;
  TRY(launch_cost(h, 0, nullptr, dJk));
  hipLaunchKernelGGL(k_to_host, grid_b(h, P.N), dim3(BLOCK), 0, h->stream, dJk, h->stage, P.N, 0, P.N, P.B);
  HIPCHECK(hipGetLastError());
  HIPCHECK(hipMemcpyAsync(Jk, h->stage, sizeof(double) * P.N * P.B, hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(hipStreamSynchronize(h->stream));
  return TO_OK;
}
int to_expand(to_handle* h) {
  CHECK_H(h); CHECK_IDLE(h); TRY(use_device(h));
  TRY(launch_set_active(h, 1)); TRY(launch_expand(h));
  HIPCHECK(hipStreamSynchronize(h->stream));
  return check_guards(h, "to_expand");
}
int to_backward(to_handle* h) {
  CHECK_H(h); CHECK_IDLE(h); TRY(use_device(h));
  TRY(launch_set_active(h, 1));
  const DevProblem& P = h->a.P;
  if (h->scan == 2 && h->a.h_diag && P.expand_variant == 0) {
    // TRAJOPT_SCAN=2 (tests): the phase API runs the solve loop's scan kernel so that its gains can be read back and compared
    // (it expands on its own: to_expand's arrays are not used)
    TRY(h->ops->expand_backward_scan(h));
  } else TRY(launch_backward(h));
  HIPCHECK(hipStreamSynchronize(h->stream));
  return check_guards(h, "to_backward");
}
int to_forward(to_handle* h, int32_t* ls_index, double* J_new) {
  CHECK_H(h); CHECK_IDLE(h); TRY(use_device(h));
  h->a.control = 0;
  // keep bpfail from a preceding to_backward: set active without clearing it
  TRY(launch_cost(h, 1, h->a.J, nullptr));
  TRY(launch_set_active(h, 1, 2));
  h->a.step = 0;
  TRY(launch_forward(h));
  TRY(download_int(h, ls_index, h->a.ls_index));
  TRY(download_scalar(h, J_new, h->a.Jout));
  HIPCHECK(hipStreamSynchronize(h->stream));
  return check_guards(h, "to_forward");
}
int to_ilqr_solve(to_handle* h, to_solve_stats* st) { CHECK_H(h); CHECK_IDLE(h); return solve(h, st, 0); }
int to_al_solve(to_handle* h, to_solve_stats* st) { CHECK_H(h); CHECK_IDLE(h); return solve(h, st, 1); }
int to_pn_solve(to_handle* h, to_solve_stats* st) { CHECK_H(h); CHECK_IDLE(h); return pn_solve(h, st); }
int to_altro_solve(to_handle* h, to_solve_stats* st) { CHECK_H(h); CHECK_IDLE(h); return altro_solve(h, st); }
// Asynchronous solves: the solve loop — enqueue chunks of batch steps, read the activity counters back — runs on
// a worker thread owned by the handle; the handle's stream carries the kernels as before.  Two handles (two streams) solved this
// way overlap on the device: the drained tail of one batch leaves most of the chip to the other.
static int solve_async(to_handle* h, to_solve_stats* st, int kind) {
  CHECK_H(h);
  if (h->inflight) return fail(TO_ERR_ARGUMENT, "a solve is already in flight on this handle (call to_solve_wait first)");
  h->inflight = true;
  h->async_rc = TO_OK;
  h->prog_active = h->a.P.B; h->prog_steps = 0;  // (before the worker exists: a to_solve_wait_below right behind this call must block)
  try {  // no exception may cross the C ABI (std::system_error: out of threads)
    h->worker = std::thread([h, st, kind] {
      h->async_rc = kind == 0 ? solve(h, st, 0) : kind == 1 ? solve(h, st, 1) : altro_solve(h, st);
      h->async_err = g_err;
    });
  } catch (const std::exception& e) {
    h->inflight = false;
    h->prog_active = 0;
    return fail(TO_ERR_HIP, std::string("could not start the solve thread: ") + e.what());
  }
  return TO_OK;
}
int to_ilqr_solve_async(to_handle* h, to_solve_stats* st) { return solve_async(h, st, 0); }
int to_al_solve_async(to_handle* h, to_solve_stats* st) { return solve_async(h, st, 1); }
int to_altro_solve_async(to_handle* h, to_solve_stats* st) { return solve_async(h, st, 2); }
int to_solve_wait(to_handle* h) {
  CHECK_H(h);
  if (!h->inflight) return TO_OK;
  h->worker.join();
  h->inflight = false;
  if (h->async_rc != TO_OK) g_err = h->async_err;
  return h->async_rc;
}
int to_solve_progress(to_handle* h, int32_t* active, int32_t* batch_steps, int32_t* in_flight) {
  CHECK_H(h);
  if (active) *active = h->prog_active.load();
  if (batch_steps) *batch_steps = h->prog_steps.load();
  if (in_flight) *in_flight = h->inflight ? 1 : 0;
  return TO_OK;
}
int to_solve_wait_below(to_handle* h, int32_t active_max) {
  CHECK_H(h);
  if (active_max < 0) return fail(TO_ERR_ARGUMENT, "to_solve_wait_below: the threshold must be >= 0");
  std::unique_lock<std::mutex> lk(h->prog_mu);
  h->prog_cv.wait(lk, [&] { return h->prog_active.load() <= active_max; });
  return TO_OK;
}
int to_dynamics_defect(to_handle* h, double* defect) {
  CHECK_H(h); CHECK_IDLE(h); CHECK_P(defect);
  TRY(use_device(h));
  TRY(h->ops->defect(h, h->d_tmp));
  return download_scalar(h, defect, h->d_tmp);
}

// expansion blocks -> host column-major blocks: h[r + Rr*(c + Cc*(k + K*b))] = X_k[row0+r][col0+c], rows / columns counted
// in error-state directions followed by control directions (whichever layout the backward pass uses)
enum { BLK_M = 0, BLK_H = 1 };
static int download_block(to_handle* h, double* host, int which, int row0, int Rr, int col0, int Cc, double* dev_out = nullptr) {
  if (!host && !dev_out) return TO_OK;
  const DevProblem& P = h->a.P;
  const int K = which == BLK_M ? P.N - 1 : P.N;
  const size_t cnt = (size_t)Rr * Cc * K * P.B;
  if (!dev_out) TRY(ensure_stage(h, cnt * sizeof(double)));
  double* const out = dev_out ? dev_out : h->stage;  // (dev_out: a caller-owned device buffer, no host copy)
  if (h->a.bwd_lane) {
    const int nc = P.ne + P.m;
    if (which == BLK_M)
      hipLaunchKernelGGL(k_lane_to_host, grid_b(h, K * Rr * Cc), dim3(BLOCK), 0, h->stream, h->a.Mc, out, P.ne * nc, 0, nc, row0, Rr, col0, Cc, K, P.B);
    else
      hipLaunchKernelGGL(k_lane_to_host, grid_b(h, K * Rr * Cc), dim3(BLOCK), 0, h->stream, h->a.Hc, out, nc * (nc + 1) / 2, 1, nc, row0, Rr, col0, Cc, K, P.B);
  } else if (h->a.bwd_mfma) {
    const int nep = h->ops->nep, rs = h->ops->rs;
    auto tix = [&](int i) { return i < P.ne ? i : nep + (i - P.ne); };  // tangent index (control directions start at NEP)
    const bool compact = which == BLK_H && h->a.h_compact;
    hipLaunchKernelGGL(k_tm_to_host, grid_b(h, K * Rr * Cc), dim3(BLOCK), 0, h->stream, which == BLK_M ? h->a.Mt : h->a.Ht, out,
                       which == BLK_M ? rs : rs + 1, tix(row0), Rr, tix(col0), Cc, K, P.B, compact ? h->d_crow : nullptr);
  } else {
    const int nc = P.ne + P.m;
    const int diag = (which == BLK_H && h->a.h_diag) ? 1 : 0;
    hipLaunchKernelGGL(k_col_to_host, grid_b(h, K * Rr * Cc), dim3(BLOCK), 0, h->stream, which == BLK_M ? h->a.Mc : h->a.Hc, out,
                       which == BLK_M ? (P.N - 1) * P.ne : (diag ? P.N : P.N * nc), which == BLK_M ? P.ne : nc, row0, Rr, col0, Cc, K, P.B,
                       h->R, h->G, diag);
  }
  HIPCHECK(hipGetLastError());
  if (dev_out) return TO_OK;
  HIPCHECK(hipMemcpyAsync(host, h->stage, cnt * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(hipStreamSynchronize(h->stream));
  return TO_OK;
}
static int download_gradient(to_handle* h, double* host, int col0, int Cc, double* dev_out = nullptr) {
  if (!host && !dev_out) return TO_OK;
  const DevProblem& P = h->a.P;
  const size_t cnt = (size_t)Cc * P.N * P.B;
  if (!dev_out) TRY(ensure_stage(h, cnt * sizeof(double)));
  double* const out = dev_out ? dev_out : h->stage;
  if (h->a.bwd_lane) {
    const int nc = P.ne + P.m;
    hipLaunchKernelGGL(k_lane_to_host, grid_b(h, P.N * Cc), dim3(BLOCK), 0, h->stream, h->a.gc, out, nc, 0, nc, 0, 1, col0, Cc, P.N, P.B);
  } else if (h->a.bwd_mfma) {
    const int tcol = col0 < P.ne ? col0 : h->ops->nep + (col0 - P.ne);
    hipLaunchKernelGGL(k_tmvec_to_host, grid_b(h, P.N * Cc), dim3(BLOCK), 0, h->stream, h->a.gt, out, tcol, Cc, P.N, P.B);
  } else {
    hipLaunchKernelGGL(k_col_to_host, grid_b(h, P.N * Cc), dim3(BLOCK), 0, h->stream, h->a.gc, out, P.N, 1, 0, 1, col0, Cc, P.N, P.B, h->R, h->G, 0);
  }
  HIPCHECK(hipGetLastError());
  if (dev_out) return TO_OK;
  HIPCHECK(hipMemcpyAsync(host, h->stage, cnt * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(hipStreamSynchronize(h->stream));
  return TO_OK;
}
int to_get_dynamics_jacobians(to_handle* h, double* A, double* Bm) {
  CHECK_H(h); CHECK_IDLE(h); TRY(use_device(h));
  const DevProblem& P = h->a.P;
  TRY(download_block(h, A, BLK_M, 0, P.ne, 0, P.ne));
  TRY(download_block(h, Bm, BLK_M, 0, P.ne, P.ne, P.m));
  return TO_OK;
}
int to_get_cost_expansion(to_handle* h, double* Qxx, double* Quu, double* Qux, double* qx, double* qu) {
  CHECK_H(h); CHECK_IDLE(h); TRY(use_device(h));
  const DevProblem& P = h->a.P;
  TRY(download_block(h, Qxx, BLK_H, 0, P.ne, 0, P.ne));
  TRY(download_block(h, Quu, BLK_H, P.ne, P.m, P.ne, P.m));
  TRY(download_block(h, Qux, BLK_H, P.ne, P.m, 0, P.ne));
  TRY(download_gradient(h, qx, 0, P.ne));
  TRY(download_gradient(h, qu, P.ne, P.m));
  return TO_OK;
}
int to_get_gains(to_handle* h, double* K, double* d, double* dV, double* rho) {
  CHECK_H(h); CHECK_IDLE(h); TRY(use_device(h));
  const DevProblem& P = h->a.P;
  if (K || d) {
    const size_t nK = (size_t)P.m * P.ne * (P.N - 1) * P.B, nd = (size_t)P.m * (P.N - 1) * P.B;
    TRY(ensure_stage(h, (nK + nd) * sizeof(double)));
    double *dK = h->stage, *dd = h->stage + nK;
    hipLaunchKernelGGL(k_gains_to_host, grid_b(h, (P.N - 1) * P.m * (P.ne + 1)), dim3(BLOCK), 0, h->stream, h->a.Kt, dK, dd, P.m, P.ne, P.N - 1, P.B);
    HIPCHECK(hipGetLastError());
    if (K) HIPCHECK(hipMemcpyAsync(K, dK, nK * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (d) HIPCHECK(hipMemcpyAsync(d, dd, nd * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
  }
  if (dV) {  // plain [2][Bp] -> host (2, B)
    std::vector<double> tmp(2 * (size_t)P.Bp);
    HIPCHECK(hipMemcpyAsync(tmp.data(), h->a.dV, tmp.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
    for (int b = 0; b < P.B; ++b) { dV[2 * b] = tmp[b]; dV[2 * b + 1] = tmp[(size_t)P.Bp + b]; }
  }
  TRY(download_scalar(h, rho, h->a.rho));
  return TO_OK;
}
int to_get_cost_to_go(to_handle* h, double* S, double* s) {
  CHECK_H(h); CHECK_IDLE(h); TRY(use_device(h));
  if (!S && !s) return fail(TO_ERR_NULL, "null pointer");
  const DevProblem& P = h->a.P;
  const size_t ne = P.ne, m = P.m, N = P.N, B = P.B;
  // every block in its getter layout, side by side in one device buffer, then the recursion (k_generic.h k_cost_to_go)
  const size_t nA = ne * ne * (N - 1) * B, nB = ne * m * (N - 1) * B, nxx = ne * ne * N * B, nuu = m * m * N * B, nux = m * ne * N * B,
               nx = ne * N * B, nu = m * N * B, nK = m * ne * (N - 1) * B, nd = m * (N - 1) * B;
  struct DevBuf { void* p = nullptr; ~DevBuf() { if (p) hipFree(p); } } buf;
  HIPCHECK(hipMalloc(&buf.p, (nA + nB + nxx + nuu + nux + nx + nu + nK + nd + nxx + nx) * sizeof(double)));
  double* dA = (double*)buf.p; double* dB = dA + nA; double* dxx = dB + nB; double* duu = dxx + nxx; double* dux = duu + nuu;
  double* dx = dux + nux; double* du = dx + nx; double* dK = du + nu; double* dd = dK + nK; double* dS = dd + nd; double* ds = dS + nxx;
  TRY(download_block(h, nullptr, BLK_M, 0, P.ne, 0, P.ne, dA));
  TRY(download_block(h, nullptr, BLK_M, 0, P.ne, P.ne, P.m, dB));
  TRY(download_block(h, nullptr, BLK_H, 0, P.ne, 0, P.ne, dxx));
  TRY(download_block(h, nullptr, BLK_H, P.ne, P.m, P.ne, P.m, duu));
  TRY(download_block(h, nullptr, BLK_H, P.ne, P.m, 0, P.ne, dux));
  TRY(download_gradient(h, nullptr, 0, P.ne, dx));
  TRY(download_gradient(h, nullptr, P.ne, P.m, du));
  hipLaunchKernelGGL(k_gains_to_host, grid_b(h, (P.N - 1) * P.m * (P.ne + 1)), dim3(BLOCK), 0, h->stream, h->a.Kt, dK, dd, P.m, P.ne, P.N - 1, P.B);
  hipLaunchKernelGGL(k_cost_to_go, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, h->stream, dA, dB, dxx, duu, dux, dx, du, dK, dd, dS, ds, P.ne, P.m, P.N, P.B);
  HIPCHECK(hipGetLastError());
  if (S) HIPCHECK(hipMemcpyAsync(S, dS, nxx * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (s) HIPCHECK(hipMemcpyAsync(s, ds, nx * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(hipStreamSynchronize(h->stream));
  return TO_OK;
}
int to_cost_expansion(to_handle* h, double* grad, double* hess) {
  CHECK_H(h); CHECK_IDLE(h); TRY(use_device(h));
  const DevProblem& P = h->a.P;
  const size_t nz = P.n + P.m, ng = nz * P.N * P.B, nh = nz * nz * P.N * P.B;
  TRY(ensure_stage(h, (ng + nh) * sizeof(double)));
  double* dg = h->stage; double* dh = h->stage + ng;
  TRY(h->ops->cost_derivs(h, dg, dh));
  if (grad) HIPCHECK(hipMemcpyAsync(grad, dg, ng * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (hess) HIPCHECK(hipMemcpyAsync(hess, dh, nh * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(hipStreamSynchronize(h->stream));
  return TO_OK;
}
int to_discrete_jacobian(to_handle* h, double* F) {
  CHECK_H(h); CHECK_IDLE(h); CHECK_P(F); TRY(use_device(h));
  const DevProblem& P = h->a.P;
  const size_t cnt = (size_t)P.n * (P.n + P.m) * (P.N - 1) * P.B;
  TRY(ensure_stage(h, cnt * sizeof(double)));
  TRY(h->ops->discrete_jacobian(h, h->stage));
  HIPCHECK(hipMemcpyAsync(F, h->stage, cnt * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(hipStreamSynchronize(h->stream));
  return TO_OK;
}

int to_constraint_info(const to_handle* h, int32_t id, int32_t* p, int32_t* width, int32_t* nk, int32_t* sense) {
  CHECK_H(h);
  if (id < 0 || id >= (int)h->cons.size()) return fail(TO_ERR_ARGUMENT, "constraint id out of range");
  const DevCon& ci = h->cons[id];
  if (p) *p = ci.p; if (width) *width = ci.width; if (nk) *nk = ci.k2 - ci.k1 + 1; if (sense) *sense = ci.d.sense;
  return TO_OK;
}
static int constraint_eval(to_handle* h, int32_t id, double* vals, double* jac) {
  TRY(use_device(h));
  if (id < 0 || id >= (int)h->cons.size()) return fail(TO_ERR_ARGUMENT, "constraint id out of range");
  const DevCon& ci = h->cons[id];
  const DevProblem& P = h->a.P;
  const int nk = ci.k2 - ci.k1 + 1;
  const size_t nv = (size_t)ci.p * nk * P.B, nj = (size_t)ci.p * ci.width * nk * P.B;
  TRY(ensure_stage(h, (nv + nj) * sizeof(double)));
  double* dv = h->stage; double* dj = h->stage + nv;
  TRY(h->ops->constraint_eval(h, (int)id, vals ? dv : nullptr, jac ? dj : nullptr));
  if (vals) HIPCHECK(hipMemcpyAsync(vals, dv, nv * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  if (jac) HIPCHECK(hipMemcpyAsync(jac, dj, nj * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(hipStreamSynchronize(h->stream));
  return TO_OK;
}
int to_evaluate_constraints(to_handle* h, int32_t id, double* vals) { CHECK_H(h); CHECK_IDLE(h); CHECK_P(vals); return constraint_eval(h, id, vals, nullptr); }
int to_constraint_jacobians(to_handle* h, int32_t id, double* jac) { CHECK_H(h); CHECK_IDLE(h); CHECK_P(jac); return constraint_eval(h, id, nullptr, jac); }
int to_constraint_hessians(to_handle* h, int32_t id, const double* lambda, double* H) {
  CHECK_H(h); CHECK_IDLE(h); CHECK_P(lambda); CHECK_P(H); TRY(use_device(h));
  if (id < 0 || id >= (int)h->cons.size()) return fail(TO_ERR_ARGUMENT, "constraint id out of range");
  const DevCon& ci = h->cons[id];
  const DevProblem& P = h->a.P;
  const int nk = ci.k2 - ci.k1 + 1;
  const size_t nl = (size_t)ci.p * nk * P.B, nh = (size_t)ci.width * ci.width * nk * P.B;
  TRY(ensure_stage(h, (nl + nh) * sizeof(double)));
  double* dl = h->stage; double* dh = h->stage + nl;
  HIPCHECK(hipMemcpyAsync(dl, lambda, nl * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHECK(hipMemcpyAsync(dh, H, nh * sizeof(double), hipMemcpyHostToDevice, h->stream));  // the operator ADDS (src/abstract_constraint.jl:255-266)
  TRY(h->ops->constraint_hessian(h, (int)id, dl, dh));
  HIPCHECK(hipMemcpyAsync(H, dh, nh * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIPCHECK(hipStreamSynchronize(h->stream));
  return TO_OK;
}
int to_max_violation(to_handle* h, double* c_max) {
  CHECK_H(h); CHECK_IDLE(h); CHECK_P(c_max); TRY(use_device(h));
  if (h->a.P.n_cons == 0) { std::memset(c_max, 0, sizeof(double) * h->a.P.B); return TO_OK; }
  TRY(launch_violation(h, h->d_tmp));
  return download_scalar(h, c_max, h->d_tmp);
}
int to_get_duals(to_handle* h, int32_t id, double* lambda, double* mu) {
  CHECK_H(h); CHECK_IDLE(h); TRY(use_device(h));
  if (id < 0 || id >= (int)h->cons.size()) return fail(TO_ERR_ARGUMENT, "constraint id out of range");
  const DevCon& ci = h->cons[id];
  const DevProblem& P = h->a.P;
  if (lambda) TRY(download_vec(h, lambda, h->a.lam, ci.p * (ci.k2 - ci.k1 + 1), (int)P.n_duals, (int)ci.dual_off));
  if (mu) TRY(download_vec(h, mu, h->a.mu, 1, P.n_cons, id));
  return TO_OK;
}
int to_set_duals(to_handle* h, int32_t id, const double* lambda, const double* mu) {
  CHECK_H(h); CHECK_IDLE(h); TRY(use_device(h));
  if (id < 0 || id >= (int)h->cons.size()) return fail(TO_ERR_ARGUMENT, "constraint id out of range");
  const DevCon& ci = h->cons[id];
  const DevProblem& P = h->a.P;
  if (lambda) TRY(upload_vec(h, lambda, h->a.lam, ci.p * (ci.k2 - ci.k1 + 1), (int)P.n_duals, (int)ci.dual_off));
  if (mu) TRY(upload_vec(h, mu, h->a.mu, 1, P.n_cons, id));
  return TO_OK;
}
int to_reset_duals(to_handle* h) {
  CHECK_H(h); CHECK_IDLE(h); TRY(use_device(h));
  const DevProblem& P = h->a.P;
  if (P.n_cons == 0) return TO_OK;
  HIPCHECK(hipMemsetAsync(h->a.lam, 0, sizeof(double) * (size_t)P.n_duals * P.Bp, h->stream));
  std::vector<double> mu((size_t)P.n_cons * P.Bp, P.opts.penalty_initial);
  HIPCHECK(hipMemcpyAsync(h->a.mu, mu.data(), mu.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHECK(hipStreamSynchronize(h->stream));
  return TO_OK;
}
int to_dual_update(to_handle* h) {
  CHECK_H(h); CHECK_IDLE(h); TRY(use_device(h));
  if (h->a.P.n_cons == 0) return TO_OK;
  TRY(h->ops->dual_update(h));
  HIPCHECK(hipStreamSynchronize(h->stream));
  return TO_OK;
}

// ---- multi-GPU: the batch shards across ranks as independent units; the only collective is one RCCL
// all-gather of the converged trajectories per solve.  librccl.so is loaded on first use (dlopen), so single-GPU hosts
// need no RCCL at all; a host in ANY language (Julia ccall, Python ctypes) exchanges the 128-byte id out of band
// (MPI / a file / torch.distributed) exactly as with ncclGetUniqueId.
int to_comm_unique_id(void* id128) {
  CHECK_P(id128);
  TRY(load_rccl());
  RCCLCHECK(g_rccl.GetUniqueId(id128));
  return TO_OK;
}
int to_comm_init_rank(to_handle* h, int32_t nranks, int32_t rank, const void* id128) {
  CHECK_H(h); CHECK_IDLE(h); CHECK_P(id128);
  if (nranks < 1 || rank < 0 || rank >= nranks) return fail(TO_ERR_ARGUMENT, "rank outside 0..nranks-1");
  if (h->comm) return fail(TO_ERR_ARGUMENT, "communicator already initialised (to_comm_destroy first)");
  TRY(load_rccl());
  TRY(use_device(h));
  Rccl::Id id;
  std::memcpy(id.b, id128, sizeof(id.b));
  RCCLCHECK(g_rccl.CommInitRank(&h->comm, nranks, id, rank));
  h->comm_rank = rank; h->comm_size = nranks;
  // every error from here on leaves the handle without a communicator (a half-initialised one would let a later to_allgather
  // run against peers that have already given up)
  auto abandon = [&](int code, const char* msg) {
    g_rccl.CommDestroy(h->comm);
    h->comm = nullptr; h->comm_rank = 0; h->comm_size = 1; h->comm_counts.clear();
    h->comm_offset = 0; h->comm_total = 0; h->comm_equal = true;
    return fail(code, msg);
  };
  // shard sizes of every rank (they may differ: a batch that does not divide by the number of GPUs)
  int32_t* dcnt = nullptr;
  if (hipMalloc((void**)&dcnt, sizeof(int32_t) * nranks) != hipSuccess) return abandon(TO_ERR_HIP, "to_comm_init_rank: hipMalloc failed");
  const int32_t mine = h->a.P.B;
  hipError_t e1 = hipMemcpyAsync(dcnt + rank, &mine, sizeof(int32_t), hipMemcpyHostToDevice, h->stream);
  int rc = e1 == hipSuccess ? g_rccl.AllGather(dcnt + rank, dcnt, 1, kNcclInt32, h->comm, h->stream) : 0;
  h->comm_counts.assign(nranks, 0);
  hipError_t e2 = hipMemcpyAsync(h->comm_counts.data(), dcnt, sizeof(int32_t) * nranks, hipMemcpyDeviceToHost, h->stream);
  hipError_t e3 = hipStreamSynchronize(h->stream);
  hipFree(dcnt);
  if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess || rc != 0)
    return abandon(TO_ERR_HIP, "to_comm_init_rank: exchanging the shard sizes failed");
  h->comm_offset = 0; h->comm_total = 0; h->comm_equal = true;
  for (int r = 0; r < nranks; ++r) {
    if (h->comm_counts[r] < 1) return abandon(TO_ERR_ARGUMENT, "to_comm_init_rank: a rank reported an empty shard");
    if (r < rank) h->comm_offset += h->comm_counts[r];
    h->comm_total += h->comm_counts[r];
    h->comm_equal = h->comm_equal && h->comm_counts[r] == h->comm_counts[0];
  }
  return TO_OK;
}
int to_comm_shards(const to_handle* h, int32_t* nranks, int32_t* rank, int64_t* B_total, int32_t* counts) {
  CHECK_H(h);
  if (!h->comm) return fail(TO_ERR_ARGUMENT, "to_comm_init_rank first");
  if (nranks) *nranks = h->comm_size;
  if (rank) *rank = h->comm_rank;
  if (B_total) *B_total = h->comm_total;
  if (counts) for (int r = 0; r < h->comm_size; ++r) counts[r] = h->comm_counts[r];
  return TO_OK;
}
int to_comm_destroy(to_handle* h) {
  CHECK_H(h); CHECK_IDLE(h);
  if (!h->comm) return TO_OK;
  TRY(use_device(h));
  HIPCHECK(hipStreamSynchronize(h->stream));
  RCCLCHECK(g_rccl.CommDestroy(h->comm));
  h->comm = nullptr; h->comm_rank = 0; h->comm_size = 1; h->comm_counts.clear();
  return TO_OK;
}
// Every rank's block, gathered in rank order into `all` (count[r] * per doubles / int32 from rank r at offset[r] * per):
// one in-place ncclAllGather when the shards are equal, else one grouped ncclBroadcast per rank (same bytes on the wire).
static int gather_blocks(to_handle* h, void* all, size_t per, int dtype, size_t elt) {
  char* base = (char*)all;
  char* mine = base + (size_t)h->comm_offset * per * elt;
  if (h->comm_equal) {
    RCCLCHECK(g_rccl.AllGather(mine, all, (size_t)h->a.P.B * per, dtype, h->comm, h->stream));
    return TO_OK;
  }
  RCCLCHECK(g_rccl.GroupStart());
  size_t off = 0;
  for (int r = 0; r < h->comm_size; ++r) {
    char* blk = base + off * per * elt;
    const int rc = g_rccl.Broadcast(blk, blk, (size_t)h->comm_counts[r] * per, dtype, r, h->comm, h->stream);
    if (rc != 0) { g_rccl.GroupEnd(); return fail(TO_ERR_HIP, std::string("ncclBroadcast: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "rccl error")); }
    off += (size_t)h->comm_counts[r];
  }
  RCCLCHECK(g_rccl.GroupEnd());
  return TO_OK;
}
// dX_all / dU_all: caller-owned DEVICE buffers of n*N*B_total and m*(N-1)*B_total doubles (B_total = sum of the ranks' shard
// sizes, to_comm_shards); on return (the call synchronises the handle's stream) they hold every rank's trajectories in host
// layout (n, N, B_total), rank-major = global trajectory order.  Shards may differ in size.
int to_allgather(to_handle* h, void* dX_all, void* dU_all) {
  CHECK_H(h); CHECK_IDLE(h);
  if (!dX_all && !dU_all) return fail(TO_ERR_NULL, "null pointer");
  if (!h->comm) return fail(TO_ERR_ARGUMENT, "to_comm_init_rank first");
  TRY(use_device(h));
  const DevProblem& P = h->a.P;
  const size_t px = (size_t)P.n * P.N, pu = (size_t)P.m * (P.N - 1);
  if (dX_all) {  // own shard written in place, then one in-place gather on the handle's stream
    double* mine = (double*)dX_all + px * (size_t)h->comm_offset;
    hipLaunchKernelGGL(k_to_host, grid_b(h, P.n * P.N), dim3(BLOCK), 0, h->stream, h->a.Xs, mine, P.n * P.N, 0, P.n * P.N, P.B);
    HIPCHECK(hipGetLastError());
    TRY(gather_blocks(h, dX_all, px, kNcclDouble, sizeof(double)));
  }
  if (dU_all) {
    double* mine = (double*)dU_all + pu * (size_t)h->comm_offset;
    hipLaunchKernelGGL(k_to_host, grid_b(h, P.m * (P.N - 1)), dim3(BLOCK), 0, h->stream, h->a.Us, mine, P.m * (P.N - 1), 0, P.m * (P.N - 1), P.B);
    HIPCHECK(hipGetLastError());
    TRY(gather_blocks(h, dU_all, pu, kNcclDouble, sizeof(double)));
  }
  HIPCHECK(hipStreamSynchronize(h->stream));
  return TO_OK;
}
// The small gather SURVEY.md §8e names next to the trajectories: iterations[B_total], status[B_total] (int32) and the
// objective cost J[B_total] of every rank's CURRENT trajectories, rank-major, into caller-owned HOST arrays (any may be
// NULL).  Staged through a device buffer of the library (a few bytes per trajectory), gathered over the same communicator.
int to_allgather_stats(to_handle* h, int32_t* iterations_all, int32_t* status_all, double* J_all) {
  CHECK_H(h); CHECK_IDLE(h);
  if (!h->comm) return fail(TO_ERR_ARGUMENT, "to_comm_init_rank first");
  if (!iterations_all && !status_all && !J_all) return fail(TO_ERR_NULL, "null pointer");
  TRY(use_device(h));
  const DevProblem& P = h->a.P;
  const size_t T = (size_t)h->comm_total, off = (size_t)h->comm_offset, B = (size_t)P.B;
  struct DevBuf { void* p = nullptr; ~DevBuf() { if (p) hipFree(p); } } buf;
  HIPCHECK(hipMalloc(&buf.p, T * sizeof(double)));
  auto gather_int = [&](const int* src, int32_t* host) -> int {
    if (!host) return TO_OK;
    int32_t* all = (int32_t*)buf.p;
    HIPCHECK(hipMemcpyAsync(all + off, src, B * sizeof(int32_t), hipMemcpyDeviceToDevice, h->stream));
    TRY(gather_blocks(h, all, 1, kNcclInt32, sizeof(int32_t)));
    HIPCHECK(hipMemcpyAsync(host, all, T * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
    return TO_OK;
  };
  TRY(gather_int(h->a.iterations, iterations_all));
  TRY(gather_int(h->a.status, status_all));
  if (J_all) {
    double* all = (double*)buf.p;
    TRY(launch_cost(h, 0, h->d_tmp, nullptr));
    HIPCHECK(hipMemcpyAsync(all + off, h->d_tmp, B * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    TRY(gather_blocks(h, all, 1, kNcclDouble, sizeof(double)));
    HIPCHECK(hipMemcpyAsync(J_all, all, T * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIPCHECK(hipStreamSynchronize(h->stream));
  }
  return TO_OK;
}

// ---- cones: stateless batched ops on `device` ------------------------------------------------------
static int cone_op(int device, int which, int32_t cone, int32_t dim, int64_t count, const double* x, const double* b, double* out, int32_t* status) {
  CHECK_P(x); CHECK_P(out);
  if (dim < 1 || dim > TO_MAX_P) return fail(TO_ERR_ARGUMENT, "cone dimension out of range");
  if (cone < TO_CONE_ZERO || cone > TO_CONE_IDENTITY) return fail(TO_ERR_ARGUMENT, "unknown cone");
  if (count <= 0) return TO_OK;
  HIPCHECK(hipSetDevice(device));
  const size_t nx = (size_t)dim * count, no = which == 0 ? nx : nx * dim;
  struct DevBuf {  // freed on every exit path
    void* p = nullptr;
    ~DevBuf() { if (p) hipFree(p); }
  } bx, bb, bo, bs;
  HIPCHECK(hipMalloc(&bx.p, nx * sizeof(double)));
  HIPCHECK(hipMalloc(&bo.p, no * sizeof(double)));
  HIPCHECK(hipMalloc(&bs.p, count * sizeof(int)));
  double *dx = (double*)bx.p, *dout = (double*)bo.p, *db = nullptr;
  int* dst = (int*)bs.p;
  HIPCHECK(hipMemcpy(dx, x, nx * sizeof(double), hipMemcpyHostToDevice));
  if (which == 2) { HIPCHECK(hipMalloc(&bb.p, nx * sizeof(double))); db = (double*)bb.p; HIPCHECK(hipMemcpy(db, b, nx * sizeof(double), hipMemcpyHostToDevice)); }
  const unsigned blocks = (unsigned)((count + 255) / 256);
  if (which == 0) hipLaunchKernelGGL(k_cone_projection, dim3(blocks), dim3(256), 0, 0, cone, dim, (long long)count, dx, dout, dst);
  else if (which == 1) hipLaunchKernelGGL(k_cone_jacobian, dim3(blocks), dim3(256), 0, 0, cone, dim, (long long)count, dx, dout, dst);
  else hipLaunchKernelGGL(k_cone_hessian, dim3(blocks), dim3(256), 0, 0, cone, dim, (long long)count, dx, db, dout, dst);
  HIPCHECK(hipGetLastError());
  HIPCHECK(hipMemcpy(out, dout, no * sizeof(double), hipMemcpyDeviceToHost));
  std::vector<int> st(count);
  HIPCHECK(hipMemcpy(st.data(), dst, count * sizeof(int), hipMemcpyDeviceToHost));
  for (int64_t i = 0; i < count; ++i) {
    if (st[i] < 0) return fail(TO_ERR_CONE, "Invalid second-order cone projection");
    if (status) status[i] = st[i];
  }
  return TO_OK;
}
int to_cone_projection(int device, int32_t cone, int32_t dim, int64_t count, const double* x, double* px, int32_t* status) {
  return cone_op(device, 0, cone, dim, count, x, nullptr, px, status);
}
int to_cone_projection_jacobian(int device, int32_t cone, int32_t dim, int64_t count, const double* x, double* jac) {
  return cone_op(device, 1, cone, dim, count, x, nullptr, jac, nullptr);
}
int to_cone_projection_hessian(int device, int32_t cone, int32_t dim, int64_t count, const double* x, const double* b, double* hess) {
  CHECK_P(b);
  return cone_op(device, 2, cone, dim, count, x, b, hess, nullptr);
}

}  // extern "C"

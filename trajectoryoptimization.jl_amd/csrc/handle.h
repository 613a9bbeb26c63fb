// handle.h — host-side state behind a to_handle and the per-model launch table.  The kernels of one model are
// instantiated in their own translation units (ops_*.hip: the Quadrotor forward pass alone is a minute of compile
// time); trajopt_hip.hip holds the C-ABI and the model-independent kernels and reaches the rest through ModelOps.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "common.h"

namespace to {

int fail(int code, const std::string& msg);  // records the message for to_last_error(), returns code

#define HIPCHECK(expr)                                                                         \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess)                                                                      \
      return ::to::fail(TO_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));        \
  } while (0)
#define TRY(expr) do { int r_ = (expr); if (r_ != TO_OK) return r_; } while (0)

constexpr int BLOCK = 64;  // one wave per workgroup: a small batch is spread over as many CUs as possible

struct ModelOps;

}  // namespace to

struct to_handle_s {
  int device = 0;
  hipStream_t stream = nullptr;
  int model_key = -1;
  const to::ModelOps* ops = nullptr;
  int R = 0, G = 0;  // lanes per trajectory / trajectories per wave of the column-layout kernels
  to::KArgs a;       // host copy of the kernel argument block (device pointers inside)
  std::vector<to_cost_desc> costs;
  std::vector<to::DevCon> cons;
  std::vector<double> dt;
  std::vector<int> cost_index;
  std::vector<double> step_table;  // TO_MODEL_VECTOR: host copy of the per-step model table (models.h ModelVectorModel)
  std::vector<void*> allocs;
  double* stage = nullptr;  // device staging buffer in host layout
  size_t stage_bytes = 0;
  int* counter_host = nullptr;  // pinned
  int counter_len = 0;
  int cw_base = 1, tw_base = 64;  // forward-wave shape: base, and the deep one (0: none) used once the active trajectories fit
  int cw_deep = 0, tw_deep = 0, deep_max_active = 0;
  int fwd2 = 2;           // forward pass as two-wave workgroups (roller + accountant, k_forward2): 0 never, 1 always, 2 per step (TRAJOPT_FWD2)
  int simds = 1024;       // SIMDs of the device (4 per CU)
  int scan = 0;           // solve loop: scan backward pass (k_scan.h) ahead of the fused cooperative kernel while the active trajectories are few (TRAJOPT_SCAN=0/1)
  int scan_max_active = 0;
  int fused_coop = 0;     // solve loop, cooperative path with diagonal cost blocks: one k_expand_backward_coop launch (TRAJOPT_FUSED_COOP=0 to split)
  int fused_lane = 0;     // solve loop: one k_expand_backward_lane launch instead of expansion + backward pass (lane path; TRAJOPT_FUSED_LANE=0 to split)
  int compact = 0;        // solves run with active-list compaction (KArgs::compact; TRAJOPT_COMPACT=0 switches it off)
  int expand_pack = 1;    // packed tangent-matrix expansion of the quaternion rigid body (k_expand.h PACK; TRAJOPT_EXPAND_PACK=0: the 4 x 16 kernel)
  int expand_lane = 1;    // lane layout: expansion by k_expand_lane (one lane per (trajectory, knot)); 0 = column-per-lane kernel (A/B knob TRAJOPT_EXPAND_LANE)
  int roll_min_active = -1;  // solve loop: batch steps with at least this many active trajectories store candidate controls only and accept
                             // by k_accept_roll (-1: the measured default per solver, 0: never; TRAJOPT_ACCEPT_ROLL_MIN)
  int roll_min_small = 32768;  // ... the default of the small (write-through) models
  double roll_min_frac = 0.25;  // ... which also need at least this fraction of the batch active (TRAJOPT_ACCEPT_ROLL_FRAC)
  int ls2_cwa = 0, ls2_cwb = 2;  // two-launch line search (common.h ls_phase): step sizes per round of launch A (0: off) / launch B (TRAJOPT_LS_TWO=a,b)
  int ls2_blkA = 0, ls2_dump = 0;  // candidate blocks of launch A; the dump block behind launch B's
  int accept_chunks = 1;  // grid.z of k_accept (a chunk is >= 32 elements of [X; U]: the copy is latency-bound per wave)
  // device copies of the descriptor tables
  to_cost_desc* d_costs = nullptr;
  to::DevCon* d_cons = nullptr;
  double* d_dt = nullptr;
  int* d_cost_index = nullptr;
  int* d_crow = nullptr;    // [64] compact_row table of the model (tangent-matrix getters)
  std::vector<char> gl_set; // [n_costs] cost i carries per-trajectory linear terms (all clear: DevProblem::gl goes back to null)
  double* d_gl = nullptr;   // per-trajectory linear cost terms (DevProblem::gl), tiled, L = n_costs * (n + m); allocated on first use
  double* d_cp = nullptr;   // per-trajectory constraint parameters (DevProblem::cp), tiled, L = n * n_cons; allocated on first use
  double* d_tmp = nullptr;  // [Bp] scratch for reductions / outputs
  double* d_tmp2 = nullptr;
  // multi-GPU: RCCL communicator of the batch shards (to_comm_*; librccl is dlopen'ed on first use)
  void* comm = nullptr;
  int comm_rank = 0, comm_size = 1;
  std::vector<int32_t> comm_counts;  // shard size of every rank (exchanged at to_comm_init_rank)
  long long comm_offset = 0, comm_total = 0;  // global index of this rank's first trajectory; sum of the shards
  bool comm_equal = true;            // all shards equal: one in-place all-gather, else grouped broadcasts
  // projected-Newton polish (k_pn.h): workspace and tables, allocated on first use
  double* pn_ws = nullptr;
  size_t pn_ws_bytes = 0;
  int* pn_pak = nullptr;
  long long* pn_koff = nullptr;
  int* pn_list = nullptr;        // [Bp] device: the trajectory polished in workspace slot i
  int* pn_list_host = nullptr;   // [Bp] pinned: where the lists are assembled
  int pn_tab_len = 0;
  int pn_cap = 0, pn_nbmax = 0;  // workspace slots; max_k (ne + candidate rows of knot k)
  long long pn_per = 0;          // doubles per slot
  hipStream_t pn_stream = nullptr;  // low-priority stream of the early polish (altro_solve), created on first use
  hipEvent_t pn_ev[3] = {nullptr, nullptr, nullptr};  // snapshot taken / early polish enqueued up to here / timing
  int32_t *snap_status = nullptr, *snap_active = nullptr;  // pinned [Bp]: snapshots of the per-trajectory solver state
  double* snap_cmax = nullptr;
  int pn_early = 0;              // > 0 inside altro_solve: the AL loop hands finished trajectories to the polish up to this many times
  std::vector<char> pn_done_early;  // [B] polished while the AL stage was still running
  int pn_early_slots = 0;        // workspace slots handed out so far
  to_solver_opts pn_opts;        // the options the polish runs with (the AL stage runs with a looser constraint_tolerance)
  double pn_early_ms = 0.0;
  // repacked working set (k_generic.h k_repack_*): the per-trajectory arrays a solve carries from step to step, their home pointers,
  // two working copies (B/2 and B/4 trajectories: ping-pong) and the working position -> home index maps
  struct RpArr { size_t off; int kind; int L; };  // off: byte offset of the pointer field inside KArgs
  std::vector<RpArr> rp_arr;
  std::vector<void*> rp_home, rp_work[2];
  int* rp_map[2] = {nullptr, nullptr};
  int rp_cap[2] = {0, 0};
  int rp_level = 0;      // 0: the kernels work on the home arrays; else on rp_work[(rp_level - 1) & 1]
  int rp_B = 0, rp_Bp = 0;  // the batch of the handle while a solve works on a smaller set
  double rp_at = 0.7;    // ... the fraction of the working set that has to be left for a move (TRAJOPT_REPACK_AT)
  int rp_min = 16384;    // repack once the active count has halved, while the set holds at least this many (TRAJOPT_REPACK=0: never)
  // asynchronous solves (to_*_solve_async / to_solve_wait)
  std::thread worker;
  std::atomic<bool> inflight{false};
  int async_rc = 0;
  std::string async_err;
  // progress of the solve in flight, as the solve loop last saw it (to_solve_progress / to_solve_wait_below: a host that
  // pipelines solves over several handles admits the next one when the one in flight has drained)
  std::atomic<int> prog_active{0};   // trajectories still iterating (B when a solve starts, 0 once its iLQR / AL stage has ended)
  std::atomic<int> prog_steps{0};    // batch steps whose counters have been read
  std::mutex prog_mu;
  std::condition_variable prog_cv;
  int last_steps = 0;     // batch steps and device time of the last solve (fill_stats)
  double last_ms = 0.0;
  // TRAJOPT_GUARD=1 (read at to_create): every device array of the handle sits between two red zones filled with a pattern, checked after
  // every batch step of a solve and every phase-API call — an out-of-bounds store of a kernel is reported where it happens, by array name,
  // instead of surfacing as a fault at whatever batch size happens to cross a mapping boundary (sanitizer hook; GPU ASan is
  // not available on this pool)
  bool guard = false;
  struct GuardRec { void* base; void* payload; size_t bytes; std::string name; };
  std::vector<GuardRec> guards;
  void* guard_tab = nullptr;   // device copy of the zone addresses (rebuilt when `guards` changes)
  int* guard_bad = nullptr;    // device: index of the first damaged zone, else INT_MAX
  bool guard_dirty = true;
  // measurement
  bool profile = false;
  std::vector<hipEvent_t> ev;  // event pool, 4 per batch step
  hipEvent_t sev[4] = {nullptr, nullptr, nullptr, nullptr};  // solve(): start, stop, two chunk read-back events (created once)
  double prof_ms[TO_PROFILE_SLOTS] = {0, 0, 0, 0};
  int64_t prof_launches[TO_PROFILE_SLOTS] = {0, 0, 0, 0};
};

namespace to {

// Launchers of the model-templated kernels (all enqueue on h->stream and return TO_OK / TO_ERR_HIP).
struct ModelOps {
  // model traits the host logic needs
  bool write_through = false;  // M::accept_write_through
  bool mfma_backward = false;  // M::mfma_backward: tangent-matrix expansion + one-wave-per-trajectory Riccati
  bool coop_backward = true;   // M::coop_backward: column-layout expansion + cooperative LDS Riccati
  bool lane_backward = false;  // M::lane_backward: lane-layout expansion + one-lane-per-trajectory Riccati (ne + m <= 6)
  bool lds_gains = false;      // forward pass stages gains through LDS
  int expand_knots = 1;
  int ls_first_round = 16;     // M::ls_first_round
  int gains_lds_pieces = 0;    // 16-byte pieces of one gains row (LDS sizing of the forward pass)
  int crow[64];                // compact_row(g, c) of the tangent-matrix layout
  int nep = 0, rs = 0;         // Tm<M>::NEP, Tm<M>::RS
  int (*rollout)(to_handle*) = nullptr;
  int (*cost)(to_handle*, int with_al, double* out, double* Jk) = nullptr;
  int (*violation)(to_handle*, double* out) = nullptr;
  int (*dual_update)(to_handle*) = nullptr;
  int (*outer)(to_handle*) = nullptr;
  int (*cost_derivs)(to_handle*, double* grad, double* hess) = nullptr;
  int (*discrete_jacobian)(to_handle*, double* F) = nullptr;
  int (*constraint_eval)(to_handle*, int ci, double* vals, double* jac) = nullptr;
  int (*constraint_hessian)(to_handle*, int ci, const double* lambda, double* H) = nullptr;
  int (*expand)(to_handle*) = nullptr;
  int (*expand_const)(to_handle*) = nullptr;     // packed expansion: the constant columns of [A B], written once per handle (k_expand_const_columns)
  int (*backward)(to_handle*) = nullptr;
  int (*expand_lane_k)(to_handle*) = nullptr;    // lane-layout expansion, one lane per (trajectory, knot) (ops_lane.h; null: column-per-lane kernel)
  int (*expand_backward)(to_handle*) = nullptr;  // fused lane expansion + Riccati (small models; null elsewhere)
  int (*expand_backward_scan)(to_handle*) = nullptr;  // fused expansion + scan Riccati, one wave per trajectory (k_scan.h)
  int (*expand_backward_coop)(to_handle*) = nullptr;  // fused expansion + cooperative Riccati (small models with <= 8 directions)
  int (*pn_prepare)(to_handle*, int want) = nullptr;  // projected-Newton polish (k_pn.h): tables + workspace slots ...
  int (*pn_launch)(to_handle*, int slot0, int count, hipStream_t stream, const to_solver_opts* opts) = nullptr;  // ... and its launches
  int (*defect)(to_handle*, double* out) = nullptr;             // max dynamics / initial-condition defect of the nominal trajectory
  int (*infeasible_controls)(to_handle*) = nullptr;             // InfeasibleModel only: slack controls from the current states (k_misc.h)
  int (*accept_roll)(to_handle*) = nullptr;  // accept by re-rolling the stored controls (k_forward.h; models without write-through)
  int (*forward[32])(to_handle*) = {};  // by kernel variant (k_forward.h MODE bits); variants a model never uses stay null
  int (*forward2[32])(to_handle*) = {};  // the same variants as two-wave workgroups (k_forward2; models with LDS-staged gains)
};

// each ops_*.hip fills the entries it instantiates; table indexed by model key (0..2 double integrator D=1..3, 3 Cartpole,
// 4 Quadrotor, 5 Quadrotor{MRP}, 6 Quadrotor{RodriguesParam}, 7 hybrid double integrator, 8 general model vector)
constexpr int N_MODEL_KEYS = 12;  // ... 9, 10 InfeasibleModel over the 1-D / 2-D double integrator, 11 over the Cartpole
void fill_ops_small(ModelOps* table);
void fill_ops_small_forward(ModelOps* table);
void fill_ops_small_lane(ModelOps* table);
void fill_ops_quad_misc(ModelOps* table);
void fill_ops_quad_expand(ModelOps* table);
void fill_ops_quad_backward(ModelOps* table);
void fill_ops_quad_forward_a(ModelOps* table);
void fill_ops_quad_forward_b(ModelOps* table);
void fill_ops_quad_forward_c(ModelOps* table);
void fill_ops_quad_forward2_a(ModelOps* table);
void fill_ops_quad_forward2_b(ModelOps* table);
void fill_ops_quad_forward2_c(ModelOps* table);
void fill_ops_quadatt_misc(ModelOps* table);
void fill_ops_quadmrp_expand(ModelOps* table);
void fill_ops_quadrp_expand(ModelOps* table);
void fill_ops_quadmrp_forward(ModelOps* table);
void fill_ops_quadrp_forward(ModelOps* table);
void fill_ops_hybrid(ModelOps* table);
void fill_ops_small_forward2(ModelOps* table);
void fill_ops_small_scan(ModelOps* table);
void fill_ops_pn(ModelOps* table);
void fill_ops_vector(ModelOps* table);
void fill_ops_infeasible_a(ModelOps* table);
void fill_ops_infeasible_b(ModelOps* table);

// handle-owned device memory (red zones around it in guard mode); g_free accepts what g_malloc returned
int g_malloc(to_handle* h, void** p, size_t bytes, const char* name);
void g_free(to_handle* h, void* p);
int check_guards(to_handle* h, const char* where);

inline dim3 grid_b(const to_handle* h, int y = 1, int z = 1) { return dim3(h->a.P.Bp / BLOCK, y, z); }

}  // namespace to

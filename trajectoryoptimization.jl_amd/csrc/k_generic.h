// k_generic.h — model-independent kernels (accept copy, solver-state initialisation, layout transposes, cones).
// Non-template __global__ functions: include from ONE translation unit only (trajopt_hip.hip).
#pragma once
#include "common.h"

namespace to {


// Accepting a step = copying the accepted candidate's slot onto slot 0, so the nominal trajectory of every lane sits in
// ONE slot and every kernel reads it with full-line coalesced loads (a per-trajectory slot index turned each nominal
// load of a wave into up to 64 separate lines: measured 2x on the quadrotor forward pass).  Runs once per forward pass:
// later rounds only store into the slots of lanes that are still searching.  grid (tiles, 1, chunks): wave (tile, z)
// copies chunk z; every lane reads ITS accepted slot (a gather: as many lines as a per-slot pass would touch) and the
// stores to slot 0 are whole 512-byte rows.
__global__ void __launch_bounds__(64) k_accept(KArgs a) {
  TILE_LANE();
  const DevProblem& P = a.P;
  const int s = b < P.B ? a.acc[b] : 0;
  if (__ballot(s != 0) == 0) return;
  const int Lx = P.N * P.n, Lu = (P.N - 1) * P.m;
  const int per = (Lx + Lu + gridDim.z - 1) / gridDim.z;
  const int e0 = blockIdx.z * per, e1 = min(Lx + Lu, e0 + per);
  if (s == 0) return;
  const double* sx = X_SLOT_PTR(a, b, s);
  double* dx = X_SLOT_PTR(a, b, 0);
  const double* su = U_SLOT_PTR(a, b, s);
  double* du = U_SLOT_PTR(a, b, 0);
#pragma unroll 16
  for (int e = e0; e < min(e1, Lx); ++e) EL(dx, e) = EL(sx, e);
#pragma unroll 16
  for (int e = max(e0, Lx) - Lx; e < e1 - Lx; ++e) EL(du, e) = EL(su, e);
}

__global__ void k_clear_acc(KArgs a) {
  TILE_LANE();
  if (b < a.P.Bp) a.acc[b] = 0;
}

// start of a solve: reset the per-trajectory solver state.  J must already hold the (AL) cost of the rollout.
__global__ void k_solve_init(KArgs a, int reset_duals) {
  TILE_LANE();
  const DevProblem& P = a.P;
  if (b >= P.Bp) return;
  if (b < 2) a.ocount[b] = 0;  // both parities of the outer-update list (common.h)
  const bool live = b < P.B;
  if (a.compact) {  // step 0 works on the whole batch
    if (live) a.alist[b] = b;
    const int tot0 = a.al_mode ? P.opts.iterations_total : P.opts.iterations;
    if (b == 0) { a.acount[0] = (tot0 > 0 && P.opts.iterations > 0) ? P.B : 0; a.acount[1] = 0; }
  }
  a.rho[b] = P.opts.bp_reg_initial; a.drho[b] = 0.0;
  a.dJzero[b] = 0; a.it_inner[b] = 0; a.iterations[b] = 0; a.outer[b] = 0;
  a.status[b] = TO_UNSOLVED; a.ls_index[b] = -1; a.bpfail[b] = 0; a.acc[b] = 0; a.oflag[b] = 0;
  a.dJ[b] = 0.0; a.grad[b] = 0.0; a.cmax[b] = 0.0;
  const int tot = a.al_mode ? P.opts.iterations_total : P.opts.iterations;
  a.budget[b] = tot < P.opts.iterations ? tot : P.opts.iterations;
  a.active[b] = (live && a.budget[b] > 0) ? 1 : 0;
  if (live && a.budget[b] <= 0) a.status[b] = TO_MAX_ITERATIONS;
  if (reset_duals) {
    double* lam0 = TILE_PTR(a.lam, P.n_duals);
    double* mu0 = TILE_PTR(a.mu, P.n_cons);
    for (long long r = 0; r < P.n_duals; ++r) EL(lam0, r) = 0.0;
    for (int ci = 0; ci < P.n_cons; ++ci) EL(mu0, ci) = P.opts.penalty_initial;
  }
}

// Stable compaction of the active flags into alist[(step+1)&1] (ascending trajectory index) and acount[(step+1)&1].
// ONE workgroup of 1024 threads: wave w owns the contiguous segment [w*seg, (w+1)*seg) of the batch, seg a multiple of 64;
// pass 1 counts (ballot + popcount), a 16-entry scan in LDS gives every wave its offset, pass 2 writes the indices.
__global__ void __launch_bounds__(1024) k_compact(KArgs a) {
  __shared__ int wcount[16];
  const int Bp = a.P.Bp, B = a.P.B;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int seg = (((Bp + 15) / 16) + 63) / 64 * 64;
  const int lo = wave * seg, hi = min(Bp, lo + seg);
  int cnt = 0;
  for (int i = lo; i < hi; i += 64) {
    const int b = i + lane;
    cnt += __popcll(__ballot(b < B && a.active[b] != 0));
  }
  if (lane == 0) wcount[wave] = cnt;
  __syncthreads();
  int off = 0, total = 0;
  for (int w = 0; w < 16; ++w) { off += (w < wave) ? wcount[w] : 0; total += wcount[w]; }
  int* out = a.alist + (size_t)((a.step + 1) & 1) * Bp;
  for (int i = lo; i < hi; i += 64) {
    const int b = i + lane;
    const bool on = b < B && a.active[b] != 0;
    const unsigned long long m = __ballot(on);
    if (on) out[off + __popcll(m & ((1ull << lane) - 1ull))] = b;
    off += __popcll(m);
  }
  if (threadIdx.x == 0) a.acount[(a.step + 1) & 1] = total;
}
// (Round 6 tried this kernel as FOUR waves instead of sixteen — a 1024-thread workgroup needs a compute unit with four free wave slots on
// every SIMD at once, and next to another handle's dense k_expand launch it sat 580 us per launch in the trace of pipelined C5 solves.
// That wait was not a bottleneck: the chip is saturated in that regime, the pipelined throughput was the same with either version
// (1.418 vs 1.42 M it/s; the time moved into k_expand's durations), and alone the four-wave version is slower — 10 vs 5 us per launch at
// B = 4096, 24 vs 9 us at 16 384: -4 % at the B = 32 768 sweep point.  Reverted; profiles/r06_ab/.)

// The same for large batches, in two launches of NB workgroups (the single workgroup takes 87 us for 131 072 flags, 9 % of a
// batch step of the large-batch sweep): k_compact_count leaves every workgroup's number of active trajectories in ccount[],
// k_compact_write sums the counts of the workgroups before it and writes its indices.  Workgroup g owns the flags
// [g*per, (g+1)*per), per a multiple of 1024; wave w of it the 64-aligned slices w*64 + i*1024.
__global__ void __launch_bounds__(1024) k_compact_count(KArgs a, int per) {
  __shared__ int wcount[16];
  const int B = a.P.B, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int lo = blockIdx.x * per;
  int cnt = 0;
  for (int i = lo + wave * 64; i < lo + per; i += 1024) {
    const int b = i + lane;
    cnt += __popcll(__ballot(b < B && a.active[b] != 0));
  }
  if (lane == 0) wcount[wave] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < 16; ++w) t += wcount[w];
    a.ccount[blockIdx.x] = t;
  }
}
__global__ void __launch_bounds__(1024) k_compact_write(KArgs a, int per) {
  __shared__ int scount[16 * 64];  // active trajectories of slice (i, w) of this workgroup, i < per/1024 <= 64
  __shared__ int base_s;
  const int B = a.P.B, Bp = a.P.Bp, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int lo = blockIdx.x * per, nsl = per / 1024;
  for (int i = 0; i < nsl; ++i) {
    const int b = lo + i * 1024 + wave * 64 + lane;
    const int c = __popcll(__ballot(b < B && a.active[b] != 0));
    if (lane == 0) scount[i * 16 + wave] = c;
  }
  if (threadIdx.x == 0) {
    int t = 0;
    for (int g = 0; g < (int)blockIdx.x; ++g) t += a.ccount[g];
    base_s = t;
    if (blockIdx.x == gridDim.x - 1) a.acount[(a.step + 1) & 1] = t + a.ccount[blockIdx.x];
  }
  __syncthreads();
  int* out = a.alist + (size_t)((a.step + 1) & 1) * Bp;
  for (int i = 0; i < nsl; ++i) {
    int off = base_s;  // slices are numbered (i, wave) in index order: i*16 + wave
    for (int q = 0; q < i * 16 + wave; ++q) off += scount[q];
    const int b = lo + i * 1024 + wave * 64 + lane;
    const bool on = b < B && a.active[b] != 0;
    const unsigned long long m = __ballot(on);
    if (on) out[off + __popcll(m & ((1ull << lane) - 1ull))] = b;
  }
}

// ------------------------------------------------------------------------------------------------ repacking the working set
// Large batches drain unevenly, and the active lists are in index order: once half of a batch has converged, a wave's 64 trajectories
// sit in two or more tiles and every nominal load / store of every kernel touches that many 512-byte rows for 64 values (r05 trace,
// Cartpole at B = 1 048 576: 95 M trajectory-iterations/s while all are active, 60 M over the rest of the solve).  When the active count
// has halved, the solve loop therefore MOVES the per-trajectory state of the trajectories still being solved into a dense working set
// (position j <- the j-th active trajectory, in index order) and goes on there with B = the number that is left; the converged ones stay
// where they are (first move) or are written back to their home position (later moves, end of the solve).  Which position holds a
// trajectory changes nothing in its arithmetic: bit-identical solves (tests/test_gpu_parity.py::test_repacked_working_set).
constexpr int RP_MAX = 32;
struct RpArgs {
  int n;                // arrays
  int kind[RP_MAX];     // 0: tiled doubles, L per trajectory; 1: plain double [B]; 2: plain int [B]; 3: tiled doubles a solve only READS (moved, never copied home)
  int L[RP_MAX];
  const void* src[RP_MAX];
  void* dst[RP_MAX];
};
// dst position j <- src position list[j], j < count; omap_new[j] = home index of that trajectory.  grid (ceil(count/64), n arrays)
__global__ void __launch_bounds__(64) k_repack_move(RpArgs r, const int* __restrict__ list, int count, const int* __restrict__ omap_old,
                                                    int* __restrict__ omap_new) {
  const int j = blockIdx.x * 64 + threadIdx.x, ai = blockIdx.y;
  if (j >= count) return;
  const int p = list[j];
  if (ai == 0) omap_new[j] = omap_old ? omap_old[p] : p;
  if (r.kind[ai] == 0 || r.kind[ai] == 3) {
    const int L = r.L[ai];
    const double* s = (const double*)r.src[ai] + ((size_t)(p >> 6) * (size_t)L) * 64 + (p & 63);
    double* d = (double*)r.dst[ai] + ((size_t)(j >> 6) * (size_t)L) * 64 + (j & 63);
#pragma unroll 8
    for (int e = 0; e < L; ++e) d[(size_t)e * 64] = s[(size_t)e * 64];
  } else if (r.kind[ai] == 1) ((double*)r.dst[ai])[j] = ((const double*)r.src[ai])[p];
  else ((int*)r.dst[ai])[j] = ((const int*)r.src[ai])[p];
}
// working position j -> home position omap[j], for the trajectories that are finished (all != 0: every position).  src = working, dst = home
__global__ void __launch_bounds__(64) k_repack_home(RpArgs r, const int* __restrict__ active, int count, const int* __restrict__ omap, int all) {
  const int j = blockIdx.x * 64 + threadIdx.x, ai = blockIdx.y;
  if (j >= count) return;
  if (!all && active[j] != 0) return;
  const int q = omap[j];
  if (r.kind[ai] == 3) return;
  if (r.kind[ai] == 0) {
    const int L = r.L[ai];
    const double* s = (const double*)r.src[ai] + ((size_t)(j >> 6) * (size_t)L) * 64 + (j & 63);
    double* d = (double*)r.dst[ai] + ((size_t)(q >> 6) * (size_t)L) * 64 + (q & 63);
#pragma unroll 8
    for (int e = 0; e < L; ++e) d[(size_t)e * 64] = s[(size_t)e * 64];
  } else if (r.kind[ai] == 1) ((double*)r.dst[ai])[q] = ((const double*)r.src[ai])[j];
  else ((int*)r.dst[ai])[q] = ((const int*)r.src[ai])[j];
}
// the active list of the step after a move: everybody, in place
__global__ void k_repack_list(int* __restrict__ list, int* __restrict__ acount, int count) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j < count) list[j] = j;
  if (j == 0) acount[0] = count;
}

// The two compaction launches for an arbitrary flag array: flags[0 .. n) -> out (ascending index), *outcount; the flags are cleared on the
// way (each is read by exactly one thread of each launch).  Used by the two-launch line search (KArgs::pending -> plist, pcount).
__global__ void __launch_bounds__(1024) k_flags_count(const int* __restrict__ flags, int n, int per, int* __restrict__ ccount) {
  __shared__ int wcount[16];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int lo = blockIdx.x * per;
  int cnt = 0;
  for (int i = lo + wave * 64; i < lo + per; i += 1024) {
    const int b = i + lane;
    cnt += __popcll(__ballot(b < n && flags[b] != 0));
  }
  if (lane == 0) wcount[wave] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < 16; ++w) t += wcount[w];
    ccount[blockIdx.x] = t;
  }
}
__global__ void __launch_bounds__(1024) k_flags_write(int* __restrict__ flags, int n, int per, const int* __restrict__ ccount, int* __restrict__ out,
                                                      int* __restrict__ outcount) {
  __shared__ int scount[16 * 64];
  __shared__ int base_s;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int lo = blockIdx.x * per, nsl = per / 1024;
  for (int i = 0; i < nsl; ++i) {
    const int b = lo + i * 1024 + wave * 64 + lane;
    const int c = __popcll(__ballot(b < n && flags[b] != 0));
    if (lane == 0) scount[i * 16 + wave] = c;
  }
  if (threadIdx.x == 0) {
    int t = 0;
    for (int g = 0; g < (int)blockIdx.x; ++g) t += ccount[g];
    base_s = t;
    if (blockIdx.x == gridDim.x - 1) outcount[0] = t + ccount[blockIdx.x];
  }
  __syncthreads();
  for (int i = 0; i < nsl; ++i) {
    int off = base_s;
    for (int q = 0; q < i * 16 + wave; ++q) off += scount[q];
    const int b = lo + i * 1024 + wave * 64 + lane;
    const bool on = b < n && flags[b] != 0;
    const unsigned long long m = __ballot(on);
    if (on) { out[off + __popcll(m & ((1ull << lane) - 1ull))] = b; flags[b] = 0; }
  }
}

__global__ void k_set_active(KArgs a, int value, int clear_bpfail) {
  TILE_LANE();
  if (b >= a.P.Bp) return;
  a.active[b] = (b < a.P.B) ? value : 0;
  if (clear_bpfail == 1) a.bpfail[b] = 0;
}

__global__ void k_penalty_max(KArgs a, double* out) {
  TILE_LANE();
  if (b >= a.P.B) return;
  const double* mu0 = TILE_PTR(a.mu, a.P.n_cons);
  double mx = 0.0;
  for (int ci = 0; ci < a.P.n_cons; ++ci) mx = fmax(mx, EL(mu0, ci));
  out[b] = mx;
}

// ------------------------------------------------------------------------------------------------ layout transposes
// host layout h[e + L*b] (e = i + dim*k, column-major (dim, K, B))  <->  tiled device layout, L elements per trajectory.
// grid: (tiles, L).  Host-side access is strided, device-side coalesced; these are API-boundary copies, not hot.
// (cnt host elements per trajectory map to device elements e0 .. e0+cnt-1 of an array with L per trajectory)
__global__ void k_to_device(const double* __restrict__ h, double* __restrict__ d, int L, int e0, int cnt, int B) {
  TILE_LANE();
  const int e = blockIdx.y;
  if (b >= B) return;
  EL(TILE_PTR(d, L), e0 + e) = h[(size_t)e + (size_t)cnt * b];
}
__global__ void k_to_host(const double* __restrict__ d, double* __restrict__ h, int L, int e0, int cnt, int B) {
  TILE_LANE();
  const int e = blockIdx.y;
  if (b >= B) return;
  h[(size_t)e + (size_t)cnt * b] = EL(TILE_PTR(d, L), e0 + e);
}
__global__ void k_fill_uniform(double* d, const double* u, int dim, int L, int B) {
  TILE_LANE();
  const int e = blockIdx.y;
  if (b >= B) return;
  EL(TILE_PTR(d, L), e) = u[e % dim];
}
// gains: Kt rows (trajectory-major, [r][ne gains + 1 feed-forward]) -> host K[m,ne,N-1,B] / d[m,N-1,B] column-major.
// grid (tiles, (N-1)*m*(ne+1)): e = (k*m + r)*(ne+1) + i
__global__ void k_gains_to_host(const double* __restrict__ Kt, double* __restrict__ hK, double* __restrict__ hd, int m, int ne, int K, int B) {
  TILE_LANE();
  const int e = blockIdx.y;
  if (b >= B) return;
  const int i = e % (ne + 1), r = (e / (ne + 1)) % m, k = e / ((ne + 1) * m);
  const double v = Kt[((size_t)b * K + k) * (m * (ne + 1)) + r * (ne + 1) + i];
  if (i < ne) { if (hK) hK[(size_t)r + (size_t)m * (i + (size_t)ne * (k + (size_t)K * b))] = v; }
  else if (hd) hd[(size_t)r + (size_t)m * (k + (size_t)K * b)] = v;
}

// column-layout arrays -> host column-major blocks: h[r + Rr*(c + Cc*(k + K*b))] = col_array(b, column c0+c, entry k*rows_per_knot + r0 + r)
// grid (ceil(B/64), K*Rr*Cc), one thread per trajectory.
// diag: the array holds one row per knot, entry (c, c) of the block (KArgs::h_diag); every other entry reads as zero.
__global__ void k_col_to_host(const double* __restrict__ src, double* __restrict__ h, int E, int rows_per_knot, int r0, int Rr, int c0,
                              int Cc, int K, int B, int R, int G, int diag) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  const int e = blockIdx.y;  // (k*Cc + c)*Rr + r
  if (b >= B) return;
  const int r = e % Rr, c = (e / Rr) % Cc, k = e / (Rr * Cc);
  const int gtile = b / G, g = b % G;
  double v;
  if (diag) v = (r0 + r == c0 + c) ? src[((size_t)gtile * E + k) * 64 + g * R + (c0 + c)] : 0.0;
  else v = src[((size_t)gtile * E + (size_t)k * rows_per_knot + r0 + r) * 64 + g * R + (c0 + c)];
  h[(size_t)r + (size_t)Rr * (c + (size_t)Cc * (k + (size_t)K * b))] = v;
}

// lane layout (k_backward.h, LaneLay) -> host column-major blocks: h[r + Rr*(c + Cc*(k + K*b))] = X_k[row0 + r][col0 + c].
// E entries of 64 lanes per knot; sym: X symmetric, upper triangle stored column by column; else row-major with ld columns
// (a vector: Rr = 1, row0 = 0, ld = its length).
__global__ void k_lane_to_host(const double* __restrict__ src, double* __restrict__ h, int E, int sym, int ld, int row0, int Rr, int col0,
                               int Cc, int K, int B) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  const int e = blockIdx.y;  // (k*Cc + c)*Rr + r
  if (b >= B) return;
  const int r = e % Rr, c = (e / Rr) % Cc, k = e / (Rr * Cc);
  const int row = row0 + r, col = col0 + c;
  const int lo = row < col ? row : col, hi = row < col ? col : row;
  const int ent = sym ? hi * (hi + 1) / 2 + lo : row * ld + col;
  h[(size_t)r + (size_t)Rr * (c + (size_t)Cc * (k + (size_t)K * b))] = src[(((size_t)(b >> 6) * K + k) * E + ent) * 64 + (b & 63)];
}

// tangent-matrix layout (k_backward.h) -> host column-major blocks: h[r + Rr*(c + Cc*(k + K*b))] = X_k[row0 + r][col0 + c] of
// trajectory b, X stored with `regs` 64-lane rows per knot.  compact: one row per knot holding, for lane (g, col), the
// entry of row crow[g*16 + col] (device table, -1: none); everything else is zero.
__global__ void k_tm_to_host(const double* __restrict__ src, double* __restrict__ h, int regs, int row0, int Rr, int col0, int Cc, int K,
                             int B, const int* __restrict__ crow) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  const int e = blockIdx.y;  // (k*Cc + c)*Rr + r
  if (b >= B) return;
  const int r = e % Rr, c = (e / Rr) % Cc, k = e / (Rr * Cc);
  const int row = row0 + r, col = col0 + c, g = row & 3;
  double v;
  if (crow) v = (crow[g * 16 + col] == row) ? src[((size_t)b * K + k) * 64 + g * 16 + col] : 0.0;
  else v = src[(((size_t)b * K + k) * regs + (row >> 2)) * 64 + g * 16 + col];
  h[(size_t)r + (size_t)Rr * (c + (size_t)Cc * (k + (size_t)K * b))] = v;
}
// gradient vectors of the tangent-matrix layout: gt[(b*K + k)*16 + col] -> h[c + Cc*(k + K*b)]
__global__ void k_tmvec_to_host(const double* __restrict__ gt, double* __restrict__ h, int col0, int Cc, int K, int B) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  const int e = blockIdx.y;  // k*Cc + c
  if (b >= B) return;
  const int c = e % Cc, k = e / Cc;
  h[(size_t)c + (size_t)Cc * (k + (size_t)K * b)] = gt[((size_t)b * K + k) * 16 + col0 + c];
}

// Cost-to-go of the backward pass (to_get_cost_to_go), recomputed from the expansion and the gains of the last backward pass: one
// thread per trajectory walks S_N = lxx_N, s_N = lx_N;  Q = l + [A B]' S [A B];  S_k = Qxx + K'Quu K + K'Qux + Qux'K (symmetrised),
// s_k = Qx + K'Quu d + K'Qu + Qux'd — the un-regularised update the solve's recursion makes.  All arrays in the host layouts of the
// getters (column-major, trajectory slowest), in device memory.  An introspection getter, not a hot path: runtime dimensions, local
// arrays in scratch.
__global__ void k_cost_to_go(const double* __restrict__ A, const double* __restrict__ Bm, const double* __restrict__ lxx, const double* __restrict__ luu,
                             const double* __restrict__ lux, const double* __restrict__ lx, const double* __restrict__ lu, const double* __restrict__ K,
                             const double* __restrict__ d, double* __restrict__ Sout, double* __restrict__ sout, int ne, int m, int N, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  constexpr int NE = TO_MAX_N, MM = TO_MAX_M;
  double S[NE * NE], s[NE], SA[NE * NE], SB[NE * MM], Qxx[NE * NE], Qux[MM * NE], Quu[MM * MM], Qx[NE], Qu[MM], KtQuu[NE * MM], Sn[NE * NE];
  auto at = [&](const double* p, int R, int Cc, int Kn, int r, int c, int k) { return p[(size_t)r + (size_t)R * (c + (size_t)Cc * (k + (size_t)Kn * b))]; };
  for (int i = 0; i < ne; ++i) { for (int j = 0; j < ne; ++j) S[i * ne + j] = at(lxx, ne, ne, N, i, j, N - 1); s[i] = at(lx, ne, 1, N, i, 0, N - 1); }
  for (int i = 0; i < ne; ++i) { for (int j = 0; j < ne; ++j) Sout[(size_t)i + (size_t)ne * (j + (size_t)ne * ((N - 1) + (size_t)N * b))] = S[i * ne + j]; sout[(size_t)i + (size_t)ne * ((N - 1) + (size_t)N * b)] = s[i]; }
  for (int k = N - 2; k >= 0; --k) {
    for (int i = 0; i < ne; ++i) {
      for (int j = 0; j < ne; ++j) { double v = 0.0; for (int r = 0; r < ne; ++r) v += S[i * ne + r] * at(A, ne, ne, N - 1, r, j, k); SA[i * ne + j] = v; }
      for (int j = 0; j < m; ++j) { double v = 0.0; for (int r = 0; r < ne; ++r) v += S[i * ne + r] * at(Bm, ne, m, N - 1, r, j, k); SB[i * m + j] = v; }
    }
    for (int i = 0; i < ne; ++i) {
      for (int j = 0; j < ne; ++j) { double v = at(lxx, ne, ne, N, i, j, k); for (int r = 0; r < ne; ++r) v += at(A, ne, ne, N - 1, r, i, k) * SA[r * ne + j]; Qxx[i * ne + j] = v; }
      double v = at(lx, ne, 1, N, i, 0, k); for (int r = 0; r < ne; ++r) v += at(A, ne, ne, N - 1, r, i, k) * s[r]; Qx[i] = v;
    }
    for (int i = 0; i < m; ++i) {
      for (int j = 0; j < ne; ++j) { double v = at(lux, m, ne, N, i, j, k); for (int r = 0; r < ne; ++r) v += at(Bm, ne, m, N - 1, r, i, k) * SA[r * ne + j]; Qux[i * ne + j] = v; }
      for (int j = 0; j < m; ++j) { double v = at(luu, m, m, N, i, j, k); for (int r = 0; r < ne; ++r) v += at(Bm, ne, m, N - 1, r, i, k) * SB[r * m + j]; Quu[i * m + j] = v; }
      double v = at(lu, m, 1, N, i, 0, k); for (int r = 0; r < ne; ++r) v += at(Bm, ne, m, N - 1, r, i, k) * s[r]; Qu[i] = v;
    }
    for (int i = 0; i < ne; ++i) for (int j = 0; j < m; ++j) { double v = 0.0; for (int r = 0; r < m; ++r) v += at(K, m, ne, N - 1, r, i, k) * Quu[r * m + j]; KtQuu[i * m + j] = v; }
    for (int i = 0; i < ne; ++i) {
      double v = Qx[i];
      for (int j = 0; j < m; ++j) v += KtQuu[i * m + j] * at(d, m, 1, N - 1, j, 0, k);
      for (int j = 0; j < m; ++j) v += at(K, m, ne, N - 1, j, i, k) * Qu[j];
      for (int j = 0; j < m; ++j) v += Qux[j * ne + i] * at(d, m, 1, N - 1, j, 0, k);
      s[i] = v;
    }
    for (int i = 0; i < ne; ++i) for (int j = 0; j < ne; ++j) {
      double v = Qxx[i * ne + j];
      for (int r = 0; r < m; ++r) v += KtQuu[i * m + r] * at(K, m, ne, N - 1, r, j, k);
      for (int r = 0; r < m; ++r) v += at(K, m, ne, N - 1, r, i, k) * Qux[r * ne + j];
      for (int r = 0; r < m; ++r) v += Qux[r * ne + i] * at(K, m, ne, N - 1, r, j, k);
      Sn[i * ne + j] = v;
    }
    for (int i = 0; i < ne; ++i) for (int j = 0; j < ne; ++j) S[i * ne + j] = 0.5 * (Sn[i * ne + j] + Sn[j * ne + i]);
    for (int i = 0; i < ne; ++i) { for (int j = 0; j < ne; ++j) Sout[(size_t)i + (size_t)ne * (j + (size_t)ne * (k + (size_t)N * b))] = S[i * ne + j]; sout[(size_t)i + (size_t)ne * (k + (size_t)N * b)] = s[i]; }
  }
}

// ------------------------------------------------------------------------------------------------ cones (src/cones.jl), stateless
// x[dim,count] column-major; one thread per vector.
__global__ void k_cone_projection(int cone, int dim, long long count, const double* x, double* px, int* status) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= count) return;
  const double* v = x + t * dim;
  double* o = px + t * dim;
  int st = 0;
  if (cone == TO_CONE_IDENTITY) { for (int i = 0; i < dim; ++i) o[i] = v[i]; }
  else if (cone == TO_CONE_ZERO) { for (int i = 0; i < dim; ++i) o[i] = 0.0; }
  else if (cone == TO_CONE_NEGATIVE_ORTHANT) { for (int i = 0; i < dim; ++i) o[i] = fmin(0.0, v[i]); }
  else if (cone == TO_CONE_POSITIVE_ORTHANT) { for (int i = 0; i < dim; ++i) o[i] = fmax(0.0, v[i]); }
  else {
    const double s = v[dim - 1];
    double a2 = 0.0;
    for (int i = 0; i < dim - 1; ++i) a2 += v[i] * v[i];
    const double a = sqrt(a2);
    if (a <= -s) { for (int i = 0; i < dim; ++i) o[i] = 0.0; st = 0; }
    else if (a <= s) { for (int i = 0; i < dim; ++i) o[i] = v[i]; st = 1; }
    else if (a >= fabs(s)) { const double c = 0.5 * (1 + s / a); for (int i = 0; i < dim - 1; ++i) o[i] = v[i] * c; o[dim - 1] = a * c; st = 2; }
    else st = -1;  // NaN input: src/cones.jl:124 throws
  }
  if (status) status[t] = st;
}

__global__ void k_cone_jacobian(int cone, int dim, long long count, const double* x, double* jac, int* status) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= count) return;
  const double* v = x + t * dim;
  double* J = jac + t * dim * dim;  // column-major: J[r + dim*c]
  for (int i = 0; i < dim * dim; ++i) J[i] = 0.0;
  int st = 0;
  if (cone == TO_CONE_IDENTITY) { for (int i = 0; i < dim; ++i) J[i + dim * i] = 1.0; }
  else if (cone == TO_CONE_NEGATIVE_ORTHANT) { for (int i = 0; i < dim; ++i) J[i + dim * i] = v[i] <= 0 ? 1.0 : 0.0; }
  else if (cone == TO_CONE_POSITIVE_ORTHANT) { for (int i = 0; i < dim; ++i) J[i + dim * i] = v[i] >= 0 ? 1.0 : 0.0; }
  else if (cone == TO_CONE_SECOND_ORDER) {
    const int nn = dim;
    const double s = v[nn - 1];
    double a2 = 0.0;
    for (int i = 0; i < nn - 1; ++i) a2 += v[i] * v[i];
    const double a = sqrt(a2);
    if (a <= -s) st = 0;
    else if (a <= s) { for (int i = 0; i < nn; ++i) J[i + nn * i] = 1.0; st = 1; }
    else if (a >= fabs(s)) {
      const double c = 0.5 * (1 + s / a);
      for (int i = 0; i < nn - 1; ++i)
        for (int j = 0; j < nn - 1; ++j) J[i + nn * j] = -0.5 * s / (a * a * a) * v[i] * v[j] + ((i == j) ? c : 0.0);
      for (int i = 0; i < nn - 1; ++i) J[i + nn * (nn - 1)] = 0.5 * v[i] / a;
      for (int i = 0; i < nn - 1; ++i) J[(nn - 1) + nn * i] = ((-0.5 * s / (a * a)) + c / a) * v[i];
      J[(nn - 1) + nn * (nn - 1)] = 0.5;
      st = 2;
    } else st = -1;
  }
  if (status) status[t] = st;
}

// Hessian of b'Π_K(x).  The second-order-cone branch restates the reference's closed form term by term (src/cones.jl:244-270:
// the H1/H2/H3 split of the v-v block, the hi/(2a) border) so that the stateless operator returns exactly what ∇²projection!
// returns; it is not on the solve path (the AL expansion uses the Moreau identities of problem_dev.h instead).
__global__ void k_cone_hessian(int cone, int dim, long long count, const double* x, const double* bvec, double* hess, int* status) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= count) return;
  const double* v = x + t * dim;
  const double* bb = bvec + t * dim;
  double* H = hess + t * dim * dim;
  for (int i = 0; i < dim * dim; ++i) H[i] = 0.0;
  int st = 0;
  if (cone == TO_CONE_SECOND_ORDER) {
    const int nn = dim - 1;
    const double s = v[nn], bs = bb[nn];
    double a2 = 0.0, vbv = 0.0;
    for (int i = 0; i < nn; ++i) { a2 += v[i] * v[i]; vbv += v[i] * bb[i]; }
    const double a = sqrt(a2);
    if (a <= -s) st = 0;
    else if (a <= s) st = 1;
    else if (a > fabs(s)) {
      for (int i = 0; i < nn; ++i) {
        double hi = 0.0;
        for (int j = 0; j < nn; ++j) hi += (-v[i] * v[j] / (a * a) + ((i == j) ? 1.0 : 0.0)) * bb[j];
        H[i + dim * nn] = hi / (2 * a);
        H[nn + dim * i] = H[i + dim * nn];
        for (int j = 0; j <= i; ++j) {
          const double vij = v[i] * v[j];
          const double H1 = hi * v[j] * (-s / (a * a * a));
          double H2 = vij * (2 * vbv) / (a * a * a * a) - v[i] * bb[j] / (a * a);
          double H3 = -vij / (a * a);
          if (i == j) { H2 -= vbv / (a * a); H3 += 1; }
          H2 *= s / a;
          H3 *= bs / a;
          H[i + dim * j] = (H1 + H2 + H3) / 2;
          H[j + dim * i] = H[i + dim * j];
        }
      }
      st = 2;
    } else st = -1;
  }
  if (status) status[t] = st;
}

}  // namespace to

// ls_round.h — lane map of the line-search rounds of the forward pass (k_forward.h).  Plain integer logic: tests/test_ls_round_host.py
// compiles it for the host and checks the map exhaustively over wave shapes and need masks.
#pragma once

namespace to {

// Lane map of one line-search round.  Round 0 — and every round of a kernel without repacking — uses the static map of the wave
// shape: hardware lane q*TW + t evaluates step size c0 + q of the wave's trajectory t.  After it, most of a wave's trajectories have
// accepted a step and the few that have not would drag the wave — and, through the slowest wave, the whole launch — through one
// round of CW step sizes after the other (C5, CW = 8: half of the launches took three rollouts, 2.1 ms instead of 1.45).  When the
// remaining depth fits that way, the LAST round is repacked: the u trajectories still searching share all 64 lanes, tw = u
// rows x (total - c0) step sizes, and the search ends with it.  Which lane evaluates a candidate does not change its value, and
// the first accepted step size is taken as before: results are bit-identical (tests/test_gpu_parity.py::test_line_search_repack).
// Repacked candidates go to a second block per wave (KArgs::repack_block0): the lanes of a trajectory that accepted earlier keep
// its candidate in the first.
struct LsRound {
  int tw, cw;  // rows (trajectories) x step sizes of this round
  int qc, tr;  // this lane evaluates step size c0 + qc of row tr ...
  int ts;      // ... the trajectory held by the wave's lane ts (< TW)
  int j;       // row of this lane's OWN trajectory (meaningful while it is still searching)
  bool has;    // row tr holds a searching trajectory
  bool repacked;
};
__device__ __forceinline__ LsRound ls_round(unsigned long long nm, int c0, int total, int CW, int TW, int q, int t, int hw, bool repack) {
  LsRound R;
  R.tw = TW; R.cw = CW; R.qc = q; R.tr = t; R.ts = t; R.j = t; R.has = ((nm >> t) & 1ull) != 0; R.repacked = false;
  if (repack && c0 > 0) {
    const int u = __popcll(nm);
    const int tw2 = u;  // one row per searching trajectory (any count: the gains staging and the lane split take arbitrary row counts)
    if (tw2 < TW && 64 / tw2 >= total - c0) {  // wave-uniform
      R.tw = tw2; R.cw = total - c0; R.repacked = true;
      R.qc = hw / tw2; R.tr = hw - R.qc * tw2;
      int cnt = 0, ts = -1, last = 0;
      for (int i = 0; i < TW; ++i)
        if ((nm >> i) & 1ull) { ts = (cnt == R.tr) ? i : ts; last = i; ++cnt; }
      R.has = ts >= 0;
      R.ts = ts >= 0 ? ts : last;  // rows past the last searching trajectory ride along on it
      R.j = __popcll(nm & ((1ull << t) - 1ull));
    }
  }
  return R;
}
// step size number i of the backtracking search: the products the sequential search forms
__device__ __forceinline__ double ls_alpha(double f, int i, int total) {
  double al = 1.0;
  for (int k = 0; k < total; ++k) al = (k < i) ? al * f : al;
  return al;
}

}  // namespace to

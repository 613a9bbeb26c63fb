// Compiles csrc/ls_round.h for the host and checks the lane map of the line-search rounds over every wave shape the forward pass
// uses and many need masks (test infrastructure; tests/test_ls_round_host.py).  A wave is simulated lane by lane.
#include <hip/hip_runtime.h>
#include "ls_round.h"

#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace to;

static int fails = 0;
#define CHECK(c, ...) do { if (!(c)) { if (fails < 20) { printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } ++fails; } } while (0)

int main() {
  long long rounds = 0, repacked = 0;
  const int shapes[][2] = {{16, 4}, {8, 8}, {4, 16}, {2, 32}, {1, 64}, {20, 3}, {32, 2}};
  for (const auto& sh : shapes) {
    const int CW = sh[0], TW = sh[1];
    for (int total : {20, 10, 33, 64}) {
      for (int trial = 0; trial < 400; ++trial) {
        unsigned long long nm = 0;
        const int density = trial % 5;  // from almost empty to full
        for (int t = 0; t < TW; ++t) if (rand() % 5 <= density) nm |= 1ull << t;
        if (nm == 0) nm = 1ull << (rand() % TW);
        for (int c0 = 0; c0 < total; c0 += CW) {
          for (int rp = 0; rp < 2; ++rp) {
            ++rounds;
            LsRound R[64];
            for (int hw = 0; hw < 64; ++hw) R[hw] = ls_round(nm, c0, total, CW, TW, hw / TW, hw % TW, hw, rp != 0);
            const LsRound& R0 = R[0];
            for (int hw = 0; hw < 64; ++hw) CHECK(R[hw].tw == R0.tw && R[hw].cw == R0.cw && R[hw].repacked == R0.repacked, "map not wave-uniform");
            const int u = __builtin_popcountll(nm);
            if (!R0.repacked) {
              CHECK(R0.tw == TW && R0.cw == CW, "static map shape");
              for (int hw = 0; hw < 64; ++hw) {
                const int q = hw / TW, t = hw % TW;
                CHECK(R[hw].qc == q && R[hw].tr == t && R[hw].ts == t && R[hw].j == t, "static map lane %d", hw);
                CHECK(R[hw].has == (((nm >> t) & 1) != 0), "static has");
              }
              // a repack was possible but not taken?  only when it is not allowed, not smaller, or does not finish the search
              if (rp && c0 > 0) CHECK(!(u < TW && 64 / u >= total - c0), "repack skipped although it fits");
              continue;
            }
            ++repacked;
            CHECK(rp && c0 > 0, "repacked without permission");
            CHECK(R0.tw < TW && R0.tw == u && R0.tw * R0.cw <= 64 && R0.cw == total - c0, "repacked shape tw %d cw %d u %d", R0.tw, R0.cw, u);
            // every (searching trajectory, remaining step size) is evaluated by exactly one lane that has a candidate
            std::vector<int> seen(TW * 64, 0);
            for (int hw = 0; hw < 64; ++hw) {
              const LsRound& r = R[hw];
              CHECK(r.qc == hw / r.tw && r.tr == hw % r.tw, "lane decomposition");
              CHECK(r.ts >= 0 && r.ts < TW && ((nm >> r.ts) & 1), "lane %d works for a trajectory that is not searching", hw);
              const bool cand = r.has && r.qc < r.cw && c0 + r.qc < total;
              if (cand) ++seen[r.ts * 64 + r.qc];
              if (r.tr < r.tw && hw < r.tw) CHECK(r.qc == 0, "row lanes");  // lanes 0..tw-1 hold the rows' trajectories (stage_gains reads b there)
            }
            for (int t = 0; t < TW; ++t)
              for (int qc = 0; qc < R0.cw; ++qc) CHECK(seen[t * 64 + qc] == (((nm >> t) & 1) ? 1 : 0), "coverage t %d qc %d: %d", t, qc, seen[t * 64 + qc]);
            // the owner lanes find their candidates: lane (qs * tw + j) works for the owner's trajectory, on step size qs
            for (int t = 0; t < TW; ++t) {
              if (!((nm >> t) & 1)) continue;
              const int j = R[t].j;  // lane t has q = 0 and trajectory t under the static map
              for (int q = 0; q * TW + t < 64; ++q) CHECK(R[q * TW + t].j == j, "row of a trajectory differs between its lanes");
              for (int qs = 0; qs < R0.cw; ++qs) {
                const int src = (qs * R0.tw + j) & 63;
                CHECK(R[src].ts == t && R[src].qc == qs && R[src].has, "owner of trajectory %d does not find step %d", t, qs);
              }
            }
          }
        }
      }
    }
  }
  // step sizes: the sequential product chain
  for (double f : {0.5, 0.7, 0.1}) {
    double a = 1.0;
    for (int i = 0; i < 25; ++i) { CHECK(ls_alpha(f, i, 25) == a, "alpha %d", i); a *= f; }
    CHECK(ls_alpha(f, 30, 25) == ls_alpha(f, 25, 25), "alpha beyond the depth");
  }
  printf("rounds %lld repacked %lld fails %d\n", rounds, repacked, fails);
  return fails ? 1 : 0;
}

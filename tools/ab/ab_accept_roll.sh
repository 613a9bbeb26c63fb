#!/bin/bash
# A/B of accept-by-rollout (TRAJOPT_ACCEPT_ROLL_MIN: forward waves from which a batch step stores candidate controls only;
# 0 = never): bench lines of C3 / C5 at several batch sizes.  Run on the GPU box: gpurun -- 'bash tools/ab/ab_accept_roll.sh'
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
for spec in "quadrotor_altro 4096" "quadrotor_altro 2048" "quadrotor_altro 16384" "quadrotor 8192" "quadrotor 16384" "quadrotor 2048"; do
  set -- $spec
  for w in 0 1; do
    TRAJOPT_ACCEPT_ROLL_MIN=$w python bench.py --workload $1 --batch $2 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/roll_$1_$2_$w.json
    python - <<PY
import json; d=json.load(open("gpurun_out/roll_$1_$2_$w.json")); k=d["roofline"]["kernels"]; print("$1", $2, $w, round(d["value"]), round(d["ms_per_step"],2), {p: round(k[p]["avg_us"]) for p in k}, d["config"].get("converged_fraction"))
PY
  done
done

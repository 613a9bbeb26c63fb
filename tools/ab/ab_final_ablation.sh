#!/bin/bash
# Ablation of the round-4 forward-pass / polish changes on ONE build: C5 (and C3) with each switched off in turn, two repetitions.
# GPU box: gpurun -- 'bash tools/ab/ab_final_ablation.sh'
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/ablation
run() {  # tag workload env...
  tag=$1; wl=$2; shift 2
  env "$@" python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/ablation/${tag}.json
  python - <<PY
import json; d=json.load(open("gpurun_out/ablation/${tag}.json")); k=d["roofline"]["kernels"]; print("${tag}", round(d["value"]), round(d["ms_per_step"],2), {p: round(k[p]["avg_us"]) for p in k})
PY
}
for rep in 1 2; do
  run c5_all_on_$rep quadrotor_altro X=1
  run c5_no_early_polish_$rep quadrotor_altro TRAJOPT_PN_EARLY=0
  run c5_no_repack_$rep quadrotor_altro TRAJOPT_LS_REPACK=0
  run c5_no_accept_roll_$rep quadrotor_altro TRAJOPT_ACCEPT_ROLL_MIN=0
  run c5_all_off_$rep quadrotor_altro TRAJOPT_PN_EARLY=0 TRAJOPT_LS_REPACK=0 TRAJOPT_ACCEPT_ROLL_MIN=0
done
run c3_all_on quadrotor X=1
run c3_all_off quadrotor TRAJOPT_LS_REPACK=0 TRAJOPT_ACCEPT_ROLL_MIN=0

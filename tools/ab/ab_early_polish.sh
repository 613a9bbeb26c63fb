#!/bin/bash
# A/B of the early polish (TRAJOPT_PN_EARLY: hand-overs of finished trajectories to the polish stream during the AL stage; 0 = one
# polish after it): the equality test, then bench lines of C5.  GPU box: gpurun -- 'bash tools/ab/ab_early_polish.sh'
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
python -m pytest tests/test_gpu_pn.py -q -m gpu -x -k "early_polish or altro_solve_vs_oracle or async" > gpurun_out/early_test.log 2>&1; grep -E "passed|failed|Error|assert" gpurun_out/early_test.log | tail -8
for e in 0 1 3; do
  TRAJOPT_PN_EARLY=$e python bench.py --workload quadrotor_altro --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/early_$e.json
  python - <<PY
import json; d=json.load(open("gpurun_out/early_$e.json")); k=d["roofline"]["kernels"]; print("early", $e, round(d["value"]), round(d["ms_per_step"],2), {p: round(k[p]["avg_us"]) for p in k}, d["config"].get("converged_fraction"), d["config"]["projected_newton"]["ms_per_solve"])
PY
done

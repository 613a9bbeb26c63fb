cd /root/repo
for spec in "quadrotor_altro 8" "quadrotor_altro 16" "quadrotor_altro 4" "quadrotor 16" "quadrotor 8"; do set -- $spec
  TRAJOPT_LS_CANDIDATES=$2 python bench.py --workload $1 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/cw_$1_$2.json
  python - <<PY
import json; d=json.load(open("gpurun_out/cw_$1_$2.json")); k=d["roofline"]["kernels"]; print("$1", "CW", $2, round(d["value"]), round(d["ms_per_step"],2), {p: round(k[p]["avg_us"]) for p in k})
PY
done

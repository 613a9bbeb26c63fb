cd /root/repo
python -m pytest tests/test_gpu_parity.py -q -m gpu -x -k "repack or accept_by_rollout or two_wave" > gpurun_out/repack_test.log 2>&1; grep -E "passed|failed|Error|assert" gpurun_out/repack_test.log | tail -8
for rp in 0 1; do for wl in quadrotor_altro quadrotor; do
  TRAJOPT_LS_REPACK=$rp python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/rp_${wl}_$rp.json
  python - <<PY
import json; d=json.load(open("gpurun_out/rp_${wl}_$rp.json")); k=d["roofline"]["kernels"]; print("$wl", $rp, round(d["value"]), round(d["ms_per_step"],2), {p: round(k[p]["avg_us"]) for p in k}, d["config"].get("converged_fraction"))
PY
done; done

#!/bin/bash
# Round-3 evidence run (final build) — full GPU suite, default bench line, rocprofv3 kernel stats + PMC passes per workload.
set -u
repo=$(pwd); out=$repo/gpurun_out/r3g2; mkdir -p "$out"
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > "$out/pytest.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest.log"; tail -4 "$out/pytest.log"
PROFILE_COPY=r03_cartpole_b1024 bash tools/run_profiles.sh r3g2/cartpole_b1024 --workload cartpole > "$out/prof_cartpole.log" 2>&1
PROFILE_COPY=r03_quadrotor_b4096 bash tools/run_profiles.sh r3g2/quadrotor_b4096 --workload quadrotor --steps 2 > "$out/prof_quadrotor.log" 2>&1
PROFILE_COPY=r03_quadrotor_al_b8192 bash tools/run_profiles.sh r3g2/quadrotor_al_b8192 --workload quadrotor_al --steps 1 > "$out/prof_quadrotor_al.log" 2>&1
timeout 900 python bench.py > "$out/bench_default.json" 2> "$out/bench_default.log"; echo "bench rc=$?"
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d "$out/probe_kt" -o kt --output-format csv -- python "$repo/bench.py" --batch 131072 --steps 1 --warmup 0 --no-cpu-baseline --no-extra --throughput-probe 0 --no-profile > "$out/probe_b131072_under_trace.json" 2> "$out/probe_kt.log"; cd "$repo"
cp $(find "$out/probe_kt" -name '*kernel_stats.csv' | head -1) "$out/probe_b131072_kernel_stats.csv" 2>/dev/null
rm -rf "$out"/*/kt "$out"/*/fetch "$out"/*/write "$out"/*/valu "$out/probe_kt" 2>/dev/null
python - <<'PY'
import json
r=json.load(open('gpurun_out/r3g2/bench_default.json'))
print('C2', round(r['value']), r['roofline']['kernel'], r['roofline']['frac'], {k:round(v['avg_us'],1) for k,v in r['roofline']['kernels'].items()}, r['config']['solver_path'])
print('probe', r.get('throughput_probe')); print('plateau', r.get('throughput_sweep',{}).get('plateau'))
for k,v in r.get('extra_workloads',{}).items(): print(k, round(v.get('value',0)), {n:round(x['avg_us'],1) for n,x in v.get('roofline',{}).get('kernels',{}).items()}, v.get('cpu_baseline',{}).get('value'))
print('cpu', r.get('cpu_baseline',{}).get('value'), r.get('cpu_baseline',{}).get('cores'), r.get('c1_cpu',{}).get('value'))
PY
ls -la "$out" "$out"/* | head -60

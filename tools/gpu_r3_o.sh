#!/bin/bash
# Round-3 GPU call O: two-wave forward pass (k_forward2) — parity, then interleaved A/B on C3 / C5.
set -u
repo=$(pwd); out=$repo/gpurun_out/r3o; mkdir -p "$out"
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x -k "two_wave" > "$out/pytest_new.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_new.log"; tail -15 "$out/pytest_new.log"
for rep in 1 2; do
  for two in 0 1; do
    TRAJOPT_FWD2=$two timeout 300 python bench.py --workload quadrotor --steps 3 --no-cpu-baseline --no-extra --throughput-probe 0 --no-probe-sweep > "$out/c3_fwd2_${two}_$rep.json" 2> "$out/c3_fwd2_${two}_$rep.log"
  done
done
for two in 0 1; do
  TRAJOPT_FWD2=$two timeout 300 python bench.py --workload quadrotor_al --steps 1 --no-cpu-baseline --no-extra --throughput-probe 0 --no-probe-sweep > "$out/c5_fwd2_$two.json" 2> "$out/c5_fwd2_$two.log"
done
cd /tmp
for two in 0 1; do
  TRAJOPT_FWD2=$two timeout 300 rocprofv3 --kernel-trace --stats -d "$out/kt$two" -o kt --output-format csv -- python "$repo/bench.py" --workload quadrotor --steps 1 --warmup 0 --no-cpu-baseline --no-extra --throughput-probe 0 --no-probe-sweep --no-profile > /dev/null 2> "$out/kt$two.log"
  cp $(find "$out/kt$two" -name '*kernel_stats.csv' | head -1) "$out/kernel_stats_fwd2_$two.csv"; rm -rf "$out/kt$two"
  head -4 "$out/kernel_stats_fwd2_$two.csv" | cut -c1-150
done
cd "$repo"
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r3o/c*_fwd2_*.json')):
    try:
        r = json.load(open(f))
        print(f.split('/')[-1], round(r['value']), {k: round(v['avg_us'], 1) for k, v in r['roofline']['kernels'].items()})
    except Exception as e:
        print(f, 'ERR', e)
PY

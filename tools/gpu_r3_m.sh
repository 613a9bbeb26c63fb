#!/bin/bash
set -u
repo=$(pwd); out=$repo/gpurun_out/r3m; mkdir -p "$out"
export TMPDIR=/tmp
L=$repo/trajectoryoptimization.jl_amd/csrc
for lib in "" _flw2; do
  TRAJOPT_HIP_LIBRARY=$L/libtrajopt_hip$lib.so timeout 900 python bench.py --no-extra --no-cpu-baseline > "$out/bench_c2$lib.json" 2> "$out/bench_c2$lib.log"
  python - <<PY
import json
r=json.load(open('gpurun_out/r3m/bench_c2$lib.json'))
print('$lib C2', round(r['value']))
print([(p.get('batch'), round(p.get('value',0)/1e6,1), round(p.get('whole_iteration_frac',0),3), p.get('error')) for p in r['throughput_sweep']['points']])
PY
done
TRAJOPT_HIP_LIBRARY=$L/libtrajopt_hip_flw2.so timeout 600 python -m pytest tests -m gpu -q -x -k "compaction or fused_lane or lane_backward" > "$out/pytest_flw2.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_flw2.log"; tail -3 "$out/pytest_flw2.log"

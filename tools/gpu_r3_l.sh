#!/bin/bash
set -u
repo=$(pwd); out=$repo/gpurun_out/r3l; mkdir -p "$out"
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x -k "compaction or fused_lane" > "$out/pytest_new.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_new.log"; tail -3 "$out/pytest_new.log"
timeout 900 python bench.py --no-extra --no-cpu-baseline > "$out/bench_c2.json" 2> "$out/bench_c2.log"
python - <<'PY'
import json
r=json.load(open('gpurun_out/r3l/bench_c2.json'))
print('C2', round(r['value']), r['roofline']['traffic_source'])
print([(p['batch'], round(p['value']/1e6,1), round(p['whole_iteration_frac'],3)) for p in r['throughput_sweep']['points']])
PY

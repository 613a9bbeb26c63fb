#!/bin/bash
# Round-3 GPU call U: full GPU suite with the scan backward pass as default, default bench line.
set -u
repo=$(pwd); out=$repo/gpurun_out/r3u; mkdir -p "$out"
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > "$out/pytest.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest.log"; tail -8 "$out/pytest.log"
timeout 900 python bench.py > "$out/bench_default.json" 2> "$out/bench_default.log"; echo "bench rc=$?"
python - <<'PY'
import json
r=json.load(open('gpurun_out/r3u/bench_default.json'))
print('C2', round(r['value']), r['roofline']['kernel'], r['roofline']['frac'], {k:round(v['avg_us'],1) for k,v in r['roofline']['kernels'].items()}, r['config']['solver_path'])
print('probe', r.get('throughput_probe')); print('plateau', r.get('throughput_sweep',{}).get('plateau'))
for k,v in r.get('extra_workloads',{}).items(): print(k, round(v.get('value',0)), {n:round(x['avg_us'],1) for n,x in v.get('roofline',{}).get('kernels',{}).items()}, v.get('cpu_baseline',{}).get('value'))
print('cpu', r.get('cpu_baseline',{}).get('value'), r.get('cpu_baseline',{}).get('cores'), r.get('c1_cpu',{}).get('value'))
PY

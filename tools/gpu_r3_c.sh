#!/bin/bash
# Round-3 GPU call C: full GPU test-suite, active-list compaction A/B, default bench line.
set -u
repo=$(pwd); out=$repo/gpurun_out/r3c; mkdir -p "$out"
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > "$out/pytest.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest.log"; tail -8 "$out/pytest.log"
ab() { # tag, env..., -- args
  tag=$1; shift
  ( while [ "$1" != "--" ]; do export "$1"; shift; done; shift
    timeout 300 python bench.py --steps 2 --warmup 1 --no-extra --no-cpu-baseline --throughput-probe 0 --no-profile "$@" 2> "$out/ab_$tag.log" | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$tag', round(r['value']), r['ms_per_step'], r['config']['batch_steps_per_solve'])" ) >> "$out/ab.txt" 2>&1
}
ab b32k_compact -- --batch 32768
ab b32k_nocompact TRAJOPT_COMPACT=0 -- --batch 32768
ab b128k_compact -- --batch 131072
ab b128k_nocompact TRAJOPT_COMPACT=0 -- --batch 131072
ab b256k_compact -- --batch 262144
ab b128k_compact_cw4 TRAJOPT_LS_CANDIDATES=4 -- --batch 131072
ab b128k_compact_cw1 TRAJOPT_LS_CANDIDATES=1 -- --batch 131072
ab b16k_default -- --batch 16384
cat "$out/ab.txt"
timeout 900 python bench.py > "$out/bench_default.json" 2> "$out/bench_default.log"; echo "bench rc=$?"
python - <<'PY'
import json
r=json.load(open('gpurun_out/r3c/bench_default.json'))
print('C2', r['value'], r['roofline']['kernels'])
print('probe', r.get('throughput_probe')); print('sweep', r.get('throughput_sweep'))
for k,v in r.get('extra_workloads',{}).items(): print(k, v.get('value'), v.get('roofline',{}).get('kernels'))
print('cpu', r.get('cpu_baseline',{}).get('value'), r.get('c1_cpu'))
PY

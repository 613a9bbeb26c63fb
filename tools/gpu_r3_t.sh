#!/bin/bash
# Round-3 GPU call T: scan backward pass — parity test, C2 A/B (interleaved), B = 4096.
set -u
repo=$(pwd); out=$repo/gpurun_out/r3t; mkdir -p "$out"
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "scan_backward or ilqr_solve_cartpole or hybrid" > "$out/pytest_new.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_new.log"; tail -25 "$out/pytest_new.log"
for rep in 1 2; do
  for sc in 0 1; do
    TRAJOPT_SCAN=$sc timeout 300 python bench.py --no-cpu-baseline --no-extra --throughput-probe 0 --no-probe-sweep > "$out/c2_scan_${sc}_$rep.json" 2> "$out/c2_scan_${sc}_$rep.log"
  done
done
for sc in 0 1; do
  TRAJOPT_SCAN=$sc TRAJOPT_SCAN_MAX=100000 timeout 300 python bench.py --batch 4096 --no-cpu-baseline --no-extra --throughput-probe 0 --no-probe-sweep > "$out/c2b4096_scan_${sc}.json" 2> "$out/c2b4096_scan_${sc}.log"
  TRAJOPT_SCAN=$sc TRAJOPT_SCAN_MAX=100000 timeout 300 python bench.py --batch 8192 --no-cpu-baseline --no-extra --throughput-probe 0 --no-probe-sweep > "$out/c2b8192_scan_${sc}.json" 2> "$out/c2b8192_scan_${sc}.log"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r3t/c2*_scan_*.json')):
    try:
        r = json.load(open(f))
        print(f.split('/')[-1], round(r['value']), {k: round(v['avg_us'], 1) for k, v in r['roofline']['kernels'].items()}, r['config'].get('converged_fraction'))
    except Exception as e:
        print(f, 'ERR', e)
PY

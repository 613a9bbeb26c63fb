#!/bin/bash
# Round-3 GPU call X: scan kernel with its own sequential pass (no second launch) — targeted tests, C2 bench.
set -u
repo=$(pwd); out=$repo/gpurun_out/r3x; mkdir -p "$out"
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "scan_backward or ilqr_solve_cartpole or hybrid or three_parameter or C2 or c2" > "$out/pytest_new.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_new.log"; tail -12 "$out/pytest_new.log"
for rep in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --no-extra --throughput-probe 0 --no-probe-sweep > "$out/c2_$rep.json" 2> "$out/c2_$rep.log"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r3x/c2_*.json')):
    r = json.load(open(f))
    print(f.split('/')[-1], round(r['value']), {k: round(v['avg_us'], 1) for k, v in r['roofline']['kernels'].items()}, r['config']['solver_path'])
PY

#!/bin/bash
# Round-3 GPU call R: full GPU suite on the build with the hybrid model + per-step two-wave forward pass; C3 / C5 A/B (interleaved).
set -u
repo=$(pwd); out=$repo/gpurun_out/r3r; mkdir -p "$out"
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > "$out/pytest.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest.log"; tail -8 "$out/pytest.log"
for rep in 1 2; do
  for two in 0 2; do
    TRAJOPT_FWD2=$two timeout 300 python bench.py --workload quadrotor --steps 3 --no-cpu-baseline --no-extra --throughput-probe 0 --no-probe-sweep > "$out/c3_fwd2_${two}_$rep.json" 2> "$out/c3_fwd2_${two}_$rep.log"
  done
done
for two in 0 2; do
  TRAJOPT_FWD2=$two timeout 300 python bench.py --workload quadrotor_al --steps 1 --no-cpu-baseline --no-extra --throughput-probe 0 --no-probe-sweep > "$out/c5_fwd2_$two.json" 2> "$out/c5_fwd2_$two.log"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r3r/c*_fwd2_*.json')):
    try:
        r = json.load(open(f))
        print(f.split('/')[-1], round(r['value']), {k: round(v['avg_us'], 1) for k, v in r['roofline']['kernels'].items()})
    except Exception as e:
        print(f, 'ERR', e)
PY

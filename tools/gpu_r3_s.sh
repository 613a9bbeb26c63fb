#!/bin/bash
# Round-3 GPU call S: full GPU suite (small-model two-wave forward pass), C2 A/B (interleaved), probe at B = 32 768.
set -u
repo=$(pwd); out=$repo/gpurun_out/r3s; mkdir -p "$out"
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > "$out/pytest.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest.log"; tail -8 "$out/pytest.log"
for rep in 1 2 3; do
  for two in 0 2; do
    TRAJOPT_FWD2=$two timeout 300 python bench.py --no-cpu-baseline --no-extra --throughput-probe 0 --no-probe-sweep > "$out/c2_fwd2_${two}_$rep.json" 2> "$out/c2_fwd2_${two}_$rep.log"
  done
done
for two in 0 2; do
  TRAJOPT_FWD2=$two timeout 300 python bench.py --batch 4096 --no-cpu-baseline --no-extra --throughput-probe 0 --no-probe-sweep > "$out/c2b4096_fwd2_${two}.json" 2> "$out/c2b4096_fwd2_${two}.log"
  TRAJOPT_FWD2=$two timeout 300 python bench.py --batch 32768 --steps 2 --no-cpu-baseline --no-extra --throughput-probe 0 --no-probe-sweep > "$out/c2b32768_fwd2_${two}.json" 2> "$out/c2b32768_fwd2_${two}.log"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r3s/c2*_fwd2_*.json')):
    try:
        r = json.load(open(f))
        print(f.split('/')[-1], round(r['value']), {k: round(v['avg_us'], 1) for k, v in r['roofline']['kernels'].items()})
    except Exception as e:
        print(f, 'ERR', e)
PY

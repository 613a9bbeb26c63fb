#!/bin/bash
# Round-3 GPU call V: where the scan + cooperative path hands over to the fused lane path (Cartpole, interleaved).
set -u
repo=$(pwd); out=$repo/gpurun_out/r3v; mkdir -p "$out"
export TMPDIR=/tmp
for B in 12288 16384 24576 32768; do
  for path in coop lane; do
    TRAJOPT_BACKWARD=$path timeout 300 python bench.py --batch $B --steps 2 --no-cpu-baseline --no-extra --throughput-probe 0 --no-probe-sweep > "$out/b${B}_$path.json" 2> "$out/b${B}_$path.log"
  done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r3v/b*.json')):
    try:
        r = json.load(open(f))
        print(f.split('/')[-1], round(r['value']), {k: round(v['avg_us'], 1) for k, v in r['roofline']['kernels'].items()}, r['config']['solver_path'])
    except Exception as e:
        print(f, 'ERR', e)
PY

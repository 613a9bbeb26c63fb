#!/bin/bash
# Round-3 GPU call AC: compact-layout cost part of the Quadrotor expansion as dot products (variant library) — parity, then C3 / C5
# interleaved against the default library.
set -u
repo=$(pwd); out=$repo/gpurun_out/r3ad; mkdir -p "$out"
export TMPDIR=/tmp
D=$repo/trajectoryoptimization.jl_amd/csrc/libtrajopt_hip.so
V=$repo/trajectoryoptimization.jl_amd/csrc/libvar_stage.so
TRAJOPT_HIP_LIBRARY=$V timeout 900 python -m pytest tests -m gpu -q -x -k "expansion or quadrotor or three_parameter or error_quadratic or zigzag or C3 or C5 or c3 or c5 or full_size" > "$out/pytest_V.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_V.log"; tail -4 "$out/pytest_V.log"
for rep in 1 2; do
  for lib in D V; do
    eval L=\$$lib
    TRAJOPT_HIP_LIBRARY=$L timeout 300 python bench.py --workload quadrotor --steps 3 --no-cpu-baseline --no-extra --throughput-probe 0 --no-probe-sweep > "$out/c3_${lib}_$rep.json" 2> "$out/c3_${lib}_$rep.log"
  done
done
for lib in D V; do
  eval L=\$$lib
  TRAJOPT_HIP_LIBRARY=$L timeout 300 python bench.py --workload quadrotor_al --steps 1 --no-cpu-baseline --no-extra --throughput-probe 0 --no-probe-sweep > "$out/c5_${lib}.json" 2> "$out/c5_${lib}.log"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r3ad/c*.json')):
    try:
        r = json.load(open(f))
        print(f.split('/')[-1], round(r['value']), {k: round(v['avg_us'], 1) for k, v in r['roofline']['kernels'].items()})
    except Exception as e:
        print(f, 'ERR', e)
PY

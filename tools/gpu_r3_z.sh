#!/bin/bash
# Round-3 GPU call Z: roller reads the whole gains block in one LDS round trip (variant library) vs one per row — C3 / C5, interleaved.
set -u
repo=$(pwd); out=$repo/gpurun_out/r3z; mkdir -p "$out"
export TMPDIR=/tmp
V=$repo/trajectoryoptimization.jl_amd/csrc/libvar_gains_once.so
D=$repo/trajectoryoptimization.jl_amd/csrc/libtrajopt_hip.so
TRAJOPT_HIP_LIBRARY=$V timeout 600 python -m pytest tests -m gpu -q -x -k "two_wave_forward_pass and not small" > "$out/pytest_var.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_var.log"; tail -3 "$out/pytest_var.log"
for rep in 1 2; do
  for lib in D V; do
    L=$D; [ $lib = V ] && L=$V
    TRAJOPT_HIP_LIBRARY=$L timeout 300 python bench.py --workload quadrotor --steps 3 --no-cpu-baseline --no-extra --throughput-probe 0 --no-probe-sweep > "$out/c3_${lib}_$rep.json" 2> "$out/c3_${lib}_$rep.log"
  done
done
for lib in D V; do
  L=$D; [ $lib = V ] && L=$V
  TRAJOPT_HIP_LIBRARY=$L timeout 300 python bench.py --workload quadrotor_al --steps 1 --no-cpu-baseline --no-extra --throughput-probe 0 --no-probe-sweep > "$out/c5_${lib}.json" 2> "$out/c5_${lib}.log"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r3z/c*.json')):
    try:
        r = json.load(open(f))
        print(f.split('/')[-1], round(r['value']), {k: round(v['avg_us'], 1) for k, v in r['roofline']['kernels'].items()})
    except Exception as e:
        print(f, 'ERR', e)
PY

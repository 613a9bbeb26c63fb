#!/bin/bash
# Round-3 GPU call AB: knots per ring group of the two-wave forward pass (barriers per rollout): D = 4 (small) / 1 (LDS gains),
# G1 = 1 / 1 (previous), G2 = 2 / 2.  Parity of the default first, then C2 and C3 interleaved.
set -u
repo=$(pwd); out=$repo/gpurun_out/r3ab; mkdir -p "$out"
export TMPDIR=/tmp
D=$repo/trajectoryoptimization.jl_amd/csrc/libtrajopt_hip.so
G1=$repo/trajectoryoptimization.jl_amd/csrc/libvar_g1.so
G2=$repo/trajectoryoptimization.jl_amd/csrc/libvar_g2.so
timeout 600 python -m pytest tests -m gpu -q -x -k "two_wave or ilqr_solve_cartpole or al_solve_cartpole or hybrid_model_vector_on_gpu or quickstart" > "$out/pytest_D.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_D.log"; tail -3 "$out/pytest_D.log"
TRAJOPT_HIP_LIBRARY=$G2 timeout 600 python -m pytest tests -m gpu -q -x -k "two_wave" > "$out/pytest_G2.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_G2.log"; tail -3 "$out/pytest_G2.log"
for rep in 1 2; do
  for lib in D G1 G2; do
    eval L=\$$lib
    TRAJOPT_HIP_LIBRARY=$L timeout 300 python bench.py --no-cpu-baseline --no-extra --throughput-probe 0 --no-probe-sweep > "$out/c2_${lib}_$rep.json" 2> "$out/c2_${lib}_$rep.log"
  done
done
for lib in D G2; do
  eval L=\$$lib
  TRAJOPT_HIP_LIBRARY=$L timeout 300 python bench.py --workload quadrotor --steps 3 --no-cpu-baseline --no-extra --throughput-probe 0 --no-probe-sweep > "$out/c3_${lib}.json" 2> "$out/c3_${lib}.log"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r3ab/c*.json')):
    try:
        r = json.load(open(f))
        print(f.split('/')[-1], round(r['value']), {k: round(v['avg_us'], 1) for k, v in r['roofline']['kernels'].items()})
    except Exception as e:
        print(f, 'ERR', e)
PY

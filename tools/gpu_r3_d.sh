#!/bin/bash
# Round-3 GPU call D: new-model tests + Quadrotor regression tests, per-phase breakdown at large batches, expansion-variant A/B.
set -u
repo=$(pwd); out=$repo/gpurun_out/r3d; mkdir -p "$out"
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -k "three_parameter or quadrotor or Quadrotor or error_quadratic or C3 or every_constraint or expansion" > "$out/pytest.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest.log"; tail -6 "$out/pytest.log"
phase() { # tag, env..., -- args : per-phase kernel times from the profiled pass
  tag=$1; shift
  ( while [ "$1" != "--" ]; do export "$1"; shift; done; shift
    timeout 300 python bench.py --steps 2 --warmup 1 --no-extra --no-cpu-baseline --throughput-probe 0 "$@" 2> "$out/ph_$tag.log" | python -c "
import sys,json; r=json.loads(sys.stdin.read()); k=r['roofline']['kernels']
print('$tag', round(r['value']), 'ms/solve', round(r['ms_per_step'],1), 'steps', r['config']['batch_steps_per_solve'], {n:(round(v['avg_us'],1), v['launches']) for n,v in k.items()})" ) >> "$out/phase.txt" 2>&1
}
phase b128k_cw2 -- --batch 131072
phase b128k_cw4 TRAJOPT_LS_CANDIDATES=4 -- --batch 131072
phase b32k_cw2 -- --batch 32768
phase b32k_cw4 TRAJOPT_LS_CANDIDATES=4 -- --batch 32768
phase b32k_cw4_nocompact TRAJOPT_LS_CANDIDATES=4 TRAJOPT_COMPACT=0 -- --batch 32768
phase b16k_coop TRAJOPT_BACKWARD=coop -- --batch 16384
phase b16k_lane_cw4 TRAJOPT_LS_CANDIDATES=4 -- --batch 16384
for lib in "" _ew2kc1 _ew1kc1 _ew2kc2; do
  phase c3$lib TRAJOPT_HIP_LIBRARY=$repo/trajectoryoptimization.jl_amd/csrc/libtrajopt_hip$lib.so -- --workload quadrotor
  phase c5$lib TRAJOPT_HIP_LIBRARY=$repo/trajectoryoptimization.jl_amd/csrc/libtrajopt_hip$lib.so -- --workload quadrotor_al
done
cat "$out/phase.txt"

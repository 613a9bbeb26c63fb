#!/bin/bash
# Round-3 GPU call E: RP NaN hunt, new tests, repeated C3/C5 A/B (expansion variants, compaction), probe with 4 step sizes.
set -u
repo=$(pwd); out=$repo/gpurun_out/r3e; mkdir -p "$out"
export TMPDIR=/tmp
timeout 300 python tools/debug_rp.py > "$out/debug_rp.txt" 2>&1; cat "$out/debug_rp.txt"
timeout 900 python -m pytest tests -m gpu -q -k "three_parameter or compaction or fused or lane_ or C3 or C5" > "$out/pytest.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest.log"; tail -6 "$out/pytest.log"
phase() {
  tag=$1; shift
  ( while [ "$1" != "--" ]; do export "$1"; shift; done; shift
    timeout 300 python bench.py --steps 2 --warmup 1 --no-extra --no-cpu-baseline --throughput-probe 0 "$@" 2> "$out/ph_$tag.log" | python -c "
import sys,json; r=json.loads(sys.stdin.read()); k=r['roofline']['kernels']
print('$tag', round(r['value']), 'ms/solve', round(r['ms_per_step'],1), 'steps', r['config']['batch_steps_per_solve'], {n:(round(v['avg_us'],1), v['launches']) for n,v in k.items()})" ) >> "$out/phase.txt" 2>&1
}
L=$repo/trajectoryoptimization.jl_amd/csrc
for rep in 1 2 3; do
  phase c3_main_$rep -- --workload quadrotor
  phase c3_main_nocompact_$rep TRAJOPT_COMPACT=0 -- --workload quadrotor
  phase c3_ew2kc1_$rep TRAJOPT_HIP_LIBRARY=$L/libtrajopt_hip_ew2kc1.so -- --workload quadrotor
  phase c3_ew1kc1_$rep TRAJOPT_HIP_LIBRARY=$L/libtrajopt_hip_ew1kc1.so -- --workload quadrotor
done
phase c5_main -- --workload quadrotor_al
phase c5_main_nocompact TRAJOPT_COMPACT=0 -- --workload quadrotor_al
phase c5_ew2kc1 TRAJOPT_HIP_LIBRARY=$L/libtrajopt_hip_ew2kc1.so -- --workload quadrotor_al
phase b128k -- --batch 131072
phase b32k -- --batch 32768
phase c2 --
cat "$out/phase.txt"

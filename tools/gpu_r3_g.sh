#!/bin/bash
# Round-3 GPU call G: fused cooperative pass with prefetch + merged symmetrisation.
set -u
repo=$(pwd); out=$repo/gpurun_out/r3g; mkdir -p "$out"
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x -k "fused_cooperative or C2 or ilqr_solve_cartpole or ragged" > "$out/pytest_new.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_new.log"; tail -4 "$out/pytest_new.log"
TRAJOPT_COOP_MERGE=0 timeout 600 python -m pytest tests -m gpu -q -x -k "fused_cooperative" > "$out/pytest_nomerge.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_nomerge.log"; tail -3 "$out/pytest_nomerge.log"
phase() {
  tag=$1; shift
  ( while [ "$1" != "--" ]; do export "$1"; shift; done; shift
    timeout 300 python bench.py --steps 5 --warmup 1 --no-extra --no-cpu-baseline --throughput-probe 0 "$@" 2> "$out/ph_$tag.log" | python -c "
import sys,json; r=json.loads(sys.stdin.read()); k=r['roofline']['kernels']
print('$tag', round(r['value']), 'ms/solve', round(r['ms_per_step'],2), 'steps', r['config']['batch_steps_per_solve'], {n:(round(v['avg_us'],1), v['launches']) for n,v in k.items()})" ) >> "$out/phase.txt" 2>&1
}
for rep in 1 2; do
  phase c2_fused_merge_$rep --
  phase c2_fused_nomerge_$rep TRAJOPT_COOP_MERGE=0 --
  phase c2_split_$rep TRAJOPT_FUSED_COOP=0 --
done
phase b2k_fused -- --batch 2048
phase b2k_split TRAJOPT_FUSED_COOP=0 -- --batch 2048
phase b4k_fused -- --batch 4096
phase b8k_fused -- --batch 8192
phase b12k_fused -- --batch 12288
phase b12k_lane TRAJOPT_BACKWARD=lane -- --batch 12288
cat "$out/phase.txt"

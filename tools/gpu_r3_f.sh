#!/bin/bash
# Round-3 GPU call F: fused cooperative pass — tests, C2 A/B (3 repetitions), full suite.
set -u
repo=$(pwd); out=$repo/gpurun_out/r3f; mkdir -p "$out"
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x -k "fused_cooperative or three_parameter" > "$out/pytest_new.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_new.log"; tail -6 "$out/pytest_new.log"
phase() {
  tag=$1; shift
  ( while [ "$1" != "--" ]; do export "$1"; shift; done; shift
    timeout 300 python bench.py --steps 5 --warmup 1 --no-extra --no-cpu-baseline --throughput-probe 0 "$@" 2> "$out/ph_$tag.log" | python -c "
import sys,json; r=json.loads(sys.stdin.read()); k=r['roofline']['kernels']
print('$tag', round(r['value']), 'ms/solve', round(r['ms_per_step'],2), 'steps', r['config']['batch_steps_per_solve'], {n:(round(v['avg_us'],1), v['launches']) for n,v in k.items()})" ) >> "$out/phase.txt" 2>&1
}
for rep in 1 2 3; do
  phase c2_fused_$rep --
  phase c2_split_$rep TRAJOPT_FUSED_COOP=0 --
done
phase b4k_fused -- --batch 4096
phase b4k_split TRAJOPT_FUSED_COOP=0 -- --batch 4096
phase b8k_fused -- --batch 8192
phase b8k_split TRAJOPT_FUSED_COOP=0 -- --batch 8192
cat "$out/phase.txt"
timeout 1500 python -m pytest tests -m gpu -q > "$out/pytest.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest.log"; tail -6 "$out/pytest.log"

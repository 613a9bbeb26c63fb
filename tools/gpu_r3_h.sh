#!/bin/bash
set -u
repo=$(pwd); out=$repo/gpurun_out/r3h; mkdir -p "$out"
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -x -k "fused_cooperative" > "$out/pytest_new.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_new.log"; tail -3 "$out/pytest_new.log"
TRAJOPT_COOP_MERGE=0 timeout 600 python -m pytest tests -m gpu -q -x -k "fused_cooperative" > "$out/pytest_nomerge.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_nomerge.log"; tail -3 "$out/pytest_nomerge.log"
phase() {
  tag=$1; shift
  ( while [ "$1" != "--" ]; do export "$1"; shift; done; shift
    timeout 300 python bench.py --steps 5 --warmup 1 --no-extra --no-cpu-baseline --throughput-probe 0 "$@" 2> "$out/ph_$tag.log" | python -c "
import sys,json; r=json.loads(sys.stdin.read()); k=r['roofline']['kernels']
print('$tag', round(r['value']), 'ms/solve', round(r['ms_per_step'],2), 'steps', r['config']['batch_steps_per_solve'], {n:(round(v['avg_us'],1), v['launches']) for n,v in k.items()})" ) >> "$out/phase.txt" 2>&1
}
phase c2_fused_merge --
phase c2_fused_nomerge TRAJOPT_COOP_MERGE=0 --
phase c2_split TRAJOPT_FUSED_COOP=0 --
phase b4k_fused_merge -- --batch 4096
phase b4k_fused_nomerge TRAJOPT_COOP_MERGE=0 -- --batch 4096
cat "$out/phase.txt"

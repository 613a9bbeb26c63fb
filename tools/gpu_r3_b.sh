#!/bin/bash
# Round-3 GPU call B: full GPU test-suite (no -x), fused lane path A/B at large batches.
set -u
repo=$(pwd); out=$repo/gpurun_out/r3b; mkdir -p "$out"
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > "$out/pytest.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest.log"; tail -8 "$out/pytest.log"
ab() { # tag, env..., -- args
  tag=$1; shift
  ( while [ "$1" != "--" ]; do export "$1"; shift; done; shift
    timeout 300 python bench.py --steps 2 --warmup 1 --no-extra --no-cpu-baseline --throughput-probe 0 --no-profile "$@" 2> "$out/ab_$tag.log" | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$tag', round(r['value']), r['ms_per_step'], r['config']['batch_steps_per_solve'])" ) >> "$out/ab.txt" 2>&1
}
ab b32k_fused -- --batch 32768
ab b32k_split TRAJOPT_FUSED_LANE=0 -- --batch 32768
ab b64k_fused -- --batch 65536
ab b128k_fused -- --batch 131072
ab b128k_split TRAJOPT_FUSED_LANE=0 -- --batch 131072
ab b256k_fused -- --batch 262144
ab b128k_fused_cw2 TRAJOPT_LS_CANDIDATES=2 -- --batch 131072
ab b128k_fused_cw1 TRAJOPT_LS_CANDIDATES=1 -- --batch 131072
ab b16k_default -- --batch 16384
ab b16k_lane TRAJOPT_BACKWARD=lane -- --batch 16384
ab b8k_lane TRAJOPT_BACKWARD=lane -- --batch 8192
cat "$out/ab.txt"

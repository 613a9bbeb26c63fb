#!/bin/bash
# Round-3 GPU call A: full GPU test-suite, the default bench line, and A/B probes of the small-model throughput path.
set -u
repo=$(pwd); out=$repo/gpurun_out/r3a; mkdir -p "$out"
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > "$out/pytest.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest.log"; tail -5 "$out/pytest.log"
timeout 600 python bench.py > "$out/bench_default.json" 2> "$out/bench_default.log"; echo "bench rc=$?"; head -c 1200 "$out/bench_default.json"; echo
ab() { # tag, env..., -- args
  tag=$1; shift
  ( while [ "$1" != "--" ]; do export "$1"; shift; done; shift
    timeout 300 python bench.py --steps 2 --warmup 1 --no-extra --no-cpu-baseline --throughput-probe 0 --no-profile "$@" 2> "$out/ab_$tag.log" | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$tag', round(r['value']), r['ms_per_step'], r['config']['batch_steps_per_solve'])" ) >> "$out/ab.txt" 2>&1
}
ab b32k_lane1 -- --batch 32768
ab b32k_lane0 TRAJOPT_EXPAND_LANE=0 -- --batch 32768
ab b32k_cw1 TRAJOPT_LS_CANDIDATES=1 -- --batch 32768
ab b32k_cw2 TRAJOPT_LS_CANDIDATES=2 -- --batch 32768
ab b32k_coop TRAJOPT_BACKWARD=coop -- --batch 32768
ab b8k_default -- --batch 8192
ab b8k_lane TRAJOPT_BACKWARD=lane -- --batch 8192
ab b4k_default -- --batch 4096
ab b4k_lane TRAJOPT_BACKWARD=lane -- --batch 4096
ab b1k_default -- --batch 1024
ab b1k_lane TRAJOPT_BACKWARD=lane -- --batch 1024
ab b128k_lane1 -- --batch 131072
cat "$out/ab.txt"

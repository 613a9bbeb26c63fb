#!/bin/bash
# Per-dispatch kernel durations of one solve (GPU box): tools/trace_durations.sh <tag> <workload>; gpurun_out/<tag>/<kernel>.txt
tag=$1; w=$2
repo=$(pwd); out=$repo/gpurun_out/$tag; mkdir -p "$out"
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace -d "$out/kt" -o kt --output-format csv -- python "$repo/bench.py" --workload $w --steps 1 --warmup 0 --no-cpu-baseline --no-extra --no-profile --throughput-probe 0 > "$out/trace.json" 2> "$out/kt.log"
cd "$repo"
csv=$(find "$out/kt" -name '*kernel_trace.csv' | head -1)
for k in k_forward k_expand k_backward "k_accept(" k_accept_roll; do python tools/kernel_durations.py "$csv" $k > "$out/${k%(}.txt"; done
python tools/kernel_durations.py "$csv" --summary > "$out/summary.txt"
rm -rf "$out/kt"

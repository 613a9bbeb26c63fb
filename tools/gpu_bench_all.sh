#!/bin/bash
# Usage (on the GPU box, via gpurun): tools/gpu_bench_all.sh <tag>
# Default bench line (C2 + extra_workloads C3/C5) and a rocprofv3 kernel-trace summary per workload under gpurun_out/<tag>/.
tag=$1
repo=$(pwd); out=$repo/gpurun_out/$tag; mkdir -p "$out"
python bench.py > "$out/bench.json" 2> "$out/bench.err"
export TMPDIR=/tmp; cd /tmp
for w in cartpole quadrotor quadrotor_altro; do
  steps=3; [ $w = quadrotor_altro ] && steps=1
  timeout 300 rocprofv3 --kernel-trace --stats -d "$out/kt_$w" -o kt --output-format csv -- python "$repo/bench.py" --workload $w --steps $steps --warmup 1 --no-cpu-baseline --no-extra --no-profile --throughput-probe 0 > "$out/trace_$w.json" 2> "$out/kt_$w.log"
  cp $(find "$out/kt_$w" -name '*kernel_stats.csv' | head -1) "$out/${w}_kernel_stats.csv" 2>/dev/null
  rm -rf "$out/kt_$w"
done
cd "$repo"

#!/bin/bash
# Profile one bench workload on the GPU box: kernel-trace stats, then one PMC pass per HBM counter (never combined with
# tracing), then the plain bench line.  Usage: tools/run_profiles.sh <tag> <bench args...>; results in gpurun_out/<tag>/.
set -u
tag=$1; shift
repo=$(pwd)
out=$repo/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d "$out/kt" -o kt --output-format csv -- python "$repo/bench.py" "$@" --no-cpu-baseline > "$out/bench_under_trace.json" 2> "$out/kt.log"
rocprofv3 --pmc FETCH_SIZE -d "$out/fetch" -o fetch --output-format csv -- python "$repo/bench.py" "$@" --steps 1 --warmup 0 --no-cpu-baseline --no-profile > /dev/null 2> "$out/fetch.log"
rocprofv3 --pmc WRITE_SIZE -d "$out/write" -o write --output-format csv -- python "$repo/bench.py" "$@" --steps 1 --warmup 0 --no-cpu-baseline --no-profile > /dev/null 2> "$out/write.log"
cd "$repo"
python tools/pmc_traffic.py "$out/hbm_traffic_pmc.json" $(find "$out/fetch" "$out/write" -name '*counter_collection.csv')
cp $(find "$out/kt" -name '*kernel_stats.csv' | head -1) "$out/kernel_stats.csv"
python bench.py "$@" > "$out/bench.json" 2> "$out/bench.log"
tail -c 600 "$out/bench.json"

#!/bin/bash
# Usage: tools/pmc_passes.sh <tag> "<counters pass1>" ["<counters pass2>" ...] -- <bench args>
# One rocprofv3 --pmc run per counter group (no tracing flags); CSVs under gpurun_out/<tag>/pN/.
tag=$1; shift
groups=()
while [ "$1" != "--" ]; do groups+=("$1"); shift; done
shift
repo=$(pwd); out=$repo/gpurun_out/$tag; mkdir -p "$out"
export TMPDIR=/tmp; cd /tmp
i=0
for g in "${groups[@]}"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $g -d "$out/p$i" -o p --output-format csv -- python "$repo/bench.py" "$@" --steps 1 --warmup 0 --no-cpu-baseline --no-profile > /dev/null 2> "$out/p$i.log"
done

#!/bin/bash
# A/B of library variants on the GPU box: tools/ab_lib.sh <tag> <lib1.so> <lib2.so> ... ; bench lines under gpurun_out/<tag>/
tag=$1; shift
mkdir -p gpurun_out/$tag
for lib in "$@"; do
  name=$(basename $lib .so)
  for w in quadrotor quadrotor_altro; do
    steps=2; [ $w = quadrotor_altro ] && steps=1
    TRAJOPT_HIP_LIBRARY=$PWD/$lib python bench.py --workload $w --steps $steps --no-cpu-baseline --no-extra --throughput-probe 0 > gpurun_out/$tag/${name}_$w.json 2>>gpurun_out/$tag/err.log
  done
done

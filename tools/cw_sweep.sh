#!/bin/bash
# first-round width of the line search (TRAJOPT_LS_CANDIDATES) on the Quadrotor workloads: bench lines under gpurun_out/cw_sweep/
mkdir -p gpurun_out/cw_sweep
for cw in 16 8 4; do
  for w in quadrotor quadrotor_altro; do
    steps=3; [ $w = quadrotor_altro ] && steps=1
    TRAJOPT_LS_CANDIDATES=$cw python bench.py --workload $w --steps $steps --no-cpu-baseline --no-extra --throughput-probe 0 > gpurun_out/cw_sweep/${w}_cw$cw.json 2>/dev/null
  done
done

#!/bin/bash
# One gpurun call: the profiles of the round (tools/run_profiles.sh per BASELINE workload, with the PMC file handed to bench.py),
# the large-batch points of the throughput sweep under the counters, then the driver-format line.  Usage: tools/gpu_round_profiles.sh r04
r=${1:-r06}
FINAL_FLAGS=--no-cpu-baseline PROFILE_COPY=${r}_cartpole_b1024 bash tools/run_profiles.sh ${r}_cartpole_b1024 --workload cartpole > /dev/null 2>&1
FINAL_FLAGS=--no-cpu-baseline PROFILE_COPY=${r}_quadrotor_b4096 bash tools/run_profiles.sh ${r}_quadrotor_b4096 --workload quadrotor > /dev/null 2>&1
FINAL_FLAGS=--no-cpu-baseline PROFILE_COPY=${r}_quadrotor_altro_b8192 bash tools/run_profiles.sh ${r}_quadrotor_altro_b8192 --workload quadrotor_altro > /dev/null 2>&1
for b in 32768 131072 1048576; do
  FINAL_FLAGS=--no-cpu-baseline PROFILE_COPY=${r}_cartpole_b$b bash tools/run_profiles.sh ${r}_cartpole_b$b --workload cartpole --batch $b --steps 1 --warmup 1 > /dev/null 2>&1
done
python bench.py > gpurun_out/${r}_bench_default.json 2> gpurun_out/${r}_bench_default.err
ls gpurun_out/${r}_*/kernel_stats.csv gpurun_out/${r}_*/hbm_traffic_pmc.json 2>&1 | head -20
tail -c 300 gpurun_out/${r}_bench_default.err

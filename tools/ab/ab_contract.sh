#!/bin/bash
# A/B: default build (-ffp-contract=fast, hipcc's default) vs -ffp-contract=on: bit-identity of the two forward kernels, bench lines
mkdir -p gpurun_out/ab_contract
L=trajectoryoptimization.jl_amd/csrc
python tools/ab/fwd2_bitwise.py > gpurun_out/ab_contract/bitwise_default.txt 2>&1
TRAJOPT_HIP_LIBRARY=$PWD/$L/libtrajopt_hip_contract_on.so python tools/ab/fwd2_bitwise.py > gpurun_out/ab_contract/bitwise_on.txt 2>&1
for rep in 1 2; do
for lib in libtrajopt_hip libtrajopt_hip_contract_on; do
  for w in cartpole quadrotor quadrotor_altro; do
    steps=10; [ $w = quadrotor ] && steps=3; [ $w = quadrotor_altro ] && steps=2
    TRAJOPT_HIP_LIBRARY=$PWD/$L/$lib.so python bench.py --workload $w --steps $steps --no-cpu-baseline --no-extra --throughput-probe 0 > gpurun_out/ab_contract/${lib}_${w}_$rep.json 2>>gpurun_out/ab_contract/err.log
  done
done
done
cat gpurun_out/ab_contract/bitwise_default.txt gpurun_out/ab_contract/bitwise_on.txt

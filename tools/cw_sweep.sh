mkdir -p gpurun_out/r2i
for cw in 4 8 16; do TRAJOPT_LS_CANDIDATES=$cw python bench.py --workload quadrotor --steps 2 --no-cpu-baseline --no-extra --throughput-probe 0 > gpurun_out/r2i/c3_cw$cw.json 2>>gpurun_out/r2i/err.log; done
for cw in 2 4 8; do TRAJOPT_LS_CANDIDATES=$cw python bench.py --workload quadrotor_al --steps 1 --no-cpu-baseline --no-extra --throughput-probe 0 > gpurun_out/r2i/c5_cw$cw.json 2>>gpurun_out/r2i/err.log; done
for cw in 4 8 16; do TRAJOPT_LS_CANDIDATES=$cw python bench.py --workload cartpole --steps 3 --no-cpu-baseline --no-extra --throughput-probe 0 > gpurun_out/r2i/c2_cw$cw.json 2>>gpurun_out/r2i/err.log; done

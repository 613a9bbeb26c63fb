#!/bin/bash
# Profile one bench workload on the GPU box: kernel-trace stats, then one PMC pass per HBM counter (never combined with
# tracing) — all of them on UNPIPELINED solves (--pipeline 0: one handle, per-kernel figures undisturbed) — then the plain bench line
# (pipelined where that is the workload's default).  Usage: tools/run_profiles.sh <tag> <bench args...>; results in gpurun_out/<tag>/.
set -u
tag=$1; shift
repo=$(pwd)
out=$repo/gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$out/kt" -o kt --output-format csv -- python "$repo/bench.py" "$@" --no-cpu-baseline --no-extra --throughput-probe 0 --no-probe-sweep --pipeline 0 > "$out/bench_under_trace.json" 2> "$out/kt.log"
timeout 300 rocprofv3 --pmc FETCH_SIZE -d "$out/fetch" -o fetch --output-format csv -- python "$repo/bench.py" "$@" --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-extra --throughput-probe 0 --no-probe-sweep --pipeline 0 > /dev/null 2> "$out/fetch.log"
timeout 300 rocprofv3 --pmc WRITE_SIZE -d "$out/write" -o write --output-format csv -- python "$repo/bench.py" "$@" --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-extra --throughput-probe 0 --no-probe-sweep --pipeline 0 > /dev/null 2> "$out/write.log"
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY -d "$out/valu" -o valu --output-format csv -- python "$repo/bench.py" "$@" --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-extra --throughput-probe 0 --no-probe-sweep --pipeline 0 > /dev/null 2> "$out/valu.log"
# matrix-core pass (k_backward_mfma runs v_mfma_f64_16x16x4_f64): instructions, MOPS (units of 512 flops), pipe-busy cycles
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d "$out/mfma" -o mfma --output-format csv -- python "$repo/bench.py" "$@" --steps 1 --warmup 0 --no-cpu-baseline --no-profile --no-extra --throughput-probe 0 --no-probe-sweep --pipeline 0 > /dev/null 2> "$out/mfma.log"
cd "$repo"
bid=$(python -c "import trajopt_amd as T; print(T.load_hip_library().build_id())")
python tools/pmc_traffic.py "$out/hbm_traffic_pmc.json" "$bid" $(find "$out/fetch" "$out/write" "$out/valu" "$out/mfma" -name '*counter_collection.csv')
cp $(find "$out/kt" -name '*kernel_stats.csv' | head -1) "$out/kernel_stats.csv"
# the plain bench line quotes the traffic just measured: bench.py reads profiles/*_<workload>_b<batch>_hbm_traffic_pmc.json and
# only accepts a file stamped with the build id of the library it loaded (PROFILE_COPY = that file name, e.g. r03_cartpole_b1024)
if [ -n "${PROFILE_COPY:-}" ]; then cp "$out/hbm_traffic_pmc.json" "profiles/${PROFILE_COPY}_hbm_traffic_pmc.json"; fi
python bench.py "$@" --no-extra --no-probe-sweep --throughput-probe 0 ${FINAL_FLAGS:-} > "$out/bench.json" 2> "$out/bench.log"
tail -c 600 "$out/bench.json"

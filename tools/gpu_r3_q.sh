#!/bin/bash
set -u
repo=$(pwd); out=$repo/gpurun_out/r3q; mkdir -p "$out"
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "hybrid or two_wave or host_api" > "$out/pytest_new.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_new.log"; tail -15 "$out/pytest_new.log"
bash tools/gpu_r3_p.sh

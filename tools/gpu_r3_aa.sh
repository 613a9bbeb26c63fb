#!/bin/bash
set -u
repo=$(pwd); out=$repo/gpurun_out/r3aa; mkdir -p "$out"
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q -k "scan_backward" > "$out/pytest_new.log" 2>&1; echo "pytest rc=$?" >> "$out/pytest_new.log"; tail -25 "$out/pytest_new.log"

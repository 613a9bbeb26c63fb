#!/usr/bin/env python3
"""Fold rocprofv3 --pmc counter_collection CSVs (one pass per counter) into per-kernel HBM bytes per launch.

Usage: pmc_traffic.py OUT.json BUILD_ID CSV [CSV ...]

BUILD_ID = to_build_id() of the library the passes ran (tools/run_profiles.sh passes it): stored under "__meta__" so
that bench.py only quotes a traffic figure measured on the binary it is timing.

FETCH_SIZE / WRITE_SIZE are reported in KB per dispatch.  Following /opt/skills/guides/MI355X_MICROARCH.md (HBM
section) FETCH_SIZE on gfx950 tallies 128-byte requests at 64 bytes, so the corrected figure doubles it; WRITE_SIZE is
taken as reported.  Both raw and corrected numbers are kept.
"""
import csv
import json
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def main(out, build_id, paths):
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for p in paths:
        with open(p, newline="") as f:
            for row in csv.DictReader(f):
                a = acc[short(row["Kernel_Name"])][row["Counter_Name"]]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
    res = {}
    for k, ctr in acc.items():
        e = {}
        fetch = write = 0.0
        for c, (tot, n) in ctr.items():
            e["launches"] = n
            if c not in ("FETCH_SIZE", "WRITE_SIZE"):
                e[c + "_per_launch"] = tot / n
                continue
            e[c + "_KB_per_launch"] = tot / n
            if c == "FETCH_SIZE":
                fetch = tot / n * 1024.0
            if c == "WRITE_SIZE":
                write = tot / n * 1024.0
        e["hbm_bytes_per_launch_raw"] = fetch + write
        e["hbm_bytes_per_launch_fetch_x2"] = 2.0 * fetch + write
        if "SQ_INSTS_VALU_FMA_F64_per_launch" in e:  # wave-level FP64 instructions -> flops (64 lanes, FMA = 2)
            e["fp64_flops_per_launch"] = 64.0 * (2.0 * e["SQ_INSTS_VALU_FMA_F64_per_launch"] + e.get("SQ_INSTS_VALU_ADD_F64_per_launch", 0.0)
                                                 + e.get("SQ_INSTS_VALU_MUL_F64_per_launch", 0.0) + e.get("SQ_INSTS_VALU_TRANS_F64_per_launch", 0.0))
        if "SQ_INSTS_VALU_MFMA_MOPS_F64_per_launch" in e:  # MOPS counts matrix-core operations in units of 512 flops
            e["mfma_f64_flops_per_launch"] = 512.0 * e["SQ_INSTS_VALU_MFMA_MOPS_F64_per_launch"]
            e["fp64_flops_per_launch_incl_mfma"] = e.get("fp64_flops_per_launch", 0.0) + e["mfma_f64_flops_per_launch"]
        res[k] = e
    res["__meta__"] = {"build_id": build_id}
    with open(out, "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3:])

#!/usr/bin/env python3
"""Per-dispatch durations of one kernel from a rocprofv3 --kernel-trace CSV: tools/kernel_durations.py <kernel_trace.csv> <substring>
(--summary instead of a substring: launches, total and mean duration per kernel, templates stripped, by total time)"""
import csv, re, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
us = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
if sys.argv[2] == "--summary":
    tot = defaultdict(lambda: [0, 0.0])
    for r in rows:
        name = re.sub(r"<.*", "", r["Kernel_Name"].replace("void ", "").replace("to::", ""))
        name = re.sub(r"\(.*", "", name)
        tot[name][0] += 1; tot[name][1] += us(r)
    for name, (cnt, t) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        print(f"{name:36s} {cnt:6d} launches {t / 1e3:10.3f} ms total {t / cnt:9.1f} us mean")
else:
    d = [us(r) for r in rows if sys.argv[2] in r["Kernel_Name"]]
    print(len(d), "dispatches; us:", " ".join(f"{x:.0f}" for x in d))

#!/usr/bin/env python3
"""List durations (us) of selected kernels in launch order.  Usage: trace_list.py CSV name [name...]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
out = []
for r in rows:
    k = r["Kernel_Name"]
    for nm in sys.argv[2:]:
        if ("to::" + nm) in k:
            out.append("%s:%.0f" % (nm[2:], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
print(" ".join(out))

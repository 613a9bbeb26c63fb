#!/usr/bin/env python3
"""Print the per-launch durations (us) of the first kernels of the last solve in a rocprofv3 kernel-trace CSV."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
seq = []
for r in rows:
    k = r["Kernel_Name"]
    if "to::k_" not in k:
        continue
    short = k.split("to::")[1].split("<")[0].split("(")[0]
    seq.append((short, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Grid_Size_X"]))
starts = [i for i, s in enumerate(seq) if s[0] == "k_solve_init"]
i0 = starts[-1]
print(" ".join("%s:%.0f" % (s[0][2:], s[1]) for s in seq[i0:i0 + n]))

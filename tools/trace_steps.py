#!/usr/bin/env python3
"""Per-batch-step timeline of the last solve in a kernel trace: kernel durations and the idle gaps between them (us)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "to::k_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seq = [(r["Kernel_Name"].split("to::")[1].split("<")[0].split("(")[0][2:], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
i0 = [i for i, s in enumerate(seq) if s[0] == "solve_init"][-1]
seq = seq[i0:]
every = int(sys.argv[2]) if len(sys.argv) > 2 else 25
step = -1; line = []; prev_end = None
for nm, st, en in seq:
    if nm == "expand":
        if line and step % every == 0: print(step, " ".join(line))
        step += 1; line = []
    gap = (st - prev_end) / 1e3 if prev_end else 0.0
    line.append("[%.0f]%s:%.0f" % (gap, nm, (en - st) / 1e3))
    prev_end = en
tot = (seq[-1][2] - seq[0][1]) / 1e3
busy = sum(en - st for _, st, en in seq) / 1e3
print("solve span %.0f us, kernels busy %.0f us, idle %.0f us, steps %d" % (tot, busy, tot - busy, step + 1))

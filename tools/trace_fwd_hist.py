#!/usr/bin/env python3
"""Sequence of k_forward launch durations of the last solve in a kernel trace (grouped per batch step)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
seq = []
for r in rows:
    k = r["Kernel_Name"]
    if "to::k_" not in k: continue
    short = k.split("to::")[1].split("<")[0].split("(")[0]
    seq.append((short, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
i0 = [i for i, s in enumerate(seq) if s[0] == "k_solve_init"][-1]
steps = []; cur = None
for s in seq[i0:]:
    if s[0] == "k_expand": cur = []; steps.append(cur)
    if s[0] == "k_forward" and cur is not None: cur.append(round(s[1]))
for i, st in enumerate(steps):
    if i % int(sys.argv[2] if len(sys.argv) > 2 else 8) == 0: print(i, st)

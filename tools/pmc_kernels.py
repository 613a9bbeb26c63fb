#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 --pmc counters.  Usage: pmc_kernels.py [--first K] [--filter SUBSTR] CSV [CSV ...]
--first K keeps only the K longest-running dispatches of each kernel (the fully occupied launches)."""
import csv, sys, re
from collections import defaultdict
args = sys.argv[1:]
first = None; flt = "to::k_"
while args and args[0].startswith("--"):
    if args[0] == "--first": first = int(args[1]); args = args[2:]
    elif args[0] == "--filter": flt = args[1]; args = args[2:]
per = defaultdict(lambda: defaultdict(list))
for p in args:
    for r in csv.DictReader(open(p)):
        k = r["Kernel_Name"]
        if flt not in k: continue
        k = re.sub(r"^void ", "", k); k = re.sub(r"\(.*$", "", k)
        per[k][r["Counter_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), float(r["Counter_Value"])))
for k, ctr in per.items():
    print(k)
    for c, lst in sorted(ctr.items()):
        if first: lst = sorted(lst, reverse=True)[:first]
        n = len(lst)
        print("   %-40s %14.1f   (n=%d, mean dur %.1f us)" % (c, sum(v for _, v in lst) / n, n, sum(d for d, _ in lst) / n / 1e3))

#!/usr/bin/env python3
"""How often do the later line-search rounds have work?  Counts k_forward launches by grid.y and duration class."""
import csv, sys
from collections import Counter
c = Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if "k_forward" not in r["Kernel_Name"]: continue
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    c[(r["Grid_Size_Y"], "busy" if d > 20 else "empty")] += 1
print(dict(c))

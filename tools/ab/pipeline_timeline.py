#!/usr/bin/env python3
"""Timeline of pipelined solves from a TRAJOPT_TRACE file: per handle and solve, when its batch crossed 100/75/50/25/10/5/1 % active,
when its stage ended, when its polish ran.  python tools/ab/pipeline_timeline.py trace.txt [B]"""
import sys
from collections import defaultdict

rows = [l.split() for l in open(sys.argv[1]) if l.strip()]
B = int(sys.argv[2]) if len(sys.argv) > 2 else max(int(r[3]) for r in rows)
t0 = min(float(r[1]) for r in rows)
byh = defaultdict(list)
for h, t, steps, act, tag in rows:
    byh[h].append((float(t) - t0, int(steps), int(act), tag))
names = {h: "H%d" % i for i, h in enumerate(sorted(byh, key=lambda k: byh[k][0][0]))}
events = []
for h, ev in byh.items():
    solve, prev_steps, marks = 0, -1, None
    for t, steps, act, tag in ev:
        if tag == "step" and (steps < prev_steps or marks is None):
            solve += 1
            marks = [0.75, 0.5, 0.25, 0.1, 0.05, 0.01]
            events.append((t, names[h], solve, "first report: step %d, %d active" % (steps, act)))
        if tag == "step":
            prev_steps = steps
            while marks and act <= marks[0] * B and act > 0:
                events.append((t, names[h], solve, "<= %g %% active (step %d, %d)" % (100 * marks.pop(0), steps, act)))
            if act == 0:
                events.append((t, names[h], solve, "stage over (step %d)" % steps))
                prev_steps = 1 << 30
        else:
            events.append((t, names[h], solve, "%s (%d trajectories)" % (tag, act)))
for t, h, solve, what in sorted(events):
    print("%9.1f ms  %s solve %d  %s" % (1e3 * t, h, solve, what))

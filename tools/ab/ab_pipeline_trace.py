#!/usr/bin/env python3
"""One short pipelined run under TRAJOPT_TRACE (set by the caller): python tools/ab/ab_pipeline_trace.py workload depth admit steps"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: E402,F401
import trajopt_amd as T  # noqa: E402
from trajectoryoptimization_jl_amd import configs  # noqa: E402
import bench  # noqa: E402

name, depth, admit, steps = sys.argv[1], int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
batch = int(sys.argv[5]) if len(sys.argv) > 5 else bench.WORKLOADS[name]["batch"]
lib = T.load_hip_library()
probs = [bench.build_problem(T, configs, name, batch, 0, 0, lib) for _ in range(depth)]
solvers = [bench.make_solver(T, configs, name, p) for p in probs]
u0 = bench.initial_controls_value(T, probs[0], name)
for s in solvers:
    T.initial_controls(s.prob, u0); s.solve()
import os
open(os.environ["TRAJOPT_TRACE"], "w").close()   # drop the warm-up solves
pipe = T.SolvePipeline(solvers, admit_below=int(admit * batch))
t0 = time.perf_counter()
for _ in range(steps):
    pipe.submit(lambda p: T.initial_controls(p, u0))
pipe.drain()
dt = time.perf_counter() - t0
print("%s depth %d admit %.2f: %.3f M it/s, %.1f ms per solve" % (name, depth, admit, pipe.total_iterations / dt / 1e6, 1e3 * dt / steps))

#!/usr/bin/env python3
"""Pipelined solves: depth x admission threshold, interleaved with the unpipelined solve, inside ONE gpurun call.
  python tools/ab/ab_pipeline.py [workload ...] > gpurun_out/ab_pipeline.jsonl
For each workload: unpipelined (mean of 3 solves), then for depth in (2, 3) and admit in (1.0, 0.75, 0.5, 0.35, 0.25, 0.125, 0.06):
`steps` pipelined solves, timed from the first submit to the last wait; the unpipelined figure again at the end."""
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401  (first: its HIP runtime serves the process)
import trajopt_amd as T  # noqa: E402
from trajectoryoptimization_jl_amd import configs  # noqa: E402
import bench  # noqa: E402


def main():
    names = sys.argv[1:] or ["quadrotor", "quadrotor_altro"]
    lib = T.load_hip_library()
    for name in names:
        batch = bench.WORKLOADS[name]["batch"]
        steps = int(os.environ.get("AB_STEPS", "12" if name == "quadrotor" else "24" if name == "cartpole" else "8"))
        grid = [(int(g.split(":")[0]), float(g.split(":")[1])) for g in os.environ.get("AB_GRID", "").split(",") if g] or \
               [(d, a) for d in (2, 3) for a in (1.0, 0.75, 0.5, 0.35, 0.25, 0.125, 0.06)]
        probs = [bench.build_problem(T, configs, name, batch, 0, 0, lib) for _ in range(max(d for d, _ in grid))]
        solvers = [bench.make_solver(T, configs, name, p) for p in probs]
        u0 = bench.initial_controls_value(T, probs[0], name)
        for s in solvers:
            T.initial_controls(s.prob, u0); s.solve()

        def unpipelined():
            ms = []
            for _ in range(3):
                T.initial_controls(probs[0], u0)
                t0 = time.perf_counter(); solvers[0].solve(); ms.append(time.perf_counter() - t0)
            return solvers[0].total_iterations / float(np.mean(ms)), 1e3 * float(np.mean(ms))

        v, ms = unpipelined()
        print(json.dumps({"workload": name, "depth": 1, "value": v, "ms_per_solve": ms}), flush=True)
        for depth, admit in grid:
            if True:
                pipe = T.SolvePipeline(solvers[:depth], admit_below=int(admit * batch))
                t0 = time.perf_counter()
                for _ in range(steps):
                    pipe.submit(lambda p: T.initial_controls(p, u0))
                pipe.drain()
                dt = time.perf_counter() - t0
                print(json.dumps({"workload": name, "hop": os.environ.get("TRAJOPT_HOP_FRAC", "default"), "depth": depth, "admit": admit, "steps": steps, "value": pipe.total_iterations / dt,
                                  "ms_per_solve": 1e3 * dt / steps}), flush=True)
        v, ms = unpipelined()
        print(json.dumps({"workload": name, "depth": 1, "value": v, "ms_per_solve": ms, "when": "after"}), flush=True)
        del solvers, probs, pipe, s
        import gc; gc.collect()


if __name__ == "__main__":
    main()

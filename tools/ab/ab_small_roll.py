#!/usr/bin/env python3
"""A/B of accept-by-rollout on the small models (TRAJOPT_ACCEPT_ROLL_MIN: active trajectories from which a batch step stores candidate
controls only and re-rolls the accepted ones; 0 = never): Cartpole C2 shape at several batch sizes, thresholds interleaved inside one
process (one `gpurun` call), `reps` timed solves each after a warm one, per-phase hipEvent times from a separate profiled solve.
  gpurun -- 'python tools/ab/ab_small_roll.py 32768,131072,1048576 0,1,32768 3 > gpurun_out/ab_small_roll.json'"""
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import numpy as np

try:
    import torch  # noqa: F401  (its HIP runtime first, as in bench.py)
    torch.cuda.is_available()
except ImportError:
    pass
import trajopt_amd as T
from trajectoryoptimization_jl_amd import configs

lib = T.load_hip_library()
batches = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "32768,131072,1048576").split(",")]
# variants: "0,1,32768" = values of TRAJOPT_ACCEPT_ROLL_MIN ("default": unset), or "name:ENV=v+ENV=v,name2:..." for arbitrary knobs
thresholds = (sys.argv[2] if len(sys.argv) > 2 else "0,1").split(",")
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
W = 48064.0
out = []
for B in batches:
    for rnd in range(2):  # interleaved repetitions
        for thr in thresholds:
            for k in [k for k in os.environ if k.startswith("TRAJOPT_")]:
                del os.environ[k]
            if ":" in thr:
                for kv in thr.split(":", 1)[1].split("+"):
                    if kv:
                        k, v = kv.split("=")
                        os.environ[k] = v
            elif thr != "default":
                os.environ["TRAJOPT_ACCEPT_ROLL_MIN"] = thr
            p = configs.cartpole_problem(batch=B, lib=lib)
            s = T.iLQRSolver(p)
            u0 = np.full(p.m, 0.01)
            s.solve()
            ms, its = [], 0
            for _ in range(reps):
                T.initial_controls(p, u0)
                t = time.perf_counter(); s.solve(); ms.append(1e3 * (time.perf_counter() - t))
                its = s.total_iterations
            p._call("set_profiling", 1); p._call("reset_profile")
            T.initial_controls(p, u0); s.solve()
            import ctypes as C
            pm, pl = (C.c_double * 4)(), (C.c_int64 * 4)()
            p._call("get_profile", pm, pl)
            p._call("set_profiling", 0)
            r = {"batch": B, "roll_min": thr, "round": rnd, "ms": [round(x, 2) for x in ms], "M_it_s": round(its / (np.mean(ms) * 1e-3) / 1e6, 3),
                 "frac": round(W * its / (np.mean(ms) * 1e-3) / 8e12, 4), "batch_steps": int(s.batch_steps),
                 "phase_us_per_step": [round(1e3 * pm[i] / max(1, pl[i]), 1) for i in range(3)]}
            print(json.dumps(r), flush=True)
            out.append(r)
            del s, p

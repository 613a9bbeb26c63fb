#!/usr/bin/env python3
"""Dev probe: phase-kernel launch times (hipEvent profile slots) on a fresh vs a converged batch, all lanes active."""
import ctypes as C, sys
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import trajopt_amd as T
from trajectoryoptimization_jl_amd import configs, internal

def prof(prob, label):
    prob._call("reset_profile"); prob._call("set_profiling", 1)
    for _ in range(5):
        internal.expand(prob); internal.backwardpass(prob); internal.forwardpass(prob)
    prob._call("set_profiling", 0)
    ms = (C.c_double * 4)(); ln = (C.c_int64 * 4)()
    prob._call("get_profile", ms, ln)
    print(label, {k: round(1e3 * ms[i] / max(1, ln[i]), 1) for i, k in enumerate(["expand", "backward", "forward"])})

for name in sys.argv[1:]:
    prob = configs.cartpole_problem() if name == "cartpole" else configs.quadrotor_problem()
    T.rollout(prob)
    prof(prob, name + " fresh    ")
    s = T.iLQRSolver(prob); s.solve()
    prof(prob, name + " converged")

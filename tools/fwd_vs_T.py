#!/usr/bin/env python3
"""Dev probe: run the phase API (expand, backward, forward) a few times; used under rocprofv3 --kernel-trace with
TRAJOPT_LS_CANDIDATES=T to see how the k_forward launch time depends on the number of concurrent candidate waves."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import trajopt_amd as T
from trajectoryoptimization_jl_amd import configs, internal
name = sys.argv[1]
prob = configs.cartpole_problem() if name == "cartpole" else configs.quadrotor_problem()
T.rollout(prob)
for _ in range(4):
    internal.expand(prob); internal.backwardpass(prob); internal.forwardpass(prob)

#!/usr/bin/env python3
"""Histogram of the accepted line-search index per iLQR iteration (phase API), to size the concurrent first round."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import trajopt_amd as T
from trajopt_amd import internal as I
from trajectoryoptimization_jl_amd import configs
lib = T.load_hip_library()
name = sys.argv[1] if len(sys.argv) > 1 else "quadrotor"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
p = configs.quadrotor_problem(batch=B, lib=lib) if name == "quadrotor" else configs.cartpole_problem(batch=B, lib=lib)
T.rollout(p)
for it in range(40):
    I.expand(p); I.backwardpass(p)
    ls, J = I.forwardpass(p)
    h = np.bincount(np.where(ls < 0, 20, ls), minlength=21)
    print(it, 'J mean %.4f' % J.mean(), 'ls hist', {i: int(c) for i, c in enumerate(h) if c})

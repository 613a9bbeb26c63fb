"""A/B of the forward-wave shape on the GPU: the same constrained Quadrotor batch stepped through the phase API with 16 and
with 8 line-search candidates per round (TRAJOPT_LS_CANDIDATES) must stay bit-identical; prints the first divergence.
Found the spill-placement miscompile recorded in DESIGN.md (tools/check_exec_spill.py scans for it)."""
import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import torch; torch.cuda.is_available()
import trajopt_amd as T
from trajopt_amd import internal as I
from trajectoryoptimization_jl_amd import configs
lib = T.load_hip_library()
os.environ["TRAJOPT_LS_DEEP"] = "0"
probs = []
for cw in ("16", "8"):
    os.environ["TRAJOPT_LS_CANDIDATES"] = cw
    o = T.SolverOptions(lib=lib, constraint_tolerance=1e-4)
    p = configs.quadrotor_problem(batch=24, N=101, tf=5.0, constrained=True, lib=lib, options=o)
    T.rollout(p); probs.append(p)
for it in range(60):
    out = []
    for p in probs:
        if it % 15 == 14: I.dual_update(p)
        I.expand(p); I.backwardpass(p)
        ls, J = I.forwardpass(p)
        out.append((ls, J, T.states(p)))
    (l0, J0, X0), (l1, J1, X1) = out
    same = np.array_equal(l0, l1) and np.array_equal(J0, J1) and np.array_equal(X0, X1)
    print(it, 'ls16', l0.tolist(), 'same' if same else 'DIFF ls8 %s dJ %s' % (l1.tolist(), np.abs(J0 - J1).max()))
    if not same: break

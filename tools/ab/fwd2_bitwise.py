"""k_forward vs k_forward2 on the same inputs: are J / X / U bit-identical?  (TRAJOPT_HIP_LIBRARY selects the build)"""
import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import torch; torch.cuda.is_available()
import trajopt_amd as T
from trajopt_amd import internal as I
from trajectoryoptimization_jl_amd import configs
lib = T.load_hip_library()
print("library", os.environ.get("TRAJOPT_HIP_LIBRARY", "default"), lib.build_id())
os.environ["TRAJOPT_LS_DEEP"] = "0"
for name, build in (("quad_con", lambda: configs.quadrotor_problem(batch=24, N=101, tf=5.0, constrained=True, lib=lib, options=T.SolverOptions(lib=lib, constraint_tolerance=1e-4))),
                    ("cartpole", lambda: configs.cartpole_problem(batch=40, lib=lib)),
                    ("cartpole_con", lambda: configs.cartpole_problem(batch=40, constrained=True, lib=lib))):
    probs = []
    for two in ("0", "1"):
        os.environ["TRAJOPT_FWD2"] = two
        p = build(); T.rollout(p); probs.append(p)
    worstJ = worstX = 0.0; ndiff = 0
    for it in range(30):
        out = []
        for p in probs:
            if p.constraints and it % 10 == 9: I.dual_update(p)
            I.expand(p); I.backwardpass(p)
            ls, J = I.forwardpass(p)
            out.append((ls, J, T.states(p), T.controls(p)))
        (l0, J0, X0, U0), (l1, J1, X1, U1) = out
        same = np.array_equal(l0, l1) and np.array_equal(J0, J1) and np.array_equal(X0, X1) and np.array_equal(U0, U1)
        ndiff += not same
        worstJ = max(worstJ, np.abs(J0 - J1).max() / np.abs(J0).max()); worstX = max(worstX, np.abs(X0 - X1).max())
        if not np.array_equal(l0, l1): print("  ls index differs at", it); break
    print(f"{name}: iterations with any bit difference {ndiff}/30, worst rel dJ {worstJ:.2e}, worst |dX| {worstX:.2e}")

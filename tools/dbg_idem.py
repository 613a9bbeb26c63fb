import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import trajopt_amd as T
from trajectoryoptimization_jl_amd import configs
p = configs.cartpole_problem(batch=1024)
T.rollout(p)
s = T.iLQRSolver(p).solve()
X = T.states(p).copy(); U = T.controls(p).copy()
T.rollout(p)
X2 = T.states(p).copy()
T.rollout(p)
X3 = T.states(p).copy()
print("rollout idempotent:", np.array_equal(X2, X3))
d = np.abs(X2 - X)
print("X shape", X.shape, "max diff", d.max())
bad = np.argwhere(d > 0)
print("first mismatches (b,k,i):", bad[:5].tolist())
# per knot count
print("mismatch count per knot (first 12):", [(int((d[:, k] > 0).sum())) for k in range(12)])
print("its of mismatching trajectories:", np.unique(s.stats["iterations"][np.unique(bad[:,0])])[:20], "status", np.unique(s.stats["status"][np.unique(bad[:,0])]))
print("ls index", None)

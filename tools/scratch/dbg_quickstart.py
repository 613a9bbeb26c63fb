import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
import trajopt_amd as T
from trajectoryoptimization_jl_amd import configs
from oracle_binding import load_oracle
oracle = load_oracle(); hip = T.load_hip_library()
def mk(lib, B=3):
    p = configs.quickstart_problem(batch=B, lib=lib); T.initial_controls(p, np.array([0.1, 0.0])); return p
for tol in (1e-3, 1e-6):
  for outer in (3,4,5,6,7):
    r=[]
    for lib in (hip, oracle):
        p = mk(lib)
        s=T.ALSolver(p, iterations_outer=outer, constraint_tolerance=tol).solve()
        r.append((int(s.stats['iterations'][0]), int(s.stats['iterations_outer'][0]), int(s.stats['status'][0]), float(s.stats['c_max'][0]), float(s.stats['dJ'][0])))
    print(tol, outer, r)
for lib in (hip, oracle):
    p = mk(lib); s=T.ALTROSolver(p).solve()
    print({k: s.stats[k] for k in ('iterations','iterations_outer','iterations_pn','status','c_max')})

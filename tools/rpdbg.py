import sys, os; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
import torch; torch.cuda.is_available()
import trajopt_amd as T
from trajectoryoptimization_jl_amd import configs
lib=T.load_hip_library()
os.environ["TRAJOPT_REPACK"]="8192"; os.environ["TRAJOPT_REPACK_AT"]="0.35"
p=configs.cartpole_problem(batch=int(os.environ.get("RPB","1048576")), lib=lib)
s=T.iLQRSolver(p)
for i in range(3):
    T.initial_controls(p, np.full(1,0.01)); s.solve(); print("solve", i, s.total_iterations, flush=True)

"""trajectoryoptimization.jl_amd — MI355X-native batched iLQR / augmented-Lagrangian hot path behind
TrajectoryOptimization.jl's Problem / Objective / AbstractConstraint / KnotPoint API surface.

All arithmetic lives in ``csrc/libtrajopt_hip.so`` (hand-written HIP for gfx950, C-ABI in
``include/trajopt_hip.h``); this package is the host-side mirror of the reference interface.
Import as ``import trajopt_amd`` (root shim) because the directory name contains a dot.
"""
from . import _capi as capi
from ._capi import load_hip_library, HipLibraryMissing, HipError
from .api import *  # noqa: F401,F403
from .api import stage_costs
from . import internal

__version__ = "0.1.0"

"""Builds csrc/libtrajopt_hip.so with hipcc for gfx950 (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

CSRC = Path(__file__).resolve().parent / "csrc"
INCLUDE = Path(__file__).resolve().parent.parent / "include"
TARGET = CSRC / "libtrajopt_hip.so"
SOURCES = ["trajopt_hip.hip"]
HEADERS = ["kernels.h", "models.h", "problem_dev.h"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value"]


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found")


def is_stale():
    if not TARGET.exists():
        return True
    t = TARGET.stat().st_mtime
    deps = [CSRC / s for s in SOURCES + HEADERS] + [INCLUDE / "trajopt_hip.h"]
    return any(d.stat().st_mtime > t for d in deps)


def build_hip(force=False, verbose=False, extra_flags=()):
    """Compile the HIP kernels + C-ABI into csrc/libtrajopt_hip.so.  Returns the path."""
    if not force and not is_stale():
        return TARGET
    cmd = [hipcc_path(), *FLAGS, *extra_flags, "-o", str(TARGET), *[str(CSRC / s) for s in SOURCES]]
    res = subprocess.run(cmd, capture_output=True, text=True, cwd=str(CSRC))
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + res.stderr[-4000:])
    if verbose:
        print(res.stderr)
    return TARGET

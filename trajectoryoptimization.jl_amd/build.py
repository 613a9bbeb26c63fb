"""Builds csrc/libtrajopt_hip.so with hipcc for gfx950 (cross-compiles without a GPU).

The binary is stamped with a hash of the sources it was compiled from (``to_build_id()``); ``build_hip`` recompiles
whenever the stamp of the existing binary differs from the tree — modification times are not trusted (a ``*.so`` is
git-ignored and travels with snapshots, so a stale binary would otherwise go unnoticed)."""
from __future__ import annotations

import hashlib
import os
import re
import shutil
import subprocess
from pathlib import Path

CSRC = Path(__file__).resolve().parent / "csrc"
INCLUDE = Path(__file__).resolve().parent.parent / "include"
TARGET = CSRC / "libtrajopt_hip.so"
SOURCES = ["trajopt_hip.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value"]


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found")


def source_files():
    return sorted(CSRC.glob("*.hip")) + sorted(CSRC.glob("*.h")) + [INCLUDE / "trajopt_hip.h"]


def source_id(extra_flags=()):
    """Hash of every source the library is compiled from, the flags included."""
    h = hashlib.sha256()
    for f in source_files():
        h.update(f.name.encode())
        h.update(f.read_bytes())
    h.update(" ".join(list(FLAGS) + list(extra_flags)).encode())
    return h.hexdigest()[:16]


def binary_id(path=TARGET):
    """Build stamp of an existing binary, read from the file (not dlopen'ed: a library loaded before a rebuild stays
    mapped under the same path).  None when absent or unstamped."""
    path = Path(path)
    if not path.exists():
        return None
    m = re.search(rb"TO_BUILD_ID=([0-9a-f]{16})", path.read_bytes())
    return m.group(1).decode() if m else None


def is_stale(extra_flags=()):
    return binary_id() != source_id(extra_flags)


def build_hip(force=False, verbose=False, extra_flags=(), target=TARGET):
    """Compile the HIP kernels + C-ABI into csrc/libtrajopt_hip.so.  Returns the path."""
    target = Path(target)
    sid = source_id(extra_flags)
    if not force and binary_id(target) == sid:
        return target
    cmd = [hipcc_path(), *FLAGS, *extra_flags, f'-DTO_BUILD_ID="{sid}"', "-o", str(target), *[str(CSRC / s) for s in SOURCES]]
    res = subprocess.run(cmd, capture_output=True, text=True, cwd=str(CSRC))
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + res.stderr[-4000:])
    if verbose:
        print(res.stderr)
    return target

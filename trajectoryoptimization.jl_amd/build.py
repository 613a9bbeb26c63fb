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
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]
# Per translation unit.  The Quadrotor expansion: SimplifyCFG's common-code sinking merges the per-row blocks of the AL
# terms into one block that takes the ADDRESSES of the gradient / Hessian-vector register arrays (a phi of pointers),
# which forces those arrays into scratch memory (96 B per lane, a memory round trip per use at one wave per SIMD).
FILE_FLAGS = {"ops_quad_expand.hip": ["-mllvm", "-simplifycfg-sink-common=false"]}
# The forward-pass translation units are compiled with -ffp-contract=on: fused multiply-adds are then formed per SOURCE expression
# (hipcc's default, fast, also fuses across statements, depending on what the surrounding code looks like after inlining).  The
# one-wave and the two-wave forward kernels (k_forward / k_forward2) share their source expressions but not their surroundings;
# the solve loop picks between them per batch step from the number of active trajectories, so with `fast` a trajectory's result
# depended — at the 1e-13 level, 2e-14 in J — on how far the REST of its batch had converged.  With `on` the two kernels are
# bit-identical (tests/test_gpu_parity.py::test_two_wave_forward_pass asserts equality; interleaved A/B in profiles/r04_ab:
# forward phase C3 600 -> 597 us, C2 81.7 -> 83.4 us per step).
for _f in ("ops_quad_forward_a", "ops_quad_forward_b", "ops_quad_forward_c", "ops_quad_forward2_a", "ops_quad_forward2_b", "ops_quad_forward2_c",
           "ops_quadmrp_forward", "ops_quadrp_forward", "ops_small_forward", "ops_small_forward2", "ops_hybrid", "ops_vector",
           "ops_infeasible_a", "ops_infeasible_b"):
    FILE_FLAGS.setdefault(_f + ".hip", []).append("-ffp-contract=on")
# The lane expansion kernels and the scan kernel differentiate the RK step with chunk-mode dual numbers whose seeds are unit vectors:
# a third of their FP64 instructions were products with a literal 0.0, which IEEE semantics forbid folding (0 * NaN).  These three
# flags allow exactly that folding (no reassociation, no reciprocal or contraction changes: finite results are bit-identical up to the
# sign of a zero); the NaN-sensitive tests in those kernels work on bit patterns (common.h not_positive).
for _f in ("ops_small_lane", "ops_small_scan"):
    FILE_FLAGS.setdefault(_f + ".hip", []).extend(["-fno-honor-nans", "-fno-honor-infinities", "-fno-signed-zeros"])
OBJDIR = CSRC / "build"


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found")


def source_files():
    return sorted(CSRC.glob("*.hip")) + sorted(CSRC.glob("*.h")) + [INCLUDE / "trajopt_hip.h"]


def flags_for(name, extra_flags=()):
    """hipcc flags of one translation unit (the tools that re-compile a file for analysis use the same ones)."""
    return [*FLAGS, *FILE_FLAGS.get(name, []), *extra_flags]


def source_id(extra_flags=()):
    """Hash of every source the library is compiled from, the flags included."""
    h = hashlib.sha256()
    for f in source_files():
        h.update(f.name.encode())
        h.update(f.read_bytes())
    h.update(" ".join(list(FLAGS) + list(extra_flags)).encode())
    h.update(repr(sorted(FILE_FLAGS.items())).encode())
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


def _compile(args):
    cmd, src = args
    res = subprocess.run(cmd, capture_output=True, text=True, cwd=str(CSRC))
    return src, res.returncode, res.stderr


def build_hip(force=False, verbose=False, extra_flags=(), target=TARGET, jobs=None):
    """Compile every csrc/*.hip (one translation unit per model / phase group, in parallel) for gfx950 and link
    csrc/libtrajopt_hip.so.  Returns the path.  Objects are cached per (source, stamp) under csrc/build/."""
    from concurrent.futures import ThreadPoolExecutor
    target = Path(target)
    sid = source_id(extra_flags)
    if not force and binary_id(target) == sid:
        return target
    hipcc = hipcc_path()
    OBJDIR.mkdir(exist_ok=True)
    sources = sorted(CSRC.glob("*.hip"))
    work, objs = [], []
    for src in sources:
        obj = OBJDIR / f"{src.stem}.{sid}.o"
        objs.append(obj)
        if force or not obj.exists():
            flags = flags_for(src.name, extra_flags) + ([f'-DTO_BUILD_ID="{sid}"'] if src.name == "trajopt_hip.hip" else [])
            work.append(([hipcc, *flags, "-c", "-o", str(obj), str(src)], src))
    with ThreadPoolExecutor(max_workers=jobs or os.cpu_count() or 4) as ex:
        for src, rc, err in ex.map(_compile, work):
            if rc != 0:
                raise RuntimeError(f"hipcc failed on {src.name}:\n" + err[-6000:])
            if verbose:
                print(src.name, err)
    for old in OBJDIR.glob("*.o"):  # objects of earlier stamps
        if old not in objs:
            old.unlink()
    res = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(target), *map(str, objs)],
                         capture_output=True, text=True, cwd=str(CSRC))
    if res.returncode != 0:
        raise RuntimeError("link failed:\n" + res.stderr[-4000:])
    return target

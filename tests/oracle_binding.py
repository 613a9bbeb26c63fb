"""Loads oracle/build/liboracle.so (building it with oracle/Makefile when missing) and binds it with
the product's ctypes ``Library`` class under the ``oracle_`` prefix.  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import subprocess
from pathlib import Path

import trajopt_amd as T

ROOT = Path(__file__).resolve().parent.parent
ORACLE_SO = ROOT / "oracle" / "build" / "liboracle.so"
NATIVE_SO = ROOT / "oracle" / "build" / "liboracle_native.so"
_lib = None
_native = None


def build_oracle(force=False):
    srcs = [ROOT / "oracle" / "trajopt_oracle.cpp", ROOT / "oracle" / "oracle_math.h", ROOT / "oracle" / "oracle_pn.h",
            ROOT / "include" / "trajopt_hip.h"]
    stale = (not ORACLE_SO.exists()) or any(s.stat().st_mtime > ORACLE_SO.stat().st_mtime for s in srcs)
    if force or stale:
        subprocess.run(["make", "-C", str(ROOT / "oracle")], check=True, capture_output=True)
    return ORACLE_SO


def _bind(path):
    lib = T.capi.Library(path, prefix="oracle_", hip=False)
    for name, argtypes in {
        "set_threads": [C.c_void_p, C.c_int],
        "dynamics": [C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)],
        "discrete_dynamics": [C.c_int32, C.POINTER(C.c_double), C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, C.POINTER(C.c_double)],
        "state_diff": [C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)],
        "state_add": [C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)],
    }.items():
        f = getattr(lib.dll, "oracle_" + name)
        f.argtypes, f.restype = argtypes, C.c_int
        lib._fn[name] = f
    mt = lib.dll.oracle_max_threads
    mt.argtypes, mt.restype = [], C.c_int
    lib.max_threads = mt
    return lib


def load_oracle_native():
    """The oracle compiled with -march=native ON THIS MACHINE (bench.py's cpu_baseline leg: SURVEY.md §8d asks for the tuned
    build; it is rebuilt on every machine that times it, the portable build stays the checker).  Returns (library, flags
    note); falls back to the portable build when the compiler is unavailable."""
    global _native
    if _native is None:
        try:
            NATIVE_SO.unlink(missing_ok=True)  # never trust a binary tuned for another host
            subprocess.run(["make", "-C", str(ROOT / "oracle"), "native"], check=True, capture_output=True)
            _native = (_bind(NATIVE_SO), "-O3 -march=native -ffp-contract=off, built on this host")
        except Exception as e:  # noqa: BLE001 - any failure means: time the portable build instead
            _native = (load_oracle(), f"-O3 -ffp-contract=off portable build (native build failed: {type(e).__name__})")
    return _native


def load_oracle():
    global _lib
    if _lib is None:
        build_oracle()
        _lib = _bind(ORACLE_SO)
    return _lib


def set_threads(prob, threads):
    prob._lib.call("set_threads", prob._h, int(threads))

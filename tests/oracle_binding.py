"""Loads oracle/build/liboracle.so (building it with oracle/Makefile when missing) and binds it with
the product's ctypes ``Library`` class under the ``oracle_`` prefix.  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import subprocess
from pathlib import Path

import trajopt_amd as T

ROOT = Path(__file__).resolve().parent.parent
ORACLE_SO = ROOT / "oracle" / "build" / "liboracle.so"
_lib = None


def build_oracle(force=False):
    srcs = [ROOT / "oracle" / "trajopt_oracle.cpp", ROOT / "oracle" / "oracle_math.h", ROOT / "include" / "trajopt_hip.h"]
    stale = (not ORACLE_SO.exists()) or any(s.stat().st_mtime > ORACLE_SO.stat().st_mtime for s in srcs)
    if force or stale:
        subprocess.run(["make", "-C", str(ROOT / "oracle")], check=True, capture_output=True)
    return ORACLE_SO


def load_oracle():
    global _lib
    if _lib is None:
        build_oracle()
        _lib = T.capi.Library(ORACLE_SO, prefix="oracle_", hip=False)
        for name, argtypes in {
            "set_threads": [C.c_void_p, C.c_int],
            "dynamics": [C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)],
            "discrete_dynamics": [C.c_int32, C.POINTER(C.c_double), C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, C.POINTER(C.c_double)],
            "state_diff": [C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)],
        }.items():
            f = getattr(_lib.dll, "oracle_" + name)
            f.argtypes, f.restype = argtypes, C.c_int
            _lib._fn[name] = f
        mt = _lib.dll.oracle_max_threads
        mt.argtypes, mt.restype = [], C.c_int
        _lib.max_threads = mt
    return _lib


def set_threads(prob, threads):
    prob._lib.call("set_threads", prob._h, int(threads))

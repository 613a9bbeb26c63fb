#!/usr/bin/env python3
"""Bit-identity of the projected-Newton kernel source across edits, on the host: compiles csrc/k_pn.h with g++ (TO_PN_HOST, the
harness of tests/test_pn_host.py), polishes that test's cases and saves every output; run it on two revisions of the tree and compare.
  python tools/pn_host_bits.py save /tmp/a.npz      (on revision A)
  python tools/pn_host_bits.py save /tmp/b.npz      (on revision B)
  python tools/pn_host_bits.py cmp /tmp/a.npz /tmp/b.npz
Round 5's restructurings of the factorisation and the sweeps were each checked this way: identical trajectories,
linearisation counts and violations to the last bit.  Uses oracle/ only as the harness's problem source (test infrastructure)."""
import ctypes as C
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def save(path):
    import test_pn_host as H
    from oracle_binding import load_oracle
    so = Path(tempfile.mkdtemp()) / "libpn_host.so"
    subprocess.run(["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-fPIC", "-shared", "-I", str(ROOT / "tests" / "host_shim"),
                    "-I", str(ROOT / "trajectoryoptimization.jl_amd" / "csrc"), "-o", str(so),
                    str(ROOT / "tests" / "host_shim" / "pn_harness.cpp")], check=True)
    lib = C.CDLL(str(so))
    lib.pn_host_solve.restype = C.c_int
    lib.pn_host_last_error.restype = C.c_char_p
    oracle = load_oracle()
    out = {}
    for name, (build, scale) in H.CASES.items():
        prob = H.al_then_perturb(lambda: build(oracle), scale)
        X, U, st, ip, cm = H.host_polish(lib, prob)
        out.update({name + "_X": X, name + "_U": U, name + "_status": st, name + "_it_pn": ip, name + "_cmax": cm})
    np.savez(path, **out)
    print("saved", path, "cases:", list(H.CASES))


def cmp(a, b):
    A, B = np.load(a), np.load(b)
    ok = True
    for k in A.files:
        same = np.array_equal(A[k], B[k])
        ok = ok and same
        print(("identical " if same else "DIFFERENT ") + k + ("" if same else f"  max |diff| = {np.abs(A[k] - B[k]).max():.3e}"))
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "save":
        save(sys.argv[2])
    elif len(sys.argv) == 4 and sys.argv[1] == "cmp":
        cmp(sys.argv[2], sys.argv[3])
    else:
        sys.exit(__doc__)

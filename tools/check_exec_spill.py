#!/usr/bin/env python3
"""Scans the gfx950 assembly of csrc/*.hip for the hipcc (ROCm 7.2) spill-placement hazard described in DESIGN.md §6:
a VGPR spill (v_accvgpr_write / scratch_store) emitted at the top of a join block BEFORE the `s_or_b64 exec, exec, …`
that re-enables the lanes which skipped the divergent region — the spilled value is then lost in those lanes.
Usage: tools/check_exec_spill.py [file.hip ...]   (exit status 1 when a kernel has such a site)"""
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

CSRC = Path(__file__).resolve().parent.parent / "trajectoryoptimization.jl_amd" / "csrc"
# VGPR->AGPR copies, and scratch stores the compiler marks as spills (a plain scratch store is the program's own store to a
# stack array: the tail of a divergent branch legitimately ends with those)
SPILL = re.compile(r"^\s*(v_accvgpr_write_b32\s+a\d+,\s*v\d+|scratch_store_\w+.*Folded Spill)")
RESTORE = re.compile(r"^\s*s_or_b64\s+exec,\s*exec,")
LABEL = re.compile(r"^(\.LBB\S+|_Z\S+):")
BRANCH = re.compile(r"^\s*(s_cbranch|s_branch|s_endpgm|s_setpc)")


def _build_flags(name):
    """The flags csrc/ is built with (trajectoryoptimization.jl_amd/build.py), loaded without importing the package."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_to_build", CSRC.parent / "build.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.flags_for(name)


def scan_text(asm):
    """[(kernel, block label, line number, instruction)]: spills found between a block label and the first `s_or_b64 exec,
    exec, ...` of that block (the run ends at any branch or other write to EXEC)."""
    hits, kernel, pending, label = [], None, [], None
    for ln, line in enumerate(asm.splitlines(), 1):
        m = LABEL.match(line)
        if m:
            if m.group(1).startswith("_Z"):
                kernel = m.group(1)
            label, pending = m.group(1), []
            continue
        if label is None or line.lstrip().startswith(";") or not line.strip():
            continue
        if RESTORE.match(line):
            hits += [(kernel, label, ln, s) for s in pending]
            label = None  # only the run of instructions before the first exec restore of a block is of interest
        elif SPILL.match(line):
            pending.append(line.strip())
        else:
            body = line.split(";")[0].split(None, 1)
            dest = body[1].split(",")[0] if len(body) > 1 else ""
            if BRANCH.match(line) or "saveexec" in line or re.search(r"\bexec\b", dest):
                label = None
    return hits


def scan(src):
    cmd = ["/opt/rocm/bin/hipcc", *_build_flags(Path(src).name), "--cuda-device-only", "-S", "-o", "-", str(src)]
    asm = subprocess.run(cmd, capture_output=True, text=True, cwd=str(CSRC)).stdout
    return src.name, scan_text(asm)


def main():
    files = [Path(a).resolve() for a in sys.argv[1:]] or sorted(CSRC.glob("*.hip"))
    bad = 0
    with ThreadPoolExecutor(8) as ex:
        for name, hits in ex.map(scan, files):
            print(f"{name}: {len(hits)} spill(s) ahead of an exec restore")
            for kernel, label, ln, ins in hits:
                dem = subprocess.run(["c++filt", kernel or "?"], capture_output=True, text=True).stdout.strip()
                print(f"   {dem.replace('void to::', '').replace('(to::KArgs)', '')}  {label} line {ln}: {ins}")
            bad += len(hits)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

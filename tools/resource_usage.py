#!/usr/bin/env python3
"""Register / LDS / occupancy table of every kernel in csrc/ops_*.hip and trajopt_hip.hip from
`hipcc -Rpass-analysis=kernel-resource-usage` (cross-compiles without a GPU).  Usage: tools/resource_usage.py [file.hip ...] > table.md"""
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

CSRC = Path(__file__).resolve().parent.parent / "trajectoryoptimization.jl_amd" / "csrc"


def _build_flags(name):
    """The flags csrc/ is built with (trajectoryoptimization.jl_amd/build.py), loaded without importing the package."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_to_build", CSRC.parent / "build.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.flags_for(name)


def analyse(src):
    cmd = ["/opt/rocm/bin/hipcc", *_build_flags(Path(src).name), "-c", "-o", "/dev/null",
           str(src), "-Rpass-analysis=kernel-resource-usage"]
    err = subprocess.run(cmd, capture_output=True, text=True, cwd=str(CSRC)).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark: .*?(Function Name|Name): (\S+)", line)
        if m:
            name = subprocess.run(["c++filt", m.group(2)], capture_output=True, text=True).stdout.strip()
            cur = {"kernel": re.sub(r"\(to::KArgs.*", "", name).replace("void to::", "").replace("to::", "")}
            rows.append(cur)
            continue
        m = re.search(r"remark: .*?\s(TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).split()[0]] = int(m.group(2))
    return src.name, rows


def main():
    files = [Path(a).resolve() for a in sys.argv[1:]] or sorted(CSRC.glob("*.hip"))
    with ThreadPoolExecutor(8) as ex:
        for name, rows in ex.map(analyse, files):
            print(f"\n### {name}\n\n| kernel | VGPR | AGPR | SGPR | scratch B/lane | LDS B/block | waves/SIMD |\n|---|---|---|---|---|---|---|")
            for r in rows:
                print(f"| `{r['kernel']}` | {r.get('VGPRs')} | {r.get('AGPRs')} | {r.get('TotalSGPRs')} | {r.get('ScratchSize')} | {r.get('LDS')} | {r.get('Occupancy')} |")


if __name__ == "__main__":
    main()

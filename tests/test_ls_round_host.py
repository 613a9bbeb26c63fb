"""The lane map of the forward pass's line-search rounds (csrc/ls_round.h: static rounds, repacked last round, step sizes) compiled
for the HOST and checked over every wave shape and thousands of need masks: wave-uniform shape, every (searching trajectory, step
size) evaluated by exactly one lane, the owner lanes finding their candidates, rows' trajectories in lanes 0..tw-1."""
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def test_ls_round_lane_map(tmp_path):
    exe = tmp_path / "ls_round_harness"
    subprocess.run(["g++", "-std=c++17", "-O1", "-I", str(ROOT / "tests" / "host_shim"), "-I", str(ROOT / "trajectoryoptimization.jl_amd" / "csrc"),
                    str(ROOT / "tests" / "host_shim" / "ls_round_harness.cpp"), "-o", str(exe)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    words = r.stdout.split()
    assert int(words[words.index("repacked") + 1]) > 1000, r.stdout   # the repacked branch was exercised

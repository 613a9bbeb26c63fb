#!/usr/bin/env python3
"""Where do a kernel's SGPR spill lanes (v_writelane / v_readlane) sit?  VERDICT r05: k_forward2<CartpoleModel, 5> carries 94 / 190 of them —
inside the knot loop they would be on the rollout's critical path.  Compiles one translation unit to gfx950 assembly with the build's flags,
finds every loop of the named kernel (a label with a later backward branch to it) and counts the lane moves inside each.
  python tools/spill_lanes_in_loops.py ops_small_forward2.hip _ZN2to10k_forward2INS_13CartpoleModelELi5E"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import trajopt_amd  # noqa: E402,F401
from trajectoryoptimization_jl_amd import build as B  # noqa: E402

tu, prefix = sys.argv[1], sys.argv[2]
with tempfile.TemporaryDirectory() as d:
    out = Path(d) / "k.s"
    subprocess.run([B.hipcc_path(), *B.flags_for(tu), "-S", "--cuda-device-only", "-o", str(out), str(B.CSRC / tu)], check=True, capture_output=True)
    txt = out.read_text().split("\n")
start = next(i for i, l in enumerate(txt) if l.startswith(prefix))
end = next(i for i in range(start + 1, len(txt)) if txt[i].startswith(".Lfunc_end"))
body = txt[start:end]
labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
loops = sorted({(labels[m.group(1)], i) for i, l in enumerate(body)
                if (m := re.match(r"\s+s_c?branch\w*\s+(\.LBB\d+_\d+)", l)) and m.group(1) in labels and labels[m.group(1)] < i})
isn = lambda x: bool(re.match(r"\s+(v_|s_|ds_|global_|buffer_|scratch_)", x))
print(f"{prefix}: {sum(map(isn, body))} instructions, {sum('v_readlane' in x for x in body)} v_readlane, {sum('v_writelane' in x for x in body)} v_writelane, {len(loops)} loops")
for a, b in loops:
    seg = body[a:b]
    rl, wl = sum("v_readlane" in x for x in seg), sum("v_writelane" in x for x in seg)
    fp = sum(bool(re.match(r"\s+v_\w+_f64", x)) for x in seg)
    print(f"  loop lines {a:5d}..{b:5d}: {sum(map(isn, seg)):5d} instructions ({fp} FP64), {rl:3d} v_readlane, {wl:3d} v_writelane" + ("   <-- lane moves inside" if rl + wl else ""))

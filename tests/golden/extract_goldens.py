#!/usr/bin/env python3
"""Extract the reference's own golden vectors for the hot path from its example notebooks.

Run in the build container (needs /root/reference); writes tests/golden/reference_goldens.json, which is
committed so the tests run where the reference does not exist (the GPU box).  Nothing here is hand-typed:
every number is parsed out of a saved notebook output cell.  Cell indices and the SURVEY.md §8c labels
(G1..G6) are recorded next to each value.
"""
import json
import re
import sys
from pathlib import Path

REF = Path(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
OUT = Path(__file__).resolve().parent / "reference_goldens.json"
NUM = r"[-+]?(?:\d+\.\d*|\.\d+|\d+)(?:[eE][-+]?\d+)?"


def cells(nb_path):
    nb = json.loads(Path(nb_path).read_text())
    out = []
    for c in nb["cells"]:
        texts = []
        for o in c.get("outputs", []):
            if "text" in o:
                texts.append("".join(o["text"]))
            elif "data" in o and "text/plain" in o["data"]:
                texts.append("".join(o["data"]["text/plain"]))
        out.append(("".join(c["source"]), texts))
    return out


def matrix_rows(text):
    rows = []
    for line in text.splitlines()[1:]:
        line = line.replace("…", " ELL ").replace("⋮", "")
        toks = line.split()
        if toks:
            rows.append([None if t == "ELL" else float(t) for t in toks])
    return rows


g = {}
api = cells(REF / "examples" / "Internal API.ipynb")
# G1: Quadrotor rollout final state (cell 6)
src, outs = api[6]
assert "rollout!" in src and "u0 += [1,0,1,0]*1e-2" in src
g["G1_quadrotor_rollout"] = {
    "source": "examples/Internal API.ipynb cell 6 (N=51, tf=5, x0 r=[1,2,1], u=hover+[1,0,1,0]*1e-2, RK4)",
    "x_final": [float(v) for v in outs[0].splitlines()[1:]],
}
assert len(g["G1_quadrotor_rollout"]["x_final"]) == 13
# G2: error-state A, B at k=1 (cell 12); A is printed with elided columns
src, outs = api[12]
assert "error_expansion" in src
A_rows, B_rows = matrix_rows(outs[0]), matrix_rows(outs[1])
assert len(A_rows) == 12 and len(B_rows) == 12
A_entries = []
for i, row in enumerate(A_rows):
    if None in row:
        ell = row.index(None)
        left, right = row[:ell], row[ell + 1:]
    else:  # only every 5th printed row carries the ellipsis marker; the visible columns are the same
        left, right = row[:7], row[7:]
    assert len(left) == 7 and len(right) == 2
    for j, v in enumerate(left):
        A_entries.append([i, j, v])
    for j, v in enumerate(right):
        A_entries.append([i, 12 - len(right) + j, v])
g["G2_error_state_jacobians"] = {
    "source": "examples/Internal API.ipynb cell 12 (display precision ~6 significant digits; stack may predate RK4)",
    "A_entries_0based": A_entries,
    "B": B_rows,
}
# G5: stage cost at k=1 (cell 20), legacy dt-scaled
g["G5_stage_cost_k1"] = {"source": "examples/Internal API.ipynb cell 20 (stage cost multiplied by dt=0.1 on the saved stack)",
                         "J1": float(api[20][1][0])}
# G6: error-state cost Hessian attitude block (cell 35)
rows = matrix_rows(api[35][1][0])
g["G6_error_state_cost_hessian"] = {"source": "examples/Internal API.ipynb cell 35: E[N-1].Q[4:6,4:6] (dt-scaled stack)", "Q_att": rows}

cp = cells(REF / "examples" / "Cartpole.ipynb")
txt = cp[25][1][0]
assert "iLQR" in txt
g["G3_cartpole_ilqr"] = {
    "source": "examples/Cartpole.ipynb cell 25 (legacy stack: RK3 + dt-scaled stage costs)",
    "iterations": int(re.search(r"Total Iterations: (\d+)", txt).group(1)),
    "cost": float(re.search(r"Terminal Cost: (%s)" % NUM, txt).group(1)),
    "dJ": float(re.search(r"Terminal dJ: \S*?(%s)\n" % NUM, txt).group(1)),
}
txt = cp[17][1][0]
U = [float(m.group(1)) for m in re.finditer(r"\[(%s)\]" % NUM, cp[21][1][0])]
g["G4_cartpole_altro"] = {
    "source": "examples/Cartpole.ipynb cells 17,19,21 (ALTRO = AL-iLQR + projected Newton; sanity only)",
    "iterations": int(re.search(r"Total Iterations: (\d+)", txt).group(1)),
    "cost": float(re.search(r"Terminal Cost: (%s)" % NUM, txt).group(1)),
    "c_max": float(re.search(r"max_violation: (%s)" % NUM, cp[19][1][0]).group(1)),
    "U_head": U[:13], "U_tail": U[13:],
}
g["G4_cartpole_ipopt"] = {
    "source": "examples/Cartpole.ipynb cell 31 (Ipopt on the same constrained problem: the converged local optimum)",
    "cost": float(re.search(r"cost:\s+(%s)" % NUM, cp[31][1][0]).group(1)),
    "c_max": float(re.search(r"max_violation: (%s)" % NUM, cp[31][1][0]).group(1)),
}
qd = cells(REF / "examples" / "Quadrotor.ipynb")
txt = qd[22][1][-1]
g["G4_quadrotor_altro"] = {
    "source": "examples/Quadrotor.ipynb cell 22 (sanity only)",
    "iterations": int(re.search(r"Iterations: (\d+)", txt).group(1)),
    "cost": float(re.search(r"Cost: (%s)" % NUM, txt).group(1)),
    "c_max": float(re.search(r"Constraint violation: (%s)" % NUM, txt).group(1)),
}
OUT.write_text(json.dumps(g, indent=1))
print("wrote", OUT, "keys:", list(g))

"""julia/TrajOptHIP.jl is the binding a TrajectoryOptimization.jl host loads.  Julia is not installed in
the build image, so the file is checked structurally against include/trajopt_hip.h: every `to_*` entry point must be
bound by a `ccall` with the header's arity and return type class, and every struct must mirror the header field by field."""
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
HEADER = (ROOT / "include" / "trajopt_hip.h").read_text()
SHIM = (ROOT / "julia" / "TrajOptHIP.jl").read_text()


def header_functions():
    h = re.sub(r"/\*.*?\*/", "", HEADER, flags=re.S)
    out = {}
    for m in re.finditer(r"^(int|const char\*|void\*)\s+(to_[a-z_0-9]+)\s*\(([^;]*?)\)\s*;", h, flags=re.M | re.S):
        args = " ".join(m.group(3).split())
        out[m.group(2)] = (m.group(1), 0 if args in ("void", "") else args.count(",") + 1)
    return out


def split_top(s):
    """Split at top-level commas (outside braces / parentheses); empty trailing items dropped."""
    items, depth, cur = [], 0, ""
    for ch in s:
        if ch in "({[":
            depth += 1
        elif ch in ")}]":
            depth -= 1
        if ch == "," and depth == 0:
            items.append(cur.strip()); cur = ""
        else:
            cur += ch
    items.append(cur.strip())
    return [i for i in items if i]


def shim_ccalls():
    """name -> list of (return type, [argument types]) for every ccall((:name, lib), ...) in the shim."""
    calls = {}
    for m in re.finditer(r"ccall\(\(:(to_[a-z_0-9]+), lib\),", SHIM):
        i, depth, j = m.end(), 1, m.end()
        while depth:                       # closing parenthesis of this ccall
            depth += {"(": 1, ")": -1}.get(SHIM[j], 0)
            j += 1
        parts = split_top(SHIM[i:j - 1])
        assert parts[1].startswith("(") and parts[1].endswith(")"), (m.group(1), parts[:2])
        types = split_top(parts[1][1:-1])
        assert len(parts) - 2 == len(types), f"{m.group(1)}: {len(types)} argument types but {len(parts) - 2} arguments"
        calls.setdefault(m.group(1), []).append((parts[0], types))
    return calls


def test_every_entry_point_is_bound_with_the_right_arity():
    funcs, calls = header_functions(), shim_ccalls()
    assert len(funcs) >= 55
    missing = sorted(set(funcs) - set(calls))
    assert not missing, f"not bound in julia/TrajOptHIP.jl: {missing}"
    unknown = sorted(set(calls) - set(funcs))
    assert not unknown, f"ccall to symbols the header does not declare: {unknown}"
    ret = {"int": "Cint", "const char*": "Cstring", "void*": "Ptr{Cvoid}"}
    for name, (rtype, nargs) in funcs.items():
        for jret, jtypes in calls[name]:
            assert jret == ret[rtype], (name, jret)
            assert len(jtypes) == nargs, f"{name}: header has {nargs} parameters, ccall passes {len(jtypes)}"


def c_struct_fields(name):
    body = re.search(r"typedef struct \{((?:(?!typedef struct).)*?)\}\s*" + name + r"\s*;", re.sub(r"/\*.*?\*/", "", HEADER, flags=re.S), flags=re.S).group(1)
    fields = []
    for decl in body.split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        _, rest = decl.replace("const ", "").split(" ", 1)
        for f in rest.split(","):
            fields.append(re.sub(r"\[.*\]", "", f).replace("*", "").strip())
    return fields


def julia_struct_fields(name):
    body = re.search(r"struct " + name + r"\b.*?\n(.*?)\nend", SHIM, flags=re.S).group(1)
    return [m.group(1) for m in re.finditer(r"^\s*(\w+)::", body, flags=re.M)]


def test_structs_mirror_the_header():
    for cname, jname in [("to_cost_desc", "CostDesc"), ("to_constraint_desc", "ConstraintDesc"), ("to_problem_desc", "ProblemDesc"),
                         ("to_solver_opts", "SolverOpts"), ("to_solve_stats", "SolveStats")]:
        assert julia_struct_fields(jname) == c_struct_fields(cname), jname


def test_reference_verbs_are_extended():
    """The shim hangs the entry points on the reference's own function names (src/TrajectoryOptimization.jl:29-71 and the
    operator API of src/abstract_constraint.jl:200-280, src/cones.jl:96-276, src/problem.jl:242-340, src/objective.jl:198-212)."""
    for verb in ["rollout!", "cost", "states", "controls", "initial_controls!", "initial_states!", "set_initial_state!",
                 "set_goal_state!", "update_trajectory!", "evaluate_constraints!", "constraint_jacobians!", "∇constraint_jacobians!",
                 "max_violation", "num_constraints", "projection!", "∇projection!", "∇²projection!", "get_model", "get_objective",
                 "get_constraints", "get_trajectory", "gettimes"]:
        assert re.search(r"TO\.(\$f|" + re.escape(verb) + r")\(", SHIM) or f":{verb}" in SHIM, verb
    for t in ["GoalConstraint", "BoundConstraint", "StateBound", "ControlBound", "NormConstraint", "CircleConstraint", "SphereConstraint",
              "LinearConstraint", "CollisionConstraint", "QuatVecEq", "IndexedConstraint", "DiagonalCost", "QuadraticCost",
              "DiagonalQuatCost", "ErrorQuadratic"]:
        assert f"::TO.{t}" in SHIM, t

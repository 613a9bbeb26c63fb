"""Host-side mirror of the reference interface: bookkeeping and error behaviour, CPU only.  The product library
validates descriptors before it touches the device, so the reference's exception classes can be checked here
without a GPU; with valid input and no GPU it must fail loudly (no CPU fallback)."""
import ctypes as C
import re
from pathlib import Path

import numpy as np
import pytest

import trajopt_amd as T

ROOT = Path(__file__).resolve().parent.parent


def test_library_exports_every_declared_symbol():
    lib = T.load_hip_library()
    header = (ROOT / "include" / "trajopt_hip.h").read_text()
    declared = set(re.findall(r"^(?:int|const char\*|void\*)\s+(to_[a-z_0-9]+)\s*\(", header, flags=re.M))
    assert len(declared) >= 45
    dll = C.CDLL(lib.path)
    missing = [s for s in sorted(declared) if not hasattr(dll, s)]
    assert not missing, missing
    bound = {"to_" + k for k in list(T.capi.SIGNATURES) + list(T.capi.HIP_ONLY)} | {"to_last_error", "to_stream", "to_build_id"}
    assert declared <= bound, sorted(declared - bound)
    assert lib.abi_version() == T.capi.TO_ABI_VERSION


def test_binary_matches_the_sources_in_the_tree():
    """The shipped libtrajopt_hip.so carries a hash of the sources it was compiled from (to_build_id); a stale binary —
    it is git-ignored and travels with snapshots — fails here instead of silently testing old kernels."""
    from trajectoryoptimization_jl_amd import build as B
    lib = T.load_hip_library()
    assert re.fullmatch(r"[0-9a-f]{16}", lib.build_id())
    assert lib.build_id() == B.binary_id(lib.path)
    assert lib.build_id() == B.source_id(), "rebuild: python -c 'import __graft_entry__ as g; g.build()'"


def test_struct_layouts_match_header():
    """ctypes mirrors vs the sizes the C compiler computes (guards against silent ABI drift)."""
    import subprocess, tempfile
    src = '#include <stdio.h>\n#include "trajopt_hip.h"\nint main(){printf("%zu %zu %zu %zu %zu\\n", sizeof(to_cost_desc), sizeof(to_constraint_desc), sizeof(to_problem_desc), sizeof(to_solver_opts), sizeof(to_solve_stats));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        (Path(d) / "s.c").write_text(src)
        subprocess.run(["gcc", "-I", str(ROOT / "include"), "-o", f"{d}/s", f"{d}/s.c"], check=True)
        sizes = list(map(int, subprocess.run([f"{d}/s"], capture_output=True, text=True, check=True).stdout.split()))
    assert sizes == [C.sizeof(T.capi.CostDesc), C.sizeof(T.capi.ConstraintDesc), C.sizeof(T.capi.ProblemDesc),
                     C.sizeof(T.capi.SolverOpts), C.sizeof(T.capi.SolveStats)]


def test_no_cpu_fallback_without_gpu():
    lib = T.load_hip_library()
    try:
        ndev = lib.device_count()
    except T.HipError:
        ndev = 0
    if ndev > 0:
        pytest.skip("a GPU is visible")
    model = T.Cartpole()
    obj = T.LQRObjective(np.ones(4), np.ones(1), np.ones(4), np.zeros(4), 11)
    with pytest.raises(T.HipError):
        T.Problem(model, obj, np.zeros(4), 1.0)
    with pytest.raises(T.HipError):
        T.projection(T.SecondOrderCone(), np.array([1.0, 2, 3]))


def test_constraint_list_bookkeeping():
    """test/constraint_list.jl:35-76."""
    n, m, N = 4, 1, 11
    xf = np.array([0, np.pi, 0, 0.0])
    cir = T.CircleConstraint(n, [1.0, 2], [1.0, 2], [0.1, 0.2])
    goal = T.GoalConstraint(xf)
    lin = T.LinearConstraint(n, m, np.ones((3, 5)), np.ones(3), T.Inequality())
    bnd = T.BoundConstraint(n, m, x_min=-np.ones(n), x_max=np.ones(n), u_min=-1, u_max=1)
    cons = T.ConstraintList(n, m, N)
    T.add_constraint(cons, cir, range(1, N + 1))
    assert cons[0] is cir and cons.inds[0] == (1, N) and cons.p == [2] * N
    T.add_constraint(cons, goal, N)
    assert cons[1] is goal and cons.inds[1] == (N, N) and cons.p[:-1] == [2] * (N - 1) and cons.p[-1] == 2 + 4
    T.add_constraint(cons, lin, range(1, 5), 1)          # insert at the front (idx is 1-based like the reference)
    assert cons[0] is lin and cons[1] is cir and cons[-1] is goal and cons.inds[0] == (1, 4)
    assert cons.p[:4] == [5] * 4 and cons.p[4:N - 1] == [2] * (N - 5) and len(cons) == 3
    cons2 = cons.copy()
    T.add_constraint(cons, bnd, range(1, N))
    assert len(cons) == 4 and len(cons2) == 3 and cons[-1] is bnd and bnd.p == 2 * (n + m)
    lin2 = T.LinearConstraint(2, 1, np.ones((3, 2)), np.ones(3), T.Inequality(), [1, 2])
    with pytest.raises(T.DimensionMismatch):
        T.add_constraint(cons, lin2, range(1, 5))                                                     # :69-70
    with pytest.raises(T.ArgumentError):
        T.add_constraint(cons, goal, N, idx=9)
    assert [c for c in cons] == [lin, cir, goal, bnd]
    assert [T.sense(c) for c in cons] == [T.Inequality(), T.Inequality(), T.Equality(), T.Inequality()]


def test_bound_constraint_constructor():
    """test/constraint_tests.jl:209-266 and src/constraints.jl:660-719."""
    n, m = 3, 2
    bnd = T.BoundConstraint(n, m, x_max=[1, np.inf, 3], x_min=[-1, -2, -np.inf], u_max=4, u_min=[-np.inf, -5])
    assert bnd.p == 2 + 2 + 2 + 1
    assert bnd.inds == [1, 3, 4, 5, 6, 7, 10]   # finite entries of [-z_max; z_min], 1-based
    with pytest.raises(T.ArgumentError):
        T.BoundConstraint(n, m, x_max=0.0, x_min=1.0)
    with pytest.raises(AssertionError):
        T.NormConstraint(n, m, -1.0, T.Inequality())
    assert T.NormConstraint(n, m, 5.0, T.SecondOrderCone(), "control").p == m + 1
    assert T.NormConstraint(n, m, 5.0, T.Equality(), "state").inds == [1, 2, 3]


def _create_rc(desc_mut):
    """Call to_create on a mutated, otherwise valid, Cartpole descriptor; return (rc, message)."""
    lib = T.load_hip_library()
    model = T.Cartpole()
    N = 11
    obj = T.LQRObjective(np.ones(4), np.ones(1), np.ones(4), np.zeros(4), N)
    uniq, index = obj._descs()
    costs = (T.capi.CostDesc * len(uniq))(*[c._desc() for c in uniq])
    cons = T.ConstraintList(4, 1, N)
    T.add_constraint(cons, T.GoalConstraint(np.zeros(4)), N)
    cdesc = cons._descs()
    d = T.capi.ProblemDesc()
    d.abi_version, d.model, d.integrator, d.n, d.m, d.N, d.B = T.capi.TO_ABI_VERSION, model.model_id, T.RK4, 4, 1, N, 2
    d.model_params[:4] = model.params()
    d.t0, d.tf = 0.0, 1.0
    d.n_costs, d.costs, d.cost_index = len(uniq), costs, None
    d.n_constraints, d.constraints = 1, cdesc
    desc_mut(d, cdesc)
    h = C.c_void_p()
    rc = lib.raw("create")(C.byref(d), None, 0, C.byref(h))
    if rc == 0:
        lib.raw("destroy")(h)
    return rc, lib.last_error()


@pytest.mark.parametrize("mut,code", [
    (lambda d, c: setattr(d, "n", 5), T.capi.TO_ERR_DIMENSION_MISMATCH),          # src/problem.jl:65-68
    (lambda d, c: setattr(d, "tf", -1.0), T.capi.TO_ERR_ASSERTION),               # src/problem.jl:50  tf > t0
    (lambda d, c: setattr(d, "N", 1), T.capi.TO_ERR_ASSERTION),
    (lambda d, c: setattr(d, "B", 0), T.capi.TO_ERR_ARGUMENT),
    (lambda d, c: setattr(d, "abi_version", 99), T.capi.TO_ERR_ARGUMENT),
    (lambda d, c: setattr(d, "model", 17), T.capi.TO_ERR_UNSUPPORTED),
    (lambda d, c: setattr(c[0], "k_last", 12), T.capi.TO_ERR_ASSERTION),           # src/constraint_list.jl:112
    (lambda d, c: c[0].inds.__setitem__(0, 9), T.capi.TO_ERR_DIMENSION_MISMATCH),  # src/constraint_list.jl:109
    (lambda d, c: setattr(c[0], "p", 3), T.capi.TO_ERR_DIMENSION_MISMATCH),
    (lambda d, c: setattr(c[0], "kind", 77), T.capi.TO_ERR_UNSUPPORTED),
    (lambda d, c: setattr(d, "n_costs", 1), T.capi.TO_ERR_ARGUMENT),
])
def test_descriptor_validation_error_classes(mut, code):
    rc, msg = _create_rc(mut)
    assert rc == code, (rc, msg)
    assert msg


def test_dt_vector_validation():
    """test/problems_tests.jl:78-85,124-132: a dt vector must be positive and sum to tf."""
    dt = np.full(10, 0.1)
    keep = []

    def good(d, c):
        keep.append(dt.copy()); d.dt = keep[-1].ctypes.data_as(C.POINTER(C.c_double))

    def bad(d, c):
        keep.append(dt * 1.5); d.dt = keep[-1].ctypes.data_as(C.POINTER(C.c_double))

    rc, _ = _create_rc(bad)
    assert rc == T.capi.TO_ERR_ASSERTION
    rc, _ = _create_rc(good)
    assert rc in (0, T.capi.TO_ERR_HIP)  # valid descriptor: succeeds on a GPU box, HipError (no fallback) here


def test_problem_constructor_errors():
    model = T.Cartpole()
    obj = T.LQRObjective(np.ones(4), np.ones(1), np.ones(4), np.zeros(4), 11)
    with pytest.raises(T.ArgumentError):
        T.Problem(model, obj, np.zeros(4), 1.0, x0=np.zeros(4))                               # src/problem.jl:87-91
    with pytest.raises(T.DimensionMismatch):
        T.Problem(model, obj, np.zeros(4), 1.0, constraints=T.ConstraintList(5, 1, 11))      # src/problem.jl:64
    with pytest.raises(T.DimensionMismatch):
        T.Problem(T.Quadrotor(), obj, np.zeros(13), 1.0)                                     # src/problem.jl:66-68
    with pytest.raises(T.DimensionMismatch):
        T.LQRCost(np.ones(4), np.ones(1), np.zeros(3))


def test_lqr_objective_structure():
    """src/objective.jl:159-183 / test/objective_tests.jl:100-120."""
    n, m, N = 4, 2, 7
    Q, R, Qf = np.arange(1, 5.0), np.array([0.1, 0.2]), np.arange(10, 14.0)
    xf, uf = np.array([1.0, 2, 3, 4]), np.array([0.5, -0.5])
    obj = T.LQRObjective(Q, R, Qf, xf, N, uf=uf)
    assert len(obj) == N and obj[0] is obj[N - 2] and obj[-1] is not obj[0]
    assert isinstance(obj[0], T.DiagonalCost) and obj[-1].terminal and not obj[0].terminal
    np.testing.assert_allclose(obj[0].q, -Q * xf); np.testing.assert_allclose(obj[0].r, -R * uf)
    assert obj[0].c == pytest.approx(0.5 * xf @ (Q * xf) + 0.5 * uf @ (R * uf))
    np.testing.assert_allclose(obj[-1].Q, Qf); np.testing.assert_allclose(obj[-1].R, R)
    assert obj[-1].c == pytest.approx(0.5 * xf @ (Qf * xf))
    dense = T.LQRObjective(np.diag(Q) + 0.1, np.diag(R), np.diag(Qf), xf, N)
    assert isinstance(dense[0], T.QuadraticCost) and dense[0].kind == T.capi.COST_QUADRATIC
    uniq, index = obj._descs()
    assert len(uniq) == 2 and index == [0] * (N - 1) + [1]
    q = T.QuatLQRCost(np.ones(13), np.ones(4), np.arange(13.0), w=2.0)
    assert q.kind == T.capi.COST_DIAGONAL_QUAT and q.w == 2.0 and list(q.q_ref) == [3, 4, 5, 6] and q.q_ind == (4, 5, 6, 7)
    with pytest.raises(AssertionError):
        T.QuatLQRCost(np.ones(13), np.ones(4), np.arange(13.0), quat_ind=(4, 5, 6))


def test_host_layout_helpers(oracle):
    """[B, knot, dim] numpy arrays are the C-ABI's column-major (dim, knot, B) memory."""
    from trajectoryoptimization_jl_amd import configs
    p = configs.cartpole_problem(batch=3, N=6, tf=0.5, lib=oracle)
    U = np.arange(3 * 5 * 1, dtype=float).reshape(3, 5, 1)
    T.initial_controls(p, U)
    np.testing.assert_array_equal(T.controls(p), U)
    T.initial_controls(p, np.array([0.25]))
    assert np.all(T.controls(p) == 0.25)
    X = np.random.default_rng(0).standard_normal((3, 6, 4))
    T.initial_states(p, X)
    np.testing.assert_array_equal(T.states(p), X)
    T.initial_states(p, X[1].T)            # a single (n, N) Julia-layout trajectory is replicated
    np.testing.assert_array_equal(T.states(p)[2], X[1])
    with pytest.raises(T.DimensionMismatch):
        T.initial_states(p, np.zeros((2, 6, 4)))
    np.testing.assert_array_equal(T.get_initial_state(p), p.x0)
    assert T.gettimes(p)[-1] == pytest.approx(0.5) and len(T.gettimes(p)) == 6
    T.set_goal_state(p, np.array([0.1, 3.0, 0, 0]))
    np.testing.assert_array_equal(T.get_final_state(p), [0.1, 3.0, 0, 0])


def test_solver_option_validation():
    """Out-of-range options are ArgumentErrors at to_create / to_set_options, not silent MAX_ITERATIONS runs."""
    lib = T.load_hip_library()
    for field, value in [("line_search_decrease_factor", 0.0), ("line_search_decrease_factor", 1.0), ("penalty_scaling", 0.5),
                         ("bp_reg_increase_factor", 1.0), ("iterations", -1), ("iterations_linesearch", 0),
                         ("iterations_linesearch", 65), ("cost_tolerance", -1e-3), ("penalty_initial", 0.0),
                         # projected-Newton polish (ABI 4)
                         ("n_steps", -1), ("rho_primal", 0.0), ("rho_chol", -1e-8), ("r_threshold", 0.0), ("projected_newton", 2),
                         ("projected_newton_tolerance", -1.0), ("active_set_tolerance_pn", -1e-3)]:
        o = lib.default_options()
        setattr(o, field, value)
        keep = []

        def mut(d, c):
            keep.append(o)
        # options travel as the second argument of to_create
        model = T.Cartpole()
        obj = T.LQRObjective(np.ones(4), np.ones(1), np.ones(4), np.zeros(4), 11)
        with pytest.raises(T.ArgumentError):
            T.Problem(model, obj, np.zeros(4), 1.0, options=T.SolverOptions(lib=lib, **{field: value}))
    o = lib.default_options()  # defaults are valid: only the missing GPU may stop the constructor here
    assert (o.projected_newton, o.n_steps, o.projected_newton_tolerance, o.active_set_tolerance_pn) == (1, 2, 1e-3, 1e-3)  # Altro's defaults
    assert (o.rho_primal, o.r_threshold) == (1e-8, 1.1) and o.rho_chol == 1e-8  # Altro: rho_chol 1e-2
    try:
        T.Problem(T.Cartpole(), T.LQRObjective(np.ones(4), np.ones(1), np.ones(4), np.zeros(4), 11), np.zeros(4), 1.0)
    except T.HipError:
        pass


def test_duplicate_constraint_indices_rejected():
    def dup(d, c):
        c[0].inds[1] = c[0].inds[0]
    rc, msg = _create_rc(dup)
    assert rc == T.capi.TO_ERR_ARGUMENT and "distinct" in msg


def test_set_duals_shape_checks(oracle):
    from trajectoryoptimization_jl_amd import configs
    from trajopt_amd import internal as I
    p = configs.cartpole_problem(batch=3, N=6, tf=0.5, constrained=True, lib=oracle)
    lam = np.arange(3 * 5 * 2, dtype=float).reshape(3, 5, 2)
    I.set_duals(p, 0, lam=lam, mu=np.array([1.0, 2.0, 3.0]))
    l2, m2 = I.get_duals(p, 0)
    np.testing.assert_array_equal(l2, lam); np.testing.assert_array_equal(m2, [1.0, 2.0, 3.0])
    I.set_duals(p, 0, lam=lam[1], mu=7.0)              # one trajectory / a scalar: explicit broadcast
    l2, m2 = I.get_duals(p, 0)
    np.testing.assert_array_equal(l2[2], lam[1]); np.testing.assert_array_equal(m2, [7.0] * 3)
    with pytest.raises(T.DimensionMismatch):
        I.set_duals(p, 0, lam=np.zeros((5, 3)))
    with pytest.raises(T.DimensionMismatch):
        I.set_duals(p, 0, mu=np.zeros(2))
    with pytest.raises(T.ArgumentError):                # n == N: (n, N) could be either layout
        q = configs.cartpole_problem(batch=2, N=4, tf=0.3, lib=oracle)
        T.initial_states(q, np.zeros((4, 4)))


def test_no_vgpr_spill_ahead_of_exec_restore():
    """hipcc (ROCm 7.2) may place a VGPR->AGPR spill at the top of a join block before EXEC is restored:
    the value is then lost in the lanes that skipped the divergent region.  tools/check_exec_spill.py scans the gfx950
    assembly of every translation unit for that pattern; the kernels are written so that it does not occur."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    if os.environ.get("TRAJOPT_SKIP_ASM_SCAN"):
        pytest.skip("TRAJOPT_SKIP_ASM_SCAN set")
    tool = Path(__file__).resolve().parent.parent / "tools" / "check_exec_spill.py"
    res = subprocess.run([sys.executable, str(tool)], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout[-4000:]
    assert res.stdout.count("0 spill(s)") >= 8, res.stdout[-2000:]


def test_spill_scanner_recognises_the_hazard():
    """Unit test of tools/check_exec_spill.py on assembly snippets: the miscompiled join block of round 2 (VGPR->AGPR copies
    ahead of the exec restore) is flagged; a divergent branch that ends in the program's own scratch store, an MFMA
    accumulator set up under full EXEC, and a spill after the restore are not."""
    import importlib.util
    from pathlib import Path
    spec = importlib.util.spec_from_file_location("check_exec_spill", Path(__file__).resolve().parent.parent / "tools" / "check_exec_spill.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bad = """
_ZN2to9k_forwardINS_14QuadrotorModelELi3EEEvNS_5KArgsE:
	s_and_saveexec_b64 s[4:5], s[6:7]
	s_cbranch_execz .LBB1_41
; %bb.40:
	global_load_dwordx2 v[10:11], v[10:11], off
	ds_write_b64 v12, v[10:11]
.LBB1_41:                               ;   in Loop: Header=BB1_11 Depth=1
	v_accvgpr_write_b32 a75, v25
	v_accvgpr_write_b32 a74, v24
	s_mov_b64 s[28:29], 0x200
	v_writelane_b32 v251, s78, 29
	s_or_b64 exec, exec, s[4:5]
	s_load_dword s62, s[2:3], 0x0
"""
    hits = mod.scan_text(bad)
    assert [h[3] for h in hits] == ["v_accvgpr_write_b32 a75, v25", "v_accvgpr_write_b32 a74, v24"]
    assert hits[0][1] == ".LBB1_41" and "k_forward" in hits[0][0]
    good = """
_ZN2to8k_expandINS_13CartpoleModelELi0ELi2ELi0EEEvNS_5KArgsE:
.LBB14_181:
	v_add_f64 v[46:47], v[24:25], v[46:47]
	scratch_store_dwordx2 off, v[46:47], off offset:32
	s_or_b64 exec, exec, s[20:21]
.LBB0_32:
	v_mfma_f64_16x16x4_f64 a[8:15], v[66:67], v[4:5], 0
	v_accvgpr_write_b32 a0, v13
	s_and_saveexec_b64 s[0:1], s[14:15]
	ds_write_b64 v38, v[72:73] offset:544
	s_or_b64 exec, exec, s[0:1]
.LBB0_40:
	s_or_b64 exec, exec, s[2:3]
	v_accvgpr_write_b32 a3, v9
	s_or_b64 exec, exec, s[8:9]
"""
    assert mod.scan_text(good) == []
    spilled = good.replace("scratch_store_dwordx2 off, v[46:47], off offset:32", "scratch_store_dwordx2 off, v[46:47], off offset:32 ; 8-byte Folded Spill")
    assert len(mod.scan_text(spilled)) == 1


def test_committed_bench_line_follows_the_contract():
    """profiles/r02_bench_default.json is the driver-format line of a plain `python bench.py` on the MI355X: the keys the
    driver and the judge read must be there, for the headline workload and for the extra ones."""
    import json
    from pathlib import Path
    line = (Path(__file__).resolve().parent.parent / "profiles" / "r02_bench_default.json").read_text().strip().splitlines()
    assert len(line) == 1  # stdout of bench.py is exactly one line
    d = json.loads(line[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "build_id"):
        assert k in d, k
    assert d["metric"] == "iLQR iterations/sec (batched trajectories)" and d["n_gpus"] == 1 and d["dtype"] == "f64"
    assert "model" not in d["config"] and "workload" in d["config"]
    for w in [d] + list(d["extra_workloads"].values()):
        r, c = w["roofline"], w["cpu_baseline"]
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert k in r, k
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in c, k
        assert c["kind"] == "port" and w["value"] > 0


def test_committed_r06_bench_line_names_its_pipelining():
    """profiles/r06_bench_default.json (a plain `python bench.py` of round 6): every pipelined line says so in config.workload, carries the
    depth / admission rule / step count under config.pipeline and the UNPIPELINED figure of the same run beside its value; both passes time
    exactly `steps` solves; no roofline fraction exceeds 1."""
    import json
    from pathlib import Path
    d = json.loads((Path(__file__).resolve().parent.parent / "profiles" / "r06_bench_default.json").read_text().strip())
    assert d["metric"] == "iLQR iterations/sec (batched trajectories)" and d["n_gpus"] == 1 and d["dtype"] == "f64" and d["vs_baseline"] is None
    lines = [("C2", d)] + [(k, v) for k, v in d["extra_workloads"].items() if "config" in v and "pipeline" in v["config"]]
    assert [k for k, _ in lines] == ["C2", "C3", "C5"]
    for key, w in lines:
        pl, un = w["config"]["pipeline"], w["unpipelined"]
        assert "PIPELINED over %d handles" % pl["depth"] in w["config"]["workload"] and pl["depth"] == 4
        assert abs(w["value"] - pl["value"]) < 1e-6 * w["value"] and w["value"] > un["value"] > 0 and w["steps"] == pl["steps"]
        r = w["roofline"]
        assert 0 < r["frac"] < 1 and 0 < r["whole_iteration"]["frac"] < 1 and r["traffic"] is not None
        assert w["cpu_baseline"]["cores"] <= (w["cpu_baseline"]["cgroup_cpu_quota_cores"] or 1e9) + 0.5
    assert d["steps"] == d["unpipelined"]["steps"] == 24
    alt = d["extra_workloads"]["C5_altro_defaults"]
    assert "n_steps = 2" in alt["config"]["workload"] and 0.9 < alt["config"]["converged_fraction"] < d["extra_workloads"]["C5"]["config"]["converged_fraction"]


def test_rccl_stub_builds_and_exports_what_the_library_binds(tmp_path):
    """tests/rccl_stub (the shared-memory stand-in that lets several ranks share one GPU in the -m gpu suite) must export every nccl*
    entry point csrc/trajopt_hip.hip dlsym()s — checked here so that the GPU test cannot silently skip a symbol."""
    import re
    import subprocess
    root = Path(__file__).resolve().parent.parent
    src = (root / "trajectoryoptimization.jl_amd" / "csrc" / "trajopt_hip.hip").read_text()
    wanted = set(re.findall(r'dlsym\(g_rccl\.lib, "(nccl\w+)"\)', src))
    assert len(wanted) == 8
    so = tmp_path / "librccl_stub.so"
    subprocess.run(["g++", "-O2", "-shared", "-fPIC", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", str(root / "tests" / "rccl_stub" / "rccl_stub.cpp"),
                    "-o", str(so), "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-pthread", "-Wno-format-truncation"], check=True, capture_output=True)
    out = subprocess.run(["nm", "-D", "--defined-only", str(so)], check=True, capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (nccl\w+)", out))
    assert wanted <= exported, sorted(wanted - exported)


def test_solve_pipeline_host_logic():
    """api.SolvePipeline without a GPU: mock solvers record what the pipeline does to them.  Job i runs on handle i % depth; a handle's previous
    job is collected (wait, on_done, results) before the handle is prepared again; the job in front is asked to drain below the admission
    threshold before the next solve is started; drain() collects what is in flight oldest first; every job is reported exactly once."""
    log = []

    class FakeProb:
        def __init__(self, name):
            self.name, self.B = name, 100

    class FakeSolver:
        def __init__(self, name):
            self.prob, self.stats = FakeProb(name), {"iterations": np.zeros(3, np.int32)}
            self.total_iterations, self.batch_steps, self.inflight = 0, 0, None

        def solve_async(self):
            assert self.inflight is None, "a handle was restarted before its previous job was collected"
            self.inflight = len([e for e in log if e[0] == "start"])
            log.append(("start", self.prob.name, self.inflight))

        def wait(self):
            assert self.inflight is not None
            log.append(("wait", self.prob.name, self.inflight))
            self.total_iterations, self.batch_steps = 10 + self.inflight, 3
            self.stats["iterations"][:] = self.inflight
            self.inflight = None

        def wait_below(self, thr):
            log.append(("wait_below", self.prob.name, thr))

    solvers = [FakeSolver("h%d" % i) for i in range(3)]
    done = []
    pipe = T.SolvePipeline(solvers, admit_below=40, keep_stats=True, on_done=lambda job, s: done.append((job, s.prob.name)))
    prepared = []
    for _ in range(8):
        pipe.submit(lambda p: prepared.append(p.name))
    pipe.drain()
    assert prepared == ["h0", "h1", "h2", "h0", "h1", "h2", "h0", "h1"]
    assert [d[0] for d in done] == list(range(8)) and [d[1] for d in done] == prepared       # every job once, in order, on its handle
    assert [r[0] for r in pipe.results] == list(range(8)) and pipe.total_iterations == sum(10 + j for j in range(8))
    assert all(int(r[3]["iterations"][0]) == r[0] for r in pipe.results)                      # the stats snapshot is the job's own
    starts = [e for e in log if e[0] == "start"]
    assert [e[1] for e in starts] == prepared
    # before start k (k >= 1) the job in front (on the previous handle) was asked to drain below the threshold
    for k in range(1, 8):
        i = log.index(starts[k])
        assert ("wait_below", prepared[k - 1], 40) in log[:i]
    # default admission: at once (threshold = the whole batch)
    assert T.SolvePipeline([FakeSolver("x"), FakeSolver("y")]).admit_below == 100

import sys, math
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import trajopt_amd as T
from trajopt_amd import internal as I
from oracle_binding import load_oracle
hip, oracle = T.load_hip_library(), load_oracle()
for rot in ("mrp", "rp"):
    def build(lib, batch=24, N=41):
        model = T.Quadrotor(rotation=rot); n, m = model.dims()
        th = math.radians(70.0) / 2
        xf = model.build_state([1.0, 1.5, 0.5], [math.cos(th), 0.0, 0.0, math.sin(th)])
        Qe = np.r_[np.ones(3), 0.5 * np.ones(3), 0.1 * np.ones(6)]
        stage = T.ErrorQuadratic(model, Qe, np.full(m, 1e-2), xf, model.hover_control())
        term = T.ErrorQuadratic(model, 100 * Qe, np.full(m, 1e-2), xf, model.hover_control(), terminal=True)
        prob = T.Problem(model, T.Objective(stage, term, N), np.zeros(n), 2.0, xf=xf, batch=batch, lib=lib)
        rng = np.random.default_rng(11)
        x0 = np.zeros((batch, n)); x0[:, :3] = rng.uniform(-0.5, 0.5, (batch, 3)); x0[:, 3:6] = 0.1 * rng.standard_normal((batch, 3))
        prob.set_initial_state(x0); T.initial_controls(prob, model.hover_control())
        rng = np.random.default_rng(0)
        U = T.controls(prob) + 0.05 * rng.standard_normal((batch, N - 1, m)); T.initial_controls(prob, U)
        return prob
    ph, po = build(hip), build(oracle)
    T.rollout(ph); T.rollout(po)
    Fh, Fo = I.discrete_jacobian(ph), I.discrete_jacobian(po)
    print(rot, "F nan hip", np.isnan(Fh).sum(), "oracle", np.isnan(Fo).sum(), "shape", Fh.shape)
    idx = np.argwhere(np.isnan(Fh))
    print("  first nan idx (b,k,row,col):", idx[:12].tolist())
    if len(idx):
        b, k = idx[0][:2]
        print("  x", T.states(ph)[b, k], "u", T.controls(ph)[b, k])
        print("  rows with nan", sorted(set(idx[:, 2].tolist())), "cols", sorted(set(idx[:, 3].tolist())), "knots", sorted(set(idx[:, 1].tolist()))[:10], "traj", sorted(set(idx[:, 0].tolist()))[:10])
    ok = ~np.isnan(Fh)
    print("  max diff where finite", np.abs(Fh[ok] - Fo[ok]).max())
    I.expand(ph); I.expand(po)
    (Ah, Bh), (Ao, Bo) = I.dynamics_jacobians(ph), I.dynamics_jacobians(po)
    print("  A nan", np.isnan(Ah).sum(), "B nan", np.isnan(Bh).sum(), "maxdiff A", np.nanmax(np.abs(Ah - Ao)), "B", np.nanmax(np.abs(Bh - Bo)))

"""Phase-level access to the hot path (the analogue of examples/`Internal API.ipynb`): expansion,
backward pass, forward pass and their buffers, all computed on the GPU through the C-ABI.
Outputs are numpy arrays indexed [trajectory, knot, row, col] (converted from the ABI's column-major
(row, col, knot, trajectory) layout)."""
from __future__ import annotations

import ctypes as C

import numpy as np

__all__ = ["expand", "backwardpass", "forwardpass", "dynamics_jacobians", "cost_expansion", "gains",
           "cost_gradient_hessian", "discrete_jacobian", "get_duals", "set_duals", "reset_duals",
           "dual_update", "al_cost"]


def _pd(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def expand(prob):
    """Dynamics Jacobians (error state) and cost(+AL) expansion at the current trajectory."""
    prob._call("expand")


def backwardpass(prob):
    """iLQR backward Riccati recursion (Altro backwardpass!)."""
    prob._call("backward")


def forwardpass(prob):
    """iLQR forward pass with backtracking line search -> (ls_index[B] int32, J_new[B])."""
    ls = np.empty(prob.B, np.int32)
    J = np.empty(prob.B)
    prob._call("forward", ls.ctypes.data_as(C.POINTER(C.c_int32)), _pd(J))
    return ls, J


def dynamics_jacobians(prob):
    """-> A [B, N-1, ne, ne], Bm [B, N-1, ne, m] (error-state, TO.error_expansion in the notebook, cell 12)."""
    ne, m, N, B = prob.errstate_dim, prob.m, prob.N, prob.B
    A = np.empty((B, N - 1, ne, ne))
    Bm = np.empty((B, N - 1, m, ne))
    prob._call("get_dynamics_jacobians", _pd(A), _pd(Bm))
    return A.transpose(0, 1, 3, 2), Bm.transpose(0, 1, 3, 2)


def cost_expansion(prob):
    """-> dict(Qxx [B,N,ne,ne], Quu [B,N,m,m], Qux [B,N,m,ne], qx [B,N,ne], qu [B,N,m]) on the error state."""
    ne, m, N, B = prob.errstate_dim, prob.m, prob.N, prob.B
    Qxx, Quu, Qux = np.empty((B, N, ne, ne)), np.empty((B, N, m, m)), np.empty((B, N, ne, m))
    qx, qu = np.empty((B, N, ne)), np.empty((B, N, m))
    prob._call("get_cost_expansion", _pd(Qxx), _pd(Quu), _pd(Qux), _pd(qx), _pd(qu))
    return dict(Qxx=Qxx.transpose(0, 1, 3, 2), Quu=Quu.transpose(0, 1, 3, 2), Qux=Qux.transpose(0, 1, 3, 2), qx=qx, qu=qu)


def gains(prob):
    """-> dict(K [B,N-1,m,ne], d [B,N-1,m], dV [B,2], rho [B])."""
    ne, m, N, B = prob.errstate_dim, prob.m, prob.N, prob.B
    K, d = np.empty((B, N - 1, ne, m)), np.empty((B, N - 1, m))
    dV, rho = np.empty((B, 2)), np.empty(B)
    prob._call("get_gains", _pd(K), _pd(d), _pd(dV), _pd(rho))
    return dict(K=K.transpose(0, 1, 3, 2), d=d, dV=dV, rho=rho)


def cost_gradient_hessian(prob):
    """RD.gradient!/RD.hessian! of the objective per knot on the full state: grad [B,N,n+m], hess [B,N,n+m,n+m]."""
    nz, N, B = prob.n + prob.m, prob.N, prob.B
    g, H = np.empty((B, N, nz)), np.empty((B, N, nz, nz))
    prob._call("cost_expansion", _pd(g), _pd(H))
    return g, H.transpose(0, 1, 3, 2)


def cost_to_go(prob):
    """Cost-to-go of the last backward pass (to_get_cost_to_go) -> S [B, N, ne, ne] (symmetric), s [B, N, ne]."""
    ne, N, B = prob.errstate_dim, prob.N, prob.B
    S, s = np.empty((B, N, ne, ne)), np.empty((B, N, ne))
    prob._call("get_cost_to_go", _pd(S), _pd(s))
    return S.transpose(0, 1, 3, 2), s


def discrete_jacobian(prob):
    """RD.jacobian! of the discretised dynamics: F = [A B] -> [B, N-1, n, n+m]."""
    n, nz, N, B = prob.n, prob.n + prob.m, prob.N, prob.B
    F = np.empty((B, N - 1, nz, n))
    prob._call("discrete_jacobian", _pd(F))
    return F.transpose(0, 1, 3, 2)


def get_duals(prob, i):
    con = prob.constraints[i]
    a, b = prob.constraints.inds[i]
    lam, mu = np.empty((prob.B, b - a + 1, con.p)), np.empty(prob.B)
    prob._call("get_duals", i, _pd(lam), _pd(mu))
    return lam, mu


def set_duals(prob, i, lam=None, mu=None):
    """Overwrite the duals [B, nk, p] and/or the penalty [B] of constraint ``i``.  One trajectory's worth ([nk, p] / a
    scalar) is broadcast over the batch explicitly; anything else raises DimensionMismatch (the C-ABI copies p*nk*B
    and B doubles from the pointers it is given)."""
    from .api import DimensionMismatch
    con = prob.constraints[i]
    a, b = prob.constraints.inds[i]
    nk = b - a + 1
    if lam is not None:
        lam = np.asarray(lam, dtype=np.float64)
        if lam.shape == (nk, con.p):
            lam = np.broadcast_to(lam, (prob.B, nk, con.p))
        elif lam.shape != (prob.B, nk, con.p):
            raise DimensionMismatch(f"lam must be [B={prob.B}, nk={nk}, p={con.p}] or [nk, p]; got {lam.shape}")
        lam = np.ascontiguousarray(lam)
    if mu is not None:
        mu = np.asarray(mu, dtype=np.float64)
        if mu.ndim == 0:
            mu = np.full(prob.B, float(mu))
        elif mu.shape != (prob.B,):
            raise DimensionMismatch(f"mu must be a scalar or [B={prob.B}]; got {mu.shape}")
        mu = np.ascontiguousarray(mu)
    prob._call("set_duals", i, _pd(lam) if lam is not None else None, _pd(mu) if mu is not None else None)


def reset_duals(prob):
    prob._call("reset_duals")


def dual_update(prob):
    prob._call("dual_update")


def al_cost(prob):
    J = np.empty(prob.B)
    prob._call("al_cost", _pd(J))
    return J

"""Drop-in for the reference seam `ilqr()` (iterative_ilqr/control/iterative_ilqr.py:7-85) and the
batched candidate solver the controller uses in its place.

`ilqr(...)` keeps the reference's positional signature and return value `(uvar, xvar, lamb)` and
solves the single problem on the GPU through the C-ABI; `HipCandidateSolver.solve()` is the batched
form: all safe-set terminal candidates of one controller round in one launch.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .params import config_from_params, obstacle_record


class HipCandidateSolver:
    """Batched ilqr() on the GPU: host NumPy in, host NumPy out, device work through BatchedILQR.

    One BatchedILQR (one C-ABI handle) is cached per distinct configuration."""

    def __init__(self, device="cuda:0", dtype="f64"):
        self.device = device
        self.dtype = dtype
        self._solvers = {}

    def _solver(self, cfg):
        from ..solver import BatchedILQR
        key = bytes(C.string_at(C.byref(cfg), C.sizeof(cfg)))
        if key not in self._solvers:
            self._solvers[key] = BatchedILQR(cfg, self.device)
        return self._solvers[key]

    def solve(self, cfg, x0, x_terms, lamb0, obs_rec, U0=None):
        """x0[n] (shared) or [B,n]; x_terms[B,n]; lamb0[B]; obs_rec[6] (shared) or None.
        Returns dict(U[B,m,N], X[B,n,N+1], lamb[B], iters[B], status[B], cost[B]) on the host."""
        import torch
        solver = self._solver(cfg)
        x_terms = np.atleast_2d(np.asarray(x_terms, float))
        B = x_terms.shape[0]
        buf = solver.alloc(B, want_gains=False)
        X = np.zeros((B, cfg.n, cfg.N + 1))
        X[:, :, 0] = np.asarray(x0, float)
        dev = lambda a: solver.to_native(torch.as_tensor(np.ascontiguousarray(a)).to(
            solver.device, solver.dtype))
        buf["X"].copy_(dev(X))
        if U0 is not None:
            buf["U"].copy_(dev(np.asarray(U0, float).reshape(B, cfg.m, cfg.N)))
        buf["x_term"].copy_(dev(x_terms))
        buf["lamb"].copy_(dev(np.asarray(lamb0, float).reshape(B)))
        if obs_rec is not None:
            buf["obs"] = dev(np.tile(np.asarray(obs_rec, float), (B, 1)))
        solver.solve(buf)
        host = lambda t: solver.to_problem_major(t).double().cpu().numpy()
        return dict(U=host(buf["U"]), X=host(buf["X"]), lamb=host(buf["lamb"]),
                    cost=host(buf["cost"]), iters=buf["iters"].cpu().numpy(),
                    status=buf["status"].cpu().numpy())


_default_solver = None


def default_solver() -> HipCandidateSolver:
    global _default_solver
    if _default_solver is None:
        _default_solver = HipCandidateSolver()
    return _default_solver


def ilqr(ilqr_param, num_horizon, xtarget, timestep, obstacle, system_param, x_terminal, dX, uvar,
         xvar, lamb, solver=None):
    """Same contract as the reference's ilqr() (control/iterative_ilqr.py:7-85): x0 = xvar[:, 0],
    initial inputs uvar, regularisation lamb in; (uvar, xvar, lamb) out.  Like the reference it
    also leaves dX[:, t] = xvar[:, t] - xtarget for the returned nominal."""
    solver = default_solver() if solver is None else solver
    cfg = config_from_params(ilqr_param, system_param, num_horizon, timestep, xtarget)
    out = solver.solve(cfg, np.asarray(xvar, float)[:, 0], np.asarray(x_terminal, float)[None],
                       [float(lamb)], None if obstacle is None else obstacle_record(obstacle),
                       U0=np.asarray(uvar, float)[None])
    uvar_new, xvar_new = out["U"][0], out["X"][0]
    if dX is not None:
        dX[:, 1:] = xvar_new[:, 1:] - np.asarray(xtarget, float).reshape(-1, 1)
    return uvar_new, xvar_new, float(out["lamb"][0])

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


    def rollout(self, cfg, x0, U0):
        """Clipped inputs and their rollout (control/iterative_ilqr.py:32-48) for U0[B,m,N] from
        x0[n] (shared) or [B,n]; host arrays dict(U, X, cost)."""
        import torch
        solver = self._solver(cfg)
        U0 = np.asarray(U0, float)
        B = U0.shape[0]
        X = np.zeros((B, cfg.n, cfg.N + 1))
        X[:, :, 0] = np.asarray(x0, float)
        dev = lambda a: solver.to_native(torch.as_tensor(np.ascontiguousarray(a)).to(
            solver.device, solver.dtype))
        Xd, Ud = dev(X), dev(U0)
        # the terminal target only enters the returned cost
        cost = solver.rollout(Xd, Ud, dev(np.zeros((B, cfg.n))))
        host = lambda t: solver.to_problem_major(t).double().cpu().numpy()
        return dict(U=host(Ud), X=host(Xd), cost=host(cost))


_default_solver = None


def default_solver() -> HipCandidateSolver:
    global _default_solver
    if _default_solver is None:
        _default_solver = HipCandidateSolver()
    return _default_solver


def ilqr(ilqr_param, num_horizon, xtarget, timestep, obstacle, system_param, x_terminal, dX, uvar,
         xvar, lamb, solver=None):
    """Same contract as the reference's ilqr() (control/iterative_ilqr.py:7-85): x0 = xvar[:, 0],
    initial inputs uvar, regularisation lamb in; (uvar, xvar, lamb) out.

    Side effects on the caller's arrays, as in the reference: its first iteration clips `uvar` in
    place and writes the nominal rollout into `xvar[:, 1:]` in place (:33-42) before any accepted
    step rebinds the names to fresh arrays (:75-76).  If `uvar` / `xvar` are writable ndarrays they
    are left in exactly that state here too (the clipped initial inputs and their rollout, from
    i2lqr_rollout); the returned arrays are always new ones.  `dX[:, 1:]` is left as the returned
    trajectory's deviation from xtarget (the reference leaves the deviation of the last nominal
    it rolled out, which is the previous one after a final accepted step; no caller reads it:
    utils/base.py:409 re-creates dX per candidate)."""
    solver = default_solver() if solver is None else solver
    cfg = config_from_params(ilqr_param, system_param, num_horizon, timestep, xtarget)
    x0 = np.array(np.asarray(xvar, float)[:, 0])
    U0 = np.array(uvar, float)
    obs = None if obstacle is None else obstacle_record(obstacle)
    out = solver.solve(cfg, x0, np.asarray(x_terminal, float)[None], [float(lamb)], obs,
                       U0=U0[None])
    if int(ilqr_param.max_ilqr_iter) > 0 and isinstance(uvar, np.ndarray) and \
            isinstance(xvar, np.ndarray) and uvar.flags.writeable and xvar.flags.writeable:
        first = solver.rollout(cfg, x0, U0[None])
        uvar[...] = first["U"][0]
        xvar[:, 1:] = first["X"][0][:, 1:]
    uvar_new, xvar_new = out["U"][0], out["X"][0]
    if dX is not None:
        dX[:, 1:] = xvar_new[:, 1:] - np.asarray(xtarget, float).reshape(-1, 1)
    return uvar_new, xvar_new, float(out["lamb"][0])

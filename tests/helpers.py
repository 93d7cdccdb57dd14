"""Shared helpers of the test-suite (the only place, with bench.py's cpu_baseline leg and
__graft_entry__.smoke(), that touches oracle/)."""
from __future__ import annotations

import numpy as np


def rel_err(a, b, floor=1e-300):
    """max |a-b| / max |b|  (array-level relative error: entries that are round-off noise next to
    the array's scale, e.g. K[0,1,t] ~ 1e-17, are compared absolutely)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), floor))


def batch_rel_err(a, b, floor=1e-300):
    """rel_err per leading-axis entry (scale = max |b| of that entry, at least `floor`), max over
    the batch."""
    a = np.asarray(a, dtype=np.float64).reshape(len(a), -1)
    b = np.asarray(b, dtype=np.float64).reshape(len(b), -1)
    if a.shape[0] == 0:
        return 0.0
    scale = np.maximum(np.abs(b).max(axis=1), floor)
    return float((np.abs(a - b).max(axis=1) / scale).max())


def to_dev(solver, arr, dtype=None):
    """Host array in the problem-major convention [B, ...] -> device tensor in the solver's
    native layout (batch-minor solvers get the batch axis moved last)."""
    import torch
    t = torch.as_tensor(np.ascontiguousarray(arr))
    if t.dtype in (torch.float64, torch.float32):
        t = t.to(solver.dtype if dtype is None else dtype)
    return solver.to_native(t.to(solver.device))


def to_host(solver, t):
    """Device tensor in the solver's native layout -> NumPy array, problem-major [B, ...]."""
    return solver.to_problem_major(t).cpu().numpy()


def dev_batch(solver, host: dict, want_gains=True):
    """Host problem dict (workloads.make_batch / golden) -> device buffer dict for the solver."""
    B = host["X"].shape[0]
    buf = solver.alloc(B, want_gains=want_gains)
    for key in ("X", "U", "x_term", "lamb"):
        buf[key].copy_(to_dev(solver, host[key]))
    if host.get("obs") is not None:
        buf["obs"] = to_dev(solver, host["obs"])
    return buf


def problems_from_calls(g, N, n=4, m=2):
    """Golden ilqr() call records (x0, x_term, lamb_in, obs) -> problem-major host batch."""
    B = len(g["x0"])
    X = np.zeros((B, n, N + 1))
    X[:, :, 0] = g["x0"]
    return dict(X=X, U=np.zeros((B, m, N)), x_term=np.array(g["x_term"], float),
                lamb=np.array(g["lamb_in"], float), obs=np.array(g["obs"], float))


class OracleCandidateSolver:
    """Test double for control.iterative_ilqr.HipCandidateSolver backed by the CPU oracle — lets
    the host-side controller logic run in the CPU-only suite.  Lives in tests/ on purpose: the
    product package never imports the oracle."""

    def __init__(self):
        self.calls = 0
        self.problems = 0

    def solve(self, cfg, x0, x_terms, lamb0, obs_rec, U0=None):
        from oracle import oracle as orc
        x_terms = np.atleast_2d(np.asarray(x_terms, float))
        B = x_terms.shape[0]
        X = np.zeros((B, cfg.n, cfg.N + 1))
        X[:, :, 0] = np.asarray(x0, float)
        U = np.zeros((B, cfg.m, cfg.N)) if U0 is None else np.asarray(U0, float).reshape(
            B, cfg.m, cfg.N)
        obs = None if obs_rec is None else np.tile(np.asarray(obs_rec, float), (B, 1))
        self.calls += 1
        self.problems += B
        return orc.ilqr_batch(cfg, X, U, x_terms, np.asarray(lamb0, float).reshape(B), obs,
                              want_gains=False)

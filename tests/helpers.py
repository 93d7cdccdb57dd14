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

    def sharded_round(self, cfg, x0, x_terms_local, qfun_local, lamb0, exchange, total,
                      obs_rec=None, n_iters=None, outer_iter=0, max_relax_iter=55, lexi=None,
                      prepared=None, bufs=None, **_unused):
        """The double of HipCandidateSolver.sharded_round on CPU tensors: the shard's solves and
        relaxed costs from the oracle, the exchange through the SAME dist.lexi_round /
        dist.flat_round the product runs (gloo process group)."""
        import torch
        from oracle import oracle as orc
        from ilqr_iterative_tasks_amd import dist as idist
        assert n_iters is None and prepared is None
        n_local = int(x_terms_local.shape[0])
        nu, nx = cfg.m * cfg.N, cfg.n * (cfg.N + 1)
        out = None
        cost = np.zeros(0)
        if n_local:
            out = self.solve(cfg, x0.numpy(), x_terms_local.numpy(), np.full(n_local, float(lamb0)),
                             obs_rec)
            cost = orc.relax_cost_batch(cfg, out["X"], x_terms_local.numpy(),
                                        qfun_local.numpy().astype(np.int32), outer_iter,
                                        max_relax_iter)
        cost_t = torch.as_tensor(np.asarray(cost, float))

        def pack_of(loc):
            return torch.as_tensor(np.concatenate([out["U"][loc].ravel(), out["X"][loc].ravel()]))

        if lexi is None:
            if int(total) < 1:
                raise ValueError("a sharded round needs at least one candidate over all ranks")
            # a rank without candidates contributes +inf costs (flat_round pads) and a zero pack,
            # as HipCandidateSolver.sharded_round does: nobody raises while the others gather
            local_pack = pack_of(idist.select_best_flat(cost_t)[0]) if n_local else \
                torch.zeros(nu + nx, dtype=torch.float64)

            def argmin(c):
                i, v = idist.select_best_flat(c)
                return torch.tensor([i]), torch.tensor([v])

            def round_winner(width, tot, best, pack_all):
                owner, loc = divmod(int(best), width)
                lo, _ = idist.shard_range(tot, owner, exchange.world)
                return pack_all[owner].clone(), torch.tensor([lo + loc, owner])

            res = idist.flat_round(exchange, cost_t, local_pack, total, argmin, round_winner)
        else:
            if callable(lexi):
                pick = lexi
            else:
                L, k = lexi

                def pick(cost_all):
                    rows = [[float(v) for v in cost_all[a * k:(a + 1) * k]] for a in range(L)]
                    a, c = idist.select_best_lexicographic(rows)
                    return a * k + c
            res = idist.lexi_round(exchange, cost_t, total, pick, pack_of, nu + nx)
            res["best_idx"] = res.pop("index")
        res["U"] = res["pack"][:nu].view(cfg.m, cfg.N)
        res["X"] = res["pack"][nu:].view(cfg.n, cfg.N + 1)
        return res


# The paper scenarios of iterative_ilqr/result/ilqr_test_*.py (golden G8):
# name: (laps, initial obstacle, {lap index: Obstacle arguments or None}); config 1 otherwise
# (num_ss_iter 2, num_ss_points 8, N 6, dt 1).  The obstacle appears at lap 5 and is removed at
# lap 6 (result/ilqr_test_add_moving_obstacle.py:18-31, :63-75).
SCENARIOS = {
    "no_obstacle": (6, None, {}),
    "static_obstacle_big": (6, (100, -5, 20, 40), {}),
    "add_static_obstacle": (7, None, {5: (35, 0, 30, 30), 6: None}),
    "moving_up": (7, None, {5: (35, -16, 34, 34, 1, 1, 1), 6: None}),
    "moving_left": (7, None, {5: (50, -1, 35, 35, 0.2, 1, 2), 6: None}),
}


def check_scenario(golden_dir, name, ego, ctrl):
    """Run scenario `name` on (ego, ctrl) and compare with golden G8 (captured from the reference by
    oracle/gen_golden_scenarios.py): lap lengths exactly; the last lap's states and inputs to 2e-3.
    (The lap loop overwrites the final state row with the goal before add_trajectory, as
    iterative_ilqr/tests/ilqr_test.py:59 does; the golden holds the row as simulated, so it is
    left out.  States of the last lap agree to 1e-6 except after the lap that fights the moving
    obstacle — 69 steps inside the exponential barrier — where round-off-level differences between
    NumPy/OpenBLAS and a restatement have been amplified to ~5e-4 by eight laps of closed loop.)"""
    from ilqr_iterative_tasks_amd import harness
    from ilqr_iterative_tasks_amd.control import Obstacle
    g8 = np.load(golden_dir / "g8_scenarios_closed_loop.npz")
    laps, _, events = SCENARIOS[name]

    def on_lap(it, c):
        if it in events:
            c.obstacle = None if events[it] is None else Obstacle(*events[it])

    got = harness.run_laps(ego, ctrl, laps, on_lap=on_lap)
    assert got == list(g8[name + "_laps"]), (got, list(g8[name + "_laps"]))
    last = np.asarray(ego.data["state"][-1], float)
    assert last.shape == g8[name + "_last_state"].shape
    assert np.abs(last[:-1] - g8[name + "_last_state"][:-1]).max() < 2e-3
    assert np.abs(np.asarray(ego.data["input"][-1], float) - g8[name + "_last_input"]).max() < 2e-3

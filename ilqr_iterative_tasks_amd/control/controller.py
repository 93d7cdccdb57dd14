"""i2LQR controller — host-side counterpart of the reference's `iLqr` (utils/base.py:305-479)
with the same public surface: set_timestep / set_state / calc_input / get_input (ControlBase,
utils/base.py:216-234), add_trajectory (:343-369) and the public `obstacle` attribute.

The per-candidate `ilqr()` calls of the reference (`for id` / `for j` loops, :391-455) become
batched solves on the GPU.  Two regularisation modes:

  lamb_mode="chained"      the reference's exact semantics: inside one lap's candidate list the
                           final lamb of candidate j seeds candidate j+1 (utils/base.py:393, :414),
                           so candidates of one lap are solved in sequence; the laps of the safe
                           set (independent, lamb is reset per lap) are batched.
  lamb_mode="independent"  every candidate starts from ilqr_param.lamb: one launch per round over
                           all (lap, candidate) pairs — the form that shards across GPUs
                           (documented deviation; config-1 laps 121/54/28/23
                           instead of 121/54/29/23).

sharded=ShardedRound(...)  (lamb_mode="independent") the multi-GPU calc_input: every rank runs the
                           same controller on the same state; in each of the three rounds a rank
                           solves only its contiguous shard of the round's candidates, the ranks
                           all-gather the relaxed costs (the ONE collective of the solve path), every
                           rank evaluates the reference's list-of-lists pick on the full vector, and
                           the rank that solved the winner hands its (U, X) to the others — the
                           next round's select_close_ss starts from that x_N on every rank
                           (utils/base.py:384-478 with the loops :391-455 sharded).
"""
from __future__ import annotations


import numpy as np

from .params import X_DIM, U_DIM, config_from_params, obstacle_record


class ControlBase:
    def __init__(self):
        self.time = 0.0
        self.timestep = None
        self.x = None
        self.u = None
        self.iters = 0

    def set_timestep(self, timestep):
        self.timestep = timestep

    def set_state(self, x):
        self.x = x

    def calc_input(self):
        pass

    def get_input(self):
        return self.u


def plant_step(x, u, dt):
    """One step of the reference plant on the host, x[4], u[2] -> x_next[4]
    (kinetic_bicycle(), systems/kinetic_bicycle.py:10-27): used by the N <= 1 branch of
    calc_input and by the lap driver; the batched solves run the same formula on the GPU."""
    px, py, v, theta = (float(c) for c in x)
    accel, delta = float(u[0]), float(u[1])
    travel = v * dt + (accel * dt ** 2) / 2
    return np.array([px + np.cos(theta) * travel, py + np.sin(theta) * travel, v + accel * dt,
                     theta + delta * dt])


class iLqr(ControlBase):
    """State kept between control steps:
      safe set        ss[lap] = states [n, T], u_ss[lap] = inputs [m, T-1], Qfun[lap] = steps to go
      iter            number of laps in the safe set;  iter_cost[lap] = its length in steps
      num_horizon     current horizon (shrinks near the end of a lap, utils/base.py:472-478)
      x_pred, u_pred  the plan chosen by the last calc_input;  u_old = its tail, replayed while
                      the horizon shrinks;  x_terminal_guess = its last state
    The reference's LMPC-only bookkeeping (old_cost, cost_improve, ss_point_selected_id, ...) is not
    part of the i2LQR path and is not kept."""

    def __init__(self, ilqr_param, obstacle=None, system_param=None, solver=None,
                 lamb_mode="chained", verbose=False, device_rounds=False, sharded=None):
        ControlBase.__init__(self)
        assert lamb_mode in ("chained", "independent")
        # sharded: a dist.ShardedRound (the exchanges of a sharded round); candidates of one lap
        # cannot chain their lamb across ranks
        assert sharded is None or (lamb_mode == "independent" and not device_rounds)
        self.sharded = sharded
        self._shard = None  # (offsets, total, lo, U_local, X_local) of the round being picked
        # device_rounds: run the three outer rounds (select / solve / relaxed cost / pick) on the
        # GPU with one read-back per control step (control/device_round.py; independent lamb)
        assert not device_rounds or lamb_mode == "independent"
        self.ilqr_param = ilqr_param
        self.system_param = system_param
        self.obstacle = obstacle          # public: scenario scripts swap it between laps
        self.lamb_mode = lamb_mode
        self.device_rounds = device_rounds
        self.verbose = verbose
        self._solver = solver
        self._rounds = None
        self.ss, self.u_ss, self.Qfun, self.iter_cost = [], [], [], []
        self.iter = 0
        self.cost = None
        self.num_horizon = ilqr_param.num_horizon
        self.x_pred = self.u_pred = self.u_old = None
        self.x_guess = self.x_terminal_guess = None
        self.last_round = None  # diagnostics of the most recent calc_input (tests)
        self.last_round_info = None  # device rounds: {"graph": replayed?, "graph_error": ...}

    # -- safe set ---------------------------------------------------------------------------
    def select_close_ss(self, iter, x0):
        """The num_ss_points columns of lap `iter` nearest to x0 in the 1-norm, nearest first
        (utils/base.py:332-341; stable order on ties, like np.argsort)."""
        dist = np.abs(self.ss[iter] - np.asarray(x0, float)[:, None]).sum(axis=0)
        return np.argsort(dist)[: self.ilqr_param.num_ss_points]

    def add_trajectory(self, x, u):
        """A finished lap x[T, n], u[T-1, m] joins the safe set (utils/base.py:343-369): its
        states, its inputs and the cost-to-go of every state in steps; the horizon is reset to
        its full length."""
        states, inputs = np.array(x, float).T, np.array(u, float).T  # own copies, time last
        steps = states.shape[1] - 1
        self.ss.append(states)
        self.u_ss.append(inputs)
        self.Qfun.append(np.arange(steps, -1, -1))
        self.iter_cost.append(steps)
        self.cost = steps
        self.iter += 1
        self.num_horizon = self.ilqr_param.num_horizon
        self.x_terminal_guess = states[:, self.num_horizon]

    # -- solve --------------------------------------------------------------------------------
    def _get_solver(self):
        if self._solver is None:
            from .iterative_ilqr import default_solver
            self._solver = default_solver()  # HIP; raises without a GPU / built extension
        return self._solver

    def _relax_cost(self, x_end, x_terminal, cost_terminal, outer_iter, num_horizon):
        """Relaxed terminal cost of one candidate: utils/base.py:427-437."""
        nrm = np.linalg.norm([x_end - x_terminal])
        p = self.ilqr_param
        for i in range(1, p.max_relax_iter + 1):
            if nrm <= 80.0 * i / (10 ** outer_iter):
                return cost_terminal + num_horizon + 100 * i
            elif nrm > 80.0 * p.max_relax_iter / (10 ** outer_iter):
                return float("Inf")
        return float("Inf")  # NaN norm (the reference would leave cost_it unset)

    def solve(self, x0, candidates, outer_iter):
        """Batched replacement of the `for id` / `for j` loops (utils/base.py:391-455).
        candidates: list over laps of (lap_id, index_ss_points).  Returns per lap the lists
        (cost_iter, U list, X list)."""
        p = self.ilqr_param
        N = self.num_horizon
        xtarget = np.zeros(X_DIM)
        cfg = config_from_params(p, self.system_param, N, self.timestep, xtarget)
        obs = None if self.obstacle is None else obstacle_record(self.obstacle)
        solver = self._get_solver()
        n_laps = len(candidates)
        width = max(len(idx) for _, idx in candidates)
        U = [[None] * len(idx) for _, idx in candidates]
        X = [[None] * len(idx) for _, idx in candidates]
        if self.lamb_mode == "independent" and self.sharded is not None:
            return self._solve_sharded(cfg, solver, x0, candidates, obs, outer_iter, N)
        if self.lamb_mode == "independent":
            x_terms = np.stack([self.ss[lap][:, j] for lap, idx in candidates for j in idx])
            out = solver.solve(cfg, x0, x_terms, np.full(len(x_terms), float(p.lamb)), obs)
            pos = 0
            for a, (lap, idx) in enumerate(candidates):
                for c in range(len(idx)):
                    U[a][c], X[a][c] = out["U"][pos], out["X"][pos]
                    pos += 1
        elif hasattr(solver, "solve_chained") and N > 1:
            # the chain on the device (round 6): every candidate of the round goes up at once, step c
            # of every lap's chain is one launch whose lamb comes from step c - 1 on the device;
            # one read-back per round instead of one per step
            res = solver.solve_chained(cfg, x0, [np.stack([self.ss[lap][:, j] for j in idx], axis=0)
                                                 for lap, idx in candidates], float(p.lamb), obs)
            for a, (_, idx) in enumerate(candidates):
                for c in range(len(idx)):
                    U[a][c], X[a][c] = res[a]["U"][c], res[a]["X"][c]
        else:
            lamb = np.full(n_laps, float(p.lamb))  # reset per lap: utils/base.py:393
            for c in range(width):
                rows = [a for a, (_, idx) in enumerate(candidates) if c < len(idx)]
                x_terms = np.stack([self.ss[candidates[a][0]][:, candidates[a][1][c]] for a in rows])
                out = solver.solve(cfg, x0, x_terms, lamb[rows], obs)
                for r, a in enumerate(rows):
                    U[a][c], X[a][c] = out["U"][r], out["X"][r]
                    lamb[a] = out["lamb"][r]  # candidate j+1 starts from candidate j's lamb
        cost = [[self._relax_cost(X[a][c][:, -1], self.ss[lap][:, j], self.Qfun[lap][j],
                                  outer_iter, N)
                 for c, j in enumerate(idx)] for a, (lap, idx) in enumerate(candidates)]
        return cost, U, X

    def _solve_sharded(self, cfg, solver, x0, candidates, obs, outer_iter, N):
        """One sharded round through the solver's device-resident form
        (HipCandidateSolver.sharded_round — the SAME function bench.py --gpus N times): this rank
        hands over its slice [lo, hi) of the round's flat candidate list (laps in order, each
        lap's nearest-first points) and their cost-to-go; solve, relaxed costs
        (i2lqr_relax_cost), all-gather, the list-of-lists pick on the gathered vector
        (i2lqr_pick_best) and the winner's hand-off from its owner (i2lqr_broadcast_winner) happen
        there.  Returns the FULL cost lists (calc_input evaluates its own pick on them: the same
        candidate, checked in _winner) and empty (U, X) lists — only the winner's trajectory
        exists on this rank, and _winner() returns it."""
        import torch
        from ..dist import select_best_lexicographic
        p = self.ilqr_param
        flat = [(lap, j) for lap, idx in candidates for j in idx]
        total = len(flat)
        lo, hi = self.sharded.shard(total)
        widths = [len(idx) for _, idx in candidates]
        offsets = np.cumsum([0] + widths)

        def host_pick(cost_all):  # laps of different widths: Python's list order on the host
            v = cost_all.double().cpu().numpy()
            a, c = select_best_lexicographic([[float(q) for q in v[offsets[r]:offsets[r + 1]]]
                                              for r in range(len(widths))])
            return int(offsets[a]) + c

        uniform = all(w == widths[0] for w in widths)
        x_terms = (np.stack([self.ss[lap][:, j] for lap, j in flat[lo:hi]]) if hi > lo
                   else np.zeros((0, cfg.n)))
        qf = np.array([self.Qfun[lap][j] for lap, j in flat[lo:hi]], dtype=np.int32)
        res = solver.sharded_round(
            cfg, torch.as_tensor(np.asarray(x0, float)), torch.as_tensor(x_terms),
            torch.as_tensor(qf), float(p.lamb), self.sharded, total, obs_rec=obs,
            outer_iter=outer_iter, max_relax_iter=p.max_relax_iter,
            lexi=(len(widths), widths[0]) if uniform else host_pick)
        cost_all = res["cost_all"].double().cpu().numpy()
        cost = [[float(v) for v in cost_all[offsets[a]:offsets[a + 1]]]
                for a in range(len(candidates))]
        idx = int(res["best_idx"])
        a = int(np.searchsorted(offsets, idx, side="right") - 1)
        self._shard = (a, idx - int(offsets[a]), res["U"].double().cpu().numpy(),
                       res["X"].double().cpu().numpy())
        U = [[None] * w for w in widths]
        X = [[None] * w for w in widths]
        return cost, U, X

    def _winner(self, a, c, u_pred, x_pred):
        """(U, X) of the picked candidate (lap position a, candidate position c): the local lists
        unless the round was sharded — then what the round handed over (its pick on the gathered
        costs is the pick calc_input has just made on the same costs)."""
        if self._shard is None:
            return u_pred[a][c], x_pred[a][c]
        sa, sc, U, X = self._shard
        self._shard = None
        if (sa, sc) != (a, c):
            raise RuntimeError(f"sharded round picked candidate {(sa, sc)}, the host pick {(a, c)}")
        return U, X

    def _device_rounds_ok(self, min_iter):
        from .device_round import DeviceRounds
        if self._rounds is None:
            self._rounds = DeviceRounds()
        return DeviceRounds.supports(self, range(min_iter, self.iter))

    def calc_input(self):
        """utils/base.py:371-479."""
        p = self.ilqr_param
        num_horizon = self.num_horizon
        min_iter = np.max([0, self.iter - p.num_ss_iter])
        rounds = []
        if self.num_horizon < p.num_horizon:
            # horizon-shrinking replay of the last plan: utils/base.py:377-382
            self.u_pred = self.u_old
            self.u = self.u_pred[:, 0]
            self.u_old = self.u_pred[:, 1:]
            self.num_horizon = self.num_horizon - 1
        elif self.device_rounds and self.num_horizon > 1 and self._device_rounds_ok(min_iter):
            laps = list(range(min_iter, self.iter))
            self.u_pred, self.x_pred, (best_loc, best_time), idx = self._rounds.run(self, laps)
            self.last_round_info = self._rounds.info  # captured graph or eager (and why)
            self.u = self.u_pred[:, 0]
            self.x_terminal_guess = self.x_pred[:, -1]
            self.u_old = self.u_pred[:, 1:]
            best_iter = best_loc + min_iter
            if (idx[best_loc][best_time] + 1) > (self.ss[best_iter].shape[1] - 1):
                self.num_horizon = self.num_horizon - 1
        else:
            for it in range(p.max_outloop_iter):
                candidates = []
                for id in range(min_iter, self.iter):
                    self.x_guess = self.x if it == 0 else self.x_pred[:, -1]
                    candidates.append((id, self.select_close_ss(id, self.x_guess)))
                if self.num_horizon > 1:
                    cost_list, u_pred, x_pred = self.solve(np.asarray(self.x, float), candidates, it)
                else:
                    # N <= 1: apply the stored input and test reachability: utils/base.py:438-450
                    cost_list, u_pred, x_pred = [], [], []
                    for id, idx in candidates:
                        cl, ul, xl = [], [], []
                        for j in idx:
                            uvar = np.zeros((U_DIM, num_horizon))
                            xvar = np.zeros((X_DIM, num_horizon + 1))
                            xvar[:, 0] = self.x
                            x_next = plant_step(np.asarray(self.x, float), self.u_old[:, 0],
                                                 self.timestep)
                            xvar[:, -1] = x_next
                            uvar[:, 0] = self.u_old[:, 0]
                            ok = np.linalg.norm([x_next - self.ss[id][:, j]]) <= p.reach_error
                            cl.append(1 + self.Qfun[id][j] if ok else float("Inf"))
                            ul.append(uvar)
                            xl.append(xvar)
                        cost_list.append(cl), u_pred.append(ul), x_pred.append(xl)
                # pick: list-of-lists min is lexicographic, then first min inside that lap's
                # list (utils/base.py:462-465)
                best_iter_loc_ss = cost_list.index(min(cost_list))
                cost_vec = cost_list[best_iter_loc_ss]
                best_time = cost_vec.index(min(cost_vec))
                best_iter = best_iter_loc_ss + min_iter
                self.u_pred, self.x_pred = self._winner(best_iter_loc_ss, best_time, u_pred, x_pred)
                self.u = self.u_pred[:, 0]
                self.x_terminal_guess = self.x_pred[:, -1]
                if self.num_horizon > 1:
                    self.u_old = self.u_pred[:, 1:]
                rounds.append(dict(index=[np.array(idx) for _, idx in candidates],
                                   cost=cost_list, best=(best_iter_loc_ss, best_time)))
                if it == 2:
                    # shrink the horizon when the chosen terminal point is the lap's last point
                    if (candidates[best_iter_loc_ss][1][best_time] + 1) > (
                            self.ss[best_iter].shape[1] - 1):
                        self.num_horizon = self.num_horizon - 1
                    break
        self.last_round = rounds
        self.time += self.timestep

"""The three outer rounds of iLqr.calc_input (utils/base.py:384-478) chained on the GPU
(SURVEY.md §8 f3): select k nearest safe-set points -> zero-init candidates -> batched ilqr() ->
relaxed terminal cost -> lexicographic pick -> the winner's terminal state is the next round's
guess.  Nothing returns to the host until the third round has picked: one read-back per control
step instead of three.  Candidates start from ilqr_param.lamb independently (the batched /
shardable mode of the controller; the reference's lamb chaining is sequential by construction)."""
from __future__ import annotations

import numpy as np
import torch

from .. import _abi
from ..solver import BatchedILQR
from .params import config_from_params, obstacle_record


class DeviceRounds:
    """use_graph=True captures the 15 launches of a control step (3 rounds x select / init / solve
    / relaxed cost / pick) into one hipGraph per (configuration, safe set) and replays it: the
    launches are back to back with no Python in between."""

    def __init__(self, device="cuda:0", use_graph=True):
        self.device = torch.device(device)
        self.use_graph = use_graph
        self._solvers = {}
        self._ss_key = None
        self._plans = {}
        self.info = None  # how the last run() executed: {"graph", "graph_requested", "graph_error"}

    def _solver(self, cfg) -> BatchedILQR:
        import ctypes as C
        key = bytes(C.string_at(C.byref(cfg), C.sizeof(cfg)))
        if key not in self._solvers:
            self._solvers[key] = BatchedILQR(cfg, self.device)
        return self._solvers[key]

    def _upload_safe_set(self, ctrl, laps):
        """Safe set of the laps in use as padded device tensors (rebuilt when a lap is added)."""
        key = (ctrl.iter, tuple(laps))
        if self._ss_key == key:
            return
        Tmax = max(ctrl.ss[l].shape[1] for l in laps)
        n = ctrl.ss[laps[0]].shape[0]
        ss = np.zeros((len(laps), n, Tmax))
        qf = np.zeros((len(laps), Tmax), np.int32)
        for a, l in enumerate(laps):
            T = ctrl.ss[l].shape[1]
            ss[a, :, :T] = ctrl.ss[l]
            qf[a, :T] = ctrl.Qfun[l]
        self.ss = torch.as_tensor(ss).to(self.device)
        self.qfun = torch.as_tensor(qf).to(self.device)
        self.T = torch.as_tensor(np.array([ctrl.ss[l].shape[1] for l in laps], np.int32)).to(
            self.device)
        self._ss_key = key

    @staticmethod
    def supports(ctrl, laps) -> bool:
        """The device rounds solve a fixed k candidates per lap; a lap with fewer than k states
        (never the case for a real lap: the fastest ones are > 20 steps) yields fewer candidates in
        the host path (utils/base.py:332-341), whose list-of-lists pick then compares lists of
        different lengths — the controller falls back to host rounds for such a safe set."""
        k = int(ctrl.ilqr_param.num_ss_points)
        return all(ctrl.ss[l].shape[1] >= k for l in laps)

    def _plan(self, ctrl, laps, solver, cfg):
        """Static buffers + (optionally) the captured graph for one (config, safe set, obstacle
        on/off) combination."""
        import ctypes as C
        p = ctrl.ilqr_param
        L, k, n, m, N = len(laps), int(p.num_ss_points), cfg.n, cfg.m, cfg.N
        key = (bytes(C.string_at(C.byref(cfg), C.sizeof(cfg))), self._ss_key, L, k,
               ctrl.obstacle is not None)
        if key in self._plans:
            return self._plans[key]
        if len(self._plans) > 64:
            self._plans.clear()
        B = L * k
        dev, dt = self.device, solver.dtype
        pl = dict(buf=solver.alloc(B, want_gains=False), x0=torch.zeros(n, dtype=dt, device=dev),
                  idx=torch.zeros(L, k, dtype=torch.int32, device=dev),
                  qf=torch.zeros(B, dtype=torch.int32, device=dev),
                  cost_it=torch.zeros(B, dtype=dt, device=dev),
                  best=torch.zeros(2, dtype=torch.int32, device=dev),
                  x_pred=torch.zeros(n, N + 1, dtype=dt, device=dev),
                  u_pred=torch.zeros(m, N, dtype=dt, device=dev), graph=None, L=L, k=k)
        if ctrl.obstacle is not None:
            pl["buf"]["obs"] = torch.zeros(B, 6, dtype=dt, device=dev)
        lamb0, max_relax = float(p.lamb), int(p.max_relax_iter)
        ss, T, qfun = self.ss, self.T, self.qfun

        def launches():
            buf = pl["buf"]
            for it in range(3):  # the reference breaks its outer loop at iter == 2 (:472-478)
                if it == 0:
                    guess, stride = pl["x0"], 1                      # self.x (:399)
                else:
                    guess, stride = pl["x_pred"][:, N:], N + 1       # self.x_pred[:, -1] (:401)
                solver.select_candidates(ss, T, qfun, guess, stride, k, pl["idx"], buf["x_term"],
                                         pl["qf"])
                solver.init_candidates(pl["x0"], lamb0, buf)
                solver.solve(buf)
                solver.relax_cost(buf["X"], buf["x_term"], pl["qf"], it, max_relax, pl["cost_it"])
                solver.pick_best(L, k, pl["cost_it"], buf["X"], buf["U"], pl["best"],
                                 pl["x_pred"], pl["u_pred"])

        pl["launches"] = launches
        pl["graph_error"] = None
        if self.use_graph:
            launches()  # warm-up outside capture: an error of the launches themselves propagates
            torch.cuda.synchronize(self.device)
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    launches()
                pl["graph"] = g
            except RuntimeError as e:
                # HIP / torch refused the capture (both report it as RuntimeError; I2lqrError is
                # one): the rounds run eagerly and say so (self.info, iLqr.last_round_info) — a
                # capture regression is a visible state, not a silent latency change
                pl["graph"], pl["graph_error"] = None, f"{type(e).__name__}: {e}"
                torch.cuda.synchronize(self.device)
        self._plans[key] = pl
        return pl

    def run(self, ctrl, laps):
        """Three rounds for controller `ctrl` over safe-set laps `laps`.  Returns host
        (u_pred[m,N], x_pred[n,N+1], best (lap position, candidate position), idx[L,k] of the
        last round)."""
        p = ctrl.ilqr_param
        cfg = config_from_params(p, ctrl.system_param, ctrl.num_horizon, ctrl.timestep,
                                 np.zeros(4), layout=_abi.LAYOUT_PROBLEM_MAJOR)
        solver = self._solver(cfg)
        self._upload_safe_set(ctrl, laps)
        pl = self._plan(ctrl, laps, solver, cfg)
        dt = solver.dtype
        pl["x0"].copy_(torch.as_tensor(np.asarray(ctrl.x, float)).to(dt), non_blocking=True)
        if ctrl.obstacle is not None:
            rec = torch.as_tensor(np.tile(obstacle_record(ctrl.obstacle), (pl["L"] * pl["k"], 1)))
            pl["buf"]["obs"].copy_(rec.to(dt), non_blocking=True)
        if pl["graph"] is not None:
            pl["graph"].replay()
        else:
            pl["launches"]()
        self.info = {"graph": pl["graph"] is not None, "graph_requested": self.use_graph,
                     "graph_error": pl["graph_error"]}
        return (pl["u_pred"].double().cpu().numpy(), pl["x_pred"].double().cpu().numpy(),
                tuple(int(v) for v in pl["best"].cpu().numpy()), pl["idx"].cpu().numpy())

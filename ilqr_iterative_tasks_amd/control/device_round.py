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
    def __init__(self, device="cuda:0"):
        self.device = torch.device(device)
        self._solvers = {}
        self._ss_key = None

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

    def run(self, ctrl, laps):
        """Three rounds for controller `ctrl` over safe-set laps `laps`.  Returns host
        (u_pred[m,N], x_pred[n,N+1], best (lap position, candidate position), idx[L,k] of the
        last round)."""
        p = ctrl.ilqr_param
        cfg = config_from_params(p, ctrl.system_param, ctrl.num_horizon, ctrl.timestep,
                                 np.zeros(4), layout=_abi.LAYOUT_PROBLEM_MAJOR)
        solver = self._solver(cfg)
        self._upload_safe_set(ctrl, laps)
        L, k, n, m, N = len(laps), int(p.num_ss_points), cfg.n, cfg.m, cfg.N
        B = L * k
        dev, dt = self.device, solver.dtype
        buf = solver.alloc(B, want_gains=False)
        if ctrl.obstacle is not None:
            buf["obs"] = torch.as_tensor(np.tile(obstacle_record(ctrl.obstacle), (B, 1))).to(dev, dt)
        x0 = torch.as_tensor(np.asarray(ctrl.x, float)).to(dev, dt)
        idx = torch.zeros(L, k, dtype=torch.int32, device=dev)
        qf = torch.zeros(B, dtype=torch.int32, device=dev)
        cost_it = torch.zeros(B, dtype=dt, device=dev)
        best = torch.zeros(2, dtype=torch.int32, device=dev)
        x_pred = torch.zeros(n, N + 1, dtype=dt, device=dev)
        u_pred = torch.zeros(m, N, dtype=dt, device=dev)
        for it in range(3):  # the reference breaks its outer loop at iter == 2 (:472-478)
            if it == 0:
                guess, stride = x0, 1                      # self.x (:399)
            else:
                guess, stride = x_pred[:, N:], N + 1       # self.x_pred[:, -1] (:401)
            solver.select_candidates(self.ss, self.T, self.qfun, guess, stride, k, idx,
                                     buf["x_term"], qf)
            solver.init_candidates(x0, float(p.lamb), buf)
            solver.solve(buf)
            solver.relax_cost(buf["X"], buf["x_term"], qf, it, int(p.max_relax_iter), cost_it)
            solver.pick_best(L, k, cost_it, buf["X"], buf["U"], best, x_pred, u_pred)
        out = (u_pred.double().cpu().numpy(), x_pred.double().cpu().numpy(),
               tuple(int(v) for v in best.cpu().numpy()), idx.cpu().numpy())
        return out

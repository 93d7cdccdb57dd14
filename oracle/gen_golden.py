#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE ITSELF (build container only).

The reference (HybridRobotics/ilqr-iterative-tasks, mounted read-only at /root/reference) is
imported from where it lies with an empty `casadi` stub package on PYTHONPATH (casadi is imported
at module scope, control/ilqr_helper.py:3-4, but no casadi symbol is used by the iLQR path).
Nothing of the reference is copied: the fixtures hold plain input/output arrays only.

Usage (from the repo root):   python oracle/gen_golden.py [--out tests/golden]

Fixture families:
  G1  first- and third-iteration internals of sampled ilqr() calls (f_x, f_u, l_*, k, K,
      forward pass)
  G2  whole ilqr() calls sampled from the config-1 closed loop (stratified by iteration count
      and by lamb_in decade)
  G3  obstacle scenarios of iterative_ilqr/result/*.py applied to captured states
  G4  horizons N in {2, 6, 20, 50}
  G5  controller level: per calc_input round the k-NN indices, cost_it lists, pick and input u
  G6  closed-loop lap lengths (config 1 and the pytest configuration)
  G7  dynamics: data/closed_loop_feasible.txt (output of KineticBicycle.get_traj)
"""
import argparse
import os
import sys
import tempfile
from pathlib import Path

import numpy as np

REF = Path("/root/reference")


def _setup_reference():
    if not REF.exists():
        sys.exit("reference not mounted at /root/reference: golden vectors can only be generated "
                 "in the build container")
    sys.dont_write_bytecode = True
    os.environ["MPLBACKEND"] = "Agg"
    tmp = Path(tempfile.mkdtemp(prefix="i2lqr_golden_"))
    (tmp / "stubs" / "casadi").mkdir(parents=True)
    (tmp / "stubs" / "casadi" / "__init__.py").write_text("")
    (tmp / "run" / "data").mkdir(parents=True)
    for f in (REF / "data").glob("*.txt"):
        (tmp / "run" / "data" / f.name).write_bytes(f.read_bytes())
    sys.path[:0] = [str(tmp / "stubs"), str(REF / "iterative_ilqr")]
    os.chdir(tmp / "run")  # get_traj() writes data/closed_loop_feasible.txt (utils/base.py:133)
    return tmp


def obs_record(obstacle):
    if obstacle is None:
        return np.array([0.0, 0.0, 1.0, 1.0, 0.0, -1.0])
    spd = 0.0 if obstacle.spd is None else float(obstacle.spd)
    opt = 0.0 if obstacle.moving_option is None else float(obstacle.moving_option)
    return np.array([obstacle.x, obstacle.y, obstacle.width, obstacle.height, spd, opt], float)


class Capture:
    """Wraps the reference's ilqr / backward_pass / forward_pass to record calls."""

    def __init__(self, base, core):
        self.base, self.core = base, core
        self.calls = []          # one dict per ilqr() call
        self.first_iter = {}     # call index -> dict of first-iteration internals
        self._cur = None
        self._orig_ilqr = base.ilqr
        self._orig_bwd = core.backward_pass
        self._orig_fwd = core.forward_pass
        base.ilqr = self._ilqr
        core.backward_pass = self._bwd
        core.forward_pass = self._fwd

    def restore(self):
        self.base.ilqr = self._orig_ilqr
        self.core.backward_pass = self._orig_bwd
        self.core.forward_pass = self._orig_fwd

    def _ilqr(self, ilqr_param, num_horizon, xtarget, timestep, obstacle, system_param,
              x_terminal, dX, uvar, xvar, lamb):
        rec = dict(x0=xvar[:, 0].copy(), x_term=np.array(x_terminal, float).copy(),
                   lamb_in=float(lamb), obs=obs_record(obstacle), N=int(num_horizon), n_bwd=0)
        self._cur = rec
        u, x, lam = self._orig_ilqr(ilqr_param, num_horizon, xtarget, timestep, obstacle,
                                    system_param, x_terminal, dX, uvar, xvar, lamb)
        rec.update(U=np.array(u, float).copy(), X=np.array(x, float).copy(), lamb_out=float(lam),
                   iters=rec["n_bwd"])
        self._cur = None
        self.calls.append(rec)
        return u, x, lam

    def _bwd(self, xvar, uvar, x_terminal, dX, lamb, num_horizon, timestep, ilqr_param, obstacle,
             sys_param):
        k, K = self._orig_bwd(xvar, uvar, x_terminal, dX, lamb, num_horizon, timestep, ilqr_param,
                              obstacle, sys_param)
        rec = self._cur
        if rec is not None:
            rec["n_bwd"] += 1
            if rec["n_bwd"] in (1, 3):
                core = self.core
                f_x = core.get_A_matrix(xvar[2, 1:], xvar[3, 1:], uvar[0, :], num_horizon,
                                        timestep)
                f_u = core.get_B_matrix(xvar[3, 1:], num_horizon, timestep)
                l_u, l_uu, l_x, l_xx = core.get_cost_derivation(uvar, dX, ilqr_param, num_horizon,
                                                                xvar, obstacle, sys_param)
                V_x, V_xx = core.get_cost_final(xvar, x_terminal, ilqr_param.matrix_Qterminal,
                                                obstacle, ilqr_param)
                rec["_first" if rec["n_bwd"] == 1 else "_third"] = dict(
                    X=xvar.copy(), U=uvar.copy(), lamb=float(lamb),
                                     f_x=np.array(f_x, float), f_u=np.array(f_u, float),
                                     l_u=l_u, l_uu=l_uu, l_x=l_x, l_xx=l_xx, V_x=V_x, V_xx=V_xx,
                                     k=k.copy(), K=K.copy())
        return k, K

    def _fwd(self, xvar, uvar, x_terminal, ilqr_param, timestep, num_horizon, matrix_k, matrix_K,
             sys_param):
        out = self._orig_fwd(xvar, uvar, x_terminal, ilqr_param, timestep, num_horizon, matrix_k,
                             matrix_K, sys_param)
        rec = self._cur
        if rec is not None and rec["n_bwd"] in (1, 3):
            key = "_first" if rec["n_bwd"] == 1 else "_third"
            if key in rec and "X_new" not in rec[key]:
                rec[key].update(X_new=out[0].copy(), U_new=out[1].copy(), cost_new=float(out[2]))
        return out


def run_closed_loop(base, lap_number, num_ss_iter, num_ss_points, obstacle_args, pytest_style,
                    ctrl_log=None):
    """iterative_ilqr/tests/ilqr_test.py:8-75 (pytest_style=False) or tests/ilqr_test.py:9-56."""
    x0 = np.zeros(4) if pytest_style else [0, 0, 0, 0]
    ego = base.KineticBicycle(system_param=base.KineticBicycleParam())
    ego.set_state(x0)
    ego.set_timestep(1)
    ego.get_traj()
    ego.set_zero_noise()
    obstacle = base.Obstacle(*obstacle_args) if obstacle_args is not None else None
    param = base.iLqrParam(num_ss_points=num_ss_points, num_ss_iter=num_ss_iter, timestep=1,
                           num_horizon=6, all_ss_iter=False, all_ss_point=False)
    ctrl = base.iLqr(param, obstacle=obstacle, system_param=base.KineticBicycleParam())
    ctrl.add_trajectory(ego.xcl, ego.ucl)
    ctrl.set_timestep(1)
    if pytest_style:
        ctrl.set_state(x0)
    ego.set_ctrl_policy(ctrl)
    if ctrl_log is not None:
        orig_select = ctrl.select_close_ss
        orig_calc = ctrl.calc_input

        def select(it, xg):
            idx = orig_select(it, xg)
            ctrl_log["select"].append((int(it), np.array(xg, float).copy(), np.array(idx)))
            return idx

        def calc():
            xin = np.array(ctrl.x, float).copy()
            nh_in = int(ctrl.num_horizon)
            orig_calc()
            ctrl_log["steps"].append(dict(x=xin, u=np.array(ctrl.u, float).copy(), nh_in=nh_in,
                                          nh_out=int(ctrl.num_horizon)))

        ctrl.select_close_ss = select
        ctrl.calc_input = calc
    sim = base.Simulator()
    sim.set_robotic(ego)
    sim.set_timestep(1)
    sim.set_traj()
    for it in range(lap_number):
        sim.sim(it, sim_time=50)
        if pytest_style:
            ego.data["state"][-1] = np.vstack((ego.data["state"][-1], ego.xcl[-1, :]))
        else:
            ego.data["state"][-1][-1, :] = ego.xcl[-1, :]
        ctrl.add_trajectory(ego.data["state"][-1], ego.data["input"][-1])
    laps = [len(ego.xcl)] + [len(ts) for ts in ego.data["timestamp"]]
    return laps, ego, ctrl


def stratified_sample(calls, target, rng):
    """Indices stratified by iteration count and by lamb_in decade."""
    buckets = {}
    for i, c in enumerate(calls):
        it = c["iters"]
        ib = 0 if it == 1 else 1 if it == 2 else 2 if it <= 8 else 3 if it <= 16 else \
            4 if it <= 40 else 5 if it < 150 else 6
        lb = int(np.clip(np.floor(np.log10(c["lamb_in"])), -30, 5))
        buckets.setdefault((ib, lb), []).append(i)
    picked = []
    keys = sorted(buckets)
    while len(picked) < target and keys:
        for key in list(keys):
            lst = buckets[key]
            if not lst:
                keys.remove(key)
                continue
            picked.append(lst.pop(rng.integers(len(lst))))
            if len(picked) >= target:
                break
    return sorted(picked)


def pack_calls(calls):
    return dict(
        x0=np.stack([c["x0"] for c in calls]), x_term=np.stack([c["x_term"] for c in calls]),
        lamb_in=np.array([c["lamb_in"] for c in calls]), obs=np.stack([c["obs"] for c in calls]),
        U=np.stack([c["U"] for c in calls]), X=np.stack([c["X"] for c in calls]),
        lamb_out=np.array([c["lamb_out"] for c in calls]),
        iters=np.array([c["iters"] for c in calls], np.int32))


def direct_ilqr(base, x0, x_term, lamb, obstacle, N, cap):
    """Call the reference ilqr() the way iLqr.calc_input does (utils/base.py:405-426)."""
    param = base.iLqrParam(num_horizon=N, timestep=1)
    uvar = np.zeros((2, N))
    xvar = np.zeros((4, N + 1))
    xvar[:, 0] = x0
    dX = np.zeros((4, N + 1))
    dX[:, 0] = xvar[:, 0]
    n_before = len(cap.calls)
    base.ilqr(param, N, np.array([0, 0, 0, 0]), 1, obstacle, base.KineticBicycleParam(),
              np.array(x_term, float), dX, uvar, xvar, lamb)
    return cap.calls[n_before]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=str(Path(__file__).resolve().parent.parent / "tests" /
                                         "golden"))
    ap.add_argument("--skip-pytest-config", action="store_true")
    args = ap.parse_args()
    out = Path(args.out)
    out.mkdir(parents=True, exist_ok=True)
    _setup_reference()

    import builtins
    real_print = builtins.print
    builtins.print = lambda *a, **k: None  # the reference prints in its hot loop
    from utils import base
    import control.iterative_ilqr as core

    rng = np.random.default_rng(20230228)
    cap = Capture(base, core)

    # ---- config 1: --lap-number 3 --num-ss-iters 2 --num-ss-points 8, obstacle (31,-3,8,6) ----
    log = dict(select=[], steps=[])
    laps1, ego1, ctrl1 = run_closed_loop(base, 3, 2, 8, (31, -3, 8, 6), False, ctrl_log=log)
    calls1 = list(cap.calls)
    real_print("config 1 laps", laps1, "ilqr calls", len(calls1), "iterations",
               sum(c["iters"] for c in calls1))

    # G2: sampled whole-ilqr calls
    idx = stratified_sample(calls1, 384, rng)
    g2 = pack_calls([calls1[i] for i in idx])
    g2["call_index"] = np.array(idx, np.int32)
    np.savez_compressed(out / "g2_ilqr_calls.npz", **g2)

    # G1: first-iteration internals for a subset of those
    g1_idx = idx[:: max(1, len(idx) // 96)][:96]
    firsts, src, which = [], [], []
    for i in g1_idx:
        for key, itn in (("_first", 1), ("_third", 3)):
            if key in calls1[i] and "X_new" in calls1[i][key]:
                firsts.append(calls1[i][key])
                src.append(i)
                which.append(itn)
    g1 = {key: np.stack([np.asarray(f[key], float) for f in firsts]) for key in firsts[0]}
    g1["x_term"] = np.stack([calls1[i]["x_term"] for i in src])
    g1["obs"] = np.stack([calls1[i]["obs"] for i in src])
    g1["iteration"] = np.array(which, np.int32)
    np.savez_compressed(out / "g1_first_iteration.npz", **g1)

    # G5: controller-level log of config 1 (every calc_input)
    steps = log["steps"]
    sel = log["select"]
    # calls are made in order: per calc_input, per round, per lap id, per candidate j
    np.savez_compressed(
        out / "g5_controller_config1.npz",
        step_x=np.stack([s["x"] for s in steps]), step_u=np.stack([s["u"] for s in steps]),
        step_nh_in=np.array([s["nh_in"] for s in steps], np.int32),
        step_nh_out=np.array([s["nh_out"] for s in steps], np.int32),
        select_id=np.array([s[0] for s in sel], np.int32),
        select_xguess=np.stack([s[1] for s in sel]),
        select_idx=np.stack([s[2] for s in sel]).astype(np.int32),
        # every ilqr call of the run, in call order (inputs + final state/lamb, compact)
        call_x0=np.stack([c["x0"] for c in calls1]),
        call_x_term=np.stack([c["x_term"] for c in calls1]),
        call_lamb_in=np.array([c["lamb_in"] for c in calls1]),
        call_lamb_out=np.array([c["lamb_out"] for c in calls1]),
        call_iters=np.array([c["iters"] for c in calls1], np.int32),
        call_xN=np.stack([c["X"][:, -1] for c in calls1]),
        call_u0=np.stack([c["U"][:, 0] for c in calls1]),
        laps=np.array(laps1, np.int32),
        lap_states=np.concatenate([np.asarray(s, float) for s in ego1.data["state"]]),
        lap_inputs=np.concatenate([np.asarray(s, float) for s in ego1.data["input"]]),
    )

    # ---- G3: obstacle scenarios on captured states ------------------------------------------
    scen = {
        "none": None,
        "static_31_m3": (31, -3, 8, 6),
        "static_100_m5": (100, -5, 20, 40),
        "static_35_0": (35, 0, 30, 30),
        "moving_up": (35, -16, 34, 34, 1, 1, 1),      # spd 1, timestep 1, option 1
        "moving_left": (50, -1, 35, 35, 0.2, 1, 2),   # spd 0.2, option 2
    }
    pick = [calls1[i] for i in stratified_sample(calls1, 40, rng)]
    g3_calls, g3_names = [], []
    for name, oargs in scen.items():
        for c in pick:
            obstacle = None if oargs is None else base.Obstacle(*oargs)
            rec = direct_ilqr(base, c["x0"], c["x_term"], 1.0, obstacle, 6, cap)
            g3_calls.append(rec)
            g3_names.append(name)
    g3 = pack_calls(g3_calls)
    g3["scenario"] = np.array(g3_names)
    np.savez_compressed(out / "g3_scenarios.npz", **g3)

    # ---- G4: horizons -----------------------------------------------------------------------
    traj = np.loadtxt(REF / "data" / "closed_loop_feasible.txt")
    for N in (2, 6, 20, 50):
        recs = []
        for s in range(0, 100, 4):
            x0 = traj[s]
            x_term = traj[min(s + N, 120)]
            recs.append(direct_ilqr(base, x0, x_term, 1.0, base.Obstacle(31, -3, 8, 6), N, cap))
        g4 = pack_calls(recs)
        firsts = [r["_first"] for r in recs]
        for key in ("k", "K", "X_new", "U_new", "cost_new"):
            g4["first_" + key] = np.stack([np.asarray(f[key], float) for f in firsts])
        g4["first_X"] = np.stack([f["X"] for f in firsts])
        g4["first_U"] = np.stack([f["U"] for f in firsts])
        np.savez_compressed(out / f"g4_horizon_N{N}.npz", **g4)

    # ---- G6: pytest configuration (tests/ilqr_test.py): 5 laps, num_ss_iter=1, obstacle y=-2 --
    laps_py = None
    if not args.skip_pytest_config:
        n0 = len(cap.calls)
        laps_py, ego_py, _ = run_closed_loop(base, 5, 1, 8, (31, -2, 8, 6), True)
        real_print("pytest config laps", laps_py, "ilqr calls", len(cap.calls) - n0)
    np.savez_compressed(out / "g6_closed_loop.npz", laps_config1=np.array(laps1, np.int32),
                        laps_pytest=np.array(laps_py if laps_py else [], np.int32))

    # ---- G7: dynamics known answer: the file get_traj() regenerates (utils/base.py:103-138) ---
    regenerated = np.loadtxt("data/closed_loop_feasible.txt")
    assert np.array_equal(regenerated, traj), "get_traj() no longer reproduces the data file"
    np.savez_compressed(out / "g7_dynamics.npz", closed_loop_feasible=traj,
                        ucl=np.asarray(ego1.ucl, float))
    cap.restore()
    builtins.print = real_print
    total = sum(f.stat().st_size for f in out.glob("*.npz"))
    print("wrote", sorted(p.name for p in out.glob("*.npz")), f"{total / 1024:.0f} KiB")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Golden lap lengths of the reference's paper scenarios (iterative_ilqr/result/ilqr_test_*.py),
captured by running the reference itself (build container only; see oracle/gen_golden.py for how
the reference is imported).  Writes tests/golden/g8_scenarios_closed_loop.npz.

Scenarios (all: num_ss_iters 2, num_ss_points 8, N = 6, dt = 1, zero noise):
  no_obstacle          result/ilqr_test_no_obstacle.py, 6 laps
  static_obstacle_big  result/ilqr_test_static_obstacle.py: (100,-5,20,40), 6 laps
  add_static_obstacle  result/ilqr_test_add_static_obstacle.py: (35,0,30,30) from lap 5, 7 laps
  moving_up            result/ilqr_test_add_moving_obstacle.py --moving-option up, 7 laps
  moving_left          ... --moving-option left, 7 laps
"""
import builtins
import sys
from copy import deepcopy
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent))
from gen_golden import _setup_reference  # noqa: E402


def run(base, laps, obstacle0, on_lap):
    ego = base.KineticBicycle(system_param=base.KineticBicycleParam())
    ego.set_state([0, 0, 0, 0])
    ego.set_timestep(1)
    ego.get_traj()
    ego.set_zero_noise()
    param = base.iLqrParam(num_ss_points=8, num_ss_iter=2, timestep=1, num_horizon=6)
    ctrl = base.iLqr(param, obstacle=obstacle0, system_param=base.KineticBicycleParam())
    ctrl.add_trajectory(ego.xcl, ego.ucl)
    ctrl.set_timestep(1)
    ego.set_ctrl_policy(ctrl)
    sim = base.Simulator()
    sim.set_robotic(ego)
    sim.set_timestep(1)
    sim.set_traj()
    for it in range(laps):
        on_lap(it, ctrl)
        sim.sim(it, sim_time=50)
        st = deepcopy(ego.data["state"][-1])
        st[-1, :] = ego.xcl[-1, :]
        ctrl.add_trajectory(st, ego.data["input"][-1])
    lens = [len(ego.xcl)] + [len(ts) for ts in ego.data["timestamp"]]
    return lens, np.asarray(ego.data["state"][-1], float), np.asarray(ego.data["input"][-1], float)


def main():
    out = Path(__file__).resolve().parent.parent / "tests" / "golden"
    _setup_reference()
    real_print = builtins.print
    builtins.print = lambda *a, **k: None
    from utils import base

    def none(it, c):
        pass

    def add_static(it, c):
        if it == 5:
            c.obstacle = base.Obstacle(35, 0, 30, 30)
        if it == 6:
            c.obstacle = None

    def moving(args):
        def f(it, c):
            if it == 5:
                c.obstacle = base.Obstacle(*args)
            if it == 6:
                c.obstacle = None
        return f

    res = {}
    for name, laps, ob0, hook in (
            ("no_obstacle", 6, None, none),
            ("static_obstacle_big", 6, base.Obstacle(100, -5, 20, 40), none),
            ("add_static_obstacle", 7, None, add_static),
            ("moving_up", 7, None, moving((35, -16, 34, 34, 1, 1, 1))),
            ("moving_left", 7, None, moving((50, -1, 35, 35, 0.2, 1, 2)))):
        lens, last_x, last_u = run(base, laps, ob0, hook)
        real_print(name, lens)
        res[name + "_laps"] = np.array(lens, np.int32)
        res[name + "_last_state"] = last_x
        res[name + "_last_input"] = last_u
    builtins.print = real_print
    np.savez_compressed(out / "g8_scenarios_closed_loop.npz", **res)
    print("wrote g8_scenarios_closed_loop.npz")


if __name__ == "__main__":
    main()

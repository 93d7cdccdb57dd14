"""Lap driver for closed-loop runs of the i2LQR controller on the HIP path.

Host plumbing that runs once per control step — not batch work.  It plays the part of the
reference's plant and simulator objects (utils/base.py: KineticBicycle :67-213, Simulator :693-726)
for the controller in control/controller.py, and keeps the handful of names the reference's
scripts touch (`KineticBicycle`, `Simulator`, `xcl` / `ucl`, `data`, `diagnostics`), but it is this
repository's own, smaller design: a lap is a `_LapLog` of Python lists turned into arrays once at
the end of the lap, the first lap's input schedule is built as one array, and the goal test and
the obstacle bookkeeping live in one function (`drive_lap`).  Plotting, pickling and the
`--direct-ilqr` warm start of the reference are not part of the path and are not provided.
"""
from __future__ import annotations

import time
from dataclasses import dataclass, field

import numpy as np

from .control.controller import plant_step
from .control.params import U_DIM, X_DIM

GOAL_RADIUS = 0.8    # a lap ends within this distance of the first lap's last state (utils/base.py:718)
LAP_SECONDS = 121    # ... or after this much simulated time (:709)
FIRST_LAP_SECONDS = 120


@dataclass
class _LapLog:
    """What one lap leaves behind; row i belongs to control step i (states has one row more)."""
    states: list = field(default_factory=list)
    inputs: list = field(default_factory=list)
    stamps: list = field(default_factory=list)
    solve_seconds: list = field(default_factory=list)
    feasible: list = field(default_factory=list)


def first_lap_inputs(dt: float) -> np.ndarray:
    """Open-loop schedule of the initial feasible lap, u[T, 2] with T = 120 / dt (the windows of
    utils/base.py:108-128): accelerate at 1 for the first second, brake at -1 between 4 s and 3 s
    before the end, steer +pi/6 during the first second (not at i = 0) and -pi/6 between 2 s and
    1 s before half time."""
    T = int(FIRST_LAP_SECONDS / dt)
    i = np.arange(T)
    u = np.zeros((T, U_DIM))
    u[i <= 1 / dt, 0] = 1.0
    u[(i >= T - 4 / dt) & (i <= T - 3 / dt), 0] = -1.0
    u[(i > 0) & (i <= 1 / dt), 1] = np.pi / 6
    u[(i >= T / 2 - 2 / dt) & (i <= T / 2 - 1 / dt), 1] = -np.pi / 6
    return u


class KineticBicycle:
    """The simulated vehicle: holds the true state, asks its controller for an input once per
    step and integrates the kinematic bicycle (systems/kinetic_bicycle.py:10-27)."""

    def __init__(self, direct_ctrl_policy=False, system_param=None, seed=None):
        self.system_param = system_param
        self.direct_ctrl_policy = direct_ctrl_policy
        self.timestep = None
        self.time = 0.0
        self.x = None
        self.u = None
        self.ctrl_policy = None
        self.xcl = None   # first lap, states [T+1, n]
        self.ucl = None   # first lap, inputs [T, m]
        self._noisy = True
        self._rng = np.random.default_rng(seed)
        self._lap = _LapLog()
        # finished laps, one array per lap (same keys as the reference's dictionaries)
        self.data = {"state": [], "input": [], "timestamp": []}
        self.diagnostics = {"solver_time": [], "feasibility": []}

    # -- configuration ----------------------------------------------------------------------
    def set_zero_noise(self):
        self._noisy = False

    def set_timestep(self, dt):
        self.timestep = dt

    def set_ctrl_policy(self, ctrl_policy):
        self.ctrl_policy = ctrl_policy

    def set_state(self, x):
        """Place the vehicle and start a fresh lap log from there."""
        self.x = np.array(x, float)
        self._lap = _LapLog(states=[self.x.copy()])

    # -- first lap ----------------------------------------------------------------------------
    def get_traj(self):
        """Roll the open-loop schedule out from the origin: fills `xcl`, `ucl`.  (The reference
        also writes data/closed_loop_feasible.txt, utils/base.py:133; its own input log aliases
        row 0, :132 — `ucl` here holds the inputs that were applied.)"""
        self.ucl = first_lap_inputs(self.timestep)
        xs = [np.zeros(X_DIM)]
        for u in self.ucl:
            xs.append(plant_step(xs[-1], u, self.timestep))
        self.xcl = np.array(xs)

    # -- one control step -----------------------------------------------------------------------
    def forward_one_step(self):
        """controller -> plant -> log.  A RuntimeError from the controller (C-ABI error codes
        surface as I2lqrError, a RuntimeError) marks the step infeasible and the previous input
        is applied again, as in utils/base.py:146-155."""
        ctrl, log = self.ctrl_policy, self._lap
        ctrl.set_state(self.x)
        t0 = time.perf_counter()
        try:
            ctrl.calc_input()
            self.u = ctrl.get_input()
            ok = 1
        except RuntimeError:
            ok = 0
        log.solve_seconds.append(time.perf_counter() - t0)
        log.feasible.append(ok)
        self.x = self._plant(self.x, self.u)
        self.time += self.timestep
        ctrl.set_state(self.x)
        log.states.append(self.x.copy())
        log.inputs.append(np.array(self.u, float))
        log.stamps.append(self.time)

    def _plant(self, x, u):
        xn = plant_step(x, u, self.timestep)
        if self._noisy:  # bounded actuation noise on v and theta (utils/base.py:207-211)
            dv, dth = np.clip(self._rng.normal(0.0, [0.01, 0.005]), -0.05, 0.05)
            xn[2] += 0.5 * dv
            xn[3] += 0.5 * dth
        return xn

    def finish_lap(self):
        """Archive the current lap log and put the vehicle back on the start line."""
        log = self._lap
        self.data["state"].append(np.array(log.states))
        self.data["input"].append(np.array(log.inputs).reshape(-1, U_DIM))
        self.data["timestamp"].append(np.array(log.stamps))
        self.diagnostics["solver_time"].append(np.array(log.solve_seconds))
        self.diagnostics["feasibility"].append(np.array(log.feasible))
        self.set_state(np.zeros(X_DIM))


def drive_lap(ego: KineticBicycle, goal: np.ndarray) -> bool:
    """One lap (utils/base.py:708-726): step until the vehicle is within GOAL_RADIUS of `goal` or
    the lap time is up; a moving obstacle advances once per step and returns to its start
    afterwards.  True if the goal was reached."""
    reached = False
    for _ in range(int(LAP_SECONDS / ego.timestep)):
        ego.forward_one_step()
        obstacle = ego.ctrl_policy.obstacle
        if obstacle is not None:
            obstacle.update_obstacle()
        if np.linalg.norm(ego.x - goal) <= GOAL_RADIUS:
            reached = True
            break
    ego.finish_lap()
    obstacle = ego.ctrl_policy.obstacle
    if obstacle is not None:
        obstacle.reset_obstacle()
    return reached


class Simulator:
    """The reference scripts' simulator surface (set_robotic / set_timestep / set_traj / sim) over
    drive_lap()."""

    def __init__(self):
        self.robotic = None
        self.timestep = None
        self.initial_traj = None

    def set_robotic(self, robotic):
        self.robotic = robotic

    def set_timestep(self, dt):
        self.timestep = dt

    def set_traj(self):
        self.initial_traj = self.robotic.xcl

    def sim(self, iter=None, sim_time=None):
        return drive_lap(self.robotic, self.initial_traj[-1])


def run_laps(ego, controller, lap_number, pytest_style=False, on_lap=None):
    """`lap_number` closed-loop laps, the finished lap joining the controller's safe set each time
    (iterative_ilqr/tests/ilqr_test.py:56-60).  Before that its last state is replaced by the goal
    state — or, with pytest_style (the reference's tests/ilqr_test.py:48-52), the goal state is
    appended.  `on_lap(lap, controller)` runs before each lap (scenario scripts swap the obstacle
    there).  Returns the lap lengths in steps, first lap included."""
    goal = ego.xcl[-1]
    for lap in range(lap_number):
        if on_lap is not None:
            on_lap(lap, controller)
        drive_lap(ego, goal)
        states = ego.data["state"][-1]
        if pytest_style:
            states = np.vstack((states, goal))
        else:
            states[-1] = goal
        ego.data["state"][-1] = states
        controller.add_trajectory(states, ego.data["input"][-1])
    return [len(ego.xcl)] + [len(ts) for ts in ego.data["timestamp"]]

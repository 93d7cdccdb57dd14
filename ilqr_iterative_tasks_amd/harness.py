"""Closed-loop harness — host-side counterpart of the reference's plant and simulator
(utils/base.py: KineticBicycle :67-213, Simulator :693-726; plotting / pickling omitted).  It runs
once per control step and is host plumbing, not batch work."""
from __future__ import annotations

import datetime

import numpy as np

from .control.controller import _plant_step
from .control.params import X_DIM, U_DIM, X_ID, U_ID


class KineticBicycle:
    def __init__(self, direct_ctrl_policy=False, system_param=None):
        self.system_param = system_param
        self.direct_ctrl_policy = direct_ctrl_policy
        self.time = 0.0
        self.delta_timer = None
        self.feasible = None
        self.timestep = None
        self.x = None
        self.u = None
        self.zero_noise_flag = False
        self.states, self.inputs, self.timestamps = None, None, None
        self.solver_times, self.feasibility = None, None
        self.data = {"state": [], "input": [], "timestamp": []}
        self.diagnostics = {"solver_time": [], "feasibility": []}
        self.ctrl_policy = None

    def set_zero_noise(self):
        self.zero_noise_flag = True

    def set_timestep(self, dt):
        self.timestep = dt

    def set_state(self, x):
        self.x = x
        self.states = x
        self.timestamps = None
        self.inputs = None
        self.solver_times = None
        self.feasible = None

    def get_traj(self):
        """Open-loop first lap: utils/base.py:103-138 (the data file is not written).  The
        reference's input log aliases its first row (`ucl = u`, :132); the inputs recorded here
        are the ones actually applied."""
        angle = np.pi / 6
        total = int(120 / self.timestep)
        xcl = np.zeros((1, X_DIM))
        ucl = []
        for i in range(total):
            u = np.zeros(U_DIM)
            if i <= 1 / self.timestep:
                u[U_ID["accel"]] = 1
            elif total - 4 / self.timestep <= i <= total - 3 / self.timestep:
                u[U_ID["accel"]] = -1
            if 0 < i <= 1 / self.timestep:
                u[U_ID["delta"]] = angle
            elif total / 2 - 2 / self.timestep <= i <= total / 2 - 1 / self.timestep:
                u[U_ID["delta"]] = -angle
            xcl = np.vstack((xcl, _plant_step(xcl[-1], u, self.timestep)))
            ucl.append(u)
        self.xcl = xcl
        self.ucl = np.array(ucl)

    def set_ctrl_policy(self, ctrl_policy):
        self.ctrl_policy = ctrl_policy

    def calc_ctrl_input(self):
        self.ctrl_policy.set_state(self.x)
        start = datetime.datetime.now()
        try:
            self.ctrl_policy.calc_input()
            self.u = self.ctrl_policy.get_input()
            self.delta_timer = (datetime.datetime.now() - start).total_seconds()
            self.feasible = 1
        except RuntimeError:  # utils/base.py:153-155
            self.feasible = 0

    def forward_one_step(self):
        self.calc_ctrl_input()
        self.forward_dynamics()
        self.ctrl_policy.set_state(self.x)
        self.update_memory()

    def update_memory(self):
        self.states = np.vstack((self.states, self.x))
        self.inputs = self.u if self.inputs is None else np.vstack((self.inputs, self.u))
        self.timestamps = self.time if self.timestamps is None else np.vstack(
            (self.timestamps, self.time))
        self.solver_times = self.delta_timer if self.solver_times is None else np.vstack(
            (self.solver_times, self.delta_timer))
        self.feasibility = self.feasible if self.feasibility is None else np.vstack(
            (self.feasibility, self.feasible))

    def update_memory_post_iter(self):
        self.data["state"].append(self.states)
        self.data["input"].append(self.inputs)
        self.data["timestamp"].append(self.timestamps)
        self.diagnostics["solver_time"].append(self.solver_times)
        self.diagnostics["feasibility"].append(self.feasibility)
        self.set_state(np.zeros((X_DIM,)))

    def forward_dynamics(self):
        x_next = _plant_step(np.asarray(self.x, float), self.u, self.timestep)
        if not self.zero_noise_flag:  # utils/base.py:207-211
            noise_v = np.maximum(-0.05, np.minimum(np.random.randn() * 0.01, 0.05))
            noise_theta = np.maximum(-0.05, np.minimum(np.random.randn() * 0.005, 0.05))
            x_next[X_ID["v"]] = x_next[X_ID["v"]] + 0.5 * noise_v
            x_next[X_ID["theta"]] = x_next[X_ID["theta"]] + 0.5 * noise_theta
        self.x = x_next
        self.time += self.timestep


class Simulator:
    def __init__(self):
        self.initial_traj = None
        self.robotic = None
        self.timestep = None

    def set_timestep(self, dt):
        self.timestep = dt

    def set_robotic(self, robotic):
        self.robotic = robotic

    def set_traj(self):
        self.initial_traj = self.robotic.xcl

    def sim(self, iter, sim_time=121.0):
        """One lap: utils/base.py:708-726 (sim_time is overridden to 121 there too)."""
        sim_time = 121
        steps = int(sim_time / self.timestep)
        for i in range(steps):
            self.robotic.forward_one_step()
            obstacle = self.robotic.ctrl_policy.obstacle
            if obstacle is not None:
                obstacle.update_obstacle()
            if np.linalg.norm(self.robotic.x - self.initial_traj[-1, :]) <= 0.8:
                self.robotic.update_memory_post_iter()
                if obstacle is not None:
                    obstacle.reset_obstacle()
                return True
            if i == steps - 1:
                self.robotic.update_memory_post_iter()
                if obstacle is not None:
                    obstacle.reset_obstacle()
        return False


def run_laps(ego, controller, lap_number, pytest_style=False, on_lap=None):
    """The lap loop of iterative_ilqr/tests/ilqr_test.py:56-60 (or tests/ilqr_test.py:48-52 with
    pytest_style=True).  Returns the lap lengths [len(first lap), len(lap 1), ...]."""
    sim = Simulator()
    sim.set_robotic(ego)
    sim.set_timestep(ego.timestep)
    sim.set_traj()
    for it in range(lap_number):
        if on_lap is not None:
            on_lap(it, controller)
        sim.sim(it)
        if pytest_style:
            ego.data["state"][-1] = np.vstack((ego.data["state"][-1], ego.xcl[-1, :]))
        else:
            ego.data["state"][-1][-1, :] = ego.xcl[-1, :]
        controller.add_trajectory(ego.data["state"][-1], ego.data["input"][-1])
    return [len(ego.xcl)] + [len(ts) for ts in ego.data["timestamp"]]

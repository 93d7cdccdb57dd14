"""Parameter bundles of the path, same names / defaults / attribute names as the reference:
KineticBicycleParam (utils/base.py:15-20), Obstacle data fields and update/reset
(utils/base.py:23-64, plotting omitted), iLqrParam (utils/base.py:242-302)."""
from __future__ import annotations

import numpy as np

from .. import _abi
from .._abi import I2lqrConfig

X_DIM, U_DIM = 4, 2  # utils/constants_kinetic_bicycle.py:2
X_ID = {"x": 0, "y": 1, "v": 2, "theta": 3}
U_ID = {"accel": 0, "delta": 1}


class KineticBicycleParam:
    def __init__(self, delta_max=np.pi / 2, a_max=2.0, v_max=10, v_min=0):
        self.delta_max = delta_max
        self.a_max = a_max
        self.v_max = v_max
        self.v_min = v_min


class Obstacle:
    """Elliptical obstacle (x, y, width, height), optionally moving: option 1 = up (+y),
    option 2 = left (-x), `spd` per simulator step (utils/base.py:50-58)."""

    def __init__(self, x, y, width, height, spd=None, timestep=None, moving_option=None):
        self.x0 = self.x = x
        self.y0 = self.y = y
        self.width = width
        self.height = height
        self.spd = spd
        self.timestep = timestep
        self.data = {"state": []}
        self.states = np.array([self.x0, self.y0])
        self.moving_option = moving_option

    def update_obstacle(self):
        if self.spd is not None:
            if self.moving_option == 1:
                self.y += self.spd * self.timestep
            if self.moving_option == 2:
                self.x -= self.spd * self.timestep
        self.states = np.vstack((self.states, [self.x, self.y]))

    def reset_obstacle(self):
        self.x = self.x0
        self.y = self.y0
        self.data["state"].append(self.states)
        self.states = np.array([self.x0, self.y0])


class iLqrParam:
    def __init__(self, matrix_Q=0 * np.diag([0.0, 0.0, 0.0, 0.0]),
                 matrix_R=0 * np.diag([0.05, 0.05]),
                 matrix_Qterminal=2 * np.diag([1.0, 1.0, 20.0, 0.02]), num_ss_points=8,
                 num_ss_iter=1, num_horizon=6, tuning_state_q1=1.0, tuning_state_q2=1.0,
                 tuning_ctrl_q1=1.0, tuning_ctrl_q2=1.0, tuning_obs_q1=2.74, tuning_obs_q2=2.74,
                 safety_margin=0.0, max_ilqr_iter=150, eps=1e-2, lamb=1, lamb_factor=10,
                 max_lamb=1000, reach_error=1.0, max_relax_iter=55, max_outloop_iter=50,
                 timestep=None, lap_number=None, time_ilqr=None, ss_option=None,
                 all_ss_point=False, all_ss_iter=False):
        self.matrix_Q = matrix_Q
        self.matrix_R = matrix_R
        self.matrix_Qterminal = matrix_Qterminal
        self.num_ss_points = num_ss_points
        self.num_ss_iter = num_ss_iter
        self.num_horizon = num_horizon
        self.timestep = timestep
        self.lap_number = lap_number
        self.time_ilqr = time_ilqr
        self.ss_option = ss_option
        self.all_ss_point = all_ss_point
        self.all_ss_iter = all_ss_iter
        self.tuning_state_q1 = tuning_state_q1
        self.tuning_state_q2 = tuning_state_q2
        self.tuning_ctrl_q1 = tuning_ctrl_q1
        self.tuning_ctrl_q2 = tuning_ctrl_q2
        self.tuning_obs_q1 = tuning_obs_q1
        self.tuning_obs_q2 = tuning_obs_q2
        self.safety_margin = safety_margin
        self.max_ilqr_iter = max_ilqr_iter
        self.eps = eps
        self.lamb = lamb
        self.lamb_factor = lamb_factor
        self.max_lamb = max_lamb
        self.reach_error = reach_error
        self.max_relax_iter = max_relax_iter
        self.max_outloop_iter = max_outloop_iter


def obstacle_record(obstacle) -> np.ndarray:
    """Reference Obstacle (or None) -> the 6-word obs record of include/i2lqr.h.  The reference
    raises NameError for spd != 0 with moving_option None (control/ilqr_helper.py:34-43); that
    combination is rejected here instead."""
    if obstacle is None:
        return np.array([0.0, 0.0, 1.0, 1.0, 0.0, -1.0])
    spd = 0.0 if obstacle.spd is None else float(obstacle.spd)
    if obstacle.moving_option is None and spd != 0.0:
        raise ValueError("obstacle with spd != 0 needs moving_option 1 (up) or 2 (left)")
    opt = 0.0 if obstacle.moving_option is None else float(obstacle.moving_option)
    return np.array([obstacle.x, obstacle.y, obstacle.width, obstacle.height, spd, opt], float)


def config_from_params(ilqr_param: iLqrParam, system_param: KineticBicycleParam, num_horizon: int,
                       timestep: float, xtarget=None, dtype="f64",
                       layout=_abi.LAYOUT_PROBLEM_MAJOR) -> I2lqrConfig:
    """(iLqrParam, KineticBicycleParam, N, dt, xtarget) -> i2lqr_config for the reference plant."""
    cfg = _abi.default_config("bicycle4", num_horizon, dtype, dt=float(timestep), layout=layout)
    cfg.max_iter = int(ilqr_param.max_ilqr_iter)
    cfg.eps = float(ilqr_param.eps)
    cfg.lamb_factor = float(ilqr_param.lamb_factor)
    cfg.max_lamb = float(ilqr_param.max_lamb)
    cfg.ctrl_q1, cfg.ctrl_q2 = float(ilqr_param.tuning_ctrl_q1), float(ilqr_param.tuning_ctrl_q2)
    cfg.obs_q1, cfg.obs_q2 = float(ilqr_param.tuning_obs_q1), float(ilqr_param.tuning_obs_q2)
    cfg.safety_margin = float(ilqr_param.safety_margin)
    # clipped / barriered at round(delta_max, 2): control/iterative_ilqr.py:38-39
    cfg.u_max[:] = [float(system_param.a_max), round(float(system_param.delta_max), 2), 0.0, 0.0]
    cfg.set_matrix("Q", ilqr_param.matrix_Q)
    cfg.set_matrix("R", ilqr_param.matrix_R)
    cfg.set_matrix("Qt", ilqr_param.matrix_Qterminal)
    xt = np.zeros(_abi.MAX_N)
    if xtarget is not None:
        xt[:X_DIM] = np.asarray(xtarget, float).ravel()
    cfg.xtarget[:] = xt.tolist()
    return cfg

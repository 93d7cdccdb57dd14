"""Host-side mirror of the reference's controller interface for the iLQR path
(iterative_ilqr/utils/base.py: ControlBase, iLqrParam, iLqr, KineticBicycleParam, Obstacle) and of
the per-candidate seam `ilqr()` (iterative_ilqr/control/iterative_ilqr.py:7)."""
from .params import KineticBicycleParam, Obstacle, iLqrParam, config_from_params, obstacle_record
from .controller import ControlBase, iLqr
from .iterative_ilqr import ilqr

__all__ = ["KineticBicycleParam", "Obstacle", "iLqrParam", "ControlBase", "iLqr", "ilqr",
           "config_from_params", "obstacle_record"]

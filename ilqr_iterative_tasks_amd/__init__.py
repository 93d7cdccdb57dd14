"""MI355X-native batched iLQR solver behind the i2LQR controller of
HybridRobotics/ilqr-iterative-tasks.

Layout of the package (only what the hot path needs):
  csrc/        hand-written HIP kernels for gfx950 + the C-ABI (include/i2lqr.h)
  _abi.py      ctypes mirror of the C-ABI, library loader (no CPU fallback)
  solver.py    BatchedILQR: torch device tensors -> C-ABI
  control/     host-side mirror of the reference's controller interface
  workloads.py synthetic batches of SURVEY.md §8(d) / BASELINE.json configs
  dist.py      batch sharding + the one RCCL all-gather of terminal costs
"""
from ._abi import (I2lqrConfig, default_config, load_library, F32, F64, SYS_BICYCLE4,
                   SYS_BICYCLE6, SYS_QUAD12, STATUS_NAMES)

__all__ = ["I2lqrConfig", "default_config", "load_library", "BatchedILQR", "F32", "F64",
           "SYS_BICYCLE4", "SYS_BICYCLE6", "SYS_QUAD12", "STATUS_NAMES"]


def __getattr__(name):
    # torch is imported lazily so the ABI mirror can be used without it
    if name in ("BatchedILQR", "I2lqrError"):
        from . import solver
        return getattr(solver, name)
    raise AttributeError(name)

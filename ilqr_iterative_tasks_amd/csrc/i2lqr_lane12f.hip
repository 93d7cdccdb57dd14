// The fp32 instantiations of quad12's one-problem-per-lane kernels (round 5; Q = R = 0): a
// translation unit of their own so that the library's large units compile side by side.  They run
// the GENERAL forms of the passes (library sin / cos, two-exponential barrier, Jacobi sweeps where
// Quu is not positive definite): the branch-free hot forms of LaneWorker::backward_blocked are built
// around fp64 literals in scalar registers.
#define I2LQR_LANE12F_DEFINE
#include "i2lqr_lane12.h"

namespace i2lqr {
I2LQR_LANE12F_KERNELS(template __global__)
}  // namespace i2lqr

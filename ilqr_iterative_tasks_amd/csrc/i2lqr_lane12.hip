// One-problem-per-lane kernels of quad12 (n = 12, m = 4, BASELINE.json configs[4]): explicit
// instantiations of the templates of i2lqr_lane.hpp (LaneWorker::backward_blocked), launched from
// i2lqr_abi.hip.
#define I2LQR_LANE12_DEFINE
#include "i2lqr_lane12.h"

namespace i2lqr {
I2LQR_LANE12_KERNELS(template __global__)
}  // namespace i2lqr

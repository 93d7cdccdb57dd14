// k_lane_iterate_pair: the bicycles' one-problem-per-lane kernel with a helper wavefront (round 5;
// i2lqr_lane.hpp), fp64 / fp32, with / without stage weights - a translation unit of its own so
// that the library's large units compile side by side.
#define I2LQR_LANEPAIR_DEFINE
#include "i2lqr_lane12.h"

namespace i2lqr {
I2LQR_LANEPAIR_KERNELS(template __global__)
}  // namespace i2lqr

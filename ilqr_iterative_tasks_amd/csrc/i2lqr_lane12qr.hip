// The stage-weight (Q, R != 0) instantiations of quad12's one-problem-per-lane kernels (round 5;
// utils/base.py:243-246: matrix_Q / matrix_R are constructor parameters of the reference): a
// translation unit of their own so that the library's four large units compile side by side.
#define I2LQR_LANE12QR_DEFINE
#include "i2lqr_lane12.h"

namespace i2lqr {
I2LQR_LANE12QR_KERNELS(template __global__)
}  // namespace i2lqr

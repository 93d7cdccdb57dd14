// The one-problem-per-lane kernels of quad12 (n = 12, m = 4) are compiled in a translation unit
// of their own (i2lqr_lane12.hip: the unrolled Riccati step is ~4000 instructions per horizon
// step); i2lqr_abi.hip launches them through these declarations.
#pragma once
#include "i2lqr_lane.hpp"

namespace i2lqr {

// (QR: the stage-weight instantiations, compiled in i2lqr_lane12qr.hip)
#define I2LQR_LANE12_KERNELS(DECL) I2LQR_LANE12_KERNELS_(DECL, false)
#define I2LQR_LANE12QR_KERNELS(DECL) I2LQR_LANE12_KERNELS_(DECL, true)
#define I2LQR_LANE12_KERNELS_(DECL, QR)                                                          \
  DECL void k_lane_iterate_rows<double, Quad12<double>, QR, false>(const DevCfg<double, 12, 4>,   \
                                                                   const LaneArgs<double>);       \
  DECL void k_lane_iterate_rows<double, Quad12<double>, QR, true>(const DevCfg<double, 12, 4>,    \
                                                                  const LaneArgs<double>);        \
  DECL void k_lane_rollout<double, Quad12<double>, QR, false>(                                 \
      const DevCfg<double, 12, 4>, int64_t, double*, double*, const double*, double*);            \
  DECL void k_lane_rollout<double, Quad12<double>, QR, true>(                                  \
      const DevCfg<double, 12, 4>, int64_t, double*, double*, const double*, double*);            \
  DECL void k_lane_backward<double, Quad12<double>, QR, false>(                                \
      const DevCfg<double, 12, 4>, int64_t, const double*, const double*, const double*,          \
      const double*, const double*, double*, double*);                                            \
  DECL void k_lane_backward<double, Quad12<double>, QR, true>(                                 \
      const DevCfg<double, 12, 4>, int64_t, const double*, const double*, const double*,          \
      const double*, const double*, double*, double*);                                            \
  DECL void k_lane_forward<double, Quad12<double>, QR, false>(                                 \
      const DevCfg<double, 12, 4>, int64_t, const double*, const double*, const double*,          \
      const double*, const double*, double*, double*, double*);                                   \
  DECL void k_lane_forward<double, Quad12<double>, QR, true>(                                  \
      const DevCfg<double, 12, 4>, int64_t, const double*, const double*, const double*,          \
      const double*, const double*, double*, double*, double*);

#ifndef I2LQR_LANE12_DEFINE
I2LQR_LANE12_KERNELS(extern template __global__)
#endif
#ifndef I2LQR_LANE12QR_DEFINE
I2LQR_LANE12QR_KERNELS(extern template __global__)
#endif

}  // namespace i2lqr

// The one-problem-per-lane kernels of quad12 (n = 12, m = 4) are compiled in a translation unit
// of their own (i2lqr_lane12.hip: the unrolled Riccati step is ~4000 instructions per horizon
// step); i2lqr_abi.hip launches them through these declarations.
#pragma once
#include "i2lqr_lane.hpp"

namespace i2lqr {

// (QR: the stage-weight instantiations, compiled in i2lqr_lane12qr.hip; F32: fp32, Q = R = 0,
// compiled in i2lqr_lane12f.hip)
#define I2LQR_LANE12_KERNELS(DECL) I2LQR_LANE12_KERNELS_(DECL, double, false)
#define I2LQR_LANE12QR_KERNELS(DECL) I2LQR_LANE12_KERNELS_(DECL, double, true)
#define I2LQR_LANE12F_KERNELS(DECL) I2LQR_LANE12_KERNELS_(DECL, float, false)
#define I2LQR_LANE12_KERNELS_(DECL, REAL, QR)                                                  \
  DECL void k_lane_iterate_rows<REAL, Quad12<REAL>, QR, false>(const DevCfg<REAL, 12, 4>,   \
                                                                   const LaneArgs<REAL>);       \
  DECL void k_lane_iterate_rows<REAL, Quad12<REAL>, QR, true>(const DevCfg<REAL, 12, 4>,    \
                                                                  const LaneArgs<REAL>);        \
  DECL void k_lane_rollout<REAL, Quad12<REAL>, QR, false>(                                 \
      const DevCfg<REAL, 12, 4>, int64_t, REAL*, REAL*, const REAL*, REAL*);            \
  DECL void k_lane_rollout<REAL, Quad12<REAL>, QR, true>(                                  \
      const DevCfg<REAL, 12, 4>, int64_t, REAL*, REAL*, const REAL*, REAL*);            \
  DECL void k_lane_backward<REAL, Quad12<REAL>, QR, false>(                                \
      const DevCfg<REAL, 12, 4>, int64_t, const REAL*, const REAL*, const REAL*,          \
      const REAL*, const REAL*, REAL*, REAL*);                                            \
  DECL void k_lane_backward<REAL, Quad12<REAL>, QR, true>(                                 \
      const DevCfg<REAL, 12, 4>, int64_t, const REAL*, const REAL*, const REAL*,          \
      const REAL*, const REAL*, REAL*, REAL*);                                            \
  DECL void k_lane_forward<REAL, Quad12<REAL>, QR, false>(                                 \
      const DevCfg<REAL, 12, 4>, int64_t, const REAL*, const REAL*, const REAL*,          \
      const REAL*, const REAL*, REAL*, REAL*, REAL*);                                   \
  DECL void k_lane_forward<REAL, Quad12<REAL>, QR, true>(                                  \
      const DevCfg<REAL, 12, 4>, int64_t, const REAL*, const REAL*, const REAL*,          \
      const REAL*, const REAL*, REAL*, REAL*, REAL*);

// k_lane_iterate_pair (the bicycles' lane kernel with a helper wavefront): compiled in
// i2lqr_lanepair.hip
#define I2LQR_LANEPAIR_KERNELS_(DECL, REAL, QR)                                                   \
  DECL void k_lane_iterate_pair<REAL, Bicycle4<REAL>, QR, false>(const DevCfg<REAL, 4, 2>,        \
                                                                 const LaneArgs<REAL>);           \
  DECL void k_lane_iterate_pair<REAL, Bicycle4<REAL>, QR, true>(const DevCfg<REAL, 4, 2>,         \
                                                                const LaneArgs<REAL>);            \
  DECL void k_lane_iterate_pair<REAL, Bicycle6<REAL>, QR, false>(const DevCfg<REAL, 6, 2>,        \
                                                                 const LaneArgs<REAL>);           \
  DECL void k_lane_iterate_pair<REAL, Bicycle6<REAL>, QR, true>(const DevCfg<REAL, 6, 2>,         \
                                                                const LaneArgs<REAL>);
#define I2LQR_LANEPAIR_KERNELS(DECL)                                                    \
  I2LQR_LANEPAIR_KERNELS_(DECL, double, false) I2LQR_LANEPAIR_KERNELS_(DECL, double, true) \
  I2LQR_LANEPAIR_KERNELS_(DECL, float, false) I2LQR_LANEPAIR_KERNELS_(DECL, float, true)
#ifndef I2LQR_LANEPAIR_DEFINE
I2LQR_LANEPAIR_KERNELS(extern template __global__)
#endif

#ifndef I2LQR_LANE12_DEFINE
I2LQR_LANE12_KERNELS(extern template __global__)
#endif
#ifndef I2LQR_LANE12QR_DEFINE
I2LQR_LANE12QR_KERNELS(extern template __global__)
#endif
#ifndef I2LQR_LANE12F_DEFINE
I2LQR_LANE12F_KERNELS(extern template __global__)
#endif

}  // namespace i2lqr

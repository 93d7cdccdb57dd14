// Launcher of the eight-lanes-per-problem kernel (i2lqr_group.hpp), compiled in its own translation
// unit (i2lqr_group.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/i2lqr.h"
#include "i2lqr_wave.hpp"

namespace i2lqr {

// true if the configuration can run on the eight-lane kernel (plant, Q = R = 0, LDS budget)
bool group_supported(const i2lqr_config& cfg);
// Enqueue k_group_iterate for B problems (problem-major layout).  Returns hipSuccess or the HIP
// error of the attribute call / launch.
template <class T> hipError_t group_iterate(const i2lqr_config& cfg, const IterArgs<T>& a,
                                            hipStream_t stream);

}  // namespace i2lqr

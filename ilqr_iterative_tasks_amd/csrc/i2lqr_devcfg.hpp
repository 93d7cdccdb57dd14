// Host side: i2lqr_config (include/i2lqr.h) -> the typed device copy the kernels take by value.
#pragma once
#include <cmath>
#include <cstring>

#include "../../include/i2lqr.h"
#include "i2lqr_wave.hpp"

namespace i2lqr {

// Debug build: the device word the kernels record an index violation in (i2lqr_debug.hpp), one
// per process and device, allocated on first use (i2lqr_abi.hip); null in the product build.
unsigned long long* debug_trap_word();

template <class T, int n, int m> inline DevCfg<T, n, m> make_dev_cfg(const i2lqr_config& h) {
  DevCfg<T, n, m> d;
  std::memset(&d, 0, sizeof(d));
  d.N = h.N;
  d.max_iter = h.max_iter;
  d.dt = (T)h.dt;
  d.eps = (T)h.eps;
  d.lamb_factor = (T)h.lamb_factor;
  d.max_lamb = (T)h.max_lamb;
  d.ctrl_q1 = (T)h.ctrl_q1;
  d.ctrl_q2 = (T)h.ctrl_q2;
  d.obs_q1 = (T)h.obs_q1;
  d.obs_q2 = (T)h.obs_q2;
  d.safety_margin = (T)h.safety_margin;
  bool hasQ = false, hasR = false;
  d.fast_barrier = 1;
  for (int a = 0; a < m; a++) {
    d.u_max[a] = (T)h.u_max[a];
    const double span = 2.0 * h.ctrl_q2 * h.u_max[a];
    d.ctrl_c[a] = (T)std::exp(-span);
    if (!(std::fabs(span) < 600.0)) d.fast_barrier = 0;
  }
  for (int i = 0; i < n; i++) d.xtarget[i] = (T)h.xtarget[i];
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) {
      d.Q[i * n + j] = (T)h.Q[i * I2LQR_MAX_N + j];
      d.Qt[i * n + j] = (T)h.Qt[i * I2LQR_MAX_N + j];
      hasQ |= h.Q[i * I2LQR_MAX_N + j] != 0.0;
    }
  for (int a = 0; a < m; a++)
    for (int b = 0; b < m; b++) {
      d.R[a * m + b] = (T)h.R[a * I2LQR_MAX_M + b];
      hasR |= h.R[a * I2LQR_MAX_M + b] != 0.0;
    }
  for (int q = 0; q < 8; q++) d.sys_par[q] = (T)h.sys_par[q];
  d.ctrl_q12 = d.ctrl_q1 * d.ctrl_q2;
  d.ctrl_q122 = d.ctrl_q1 * (d.ctrl_q2 * d.ctrl_q2);
  d.obs_q12 = d.obs_q1 * d.obs_q2;
  d.obs_q122 = d.obs_q1 * (d.obs_q2 * d.obs_q2);
  if (h.system_id == I2LQR_SYS_QUAD12) {  // Quad12::step_tr / jac_var / plant_const read these
    const T mass = d.sys_par[0], g = d.sys_par[1], arm = d.sys_par[2];
    const T Ix = d.sys_par[3], Iy = d.sys_par[4], Iz = d.sys_par[5], ct = d.sys_par[6];
    d.pd[0] = T(1) / mass;
    d.pd[1] = arm / Ix;
    d.pd[2] = arm / Iy;
    d.pd[3] = ct / Iz;
    d.pd[4] = (Iy - Iz) / Ix;
    d.pd[5] = (Iz - Ix) / Iy;
    d.pd[6] = (Ix - Iy) / Iz;
    d.pd[7] = mass * g;
    for (int q = 0; q < 6; q++) {
      const T v = q < 2 ? d.dt * arm / Ix : (q < 4 ? d.dt * arm / Iy : d.dt * ct / Iz);
      d.pd[8 + q] = (q & 1) ? -v : v;
    }
  }
  d.flags = (hasQ ? FLAG_HAS_Q : 0) | (hasR ? FLAG_HAS_R : 0);
  d.trap = debug_trap_word();
  return d;
}


}  // namespace i2lqr

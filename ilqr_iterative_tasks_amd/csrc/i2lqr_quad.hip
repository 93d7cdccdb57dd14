// Instantiations and launcher of the sixteen-lanes-per-problem kernel (i2lqr_quad.hpp).
#include "i2lqr_geometry.hpp"
#include "i2lqr_group.h"

#include <cstdlib>

#include "i2lqr_devcfg.hpp"
#include "i2lqr_quad.hpp"
#include "i2lqr_dryrun.hpp"  // (empty unless -DI2LQR_DRY_RUN: the ASan build)

namespace i2lqr {

namespace {

bool quad_has_stage_weights(const i2lqr_config& cfg) {
  for (int i = 0; i < cfg.n; i++)
    for (int j = 0; j < cfg.n; j++)
      if (cfg.Q[i * I2LQR_MAX_N + j] != 0.0) return true;
  for (int a = 0; a < cfg.m; a++)
    for (int b = 0; b < cfg.m; b++)
      if (cfg.R[a * I2LQR_MAX_M + b] != 0.0) return true;
  return false;
}

template <class T> hipError_t launch_quad(const i2lqr_config& cfg, const IterArgs<T>& a, void* ws,
                                          hipStream_t s) {
  using Sys = Quad12<T>;
  const auto c = make_dev_cfg<T, Sys::n, Sys::m>(cfg);
  const size_t lds = (size_t)QLayout<Sys>(cfg.N).lds_total * kQPW * sizeof(T);
  if (lds > device_geometry().default_dyn_lds) {
    hipError_t e = hipFuncSetAttribute((const void*)k_quad_iterate<T, Sys>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  const unsigned grid = (unsigned)((a.B + kQPW - 1) / kQPW);
#ifdef I2LQR_STAMPS
  IterArgs<T> a2 = a;  // diagnostic build: [B][8] u64 phase sums at the address in I2LQR_DBG_PTR
  a2.dbg = nullptr;
  if (const char* e = getenv("I2LQR_DBG_PTR")) a2.dbg = (unsigned long long*)strtoull(e, nullptr, 0);
  hipLaunchKernelGGL((k_quad_iterate<T, Sys>), dim3(grid), dim3(64), lds, s, c, a2, (T*)ws);
  return hipGetLastError();
#endif
  hipLaunchKernelGGL((k_quad_iterate<T, Sys>), dim3(grid), dim3(64), lds, s, c, a, (T*)ws);
  return hipGetLastError();
}

}  // namespace

bool quad_supported(const i2lqr_config& cfg) {
  if (cfg.system_id != I2LQR_SYS_QUAD12 || cfg.layout != I2LQR_LAYOUT_PROBLEM_MAJOR) return false;
  if (quad_has_stage_weights(cfg)) return false;
  const size_t elem = cfg.dtype == I2LQR_F64 ? 8 : 4;
  return (size_t)QLayout<Quad12<double>>(cfg.N).lds_total * kQPW * elem <= device_geometry().max_dyn_lds;
}

int64_t quad_workspace_bytes(const i2lqr_config& cfg, int64_t B) {
  if (!quad_supported(cfg)) return 0;
  const int64_t Bp = (B + kQPW - 1) / kQPW * kQPW;  // whole wavefronts
  const int64_t elem = cfg.dtype == I2LQR_F64 ? 8 : 4;
  return Bp * (int64_t)QLayout<Quad12<double>>(cfg.N).ws_total * elem;
}

template <> hipError_t quad_iterate<double>(const i2lqr_config& cfg, const IterArgs<double>& a,
                                            void* ws, hipStream_t s) {
  return launch_quad<double>(cfg, a, ws, s);
}
template <> hipError_t quad_iterate<float>(const i2lqr_config& cfg, const IterArgs<float>& a,
                                           void* ws, hipStream_t s) {
  return launch_quad<float>(cfg, a, ws, s);
}

}  // namespace i2lqr

// Instantiations and launcher of the eight-lanes-per-problem kernel (i2lqr_group.hpp).
#include "i2lqr_group.h"

#include "i2lqr_devcfg.hpp"
#include "i2lqr_geometry.hpp"
#include "i2lqr_group.hpp"
#include "i2lqr_dryrun.hpp"  // (empty unless -DI2LQR_DRY_RUN: the ASan build)

namespace i2lqr {

namespace {

template <class T, class Sys, int G = kGroup> size_t group_lds_bytes(int N) {
  return (size_t)GLayout<Sys, G>(N).wave_words() * sizeof(T);
}

bool has_stage_weights(const i2lqr_config& cfg) {
  for (int i = 0; i < cfg.n; i++)
    for (int j = 0; j < cfg.n; j++)
      if (cfg.Q[i * I2LQR_MAX_N + j] != 0.0) return true;
  for (int a = 0; a < cfg.m; a++)
    for (int b = 0; b < cfg.m; b++)
      if (cfg.R[a * I2LQR_MAX_M + b] != 0.0) return true;
  return false;
}

// Launches with more than 64 KiB of dynamic LDS need the kernel's attribute raised — per kernel
// AND per device (a second GPU used from the same thread has its own copy of the attribute):
// once per (kernel, device, size).
template <auto Kernel> hipError_t raise_lds_limit(size_t lds) {
  if (lds <= device_geometry().default_dyn_lds) return hipSuccess;
  constexpr int kMaxDev = 64;
  static thread_local int raised_for[kMaxDev] = {};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= kMaxDev || raised_for[dev] < (int)lds) {
    e = hipFuncSetAttribute((const void*)Kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < kMaxDev) raised_for[dev] = (int)lds;
  }
  return hipSuccess;
}

template <class T, class Sys, int H, int G = kGroup>
hipError_t launch_h(const i2lqr_config& cfg, const IterArgs<T>& a, hipStream_t s) {
  const auto c = make_dev_cfg<T, Sys::n, Sys::m>(cfg);
  const size_t lds = group_lds_bytes<T, Sys, G>(cfg.N);
  if (hipError_t e = raise_lds_limit<k_group_iterate<T, Sys, H, false, G>>(lds); e != hipSuccess)
    return e;
  constexpr int PW = 64 / G;
  const unsigned grid = (unsigned)((a.B + PW - 1) / PW);
  hipLaunchKernelGGL((k_group_iterate<T, Sys, H, false, G>), dim3(grid), dim3(64 * H), lds, s, c, a,
                     (T*)nullptr);
  return hipGetLastError();
}
// Sixteen lanes per problem (one problem per DPP row, four per wavefront): the backward step
// exchanges nothing through LDS (GroupWorker::backward_row).  One helper wavefront for the record
// phase (2 x 16 lanes >= the 21 records of a problem: one round) while that leaves every wavefront
// a SIMD of its own (<= 512 workgroups = 2048 problems).
template <class T, class Sys>
hipError_t launch16(const i2lqr_config& cfg, const IterArgs<T>& a, hipStream_t s) {
  const int64_t cus = device_geometry().cus;
#ifndef I2LQR_STAMPS
  if ((a.B + 3) / 4 <= 2 * cus) return launch_h<T, Sys, 2, 16>(cfg, a, s);
#endif
  return launch_h<T, Sys, 1, 16>(cfg, a, s);
}
// Two helper wavefronts for the record phase (k_group_iterate<.., 3>) while every workgroup has
// a CU to itself (<= 256 workgroups = 2048 problems): 0.2216 -> 0.2107 ms per 10 iterations at
// 1024 problems (two / four wavefronts: 0.2122 / 0.2102).  With two workgroups per CU the extra
// wavefronts crowd the main ones off their SIMDs: 0.25 -> 0.40 ms at 4096 problems.
// Workspace form (k_group_iterate<.., 1, true>): records and gains in HBM, 4 KB of LDS per problem
template <class T, class Sys>
hipError_t launch_ws(const i2lqr_config& cfg, const IterArgs<T>& a, void* ws, hipStream_t s) {
  const auto c = make_dev_cfg<T, Sys::n, Sys::m>(cfg);
  const size_t lds = (size_t)GLayout<Sys>(cfg.N, true).wave_words() * sizeof(T);
  if (hipError_t e = raise_lds_limit<k_group_iterate<T, Sys, 1, true>>(lds); e != hipSuccess) return e;
  const unsigned grid = (unsigned)((a.B + kGroupsPerWave - 1) / kGroupsPerWave);
  hipLaunchKernelGGL((k_group_iterate<T, Sys, 1, true>), dim3(grid), dim3(64), lds, s, c, a, (T*)ws);
  return hipGetLastError();
}

template <class T, class Sys>
hipError_t launch(const i2lqr_config& cfg, const IterArgs<T>& a, hipStream_t s) {
  const int64_t cus = device_geometry().cus;
#ifndef I2LQR_STAMPS  // (the diagnostic build stamps the phases of the lone wavefront)
  if ((a.B + kGroupsPerWave - 1) / kGroupsPerWave <= cus) return launch_h<T, Sys, 3>(cfg, a, s);
#endif
  return launch_h<T, Sys, 1>(cfg, a, s);
}

// Wavefronts per group of eight problems in the speculative kernel.  Three skip two rejects per
// round but share one CU's LDS three ways; the stragglers that decide a solve's duration alternate
// accept / reject, which two wavefronts cover: measured (i2lqr_solve, ms) 16 problems 0.43 (V = 3)
// / 0.46 (V = 2), 256: 0.25 / 0.26, 1024: 0.77 / 0.72, 2048: 0.70 / 0.50, tail of 65536: 2.41 /
// 2.34.  Three up to kSpecWideBatch problems where they fit, two above and in the tail.
constexpr int kSpecWideBatch = 512;

template <class T, class Sys, int V, int G = kGroup> size_t spec_lds_bytes(int N) {
  return (size_t)GSpecLayout<Sys, V, G>(N).group_words() * sizeof(T);
}

// SETIO: a.count_max problems at most (k_group_spec<.., true>, the tail of the chunked solves)
template <class T, class Sys, int V, bool SETIO, int G>
hipError_t launch_spec_v(const i2lqr_config& cfg, const IterArgs<T>& a, hipStream_t s) {
  const auto c = make_dev_cfg<T, Sys::n, Sys::m>(cfg);
  const size_t lds = spec_lds_bytes<T, Sys, V, G>(cfg.N);
  if (hipError_t e = raise_lds_limit<k_group_spec<T, Sys, V, SETIO, G>>(lds); e != hipSuccess)
    return e;
  constexpr int PW = 64 / G;
  const int64_t problems = SETIO ? (int64_t)a.count_max : a.B;
  const unsigned grid = (unsigned)((problems + PW - 1) / PW);
  hipLaunchKernelGGL((k_group_spec<T, Sys, V, SETIO, G>), dim3(grid), dim3(64 * V), lds, s, c, a);
  return hipGetLastError();
}
template <class T, class Sys, bool SETIO, int G>
hipError_t launch_spec(const i2lqr_config& cfg, const IterArgs<T>& a, hipStream_t s) {
  const DeviceGeometry& geo = device_geometry();
  const bool wide = !SETIO && a.B <= geo.scaled(kSpecWideBatch) &&
                    spec_lds_bytes<T, Sys, 3, G>(cfg.N) <= geo.max_dyn_lds;
  if (wide) return launch_spec_v<T, Sys, 3, SETIO, G>(cfg, a, s);
  return launch_spec_v<T, Sys, 2, SETIO, G>(cfg, a, s);
}
template <class T, bool SETIO>
hipError_t launch_spec_any(const i2lqr_config& cfg, const IterArgs<T>& a, hipStream_t s, int lanes) {
  const bool b4 = cfg.system_id == I2LQR_SYS_BICYCLE4;
  if (lanes == 16)
    return b4 ? launch_spec<T, Bicycle4<T>, SETIO, 16>(cfg, a, s)
              : launch_spec<T, Bicycle6<T>, SETIO, 16>(cfg, a, s);
  return b4 ? launch_spec<T, Bicycle4<T>, SETIO, kGroup>(cfg, a, s)
            : launch_spec<T, Bicycle6<T>, SETIO, kGroup>(cfg, a, s);
}

template <int G> bool spec_lds_fits(const i2lqr_config& cfg) {  // the two-wavefront form
  const size_t lds = cfg.dtype == I2LQR_F64
      ? (cfg.system_id == I2LQR_SYS_BICYCLE4 ? spec_lds_bytes<double, Bicycle4<double>, 2, G>(cfg.N)
                                             : spec_lds_bytes<double, Bicycle6<double>, 2, G>(cfg.N))
      : (cfg.system_id == I2LQR_SYS_BICYCLE4 ? spec_lds_bytes<float, Bicycle4<float>, 2, G>(cfg.N)
                                             : spec_lds_bytes<float, Bicycle6<float>, 2, G>(cfg.N));
  return lds <= device_geometry().max_dyn_lds;
}
bool spec_plant_ok(const i2lqr_config& cfg) {
  if (cfg.system_id != I2LQR_SYS_BICYCLE4 && cfg.system_id != I2LQR_SYS_BICYCLE6) return false;
  return !has_stage_weights(cfg);
}

}  // namespace

bool group_spec_supported(const i2lqr_config& cfg, int lanes) {
  if (!spec_plant_ok(cfg) || cfg.layout != I2LQR_LAYOUT_PROBLEM_MAJOR) return false;
  return lanes == 16 ? spec_lds_fits<16>(cfg) : (group_supported(cfg) && spec_lds_fits<kGroup>(cfg));
}
int group_spec_tail_lanes(const i2lqr_config& cfg) {
  if (!spec_plant_ok(cfg)) return 0;
  return spec_lds_fits<16>(cfg) ? 16 : (spec_lds_fits<kGroup>(cfg) ? kGroup : 0);
}
bool group_spec_tail_supported(const i2lqr_config& cfg) { return group_spec_tail_lanes(cfg) != 0; }

// chains (k_group_spec<.., 3, false, 16, true>): the sixteen-lane form with three wavefronts
template <class T, class Sys>
hipError_t launch_spec_chain(const i2lqr_config& cfg, const IterArgs<T>& a, hipStream_t s) {
  const auto c = make_dev_cfg<T, Sys::n, Sys::m>(cfg);
  const size_t lds = spec_lds_bytes<T, Sys, 3, 16>(cfg.N);
  if (hipError_t e = raise_lds_limit<k_group_spec<T, Sys, 3, false, 16, true>>(lds); e != hipSuccess)
    return e;
  const unsigned grid = (unsigned)((a.B + 3) / 4);
  hipLaunchKernelGGL((k_group_spec<T, Sys, 3, false, 16, true>), dim3(grid), dim3(64 * 3), lds, s, c, a);
  return hipGetLastError();
}
bool group_spec_chain_supported(const i2lqr_config& cfg, int64_t chains) {
  if (!spec_plant_ok(cfg) || cfg.layout != I2LQR_LAYOUT_PROBLEM_MAJOR) return false;
  const DeviceGeometry& geo = device_geometry();
  if (chains > geo.scaled(kSpecWideBatch)) return false;  // three wavefronts per workgroup
  const size_t lds = cfg.dtype == I2LQR_F64
      ? (cfg.system_id == I2LQR_SYS_BICYCLE4 ? spec_lds_bytes<double, Bicycle4<double>, 3, 16>(cfg.N)
                                             : spec_lds_bytes<double, Bicycle6<double>, 3, 16>(cfg.N))
      : (cfg.system_id == I2LQR_SYS_BICYCLE4 ? spec_lds_bytes<float, Bicycle4<float>, 3, 16>(cfg.N)
                                             : spec_lds_bytes<float, Bicycle6<float>, 3, 16>(cfg.N));
  return lds <= geo.max_dyn_lds;
}
template <> hipError_t group_spec_chain<double>(const i2lqr_config& cfg, const IterArgs<double>& a,
                                                hipStream_t s) {
  if (cfg.system_id == I2LQR_SYS_BICYCLE4) return launch_spec_chain<double, Bicycle4<double>>(cfg, a, s);
  return launch_spec_chain<double, Bicycle6<double>>(cfg, a, s);
}
template <> hipError_t group_spec_chain<float>(const i2lqr_config& cfg, const IterArgs<float>& a,
                                               hipStream_t s) {
  if (cfg.system_id == I2LQR_SYS_BICYCLE4) return launch_spec_chain<float, Bicycle4<float>>(cfg, a, s);
  return launch_spec_chain<float, Bicycle6<float>>(cfg, a, s);
}

template <> hipError_t group_spec_iterate<double>(const i2lqr_config& cfg, const IterArgs<double>& a,
                                                  hipStream_t s, int lanes) {
  return launch_spec_any<double, false>(cfg, a, s, lanes);
}
template <> hipError_t group_spec_iterate<float>(const i2lqr_config& cfg, const IterArgs<float>& a,
                                                 hipStream_t s, int lanes) {
  return launch_spec_any<float, false>(cfg, a, s, lanes);
}
template <> hipError_t group_spec_tail<double>(const i2lqr_config& cfg, const IterArgs<double>& a,
                                               hipStream_t s) {
  return launch_spec_any<double, true>(cfg, a, s, group_spec_tail_lanes(cfg));
}
template <> hipError_t group_spec_tail<float>(const i2lqr_config& cfg, const IterArgs<float>& a,
                                              hipStream_t s) {
  return launch_spec_any<float, true>(cfg, a, s, group_spec_tail_lanes(cfg));
}

bool group16_supported(const i2lqr_config& cfg) {
  if (cfg.system_id != I2LQR_SYS_BICYCLE4 && cfg.system_id != I2LQR_SYS_BICYCLE6) return false;
  if (cfg.layout != I2LQR_LAYOUT_PROBLEM_MAJOR || has_stage_weights(cfg)) return false;
  const size_t lds = cfg.dtype == I2LQR_F64
      ? (cfg.system_id == I2LQR_SYS_BICYCLE4 ? group_lds_bytes<double, Bicycle4<double>, 16>(cfg.N)
                                             : group_lds_bytes<double, Bicycle6<double>, 16>(cfg.N))
      : (cfg.system_id == I2LQR_SYS_BICYCLE4 ? group_lds_bytes<float, Bicycle4<float>, 16>(cfg.N)
                                             : group_lds_bytes<float, Bicycle6<float>, 16>(cfg.N));
  return lds <= device_geometry().max_dyn_lds;
}
template <> hipError_t group16_iterate<double>(const i2lqr_config& cfg, const IterArgs<double>& a,
                                               hipStream_t s) {
  if (cfg.system_id == I2LQR_SYS_BICYCLE4) return launch16<double, Bicycle4<double>>(cfg, a, s);
  return launch16<double, Bicycle6<double>>(cfg, a, s);
}
template <> hipError_t group16_iterate<float>(const i2lqr_config& cfg, const IterArgs<float>& a,
                                              hipStream_t s) {
  if (cfg.system_id == I2LQR_SYS_BICYCLE4) return launch16<float, Bicycle4<float>>(cfg, a, s);
  return launch16<float, Bicycle6<float>>(cfg, a, s);
}

bool group_supported(const i2lqr_config& cfg) {
  if (cfg.system_id != I2LQR_SYS_BICYCLE4 && cfg.system_id != I2LQR_SYS_BICYCLE6) return false;
  if (cfg.layout != I2LQR_LAYOUT_PROBLEM_MAJOR || has_stage_weights(cfg)) return false;
  const size_t lds = cfg.dtype == I2LQR_F64
      ? (cfg.system_id == I2LQR_SYS_BICYCLE4 ? group_lds_bytes<double, Bicycle4<double>>(cfg.N)
                                             : group_lds_bytes<double, Bicycle6<double>>(cfg.N))
      : (cfg.system_id == I2LQR_SYS_BICYCLE4 ? group_lds_bytes<float, Bicycle4<float>>(cfg.N)
                                             : group_lds_bytes<float, Bicycle6<float>>(cfg.N));
  return lds <= device_geometry().max_dyn_lds;
}

int64_t group_workspace_bytes(const i2lqr_config& cfg, int64_t B) {
  if (!group_supported(cfg) || B <= 0) return 0;
  const int64_t probs = (B + kGroupsPerWave - 1) / kGroupsPerWave * kGroupsPerWave;
  const int64_t words = cfg.system_id == I2LQR_SYS_BICYCLE4
      ? GLayout<Bicycle4<double>>(cfg.N, true).ws_words()
      : GLayout<Bicycle6<double>>(cfg.N, true).ws_words();
  const size_t lds = cfg.dtype == I2LQR_F64
      ? (cfg.system_id == I2LQR_SYS_BICYCLE4
             ? (size_t)GLayout<Bicycle4<double>>(cfg.N, true).wave_words() * 8
             : (size_t)GLayout<Bicycle6<double>>(cfg.N, true).wave_words() * 8)
      : (cfg.system_id == I2LQR_SYS_BICYCLE4
             ? (size_t)GLayout<Bicycle4<float>>(cfg.N, true).wave_words() * 4
             : (size_t)GLayout<Bicycle6<float>>(cfg.N, true).wave_words() * 4);
  if (lds > device_geometry().max_dyn_lds) return 0;
  return probs * words * (cfg.dtype == I2LQR_F64 ? 8 : 4);
}
template <> hipError_t group_iterate_ws<double>(const i2lqr_config& cfg, const IterArgs<double>& a,
                                                void* ws, hipStream_t s) {
  if (cfg.system_id == I2LQR_SYS_BICYCLE4) return launch_ws<double, Bicycle4<double>>(cfg, a, ws, s);
  return launch_ws<double, Bicycle6<double>>(cfg, a, ws, s);
}
template <> hipError_t group_iterate_ws<float>(const i2lqr_config& cfg, const IterArgs<float>& a,
                                               void* ws, hipStream_t s) {
  if (cfg.system_id == I2LQR_SYS_BICYCLE4) return launch_ws<float, Bicycle4<float>>(cfg, a, ws, s);
  return launch_ws<float, Bicycle6<float>>(cfg, a, ws, s);
}

template <> hipError_t group_iterate<double>(const i2lqr_config& cfg, const IterArgs<double>& a,
                                             hipStream_t s) {
  if (cfg.system_id == I2LQR_SYS_BICYCLE4) return launch<double, Bicycle4<double>>(cfg, a, s);
  return launch<double, Bicycle6<double>>(cfg, a, s);
}
template <> hipError_t group_iterate<float>(const i2lqr_config& cfg, const IterArgs<float>& a,
                                            hipStream_t s) {
  if (cfg.system_id == I2LQR_SYS_BICYCLE4) return launch<float, Bicycle4<float>>(cfg, a, s);
  return launch<float, Bicycle6<float>>(cfg, a, s);
}

}  // namespace i2lqr

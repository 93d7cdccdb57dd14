// Dry-run launches for the host sanitizers.  Compiled in ONLY with
// -DI2LQR_DRY_RUN (the AddressSanitizer / UBSan build, `make asan`); active only when the environment
// says I2LQR_DRY_RUN=1.  i2lqr_create then skips the device, and every kernel launch of the
// library becomes a RECORD — kernel, grid, workgroup size, dynamic LDS — whose pointer arguments
// (plain pointers and every pointer field of IterArgs / LaneArgs / LaneSet) must lie inside a range
// the driver has declared with i2lqr_dry_run(1, base, bytes): the registered workspace and the
// caller's arrays.  That puts the host code BEHIND a live handle — workspace carving, the chunked
// solve's scheduler, the LDS budgeting, the sharded round — under ASan / UBSan on a box without a
// GPU, and turns "a carved pointer left the workspace" into a reported violation.
// The product build never sees this header's macros.
#pragma once
#ifdef I2LQR_DRY_RUN
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "i2lqr_wave.hpp"

namespace i2lqr {
namespace dry {

bool on();                                      // I2LQR_DRY_RUN=1 (read once)
void allow(const void* base, size_t bytes);     // a range kernel pointers may point into
void reset();                                   // forget ranges, records and violations
void record(const char* kernel, dim3 grid, dim3 block, size_t lds);
void ptr(const char* kernel, const char* field, const void* p);
int64_t report(char* buf, int64_t n);           // text of records + violations; returns #violations

template <class A> inline void arg(const char*, const A&) {}  // scalars, device configs
template <class P> inline void arg(const char* k, P* p) { ptr(k, "pointer argument", (const void*)p); }
template <class T> inline void arg(const char* k, const IterArgs<T>& a) {
  ptr(k, "IterArgs.X", a.X); ptr(k, "IterArgs.U", a.U); ptr(k, "IterArgs.x_term", a.x_term);
  ptr(k, "IterArgs.lamb", a.lamb); ptr(k, "IterArgs.obs", a.obs); ptr(k, "IterArgs.cost", a.cost);
  ptr(k, "IterArgs.K", a.K); ptr(k, "IterArgs.k", a.k); ptr(k, "IterArgs.iters", a.iters);
  ptr(k, "IterArgs.status", a.status); ptr(k, "IterArgs.count", a.count);
  ptr(k, "IterArgs.orig", a.orig); ptr(k, "IterArgs.out_X", a.out_X);
  ptr(k, "IterArgs.out_U", a.out_U); ptr(k, "IterArgs.out_K", a.out_K);
  ptr(k, "IterArgs.out_k", a.out_k); ptr(k, "IterArgs.out_lamb", a.out_lamb);
  ptr(k, "IterArgs.out_cost", a.out_cost); ptr(k, "IterArgs.out_iters", a.out_iters);
  ptr(k, "IterArgs.out_status", a.out_status); ptr(k, "IterArgs.qfun", a.qfun);
  ptr(k, "IterArgs.cost_it", a.cost_it); ptr(k, "IterArgs.pick_part", a.pick_part);
  ptr(k, "IterArgs.pick_ticket", a.pick_ticket); ptr(k, "IterArgs.best_idx", a.best_idx);
  ptr(k, "IterArgs.best_cost", a.best_cost);
}
#ifdef I2LQR_DRY_RUN_LANE  // (translation units that see i2lqr_lane.hpp)
template <class T> inline void arg(const char* k, const LaneSet<T>& s) {
  ptr(k, "LaneSet.X", s.X); ptr(k, "LaneSet.U", s.U); ptr(k, "LaneSet.x_term", s.x_term);
  ptr(k, "LaneSet.obs", s.obs); ptr(k, "LaneSet.lamb", s.lamb); ptr(k, "LaneSet.cost", s.cost);
  ptr(k, "LaneSet.K", s.K); ptr(k, "LaneSet.k", s.k); ptr(k, "LaneSet.iters", s.iters);
  ptr(k, "LaneSet.status", s.status); ptr(k, "LaneSet.orig", s.orig);
}
template <class T> inline void arg(const char* k, const LaneArgs<T>& a) {
  ptr(k, "LaneArgs.X", a.X); ptr(k, "LaneArgs.U", a.U); ptr(k, "LaneArgs.x_term", a.x_term);
  ptr(k, "LaneArgs.lamb", a.lamb); ptr(k, "LaneArgs.obs", a.obs); ptr(k, "LaneArgs.cost", a.cost);
  ptr(k, "LaneArgs.K", a.K); ptr(k, "LaneArgs.k", a.k); ptr(k, "LaneArgs.iters", a.iters);
  ptr(k, "LaneArgs.status", a.status); ptr(k, "LaneArgs.wsU", a.wsU); ptr(k, "LaneArgs.wsK", a.wsK);
  ptr(k, "LaneArgs.wsk", a.wsk); ptr(k, "LaneArgs.wsX", a.wsX); ptr(k, "LaneArgs.count", a.count);
  if (a.cp.on) {
    ptr(k, "LaneArgs.cp.orig", a.cp.orig); ptr(k, "LaneArgs.cp.count_out", a.cp.count_out);
    arg(k, a.cp.dst);
    arg(k, a.cp.usr);
  }
}
#endif

template <class... A>
inline void launch(const char* kernel, dim3 grid, dim3 block, size_t lds, const A&... a) {
  record(kernel, grid, block, lds);
  (arg(kernel, a), ...);
}

}  // namespace dry
}  // namespace i2lqr

// every launch of the translation unit goes through here
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...)                                  \
  do {                                                                                             \
    if (::i2lqr::dry::on())                                                                        \
      ::i2lqr::dry::launch(#kernel, dim3(grid), dim3(block), (size_t)(lds), __VA_ARGS__);          \
    else                                                                                           \
      hipLaunchKernelGGLInternal((kernel), (grid), (block), (lds), (stream), __VA_ARGS__);         \
  } while (0)
// ... and the runtime calls between them succeed without a device
#define hipMemsetAsync(...) (::i2lqr::dry::on() ? hipSuccess : hipMemsetAsync(__VA_ARGS__))
#define hipMemcpyAsync(...) (::i2lqr::dry::on() ? hipSuccess : hipMemcpyAsync(__VA_ARGS__))
#define hipFuncSetAttribute(...) (::i2lqr::dry::on() ? hipSuccess : hipFuncSetAttribute(__VA_ARGS__))
#define hipGetLastError() (::i2lqr::dry::on() ? hipSuccess : hipGetLastError())
#define hipEventCreateWithFlags(ev, flags) \
  (::i2lqr::dry::on() ? (*(ev) = (hipEvent_t)0x10, hipSuccess) : hipEventCreateWithFlags(ev, flags))
#define hipEventDestroy(...) (::i2lqr::dry::on() ? hipSuccess : hipEventDestroy(__VA_ARGS__))
#define hipEventRecord(...) (::i2lqr::dry::on() ? hipSuccess : hipEventRecord(__VA_ARGS__))
#define hipStreamWaitEvent(...) (::i2lqr::dry::on() ? hipSuccess : hipStreamWaitEvent(__VA_ARGS__))
#define hipFree(...) (::i2lqr::dry::on() ? hipSuccess : hipFree(__VA_ARGS__))
#define hipGetDevice(p) (::i2lqr::dry::on() ? (*(p) = 0, hipSuccess) : hipGetDevice(p))
#endif  // I2LQR_DRY_RUN

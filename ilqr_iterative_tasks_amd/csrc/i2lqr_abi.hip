// C-ABI of libi2lqr_hip.so (include/i2lqr.h): argument validation, typed device configs and
// kernel dispatch.  No exception crosses the boundary; every entry point returns an int code and
// records a thread-local message for i2lqr_last_error().
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and prototypes only: the library is bound at run time (rccl_api())

#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <unordered_set>

#include "../../include/i2lqr.h"
#include "i2lqr_devcfg.hpp"
#include "i2lqr_geometry.hpp"
#include "i2lqr_group.h"
#include "i2lqr_lane.hpp"
#include "i2lqr_select.hpp"
#include "i2lqr_wave.hpp"

#include "i2lqr_lane12.h"

#define I2LQR_DRY_RUN_LANE 1
#include "i2lqr_dryrun.hpp"  // (empty unless -DI2LQR_DRY_RUN: the ASan build)

using namespace i2lqr;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess)                                                                  \
      return fail(I2LQR_ERR_LAUNCH, "%s failed: %s", #expr, hipGetErrorString(e_));        \
  } while (0)

}  // namespace

#ifdef I2LQR_DRY_RUN
#include <string>
#include <vector>
namespace i2lqr {
namespace dry {
namespace {
struct State {
  std::mutex mu;
  std::vector<std::pair<uintptr_t, uintptr_t>> ranges;
  std::string text;
  int64_t launches = 0, violations = 0;
};
State& st() {
  static State s;
  return s;
}
}  // namespace
bool on() {
  static const bool v = [] {
    const char* e = getenv("I2LQR_DRY_RUN");
    return e && e[0] == '1';
  }();
  return v;
}
void allow(const void* base, size_t bytes) {
  std::lock_guard<std::mutex> lock(st().mu);
  st().ranges.emplace_back((uintptr_t)base, (uintptr_t)base + bytes);
}
void reset() {
  std::lock_guard<std::mutex> lock(st().mu);
  st().ranges.clear();
  st().text.clear();
  st().launches = st().violations = 0;
}
void record(const char* kernel, dim3 grid, dim3 block, size_t lds) {
  std::lock_guard<std::mutex> lock(st().mu);
  char line[384];
  snprintf(line, sizeof(line), "launch %.200s grid %u block %u lds %zu\n", kernel, grid.x, block.x, lds);
  if (st().text.size() < (1u << 15)) st().text += line;
  st().launches++;
  const DeviceGeometry& g = device_geometry();
  if (grid.x == 0 || block.x == 0 || block.x > 1024 || lds > g.max_dyn_lds) {
    st().violations++;
    snprintf(line, sizeof(line), "VIOLATION %.200s: launch shape grid %u block %u lds %zu\n", kernel,
             grid.x, block.x, lds);
    st().text += line;
  }
}
void ptr(const char* kernel, const char* field, const void* p) {
  if (!p) return;
  std::lock_guard<std::mutex> lock(st().mu);
  const uintptr_t a = (uintptr_t)p;
  for (const auto& r : st().ranges)
    if (a >= r.first && a < r.second) return;
  st().violations++;
  char line[384];
  snprintf(line, sizeof(line), "VIOLATION %.200s: %s = %p lies in no declared range\n", kernel, field, p);
  st().text += line;
}
int64_t report(char* buf, int64_t n) {
  std::lock_guard<std::mutex> lock(st().mu);
  if (buf && n > 0) {
    // violations first: the buffer may be shorter than the launch log
    std::string out;
    size_t pos = 0;
    while ((pos = st().text.find("VIOLATION", pos)) != std::string::npos) {
      const size_t end = st().text.find('\n', pos);
      out += st().text.substr(pos, end == std::string::npos ? std::string::npos : end - pos + 1);
      if (end == std::string::npos) break;
      pos = end + 1;
    }
    out += st().text;
    snprintf(buf, (size_t)n, "%s", out.c_str());
  }
  st().text.clear();
  const int64_t v = st().violations;
  st().violations = 0;
  return v;
}
}  // namespace dry
}  // namespace i2lqr
#endif

namespace i2lqr {
// hipDeviceGetAttribute once per device; I2LQR_FAKE_CUS=<n> (a debug override, parity tests of the
// derived thresholds on the full chip) replaces the CU count.
const DeviceGeometry& device_geometry() {
  constexpr int kMaxDev = 64;
  static DeviceGeometry table[kMaxDev];
  static std::once_flag once[kMaxDev];
  static const DeviceGeometry fallback{};  // no device visible: the MI355X figures
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) {
    (void)hipGetLastError();
    return fallback;
  }
  std::call_once(once[dev], [dev] {
    DeviceGeometry g;
    int v = 0;
    bool ok = true;
    auto attr = [&](hipDeviceAttribute_t a) {
      v = 0;
      const bool got = hipDeviceGetAttribute(&v, a, dev) == hipSuccess && v > 0;
      if (!got) (void)hipGetLastError();
      return got;
    };
    if (attr(hipDeviceAttributeMultiprocessorCount)) g.cus = v; else ok = false;
    if (attr(hipDeviceAttributeWarpSize)) g.wave = v; else ok = false;
    if (attr(hipDeviceAttributeMaxSharedMemoryPerMultiprocessor)) g.lds_per_cu = (size_t)v; else ok = false;
    // what one workgroup can be given: the whole CU's LDS on CDNA (opt-in above the default)
    g.max_dyn_lds = g.lds_per_cu;
    if (attr(hipDeviceAttributeMaxSharedMemoryPerBlock)) {
      g.default_dyn_lds = (size_t)v < g.lds_per_cu ? (size_t)v : g.lds_per_cu;
      if (g.default_dyn_lds > 64 * 1024) g.default_dyn_lds = 64 * 1024;  // opt-in needed above 64 KiB
    }
    g.queried = ok ? 1 : 0;
    if (const char* e = getenv("I2LQR_FAKE_CUS")) {
      const long f = strtol(e, nullptr, 10);
      if (f >= 1 && f <= 4096) { g.cus = (int)f; g.faked = 1; }
    }
    table[dev] = g;
  });
  return table[dev];
}

#ifdef I2LQR_DEBUG
unsigned long long* debug_trap_word() {
  constexpr int kMaxDev = 64;
  static unsigned long long* word[kMaxDev] = {};
  static std::mutex mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDev) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  if (!word[dev]) {
    if (hipMalloc((void**)&word[dev], sizeof(unsigned long long)) != hipSuccess) return nullptr;
    (void)hipMemset(word[dev], 0, sizeof(unsigned long long));
  }
  return word[dev];
}
#else
unsigned long long* debug_trap_word() { return nullptr; }
#endif
}  // namespace i2lqr

#ifdef I2LQR_DEBUG
__global__ void k_debug_self_test(double* buf, unsigned long long* trap) {
  const auto s = i2lqr::make_slice(buf, 8, trap, i2lqr::TAG_WAVE_LDS);
  if (threadIdx.x == 0) s[8] = 1.0;  // one past the end: recorded, redirected to s[0]
}
#endif

namespace {
// Debug build: after a call's launches, wait for the stream and turn a recorded index violation
// into I2LQR_ERR_LAUNCH (the record is cleared).  The product build returns rc untouched and does
// not synchronise.
int debug_check(int rc, void* stream) {
#ifdef I2LQR_DEBUG
  if (rc != I2LQR_OK) return rc;
  unsigned long long* w = i2lqr::debug_trap_word();
  if (!w) return rc;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing((hipStream_t)stream, &cap) == hipSuccess &&
      cap != hipStreamCaptureStatusNone)
    return rc;  // a stream being captured into a graph cannot be waited for: the next eager call checks
  if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess)
    return fail(I2LQR_ERR_LAUNCH, "debug build: the stream reported an error");
  unsigned long long rec = 0;
  if (hipMemcpy(&rec, w, sizeof(rec), hipMemcpyDeviceToHost) != hipSuccess)
    return fail(I2LQR_ERR_LAUNCH, "debug build: could not read the violation record");
  if (rec) {
    (void)hipMemset(w, 0, sizeof(rec));
    static const char* const names[] = {"?", "wave-kernel LDS slice", "eight-lane LDS slice",
                                        "sixteen-lane LDS slice", "quad12 HBM workspace slot",
                                        "lane kernel row of X", "lane kernel row of U / k",
                                        "lane kernel row of K", "lane kernel LDS word",
                                        "compaction row / problem index"};
    const int tag = (int)((rec >> 56) & 0x7f);
    return fail(I2LQR_ERR_LAUNCH, "debug build: index check failed: %s, index %lld, limit %lld",
                names[tag < 10 ? tag : 0], (long long)((rec >> 28) & 0xfffffff),
                (long long)(rec & 0xfffffff));
  }
#else
  (void)stream;
#endif
  return rc;
}
}  // namespace

struct i2lqr_handle {
  i2lqr_config cfg;
  int lanes;        // lanes of a wavefront that cooperate on one problem (1: batch-minor layout)
  size_t lds_bytes; // dynamic LDS per workgroup (one wavefront)
  int device;
  void* ws;         // caller-owned scratch of the batch-minor kernels
  int64_t ws_bytes;
  int64_t compact_min_batch;  // i2lqr_solve uses the chunked, compacting form from this batch; 0: never; -1: automatic
  // scheduling options of the one-problem-per-lane kernels (i2lqr_set_option); -1 = automatic
  int opt_defer, opt_reroll, opt_lds_steps, opt_merge, opt_ckpt, opt_stagger;
  int wave_tail;  // chunked solve: finish <= this many survivors with one problem per wavefront (0: off, -1: automatic)
  int opt_pair;  // bicycles' lane kernel (fp64 and fp32, with or without stage weights): workgroups of two wavefronts (main + helper); -1 = automatic
  int opt_two_x;  // ... its second state buffer (no re-roll of accepted steps); -1 = automatic
  int opt_chunk_step;  // chunked solve: length of the chunk that follows the first (automatic: 4); a schedule to measure against
  int opt_fuse;  // chunked solve: compaction folded into the chunk kernels' exit (round 6); 0: k_lane_compact launches; -1 = automatic (on)
  int opt_final_round;  // chunked solve: the round whose tail kernel takes every survivor and ends the schedule; 0: never; -1 = automatic
  int opt_first_chunk;  // chunked solve: pinned length of the first chunk, no extension chunks (a hand-tuned schedule to measure the data-driven one against); -1 = automatic
  int opt_fstep;  // one-problem-per-wavefront kernel: per-step Jacobian matrices in LDS; -1 = automatic
  int opt_group;  // problem-major layout: lanes per problem of the fused kernels: 8, 64; -1 = automatic
  int opt_spec;   // eight-lane kernel: speculative form (2-3 wavefronts per eight problems); -1 = automatic
  int opt_group_ws;  // eight-lane kernel: workspace form (records / gains in HBM); -1 = automatic
  // i2lqr_iterate_pick: the epilogue a call asks for; `fused` is set by the launcher that folded
  // it into its kernel, otherwise the call runs the separate kernels.  The pointer to the call's
  // epilogue is THREAD-LOCAL (t_epi below), not handle state: two host threads inside
  // i2lqr_iterate_pick on one handle do not see each other's.
  struct Epilogue {
    const int32_t* qfun;
    int outer_iter, max_relax_iter;
    void* cost_it;
    void* part;
    int64_t* best_idx;
    void* best_cost;
    bool fused;
  };
  // Device words of the last-workgroup-done reduction (each wraps to 0 by itself when its launch
  // has drawn all its tickets).  kTickets words, handed out round-robin: calls on one handle that
  // are in flight at the same time (two streams) draw from different words as long as fewer than
  // kTickets of them overlap; their part[] workspaces are the caller's and must differ.
  static constexpr unsigned kTickets = 16;
  unsigned* ticket;
  std::atomic<unsigned> ticket_next;
  // i2lqr_sharded_round_flat: the event that orders the side stream behind the shard's solve and
  // the one a later round waits for before it reuses the buffers (created on first use)
  hipEvent_t ev_ready, ev_side_done;
  bool side_pending;  // ev_side_done has been recorded at least once
  DeviceGeometry geo;  // of h->device, queried in i2lqr_create (i2lqr_geometry.hpp)
};

namespace {
thread_local i2lqr_handle::Epilogue* t_epi = nullptr;
}

namespace {

// Which fused kernel a problem-major call runs on: ONE function, used by the launchers and by
// i2lqr_iterate_kernel / i2lqr_solve_kernel (what bench.py labels its results with).
enum FusedKernel { K_WAVE, K_GROUP, K_GROUP16, K_GROUP_WS, K_SPEC, K_SPEC16, K_QUAD, K_INVALID };
// (measured on the 256-CU chip; DeviceGeometry::scaled() elsewhere)
constexpr int64_t kAutoGroupBatch = 1024;  // eight-lane kernel from here (automatic)
constexpr int64_t kAutoSpecBatch = 12288;  // speculative form for solves up to here (automatic)

// *why: the message of K_INVALID (a forced option the configuration cannot honour)
FusedKernel select_fused(const i2lqr_handle* h, int64_t B, bool early_exit, const char** why) {
  static const char* none = "";
  if (!why) why = &none;
  const bool m2 = h->cfg.m == 2 && h->cfg.n + h->cfg.m <= 8;
  const bool q16 = h->cfg.n + h->cfg.m == 16;
  if (m2) {
    // Eight lanes per problem, eight problems per wavefront (i2lqr_group.hpp).  Automatic where it
    // is built (the bicycles, Q = R = 0) from 1024 problems: below that every problem gets a SIMD
    // of its own either way and the one-problem-per-wavefront kernel's iteration is ~8 % shorter
    // (tools/group_ab.py: 0.195 vs 0.212 ms per 10 iterations at 64-256 problems, 0.219 vs 0.213
    // at 1024, 0.63 vs 0.25 at 4096).
    const bool can = group_supported(h->cfg);
    if (h->opt_group == 8 && !can) {
      *why = "\"group_lanes\" = 8 needs a bicycle plant with Q = R = 0 and a horizon whose eight "
             "problem slices fit a CU's LDS";
      return K_INVALID;
    }
    // Speculative form (k_group_spec): two or three wavefronts per eight problems run the
    // iterations that follow 0, 1, (2) rejects at once; bit-identical results.  With a FIXED
    // iteration count it measures slower than the plain kernel (0.275 vs 0.215 ms per 10
    // iterations at 1024 problems), so it stays opt-in there.  Automatic for solves to
    // termination (early_exit) of at most kAutoSpecBatch problems: the launch lasts as long as
    // its slowest problem, and the slowest problems alternate accepts and rejects — i2lqr_solve
    // 0.60 -> 0.44 ms at 16 problems, 1.21 -> 0.71 ms at 1024, 1.23 -> 1.17 at 8192.
    // (sixteen lanes per problem — the DPP passes, four problems per workgroup — wherever its
    // buffers fit; "group_lanes" 8 pins the eight-lane form)
    const bool can_spec16 = group_spec_supported(h->cfg, 16), can_spec8 = group_spec_supported(h->cfg, 8);
    const bool spec16 = h->opt_group == 16 || (h->opt_group < 0 && can_spec16);
    const bool can_spec = spec16 ? can_spec16 : can_spec8;
    if (h->opt_spec == 1 && !can_spec) {
      *why = "\"speculate\" = 1 needs the eight- / sixteen-lane kernel and a horizon whose speculative "
             "buffers fit a CU's LDS";
      return K_INVALID;
    }
    if (can_spec && h->opt_group != 64 &&
        (h->opt_spec == 1 ||
         (h->opt_spec < 0 && h->opt_group < 0 && early_exit && B <= h->geo.scaled(kAutoSpecBatch))))
      return spec16 ? K_SPEC16 : K_SPEC;
    // Sixteen lanes per problem (one problem per DPP row; GroupWorker::backward_row / forward_row):
    // no LDS round trip in the serial chains.  Four problems per wavefront: automatic for every
    // batch that still leaves each wavefront a SIMD of its own (up to kGroup16Batch problems) —
    // its iteration is shorter than the one-problem-per-wavefront kernel's at ANY size (0.145
    // against 0.188 ms per 10 iterations from 1 to 512 problems, n = 6, N = 20; tools/_diag);
    // "group_lanes" 16 / 8 / 64 pins the choice.
    const bool can16 = group16_supported(h->cfg);  // (four slices: longer horizons than `can`)
    if (h->opt_group == 16 && !can16) {
      *why = "\"group_lanes\" = 16 needs a bicycle plant with Q = R = 0 and a horizon whose four "
             "problem slices fit a CU's LDS";
      return K_INVALID;
    }
    // Beyond one round of 1024 wavefronts the sixteen-lane kernel runs in rounds (0.168 ms per 4096
    // problems); the eight-lane workspace form (four wavefronts of eight problems per CU) is the
    // better choice only between 4096 and kGroupWsTop problems — and only with the caller's
    // workspace registered.
    const int64_t need = can ? group_workspace_bytes(h->cfg, B) : 0;
    const bool have_ws = need > 0 && h->ws && h->ws_bytes >= need;
    const bool ws_range = B > h->geo.scaled(kGroupWsBatch) && B <= h->geo.scaled(group_ws_top(h->cfg));
    if (h->opt_group == 16 ||
        (h->opt_group < 0 && can16 && h->opt_group_ws != 1 && !(ws_range && can && have_ws)))
      return K_GROUP16;
    if (h->opt_group == 8 || (h->opt_group < 0 && can && B >= h->geo.scaled(kAutoGroupBatch))) {
      // "group_workspace" 0 / 1 pins the choice of the eight-lane form
      if (h->opt_group_ws == 1 && !have_ws) {
        *why = "\"group_workspace\" = 1 needs a registered workspace of i2lqr_workspace_bytes() "
               "for this batch";
        return K_INVALID;
      }
      if (have_ws && (h->opt_group_ws == 1 ||
                      (h->opt_group_ws < 0 && (ws_range || (h->opt_group == 8 && B > h->geo.scaled(kGroupWsBatch))))))
        return K_GROUP_WS;
      return K_GROUP;
    }
  } else if (h->opt_group == 8) {
    *why = "\"group_lanes\" = 8 is built for the m = 2 plants only";
    return K_INVALID;
  }
  if (q16) {
    // Sixteen lanes per problem, four problems per wavefront (i2lqr_quad.hpp): needs the caller's
    // workspace (i2lqr_workspace_bytes); automatic whenever it is registered.
    const bool can = quad_supported(h->cfg);
    const bool have_ws = h->ws && h->ws_bytes >= quad_workspace_bytes(h->cfg, B);
    if (h->opt_group == 16 && !(can && have_ws)) {
      *why = "\"group_lanes\" = 16 needs Q = R = 0 and a registered workspace of "
             "i2lqr_workspace_bytes() for this batch";
      return K_INVALID;
    }
    if ((h->opt_group == 16 || h->opt_group < 0) && can && have_ws) return K_QUAD;
  } else if (h->opt_group == 16 && !m2) {
    *why = "\"group_lanes\" = 16 is built for the bicycles (DPP row form) and the n + m = 16 plant";
    return K_INVALID;
  }
  return K_WAVE;
}

// One launcher per (dtype, system); LANES fixed at 64 = one problem per wavefront.
template <class T, class Sys> struct Launch {
  static constexpr int n = Sys::n, m = Sys::m, LANES = 64;
  using Cfg = DevCfg<T, n, m>;

  static size_t lds_bytes(int N) { return (size_t)Layout<Sys>(N).total * sizeof(T) * (64 / LANES); }
  static size_t fstep_lds_bytes(int N) {
    return (size_t)Layout<Sys>(N, true).total * sizeof(T) * (64 / LANES);
  }
  static constexpr bool kHasFstep = Sys::n <= 6;  // the bicycles; quad12's F is 1.5 KB per step
  static unsigned grid(int64_t B) { return (unsigned)((B + (64 / LANES) - 1) / (64 / LANES)); }

  static int prepare(i2lqr_handle* h) {
    h->lanes = LANES;
    h->lds_bytes = lds_bytes(h->cfg.N);
    if (h->lds_bytes > h->geo.max_dyn_lds)
      return fail(I2LQR_ERR_UNSUPPORTED, "horizon %d needs %zu B of LDS per wavefront (> %zu KiB)",
                  h->cfg.N, h->lds_bytes, h->geo.max_dyn_lds / 1024);
    if (h->lds_bytes > h->geo.default_dyn_lds) {
      const int bytes = (int)h->lds_bytes;
      HIP_TRY(hipFuncSetAttribute((const void*)k_iterate<T, Sys, LANES, false>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
      HIP_TRY(hipFuncSetAttribute((const void*)k_iterate<T, Sys, LANES, true>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
      HIP_TRY(hipFuncSetAttribute((const void*)k_backward<T, Sys, LANES, false>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
      HIP_TRY(hipFuncSetAttribute((const void*)k_backward<T, Sys, LANES, true>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
      HIP_TRY(hipFuncSetAttribute((const void*)k_forward<T, Sys, LANES, false>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
      HIP_TRY(hipFuncSetAttribute((const void*)k_forward<T, Sys, LANES, true>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
      HIP_TRY(hipFuncSetAttribute((const void*)k_rollout<T, Sys, LANES, false>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
      HIP_TRY(hipFuncSetAttribute((const void*)k_rollout<T, Sys, LANES, true>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    }
    return I2LQR_OK;
  }

  static int iterate(i2lqr_handle* h, int64_t B, int n_iters, int early_exit, void* X, void* U,
                     const void* x_term, void* lamb, const void* obs, void* cost, void* K, void* k,
                     int32_t* iters, int32_t* status, hipStream_t s) {
    const Cfg c = make_dev_cfg<T, n, m>(h->cfg);
    IterArgs<T> a;
    a.B = B;
    a.n_iters = n_iters;
    a.early_exit = early_exit;
    a.X = (T*)X;
    a.U = (T*)U;
    a.x_term = (const T*)x_term;
    a.lamb = (T*)lamb;
    a.obs = (const T*)obs;
    a.cost = (T*)cost;
    a.K = (T*)K;
    a.k = (T*)k;
    a.iters = iters;
    a.status = status;
    a.dbg = nullptr;
    a.count = nullptr; a.count_max = 0; a.max_total = 0; a.set_stride = 0;
#ifdef I2LQR_STAMPS
    a.dbg = (unsigned long long*)h->ws;  // diagnostic build: caller registers [B][8] u64 here
#endif
    const char* why = "";
    const FusedKernel fk = select_fused(h, B, early_exit != 0, &why);
    if (t_epi && (fk == K_GROUP || fk == K_GROUP16 || fk == K_GROUP_WS)) {  // the eight-lane kernels carry the epilogue
      a.qfun = t_epi->qfun;
      a.outer_iter = t_epi->outer_iter;
      a.max_relax_iter = t_epi->max_relax_iter;
      a.cost_it = (T*)t_epi->cost_it;
      if (t_epi->best_idx) {
        a.pick_part = (MinPair<T>*)t_epi->part;
        a.pick_ticket = h->ticket + h->ticket_next.fetch_add(1, std::memory_order_relaxed) %
                                        i2lqr_handle::kTickets;
        a.best_idx = t_epi->best_idx;
        a.best_cost = (T*)t_epi->best_cost;
      }
      t_epi->fused = true;
    }
    switch (fk) {
      case K_INVALID:
        return fail(I2LQR_ERR_UNSUPPORTED, "%s", why);
      case K_SPEC:
      case K_SPEC16:
        if constexpr (m == 2 && n + m <= 8) {
          HIP_TRY(group_spec_iterate<T>(h->cfg, a, s, fk == K_SPEC16 ? 16 : 8));
          return I2LQR_OK;
        }
        break;
      case K_GROUP:
        if constexpr (m == 2 && n + m <= 8) {
          HIP_TRY(group_iterate<T>(h->cfg, a, s));
          return I2LQR_OK;
        }
        break;
      case K_GROUP16:
        if constexpr (m == 2 && n + m <= 8) {
          HIP_TRY(group16_iterate<T>(h->cfg, a, s));
          return I2LQR_OK;
        }
        break;
      case K_GROUP_WS:
        if constexpr (m == 2 && n + m <= 8) {
          HIP_TRY(group_iterate_ws<T>(h->cfg, a, h->ws, s));
          return I2LQR_OK;
        }
        break;
      case K_QUAD:
        if constexpr (n + m == 16) {
          HIP_TRY(quad_iterate<T>(h->cfg, a, h->ws, s));
          return I2LQR_OK;
        }
        break;
      case K_WAVE:
        break;
    }
    // Per-step F matrices (prep() writes them in parallel over t; the serial recursion then has no
    // Jacobian refresh) double the LDS slice: taken when every wavefront of the launch still fits
    // on the chip at once, i.e. in the latency-bound regime this variant exists for.
    const size_t lds_f = fstep_lds_bytes(h->cfg.N);
    const int64_t waves_per_cu = (grid(B) + h->geo.cus - 1) / h->geo.cus;
    const bool fstep = kHasFstep && lds_f <= h->geo.default_dyn_lds &&
                       (h->opt_fstep >= 0 ? h->opt_fstep != 0
                                          : waves_per_cu * lds_f <= h->geo.lds_per_cu - 10 * 1024);
    if constexpr (kHasFstep) {
      if (fstep) {
        if (c.flags)
          hipLaunchKernelGGL((k_iterate<T, Sys, LANES, true, true>), dim3(grid(B)), dim3(64), lds_f,
                             s, c, a);
        else
          hipLaunchKernelGGL((k_iterate<T, Sys, LANES, false, true>), dim3(grid(B)), dim3(64), lds_f,
                             s, c, a);
        HIP_TRY(hipGetLastError());
        return I2LQR_OK;
      }
    }
    if (c.flags)
      hipLaunchKernelGGL((k_iterate<T, Sys, LANES, true>), dim3(grid(B)), dim3(64), h->lds_bytes, s,
                         c, a);
    else
      hipLaunchKernelGGL((k_iterate<T, Sys, LANES, false>), dim3(grid(B)), dim3(64), h->lds_bytes,
                         s, c, a);
    HIP_TRY(hipGetLastError());
    return I2LQR_OK;
  }
  static int names_pair(i2lqr_handle*, int64_t, int) { return 0; }
  // L chains of k problems each in ONE launch (k_group_spec<.., CHAIN>)
  static int solve_chained(i2lqr_handle* h, int64_t L, int k, void* X, void* U, const void* x_term,
                           void* lamb, const void* obs, void* cost, void* K, void* kk, int32_t* iters,
                           int32_t* status, hipStream_t s) {
    if constexpr (m == 2 && n + m <= 8) {
      if (h->opt_spec == 0 || h->opt_group == 8 || h->opt_group == 64 ||
          !group_spec_chain_supported(h->cfg, L))
        return fail(I2LQR_ERR_UNSUPPORTED, "chains run on the sixteen-lane speculative kernel: a "
                    "bicycle plant with Q = R = 0, at most %lld chains, a horizon whose three-wavefront "
                    "buffers fit a CU's LDS (solve the chain steps one after the other instead)",
                    (long long)h->geo.scaled(512));
      IterArgs<T> a;
      a.B = L; a.n_iters = h->cfg.max_iter; a.early_exit = 1;
      a.X = (T*)X; a.U = (T*)U; a.x_term = (const T*)x_term; a.lamb = (T*)lamb;
      a.obs = (const T*)obs; a.cost = (T*)cost; a.K = (T*)K; a.k = (T*)kk;
      a.iters = iters; a.status = status; a.dbg = nullptr;
      a.count = nullptr; a.count_max = 0; a.max_total = 0; a.set_stride = 0;
      a.chain_len = k;
      HIP_TRY(group_spec_chain<T>(h->cfg, a, s));
      return I2LQR_OK;
    }
    return fail(I2LQR_ERR_UNSUPPORTED, "chains are built for the m = 2 plants");
  }
  static int rollout(i2lqr_handle* h, int64_t B, void* X, void* U, const void* x_term, void* cost,
                     hipStream_t s) {
    const Cfg c = make_dev_cfg<T, n, m>(h->cfg);
    if (c.flags)
      hipLaunchKernelGGL((k_rollout<T, Sys, LANES, true>), dim3(grid(B)), dim3(64), h->lds_bytes, s,
                         c, B, (T*)X, (T*)U, (const T*)x_term, (T*)cost);
    else
      hipLaunchKernelGGL((k_rollout<T, Sys, LANES, false>), dim3(grid(B)), dim3(64), h->lds_bytes,
                         s, c, B, (T*)X, (T*)U, (const T*)x_term, (T*)cost);
    HIP_TRY(hipGetLastError());
    return I2LQR_OK;
  }
  static int backward(i2lqr_handle* h, int64_t B, const void* X, const void* U, const void* x_term,
                      const void* lamb, const void* obs, void* K, void* k, hipStream_t s) {
    const Cfg c = make_dev_cfg<T, n, m>(h->cfg);
    if (c.flags)
      hipLaunchKernelGGL((k_backward<T, Sys, LANES, true>), dim3(grid(B)), dim3(64), h->lds_bytes,
                         s, c, B, (const T*)X, (const T*)U, (const T*)x_term, (const T*)lamb,
                         (const T*)obs, (T*)K, (T*)k);
    else
      hipLaunchKernelGGL((k_backward<T, Sys, LANES, false>), dim3(grid(B)), dim3(64), h->lds_bytes,
                         s, c, B, (const T*)X, (const T*)U, (const T*)x_term, (const T*)lamb,
                         (const T*)obs, (T*)K, (T*)k);
    HIP_TRY(hipGetLastError());
    return I2LQR_OK;
  }
  static int forward(i2lqr_handle* h, int64_t B, const void* X, const void* U, const void* x_term,
                     const void* K, const void* k, void* Xn, void* Un, void* cost_new,
                     hipStream_t s) {
    const Cfg c = make_dev_cfg<T, n, m>(h->cfg);
    if (c.flags)
      hipLaunchKernelGGL((k_forward<T, Sys, LANES, true>), dim3(grid(B)), dim3(64), h->lds_bytes, s,
                         c, B, (const T*)X, (const T*)U, (const T*)x_term, (const T*)K,
                         (const T*)k, (T*)Xn, (T*)Un, (T*)cost_new);
    else
      hipLaunchKernelGGL((k_forward<T, Sys, LANES, false>), dim3(grid(B)), dim3(64), h->lds_bytes,
                         s, c, B, (const T*)X, (const T*)U, (const T*)x_term, (const T*)K,
                         (const T*)k, (T*)Xn, (T*)Un, (T*)cost_new);
    HIP_TRY(hipGetLastError());
    return I2LQR_OK;
  }
};

// Batch-minor / batch-tiled layouts: one problem per lane (i2lqr_lane.hpp).
template <class T, class Sys, bool TILED> struct LaneLaunch {
  // (measured on the 256-CU chip: DeviceGeometry::scaled() where they are used)
  static constexpr int kAutoWaveTail = 2048;
  static constexpr int kAutoSpecTail = 12288;  // (round 5: 8192; tools/solve_bench.py --plans)
  static constexpr int64_t kAutoCompactBatch = 4096;
  static constexpr int n = Sys::n, m = Sys::m, NT = Sys::NTRIG;
  using Cfg = DevCfg<T, n, m>;
  static unsigned grid(int64_t B) { return (unsigned)((B + 63) / 64); }
  // Workspace (bytes) for B problems: candidate trajectory + gains scratch (every call), and for
  // the chunked solve two compacted work sets, scratch iters/status and one counter per round.
  static constexpr int kMaxRounds = 24;  // compaction rounds of the chunked solve (one counter each)
  static constexpr int kFirstChunk = 8;
  static constexpr int kFinalRound = 3;  // chunked solve: the round (at 24 iterations) whose tail takes everything
  static int64_t set_words(int N) { return (int64_t)(n * (N + 1) + m * N + n + 6 + 2); }
  static int64_t ws_bytes(int N, int64_t B) {
    const int64_t Bp = TILED ? (B + 63) / 64 * 64 : B;
    // (+ the second state buffer of the helper-wavefront form: its own rows since round 6 — the
    // chunks' exits pack survivors into the work set the chunk does not run on, which used to lend
    // its states)
    const int64_t words = lane_workspace_words<Sys>(N, Bp) + 2 * set_words(N) * Bp +
                          Bp * (int64_t)(n * (N + 1));
    return words * (int64_t)sizeof(T) + (2 * 3 + 2) * Bp * 4 + 16 + kMaxRounds * 4;
  }
  static int prepare(i2lqr_handle* h) {
    h->lanes = 1;
    h->lds_bytes = 0;
    return I2LQR_OK;
  }
  // stage weights Q, R != 0: the bicycles in both precisions; the row-block plant (quad12) in fp64
  // (round 5: instantiations of their own, i2lqr_lane12qr.hip)
  static constexpr bool kStageWeights = Sys::NBLK == 0 || sizeof(T) == 8;
  static int need_ws(i2lqr_handle* h, int64_t B) {
    if constexpr (!kStageWeights) {
      bool hasqr = false;
      for (int i = 0; i < I2LQR_MAX_N * I2LQR_MAX_N && !hasqr; i++) hasqr = h->cfg.Q[i] != 0.0;
      for (int i = 0; i < I2LQR_MAX_M * I2LQR_MAX_M && !hasqr; i++) hasqr = h->cfg.R[i] != 0.0;
      if (hasqr)
        return fail(I2LQR_ERR_UNSUPPORTED, "the one-problem-per-lane kernels of this plant are built "
                    "for Q = R = 0 (use the problem-major layout for stage weights)");
    }
    if (TILED && (B & 63))
      return fail(I2LQR_ERR_INVALID, "the batch-tiled layout needs a batch that is a multiple of "
                  "64 (got %lld)", (long long)B);
    const int64_t need = ws_bytes(h->cfg.N, B);
    if (!h->ws || h->ws_bytes < need)
      return fail(I2LQR_ERR_INVALID, "workspace of %lld B registered, batch %lld needs %lld B "
                  "(i2lqr_workspace_bytes / i2lqr_set_workspace)", (long long)h->ws_bytes,
                  (long long)B, (long long)need);
    return I2LQR_OK;
  }
  struct Carved {
    LaneSet<T> set[2];
    int32_t* uiters;
    int32_t* ustatus;
    int32_t* count;  // [kMaxRounds]: the live count after each compaction round
  };
  static void carve(i2lqr_handle* h, int64_t B, LaneArgs<T>& a, Carved* cv = nullptr) {
    const int N = h->cfg.N;
    // (no workspace registered: names_pair() carves on scratch arguments that are never launched —
    // a non-null base keeps the pointer arithmetic defined)
    static char no_ws[16];
    T* p = (T*)(h->ws ? h->ws : (void*)no_ws);
    a.wsU = p; p += B * (int64_t)(m * N);
    a.wsK = p; p += B * (int64_t)(m * n * N);
    a.wsk = p; p += B * (int64_t)(m * N);
    // second state buffer of the helper-wavefront form (rows of its own behind the gains scratch)
    a.wsX = h->opt_two_x != 0 ? p : nullptr;
    p += B * (int64_t)(n * (N + 1));
    a.two_max = 64 * (int)(h->opt_two_x == 1 ? pair_max_grid(h->geo) : two_x_max_grid(h->geo));
    a.count_lo = -1;
    a.count_hi = 0x7fffffff;
    a.count = nullptr;
    a.resume = 0;
    a.max_total = 0x7fffffff;
    // The gains of the first steps stay in LDS: as many steps as keep the register-limited number
    // of wavefronts per CU (one per SIMD, four per CU) resident in the 160 KiB: 36 KiB each, i.e.
    // 5 steps in fp64 and 10 in fp32 at n=6, m=2.
    const int per_step = 64 * m * (n + 1) * (int)sizeof(T);
    a.lds_steps = ((int)h->geo.lds_per_simd_wave() - kK0Bytes) / per_step;  // steps 1..lds_steps; k_0 sits behind them
    if (a.lds_steps < 0) a.lds_steps = 0;
    if (a.lds_steps > N - 1) a.lds_steps = N - 1;
    // (batch sizes in units of the chip: full = one wavefront per SIMD, 65536 problems on 256 CUs)
    const int64_t full = h->geo.full_batch();
    a.reroll = B >= full / 2 ? 1 : 0;  // pays only where the kernel sits on the HBM roof (32768)
    a.defer = 1;  // the forward pass stores no states; accepted steps re-roll them (see i2lqr_lane.hpp)
    // ... and merge the accepted candidate inputs into the one input buffer.  fp64 (HBM-bound):
    // 7.56 -> 7.23 KB per problem-iteration, +4.8 % it/s at 2^20 problems; fp32 (instruction-bound):
    // the extra row writes cost 6 %, so the per-lane buffer swap stays (tools/ab_bench.py).  Below
    // ~40000 problems the fp64 kernel is instruction-bound as well (one or two wavefronts per CU):
    // 226 -> 237 M it/s at 12800 problems without the merge, 411 -> 426 at 24576, 528 -> 545 at
    // 32768, +-0 from 40960 to 57344, 811 -> 846 WITH it at 65536.
    a.merge = sizeof(T) == 8 && B > full / 2 ? 1 : 0;
    if (h->opt_merge >= 0) a.merge = h->opt_merge;
    a.ckpt = 0;  // decided in finish_options() once the other options are final
    a.stagger = 0;
    std::memset(&a.cp, 0, sizeof(a.cp));  // plain exit; solve_compacting() fills it per chunk
    if constexpr (Sys::NBLK > 0) {
      // automatic where the launch fills the chip (one wavefront per SIMD: 65536 problems)
      a.stagger = B >= full ? 45 : 0;
    } else if (sizeof(T) == 8 && B > full * 15 / 16 && B <= full * 5 / 4) {
      // bicycles, fp64, a launch of about one wavefront per SIMD: +3 % (883 -> 912 M it/s at 65536
      // problems, tools/ab_bench.py --cold); fp32 (issue-bound) gains nothing, launches of two
      // and more rounds desynchronise by themselves and only pay the delay (-2 %)
      a.stagger = 8;
    }
    if (h->opt_stagger >= 0) a.stagger = h->opt_stagger > 999 ? 999 : h->opt_stagger;
    a.dbg = nullptr;
#ifdef I2LQR_STAMPS
    if (const char* e = getenv("I2LQR_DBG_PTR")) a.dbg = (unsigned long long*)strtoull(e, nullptr, 0);
#endif
    if (h->opt_reroll >= 0) a.reroll = h->opt_reroll;
    if (h->opt_defer >= 0) a.defer = h->opt_defer;
    if (h->opt_lds_steps >= 0 && h->opt_lds_steps < a.lds_steps) a.lds_steps = h->opt_lds_steps;
    a.lds_grow = h->opt_lds_steps < 0 ? 1 : 0;
    finish_options(h, B, a);
    if (!cv) return;
    for (int q = 0; q < 2; q++) {
      LaneSet<T>& st = cv->set[q];
      st.B = B;
      st.X = p; p += B * (int64_t)(n * (N + 1));
      st.U = p; p += B * (int64_t)(m * N);
      st.x_term = p; p += B * n;
      st.obs = p; p += B * 6;
      st.lamb = p; p += B;
      st.cost = p; p += B;
      st.K = a.wsK;
      st.k = a.wsk;
    }
    int32_t* ip = (int32_t*)(((uintptr_t)p + 15) & ~(uintptr_t)15);
    for (int q = 0; q < 2; q++) {
      cv->set[q].iters = ip; ip += B;
      cv->set[q].status = ip; ip += B;
      cv->set[q].orig = ip; ip += B;
    }
    cv->uiters = ip; ip += B;
    cv->ustatus = ip; ip += B;
    cv->count = ip;
  }
  // LDS of the state checkpointing: the states of one segment of kSeg steps for 64 lanes
  static constexpr int kSegBytes = (kSeg + 1) * n * 64 * (int)sizeof(T);
  // k_0 of 64 lanes: step 0 keeps only its feed-forward term in LDS (LaneWorker::lds_k0)
  static constexpr int kK0Bytes = 64 * m * (int)sizeof(T);
  // State checkpointing (i2lqr_lane.hpp): fp64 with deferred, merged states and the re-rolling
  // forward pass, Q = R = 0; automatic from 65536 problems (where the kernel sits on the HBM
  // roof).  Its segment buffer takes the place of LDS-resident gain steps.  Must be called again
  // after any later change of defer / reroll / merge (the early-exit path does).
  static void finish_options(const i2lqr_handle* h, int64_t B, LaneArgs<T>& a) {
    bool hasqr = false;
    for (int i = 0; i < I2LQR_MAX_N * I2LQR_MAX_N && !hasqr; i++) hasqr = h->cfg.Q[i] != 0.0;
    for (int i = 0; i < I2LQR_MAX_M * I2LQR_MAX_M && !hasqr; i++) hasqr = h->cfg.R[i] != 0.0;
    const bool can = sizeof(T) == 8 && !hasqr && a.defer && a.merge && a.reroll && h->cfg.N >= 2;
    // (tools/ab_bench.py --cold, round 3: -2 % at 65536 problems, +-0 at 131072, +5 % at 262144,
    // +4-6 % at 2^20; HBM bytes per problem-iteration 6993 -> 6243 (1.41 -> 1.26 x algorithmic):
    // on from 65536 problems, BASELINE's roofline batch and half a per-GPU shard of configs[3] —
    // where the kernel already sits on the HBM roof the bytes are what is left to win)
    a.ckpt = can && (h->opt_ckpt >= 0 ? h->opt_ckpt != 0 : B >= h->geo.full_batch());
    if (a.ckpt) {
      const int per_step = 64 * m * (n + 1) * (int)sizeof(T);
      int steps = ((int)h->geo.lds_per_simd_wave() + 1024 - kSegBytes - kK0Bytes) / per_step;
      if (steps < 0) steps = 0;
      if (a.lds_steps > steps) a.lds_steps = steps;
    }
  }
  // The helper-wavefront form (k_lane_iterate_pair; round 5): the bicycles in fp64 and fp32 (stage
  // weights: its HASQR instantiations), states not checkpointed, and a launch of at most 512 workgroups (32768 problems) - two wavefronts per
  // workgroup then still find a SIMD each.  Properties of the chip: half as many workgroups as it
  // has SIMDs (512 on 256 CUs); the second state buffer up to one workgroup per CU.
  static unsigned pair_max_grid(const DeviceGeometry& g) { return (unsigned)(g.simds() / 2); }
  static unsigned two_x_max_grid(const DeviceGeometry& g) { return (unsigned)g.cus; }
  static bool pair_built(const Cfg& c, const LaneArgs<T>& a, int opt_pair) {
    if constexpr (Sys::NBLK == 0) return !a.ckpt && opt_pair != 0;
    return false;
  }
  static bool use_pair(const DeviceGeometry& g, const Cfg& c, const LaneArgs<T>& a, int64_t B,
                       int opt_pair) {
    return pair_built(c, a, opt_pair) && (opt_pair == 1 || grid(B) <= pair_max_grid(g));
  }
  static size_t lane_lds(const LaneArgs<T>& a) {
    return (size_t)a.lds_steps * 64 * m * (n + 1) * sizeof(T) + kK0Bytes + (a.ckpt ? kSegBytes : 0);
  }
  // A launch of at most 256 / 512 workgroups puts one / two of them on a CU, so each has 160 / 80
  // KiB of LDS to itself where a full launch budgets 36 (four wavefronts per CU): the gains of that
  // many more horizon steps stay in LDS between the backward and the forward pass instead of going
  // through HBM (fp64, n = 6, m = 2: 19 of 20 steps up to 256 workgroups, 8-10 up to 512).
  // lds_steps of a launch of `workgroups` workgroups with `fixed` bytes of other dynamic LDS each,
  // at most max_dyn bytes of dynamic LDS per workgroup
  static int grown_lds_steps(const DeviceGeometry& g, const Cfg& c, const LaneArgs<T>& a,
                             unsigned workgroups, size_t fixed, size_t max_dyn) {
    if (!a.lds_grow || a.ckpt) return a.lds_steps;
    const unsigned per_cu = (workgroups + (unsigned)g.cus - 1) / (unsigned)g.cus;
    size_t budget = g.lds_per_cu / (per_cu ? per_cu : 1);
    if (budget > max_dyn) budget = max_dyn;
    const size_t per_step = (size_t)64 * m * (n + 1) * sizeof(T);
    const size_t need0 = fixed + kK0Bytes + 1024;  // (1 KiB: allocation granularity)
    int steps = budget > need0 ? (int)((budget - need0) / per_step) : 0;
    if (steps > c.N - 1) steps = c.N - 1;
    return steps > a.lds_steps ? steps : a.lds_steps;
  }
  // More than 64 KiB of dynamic LDS per workgroup has to be asked for; the attribute belongs to
  // the kernel ON THE CURRENT DEVICE, so it is set by every launch that needs it (a process may
  // drive several devices) and a refusal falls back to what 64 KiB hold.
  template <class K>
  static void grow_lds(const DeviceGeometry& g, K kernel, const Cfg& c, LaneArgs<T>& a,
                       unsigned workgroups, size_t fixed) {
    const int before = a.lds_steps;
    a.lds_steps = grown_lds_steps(g, c, a, workgroups, fixed, g.max_dyn_lds);
    if (lane_lds(a) + fixed <= g.default_dyn_lds) return;
    if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)g.max_dyn_lds) == hipSuccess)
      return;
    (void)hipGetLastError();
    a.lds_steps = before;
    a.lds_steps = grown_lds_steps(g, c, a, workgroups, fixed, g.default_dyn_lds);
  }
  template <bool TL>
  static void launch_pair(const DeviceGeometry& g, const Cfg& c, const LaneArgs<T>& a,
                          unsigned workgroups, hipStream_t s) {
    if constexpr (Sys::NBLK == 0) {
      LaneArgs<T> ap = a;
      if (c.flags) {  // stage weights: the record carries 2 Q (x_t - xtarget) as well
        using LW = LaneWorker<T, Sys, true, TL>;
        const size_t fixed = 2 * LW::kRec * 64 * sizeof(T) + 65 * sizeof(int) + 12;
        grow_lds(g, &k_lane_iterate_pair<T, Sys, true, TL>, c, ap, workgroups, fixed);
        hipLaunchKernelGGL((k_lane_iterate_pair<T, Sys, true, TL>), dim3(workgroups), dim3(128),
                           lane_lds(ap) + fixed, s, c, ap);
      } else {
        using LW = LaneWorker<T, Sys, false, TL>;
        const size_t fixed = 2 * LW::kRec * 64 * sizeof(T) + 65 * sizeof(int) + 12;
        grow_lds(g, &k_lane_iterate_pair<T, Sys, false, TL>, c, ap, workgroups, fixed);
        hipLaunchKernelGGL((k_lane_iterate_pair<T, Sys, false, TL>), dim3(workgroups), dim3(128),
                           lane_lds(ap) + fixed, s, c, ap);
      }
    }
  }
  template <bool TL>
  static void launch_iterate(const DeviceGeometry& g, const Cfg& c, const LaneArgs<T>& a, int64_t B,
                             hipStream_t s, int opt_pair = 0) {
    if (use_pair(g, c, a, B, opt_pair)) {
      launch_pair<TL>(g, c, a, grid(B), s);
      return;
    }
    if constexpr (Sys::NBLK > 0) {  // row-block plants: their own fused kernel, no LDS
      bool launched = false;
      if constexpr (kStageWeights) {  // (need_ws() has refused stage weights where they are not built)
        if (c.flags) {
          hipLaunchKernelGGL((k_lane_iterate_rows<T, Sys, true, TL>), dim3(grid(B)), dim3(64), 0, s, c, a);
          launched = true;
        }
      }
      if (!launched)
        hipLaunchKernelGGL((k_lane_iterate_rows<T, Sys, false, TL>), dim3(grid(B)), dim3(64), 0, s, c, a);
    } else {
      LaneArgs<T> ag = a;
      if (c.flags) {
        grow_lds(g, &k_lane_iterate<T, Sys, true, TL>, c, ag, grid(B), 0);
        hipLaunchKernelGGL((k_lane_iterate<T, Sys, true, TL>), dim3(grid(B)), dim3(64),
                           lane_lds(ag), s, c, ag);
      } else {
        grow_lds(g, &k_lane_iterate<T, Sys, false, TL>, c, ag, grid(B), 0);
        hipLaunchKernelGGL((k_lane_iterate<T, Sys, false, TL>), dim3(grid(B)), dim3(64),
                           lane_lds(ag), s, c, ag);
      }
    }
  }

  // The schedule of the chunked solve is STRUCTURAL and its decisions are the device's (round 5;
  // VERDICT r4 #3 — round 4 sized the first chunk from the bench workload's survivor curve, a
  // constant fitted to workloads.make_batch): a first chunk of kFirstChunk iterations on the whole
  // batch (never shorter: the tail kernel sums in another order than the lane kernels, and the
  // more iterations of a long-horizon problem it runs the further the two solves drift apart in
  // the last digits — 2.3e-8 relative at N = 50 with a first chunk of 6 against the suite's 1e-8),
  // then rounds of { compaction, tail kernel, lane chunk } with chunk lengths 4, then as many
  // iterations as are done (compaction points 8, 12, 24, 48, 96).  Which of the two kernels of a
  // round does the work is decided ON THE DEVICE from the live count the compaction leaves: the
  // tail kernel is a no-op above its cap ("wave_tail"), the lane chunk skips what the tail has
  // finished.  Measured against hand-tuned first chunks of 8 / 10 / 12 / 14 on three
  // distributions (the bench's; every problem with the obstacle; targets twice as far) at
  // 16384 ... 262144 problems (tools/solve_bench.py --plans, profiles/r05_solve_schedule.txt):
  // within 15 % of the best hand-tuned schedule in all 18 cases (worst: 32768 problems, every
  // problem with the obstacle, 1.144; 65536: 1.033 / 1.081 / 1.062).  Tried and dropped: up to three in-place
  // extension chunks of 2 iterations gated on the survivor count of the chunk before (ahead at
  // <= 32768 problems on the hard distribution, 13-41 % behind from 131072, where a chunk over
  // the whole batch is several rounds of wavefronts); chunks of 2 behind the first.
  // Chunked solve with compaction (large batches): ilqr() runs 1..max_iter iterations per
  // problem, so a wavefront of 64 problems would otherwise idle on its slowest lane.  The batch
  // is solved in chunks of 8, 4, then doubling; after every chunk the terminated problems
  // are scattered to the caller's arrays and the survivors are packed into a dense work set
  // (k_lane_compact).  No host synchronisation: the live count stays in device memory and
  // surplus wavefronts exit at once.  Results are bit-identical to the plain launch.
  static int solve_compacting(i2lqr_handle* h, int64_t B, void* X, void* U, const void* x_term,
                              void* lamb, const void* obs, void* cost, void* K, void* k,
                              int32_t* iters, int32_t* status, hipStream_t s) {
    if (int rc = need_ws(h, B)) return rc;
    const Cfg c = make_dev_cfg<T, n, m>(h->cfg);
    const int N = h->cfg.N, max_iter = h->cfg.max_iter;
    LaneArgs<T> a0;
    Carved cv;
    carve(h, B, a0, &cv);
    LaneSet<T> usr;
    usr.B = B;
    usr.X = (T*)X; usr.U = (T*)U; usr.x_term = (T*)x_term; usr.obs = (T*)obs;
    usr.lamb = (T*)lamb; usr.cost = (T*)cost; usr.K = (T*)K; usr.k = (T*)k;
    usr.iters = iters ? iters : cv.uiters;
    usr.status = status ? status : cv.ustatus;
    usr.orig = nullptr;
    if (!obs) { cv.set[0].obs = nullptr; cv.set[1].obs = nullptr; }
    // Latency tail: once few problems survive, the rest of the solve is bound by the slowest
    // problem's iteration latency, which is ~2.8x lower with one problem per WAVEFRONT.  If the
    // packed set holds <= wave_tail problems the one-problem-per-wavefront kernel finishes them
    // (it reads the live count itself and is a no-op otherwise); the lane chunks that follow skip
    // finished problems and the next compaction scatters them to the caller.
    // The survivors are the problems with long accept / reject chains (stragglers alternate
    // accept, reject, accept, ...): the speculative sixteen-lane kernel runs the iteration after
    // a reject beside the current one and needs about half the rounds; bit-identical to the
    // plain kernel.  It takes over from 12288 survivors (workgroups of four problems whose slowest
    // member decides; the hardware backfills), the one-problem-per-wavefront kernel (other
    // plants, stage weights) from 2048.
    bool spec_tail = false;
    if constexpr (m == 2 && n + m <= 8)
      spec_tail = h->opt_spec != 0 && c.flags == 0 && group_spec_tail_supported(h->cfg);
    const int wave_tail = h->wave_tail < 0
        ? (int)h->geo.scaled(spec_tail ? kAutoSpecTail : kAutoWaveTail) : h->wave_tail;
    // chunk 0 runs in place on the caller's arrays
    const int first = h->opt_first_chunk > 0 ? h->opt_first_chunk : kFirstChunk;
    int done = 0, len = max_iter < first ? max_iter : first;
    const int step = h->opt_chunk_step > 0 ? h->opt_chunk_step : 4;
    a0.B = B; a0.n_iters = len; a0.early_exit = 1;
    a0.X = usr.X; a0.U = usr.U; a0.x_term = usr.x_term; a0.lamb = usr.lamb; a0.obs = usr.obs;
    a0.cost = usr.cost; a0.K = usr.K; a0.k = usr.k; a0.iters = usr.iters; a0.status = usr.status;
    a0.max_total = max_iter;
    // one live counter per compaction round, all cleared by ONE fill in front of the first chunk (a
    // fill per round was a 5 us launch of its own in each of the rounds that follow the tail)
    HIP_TRY(hipMemsetAsync(cv.count, 0, kMaxRounds * sizeof(int32_t), s));
    // Round 6: the compaction is folded into the chunk kernels' EXIT (LaneCompact: a wavefront that
    // has finished its chunk packs its survivors into the next work set and scatters what
    // terminated, one atomic add per wavefront, while the other wavefronts still iterate) — the
    // k_lane_compact launch between every two chunks and the final scatter pass are gone
    // ("fused_compaction" 0 brings them back: the A/B and the fallback).
    const bool fused = h->opt_fuse != 0;
    auto plan_exit = [&](LaneArgs<T>& a, bool src_user, const int32_t* orig, LaneSet<T>* dst,
                         int32_t* count_out) {
      std::memset(&a.cp, 0, sizeof(a.cp));
      if (!fused) return;
      a.cp.on = 1;
      a.cp.src_is_user = src_user ? 1 : 0;
      a.cp.user_tiled = TILED ? 1 : 0;
      a.cp.orig = orig;
      if (dst) a.cp.dst = *dst;
      a.cp.count_out = count_out;
      a.cp.usr = usr;
    };
    int round = 0;  // counter of the work set the NEXT compaction fills
    int cur = 0;    // ... and its index
    const bool only_chunk = len >= max_iter;  // (max_iter <= first chunk: nothing survives it)
    plan_exit(a0, true, nullptr, only_chunk ? nullptr : &cv.set[cur], cv.count + round);
    launch_iterate<TILED>(h->geo, c, a0, B, s, h->opt_pair);
    done += len;
    const unsigned cgrid = (unsigned)((B + 255) / 256);
    bool src_user = true;
    const int32_t* count_in = nullptr;
    LaneSet<T> src = usr;
    // The LAST round (automatic: the one at 24 iterations; "final_round"): its tail kernel takes
    // whatever is left, however many, and runs it to termination — no lane chunk, no further
    // round is enqueued behind it.  Every round after the one whose tail ran used to be three or
    // four empty launches (14-19 us each round; 65536 problems of the bench workload: the rounds at
    // 24, 48 and 96 iterations).  What survives 24 iterations are the long accept / reject chains
    // the speculative kernel is built for, and their iteration counts spread over 25 ... 150: a
    // lane chunk would idle on its slowest lane.
    const int final_round = (!spec_tail || wave_tail <= 0) ? 0
        : (h->opt_final_round >= 0 ? h->opt_final_round : kFinalRound);
    while (done < max_iter) {
      int32_t* const count = cv.count + round;
      if (!fused)
        hipLaunchKernelGGL((k_lane_compact<T, TILED>), dim3(cgrid), dim3(256), 0, s, n, m, N, src,
                           src_user ? 1 : 0, count_in, cv.set[cur], count, usr, c.trap);
      const LaneSet<T>& w = cv.set[cur];
      const bool last_round = final_round > 0 && round + 1 >= final_round;
      if (wave_tail > 0 && done >= 4) {
        using WL = Launch<T, Sys>;
        const size_t lds_f = WL::fstep_lds_bytes(N);
        IterArgs<T> t;
        t.B = B; t.n_iters = max_iter; t.early_exit = 1;
        t.X = w.X; t.U = w.U; t.x_term = w.x_term; t.lamb = w.lamb; t.obs = w.obs;
        t.cost = w.cost; t.K = w.K; t.k = w.k; t.iters = w.iters; t.status = w.status;
        t.dbg = nullptr;
        t.count = count; t.count_max = last_round ? (int)B : wave_tail; t.max_total = max_iter;
        t.set_stride = B;
        if (spec_tail) {
          // the speculative tail delivers its problems to the caller's arrays itself (the
          // scatter pass of the next compaction was 82 us for 7938 problems with gains)
          t.orig = w.orig;
          t.out_X = usr.X; t.out_U = usr.U; t.out_K = usr.K; t.out_k = usr.k;
          t.out_lamb = usr.lamb; t.out_cost = usr.cost;
          t.out_iters = usr.iters; t.out_status = usr.status;
          t.out_B = usr.B; t.out_tiled = TILED ? 1 : 0;
          if constexpr (m == 2 && n + m <= 8) HIP_TRY(group_spec_tail<T>(h->cfg, t, s));
        } else if (WL::kHasFstep && lds_f <= h->geo.default_dyn_lds) {
          if constexpr (WL::kHasFstep) {
            if (c.flags)
              hipLaunchKernelGGL((k_iterate<T, Sys, 64, true, true, true>), dim3(wave_tail),
                                 dim3(64), lds_f, s, c, t);
            else
              hipLaunchKernelGGL((k_iterate<T, Sys, 64, false, true, true>), dim3(wave_tail),
                                 dim3(64), lds_f, s, c, t);
          }
        }
      }
      if (last_round && spec_tail) {  // everything has been delivered by the tail kernel
        src_user = true;  // (nothing left to scatter)
        break;
      }
      // chunk lengths after the first: one chunk of 4 (the tail's second chance), then as many
      // iterations as are done (first chunk 10: compaction points at 10, 14, 28, 56 iterations).
      // Once the tail has run, every further round is three empty launches (14 us), so few of
      // them (the schedule 8, 4, 4, 8, 8, 16, 16, 32, 32, ... spent 0.19 of 2.13 ms there at 65536
      // problems; 10, 4, 4, 10, 20, 40, 12: 85 us of 1.55 ms).  Against 4, 4, 4, 4, ... with a tail
      // of 2048: 1.69 -> 1.35 ms at 16384 problems.
      len = done < 12 ? step : done;
      if (done + len > max_iter || round + 2 >= kMaxRounds) len = max_iter - done;
      LaneArgs<T> a = a0;
      a.X = w.X; a.U = w.U; a.x_term = w.x_term; a.lamb = w.lamb; a.obs = w.obs; a.cost = w.cost;
      a.K = nullptr; a.k = nullptr;  // gains of work sets go to the scratch buffer (w.K == wsK)
      a.iters = w.iters; a.status = w.status;
      a.count = count; a.resume = 1; a.n_iters = len;
      // the chunk's exit packs its survivors into the OTHER work set (none behind the last chunk)
      plan_exit(a, false, w.orig, done + len >= max_iter ? nullptr : &cv.set[cur ^ 1],
                cv.count + round + 1);
      // a batch too large for the helper-wavefront form as a whole: its survivors may not be.
      // Both kernels are enqueued and the live count decides on the device which one runs (the
      // other is an empty launch, ~3 us): 21000 survivors of 65536 problems iterate at 46
      // instead of 67 us per iteration.  The pair's chunk takes the options of a launch of its
      // size (no state checkpoints, no merge, no re-rolling forward pass: all bit-identical).
      LaneArgs<T> ap = a;
      ap.ckpt = 0;
      ap.merge = h->opt_merge >= 0 ? h->opt_merge : 0;
      ap.reroll = h->opt_reroll >= 0 ? h->opt_reroll : 0;
      const unsigned pair_max = pair_max_grid(h->geo);
      if (pair_built(c, ap, h->opt_pair) && h->opt_pair < 0 && grid(B) > pair_max) {
        ap.count_hi = 64 * (int)pair_max;
        launch_pair<false>(h->geo, c, ap, pair_max, s);
        a.count_lo = ap.count_hi;
        launch_iterate<false>(h->geo, c, a, B, s, 0);
      } else {
        launch_iterate<false>(h->geo, c, a, B, s, h->opt_pair);
      }
      done += len;
      src = w;
      src_user = false;
      count_in = count;
      cur ^= 1;
      round++;
    }
    // every remaining problem has a terminal status now: scatter them all
    if (!fused && !src_user) {
      LaneSet<T> none;
      std::memset(&none, 0, sizeof(none));
      hipLaunchKernelGGL((k_lane_compact<T, TILED>), dim3(cgrid), dim3(256), 0, s, n, m, N, src, 0,
                         count_in, none, cv.count + round, usr, c.trap);
    }
    HIP_TRY(hipGetLastError());
    return I2LQR_OK;
  }

  static int iterate(i2lqr_handle* h, int64_t B, int n_iters, int early_exit, void* X, void* U,
                     const void* x_term, void* lamb, const void* obs, void* cost, void* K, void* k,
                     int32_t* iters, int32_t* status, hipStream_t s) {
    // automatic: chunked solve with the speculative tail from 4096 problems and more than 16
    // iterations allowed (measured 1.2-1.9x on the bench workload from 4096 to 262144 problems;
    // the chunks alone cost ~9 % when nothing terminates early)
    const int64_t cmin = h->compact_min_batch < 0
        ? (h->cfg.max_iter > 16 ? h->geo.scaled(kAutoCompactBatch) : 0) : h->compact_min_batch;
    if (early_exit && cmin > 0 && B >= cmin && n_iters > 4 && n_iters == h->cfg.max_iter)
      return solve_compacting(h, B, X, U, x_term, lamb, obs, cost, K, k, iters, status, s);
    if (int rc = need_ws(h, B)) return rc;
    const Cfg c = make_dev_cfg<T, n, m>(h->cfg);
    LaneArgs<T> a;
    carve(h, B, a);
    a.B = B; a.n_iters = n_iters; a.early_exit = early_exit;
    a.X = (T*)X; a.U = (T*)U; a.x_term = (const T*)x_term; a.lamb = (T*)lamb;
    a.obs = (const T*)obs; a.cost = (T*)cost; a.K = (T*)K; a.k = (T*)k;
    a.iters = iters; a.status = status;
    a.max_total = n_iters;
    if (early_exit) {  // solve() is bound by its slowest problem's latency, not by HBM traffic
      if (h->opt_reroll < 0) a.reroll = 0;
      if (h->opt_defer < 0) a.defer = 0;
      finish_options(h, B, a);
    }
    launch_iterate<TILED>(h->geo, c, a, B, s, h->opt_pair);
    HIP_TRY(hipGetLastError());
    return I2LQR_OK;
  }
  // 1 if the (first) launch of i2lqr_iterate / i2lqr_solve for B problems is the helper-wavefront
  // kernel: the SAME carve + option + use_pair sequence the launchers above run, on scratch
  // arguments (no launch, no device access) — i2lqr_iterate_kernel / i2lqr_solve_kernel (ADVICE r5:
  // a hand-written mirror of these conditions had drifted).
  static int solve_chained(i2lqr_handle*, int64_t, int, void*, void*, const void*, void*, const void*,
                           void*, void*, void*, int32_t*, int32_t*, hipStream_t) {
    return fail(I2LQR_ERR_UNSUPPORTED, "chains are problem-major batches (the controller's path)");
  }
  static int names_pair(i2lqr_handle* h, int64_t B, int early_exit) {
    const Cfg c = make_dev_cfg<T, n, m>(h->cfg);
    LaneArgs<T> a;
    carve(h, B, a);
    const int64_t cmin = h->compact_min_batch < 0
        ? (h->cfg.max_iter > 16 ? h->geo.scaled(kAutoCompactBatch) : 0) : h->compact_min_batch;
    const bool chunked = early_exit && cmin > 0 && B >= cmin && h->cfg.max_iter > 4;
    if (early_exit && !chunked) {  // (the single-launch solve: iterate() above)
      if (h->opt_reroll < 0) a.reroll = 0;
      if (h->opt_defer < 0) a.defer = 0;
      finish_options(h, B, a);
    }
    return use_pair(h->geo, c, a, B, h->opt_pair) ? 1 : 0;
  }
  static int rollout(i2lqr_handle* h, int64_t B, void* X, void* U, const void* x_term, void* cost,
                     hipStream_t s) {
    if (int rc = need_ws(h, B)) return rc;
    const Cfg c = make_dev_cfg<T, n, m>(h->cfg);
    LaneArgs<T> a;
    carve(h, B, a);
    bool launched = false;
    if constexpr (kStageWeights) {
      if (c.flags) {
        hipLaunchKernelGGL((k_lane_rollout<T, Sys, true, TILED>), dim3(grid(B)), dim3(64), 0, s, c, B,
                         (T*)X, (T*)U, (const T*)x_term, (T*)cost);
        launched = true;
      }
    }
    if (!launched)
      hipLaunchKernelGGL((k_lane_rollout<T, Sys, false, TILED>), dim3(grid(B)), dim3(64), 0, s, c, B,
                         (T*)X, (T*)U, (const T*)x_term, (T*)cost);
    HIP_TRY(hipGetLastError());
    return I2LQR_OK;
  }
  static int backward(i2lqr_handle* h, int64_t B, const void* X, const void* U, const void* x_term,
                      const void* lamb, const void* obs, void* K, void* k, hipStream_t s) {
    if (int rc = need_ws(h, B)) return rc;
    const Cfg c = make_dev_cfg<T, n, m>(h->cfg);
    LaneArgs<T> a;
    carve(h, B, a);
    bool launched = false;
    if constexpr (kStageWeights) {
      if (c.flags) {
        hipLaunchKernelGGL((k_lane_backward<T, Sys, true, TILED>), dim3(grid(B)), dim3(64), 0, s, c, B,
                         (const T*)X, (const T*)U, (const T*)x_term, (const T*)lamb,
                         (const T*)obs, (T*)K, (T*)k);
        launched = true;
      }
    }
    if (!launched)
      hipLaunchKernelGGL((k_lane_backward<T, Sys, false, TILED>), dim3(grid(B)), dim3(64), 0, s, c, B,
                         (const T*)X, (const T*)U, (const T*)x_term, (const T*)lamb,
                         (const T*)obs, (T*)K, (T*)k);
    HIP_TRY(hipGetLastError());
    return I2LQR_OK;
  }
  static int forward(i2lqr_handle* h, int64_t B, const void* X, const void* U, const void* x_term,
                     const void* K, const void* k, void* Xn, void* Un, void* cost_new,
                     hipStream_t s) {
    if (int rc = need_ws(h, B)) return rc;
    const Cfg c = make_dev_cfg<T, n, m>(h->cfg);
    LaneArgs<T> a;
    carve(h, B, a);
    bool launched = false;
    if constexpr (kStageWeights) {
      if (c.flags) {
        hipLaunchKernelGGL((k_lane_forward<T, Sys, true, TILED>), dim3(grid(B)), dim3(64), 0, s, c, B,
                         (const T*)X, (const T*)U, (const T*)x_term, (const T*)K, (const T*)k,
                         (T*)Xn, (T*)Un, (T*)cost_new);
        launched = true;
      }
    }
    if (!launched)
      hipLaunchKernelGGL((k_lane_forward<T, Sys, false, TILED>), dim3(grid(B)), dim3(64), 0, s, c, B,
                         (const T*)X, (const T*)U, (const T*)x_term, (const T*)K, (const T*)k,
                         (T*)Xn, (T*)Un, (T*)cost_new);
    HIP_TRY(hipGetLastError());
    return I2LQR_OK;
  }
};

// dispatch over (layout, dtype, system)
#define I2LQR_DISPATCH(h, CALL)                                                               \
  do {                                                                                        \
    const int sid_ = (h)->cfg.system_id;                                                      \
    if ((h)->cfg.layout == I2LQR_LAYOUT_BATCH_MINOR) {                                        \
      if ((h)->cfg.dtype == I2LQR_F64) {                                                      \
        if (sid_ == I2LQR_SYS_BICYCLE4)                                                       \
          return LaneLaunch<double, Bicycle4<double>, false>::CALL;                           \
        if (sid_ == I2LQR_SYS_BICYCLE6)                                                       \
          return LaneLaunch<double, Bicycle6<double>, false>::CALL;                           \
        if (sid_ == I2LQR_SYS_QUAD12)                                                         \
          return LaneLaunch<double, Quad12<double>, false>::CALL;                             \
      } else {                                                                                \
        if (sid_ == I2LQR_SYS_BICYCLE4) return LaneLaunch<float, Bicycle4<float>, false>::CALL; \
        if (sid_ == I2LQR_SYS_BICYCLE6) return LaneLaunch<float, Bicycle6<float>, false>::CALL; \
        if (sid_ == I2LQR_SYS_QUAD12) return LaneLaunch<float, Quad12<float>, false>::CALL;     \
      }                                                                                       \
      return fail(I2LQR_ERR_UNSUPPORTED, "system %d is not built for the batch-minor layout", \
                  sid_);                                                                      \
    }                                                                                         \
    if ((h)->cfg.layout == I2LQR_LAYOUT_BATCH_TILED) {                                        \
      if ((h)->cfg.dtype == I2LQR_F64) {                                                      \
        if (sid_ == I2LQR_SYS_BICYCLE4)                                                       \
          return LaneLaunch<double, Bicycle4<double>, true>::CALL;                            \
        if (sid_ == I2LQR_SYS_BICYCLE6)                                                       \
          return LaneLaunch<double, Bicycle6<double>, true>::CALL;                            \
        if (sid_ == I2LQR_SYS_QUAD12)                                                         \
          return LaneLaunch<double, Quad12<double>, true>::CALL;                              \
      } else {                                                                                \
        if (sid_ == I2LQR_SYS_BICYCLE4) return LaneLaunch<float, Bicycle4<float>, true>::CALL; \
        if (sid_ == I2LQR_SYS_BICYCLE6) return LaneLaunch<float, Bicycle6<float>, true>::CALL; \
        if (sid_ == I2LQR_SYS_QUAD12) return LaneLaunch<float, Quad12<float>, true>::CALL;      \
      }                                                                                       \
      return fail(I2LQR_ERR_UNSUPPORTED, "system %d is not built for the batch-tiled layout", \
                  sid_);                                                                      \
    }                                                                                         \
    if ((h)->cfg.dtype == I2LQR_F64) {                                                        \
      if (sid_ == I2LQR_SYS_BICYCLE4) return Launch<double, Bicycle4<double>>::CALL;          \
      if (sid_ == I2LQR_SYS_BICYCLE6) return Launch<double, Bicycle6<double>>::CALL;          \
      if (sid_ == I2LQR_SYS_QUAD12) return Launch<double, Quad12<double>>::CALL;              \
    } else {                                                                                  \
      if (sid_ == I2LQR_SYS_BICYCLE4) return Launch<float, Bicycle4<float>>::CALL;            \
      if (sid_ == I2LQR_SYS_BICYCLE6) return Launch<float, Bicycle6<float>>::CALL;            \
      if (sid_ == I2LQR_SYS_QUAD12) return Launch<float, Quad12<float>>::CALL;                \
    }                                                                                         \
    return fail(I2LQR_ERR_UNSUPPORTED, "system %d / dtype %d not built", sid_,               \
                (h)->cfg.dtype);                                                              \
  } while (0)

int prepare_dispatch(i2lqr_handle* h) { I2LQR_DISPATCH(h, prepare(h)); }
int names_pair_dispatch(i2lqr_handle* h, int64_t B, int early_exit) {
  I2LQR_DISPATCH(h, names_pair(h, B, early_exit));
}
int dispatch_solve_chained(i2lqr_handle* h, int64_t L, int k, void* X, void* U, const void* x_term,
                           void* lamb, const void* obs, void* cost, void* K, void* kk, int32_t* iters,
                           int32_t* status, void* stream) {
  I2LQR_DISPATCH(h, solve_chained(h, L, k, X, U, x_term, lamb, obs, cost, K, kk, iters, status,
                                  (hipStream_t)stream));
}
int dispatch_rollout(i2lqr_handle* h, int64_t B, void* X, void* U, const void* x_term, void* cost,
                     void* stream) {
  I2LQR_DISPATCH(h, rollout(h, B, X, U, x_term, cost, (hipStream_t)stream));
}
int dispatch_backward(i2lqr_handle* h, int64_t B, const void* X, const void* U, const void* x_term,
                      const void* lamb, const void* obs, void* K, void* k, void* stream) {
  I2LQR_DISPATCH(h, backward(h, B, X, U, x_term, lamb, obs, K, k, (hipStream_t)stream));
}
int dispatch_forward(i2lqr_handle* h, int64_t B, const void* X, const void* U, const void* x_term,
                     const void* K, const void* k, void* X_new, void* U_new, void* cost_new,
                     void* stream) {
  I2LQR_DISPATCH(h, forward(h, B, X, U, x_term, K, k, X_new, U_new, cost_new,
                            (hipStream_t)stream));
}
int dispatch_iterate(i2lqr_handle* h, int64_t B, int n_iters, int early_exit, void* X, void* U,
                     const void* x_term, void* lamb, const void* obs, void* cost, void* K, void* k,
                     int32_t* iters, int32_t* status, void* stream) {
  I2LQR_DISPATCH(h, iterate(h, B, n_iters, early_exit, X, U, x_term, lamb, obs, cost, K, k, iters,
                            status, (hipStream_t)stream));
}

// Live handles: i2lqr_destroy of NULL is a no-op, of a pointer that is not (or no longer) a live
// handle an error code instead of a double free.
std::mutex g_live_mu;
std::unordered_set<const i2lqr_handle*>& live_handles() {
  static std::unordered_set<const i2lqr_handle*> set;
  return set;
}

int check_common(const i2lqr_handle* h, int64_t B) {
  if (!h) return fail(I2LQR_ERR_INVALID, "null handle");
  if (B < 0) return fail(I2LQR_ERR_INVALID, "negative batch %lld", (long long)B);
  if (B > (int64_t)0x7fffffff) return fail(I2LQR_ERR_INVALID, "batch %lld too large", (long long)B);
  return I2LQR_OK;
}

// ---- arg-min kernels (flat, first index wins ties) -------------------------------------------
// FINAL: a single workgroup scans the whole vector and writes the result itself (small batches:
// one launch instead of two)
template <class T, bool FINAL = false>
__global__ __launch_bounds__(256) void k_argmin_partial(int64_t B, const T* cost, MinPair<T>* part,
                                                        int64_t* best_idx = nullptr,
                                                        T* best_cost = nullptr) {
  __shared__ T sv[256];
  __shared__ int64_t si[256];
  T bv = T(0);
  int64_t bi = -1;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < B; i += (int64_t)gridDim.x * 256) {
    const T v = cost[i];
    if (better(v, i, bv, bi)) { bv = v; bi = i; }
  }
  sv[threadIdx.x] = bv;
  si[threadIdx.x] = bi;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      const T ov = sv[threadIdx.x + s];
      const int64_t oi = si[threadIdx.x + s];
      if (oi >= 0 && better(ov, oi, sv[threadIdx.x], si[threadIdx.x])) {
        sv[threadIdx.x] = ov;
        si[threadIdx.x] = oi;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    if constexpr (FINAL) {
      *best_idx = si[0];
      *best_cost = si[0] >= 0 ? sv[0] : (T)INFINITY;
    } else {
      part[blockIdx.x].v = sv[0];
      part[blockIdx.x].i = si[0];
    }
  }
}

template <class T>
__global__ __launch_bounds__(256) void k_argmin_final(int nparts, const MinPair<T>* part,
                                                      int64_t* best_idx, T* best_cost) {
  __shared__ T sv[256];
  __shared__ int64_t si[256];
  T bv = T(0);
  int64_t bi = -1;
  for (int p = threadIdx.x; p < nparts; p += 256) {
    const T v = part[p].v;
    const int64_t i = part[p].i;
    if (i >= 0 && better(v, i, bv, bi)) { bv = v; bi = i; }
  }
  sv[threadIdx.x] = bv;
  si[threadIdx.x] = bi;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      const T ov = sv[threadIdx.x + s];
      const int64_t oi = si[threadIdx.x + s];
      if (oi >= 0 && better(ov, oi, sv[threadIdx.x], si[threadIdx.x])) {
        sv[threadIdx.x] = ov;
        si[threadIdx.x] = oi;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    *best_idx = si[0];
    *best_cost = si[0] >= 0 ? sv[0] : (T)INFINITY;
  }
}

// ---- sharded round (i2lqr_sharded_round_flat) --------------------------------------------------
// What a rank contributes to the exchange, ONE launch: workgroup 0 packs the trajectory of the
// shard's pick (k_pack_problem's gather; zeros for a rank without candidates), and — ragged split
// only (padded non-null) — all workgroups copy the B costs into the `width` slots every rank
// gathers and fill the rest with +inf, which never wins the pick.
template <class T>
__global__ __launch_bounds__(256) void k_round_prepare(int64_t B, int n, int m, int N, int layout,
                                                       const T* X, const T* U, const int64_t* idx,
                                                       T* pack, int64_t width, const T* cost_it,
                                                       T* padded) {
  if (padded)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < width; i += (int64_t)gridDim.x * 256)
      padded[i] = i < B ? cost_it[i] : (T)INFINITY;
  if (blockIdx.x != 0) return;
  const int nu = m * N, nx = n * (N + 1);
  if (B <= 0) {
    for (int e = threadIdx.x; e < nu + nx; e += 256) pack[e] = T(0);
    return;
  }
  int64_t b = idx[0];
  if (b < 0) b = 0;
  if (b >= B) b = B - 1;
  auto fetch = [&](const T* A, int comps, int T_, int c, int t) -> T {
    if (layout == 0) return A[(b * comps + c) * T_ + t];
    if (layout == 1) return A[((int64_t)t * comps + c) * B + b];
    return A[(((b >> 6) * T_ + t) * comps + c) * 64 + (b & 63)];
  };
  for (int e = threadIdx.x; e < nu; e += 256) pack[e] = fetch(U, m, N, e / N, e % N);
  for (int e = threadIdx.x; e < nx; e += 256)
    pack[nu + e] = fetch(X, n, N + 1, e / (N + 1), e % (N + 1));
}

// The pick over the gathered costs and the winner's hand-off in ONE workgroup: k_argmin_partial<T,
// true> (same total order: the flat arg-min with first-index tie-break, NaN never wins) followed by
// k_round_winner's translation and copy.
template <class T>
__global__ __launch_bounds__(256) void k_round_pick(int world, int64_t width, int64_t total,
                                                    int64_t pack_count, const T* cost_all,
                                                    const T* pack_all, T* best_cost, T* winner,
                                                    int64_t* best_global) {
  __shared__ T sv[256];
  __shared__ int64_t si[256];
  const int64_t G = (int64_t)world * width;
  T bv = T(0);
  int64_t bi = -1;
  for (int64_t i = threadIdx.x; i < G; i += 256) {
    const T v = cost_all[i];
    if (better(v, i, bv, bi)) { bv = v; bi = i; }
  }
  sv[threadIdx.x] = bv;
  si[threadIdx.x] = bi;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) {
      const T ov = sv[threadIdx.x + s];
      const int64_t oi = si[threadIdx.x + s];
      if (oi >= 0 && better(ov, oi, sv[threadIdx.x], si[threadIdx.x])) {
        sv[threadIdx.x] = ov;
        si[threadIdx.x] = oi;
      }
    }
    __syncthreads();
  }
  const int64_t p = si[0];
  const int64_t q = p < 0 ? 0 : p;
  const int64_t owner = q / width, loc = q - owner * width;
  const int64_t base = total / world, rem = total - base * world;
  const int64_t lo = owner * base + (owner < rem ? owner : rem);
  if (threadIdx.x == 0) {
    *best_cost = p >= 0 ? sv[0] : (T)INFINITY;
    best_global[0] = p < 0 ? -1 : lo + loc;
    best_global[1] = owner;
  }
  for (int64_t e = threadIdx.x; e < pack_count; e += 256)
    winner[e] = pack_all[owner * pack_count + e];
}

// ---- RCCL, bound at run time ---------------------------------------------------------------
// The one collective of the path is an all-gather of the candidates' terminal
// costs.  libi2lqr_hip.so does not link librccl: a process that already carries a copy (PyTorch
// ships its own librccl.so, the one torch.distributed's "nccl" backend uses) must not get a
// second one, and single-GPU users need none at all.  The first communicator call looks for a
// loaded librccl first and loads the ROCm one otherwise.
struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t,
                            hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t,
                            hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;  // optional: without them the two all-gathers of a
  ncclResult_t (*GroupEnd)() = nullptr;    // round are issued one after the other
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  bool ok = false;
  char why[256] = "";  // the loader's message, captured once (dlerror() is cleared by reading it)
};

const RcclApi& rccl_api() {
  static const RcclApi api = [] {
    RcclApi a;
    for (const char* name : {"librccl.so", "librccl.so.1"})
      if ((a.lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD))) break;
    if (!a.lib)
      for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
        if ((a.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!a.lib) {
      const char* e = dlerror();
      snprintf(a.why, sizeof(a.why), "%s", e ? e : "librccl.so not found by the dynamic loader");
      return a;
    }
    auto sym = [&](const char* n) { return dlsym(a.lib, n); };
    a.GetUniqueId = (decltype(a.GetUniqueId))sym("ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))sym("ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))sym("ncclCommDestroy");
    a.CommAbort = (decltype(a.CommAbort))sym("ncclCommAbort");
    a.CommCount = (decltype(a.CommCount))sym("ncclCommCount");
    a.CommUserRank = (decltype(a.CommUserRank))sym("ncclCommUserRank");
    a.AllGather = (decltype(a.AllGather))sym("ncclAllGather");
    a.Broadcast = (decltype(a.Broadcast))sym("ncclBroadcast");
    a.GroupStart = (decltype(a.GroupStart))sym("ncclGroupStart");
    a.GroupEnd = (decltype(a.GroupEnd))sym("ncclGroupEnd");
    a.GetErrorString = (decltype(a.GetErrorString))sym("ncclGetErrorString");
    a.ok = a.GetUniqueId && a.CommInitRank && a.CommDestroy && a.CommAbort && a.CommCount &&
           a.CommUserRank && a.AllGather && a.Broadcast && a.GetErrorString;
    if (!a.ok) snprintf(a.why, sizeof(a.why), "the loaded librccl lacks a symbol this library binds");
    return a;
  }();
  return api;
}

#define RCCL_TRY(api, expr)                                                                 \
  do {                                                                                      \
    ncclResult_t r_ = (expr);                                                               \
    if (r_ != ncclSuccess)                                                                  \
      return fail(I2LQR_ERR_LAUNCH, "%s failed: %s", #expr, (api).GetErrorString(r_));      \
  } while (0)

constexpr int kArgminBlocks = 256;
constexpr int64_t kArgminSingle = 16384;  // up to here a single workgroup scans the vector

}  // namespace

// ---------------------------------------------------------------------------------------------
extern "C" {

int i2lqr_version(void) { return I2LQR_ABI_VERSION; }

const char* i2lqr_last_error(void) { return g_err; }

int i2lqr_config_default(i2lqr_config* cfg, int system_id, int num_horizon) {
  if (!cfg) return fail(I2LQR_ERR_INVALID, "null cfg");
  std::memset(cfg, 0, sizeof(*cfg));
  cfg->struct_size = (int32_t)sizeof(*cfg);
  cfg->N = num_horizon;
  cfg->dtype = I2LQR_F64;
  cfg->layout = I2LQR_LAYOUT_PROBLEM_MAJOR;
  cfg->system_id = system_id;
  cfg->max_iter = 150;
  cfg->dt = 1.0;
  cfg->eps = 1e-2;
  cfg->lamb_factor = 10.0;
  cfg->max_lamb = 1000.0;
  cfg->ctrl_q1 = cfg->ctrl_q2 = 1.0;
  cfg->obs_q1 = cfg->obs_q2 = 2.74;
  cfg->safety_margin = 0.0;
  auto diag = [&](double* M, int i, double v) { M[i * I2LQR_MAX_N + i] = v; };
  switch (system_id) {
    case I2LQR_SYS_BICYCLE4: {
      cfg->n = 4;
      cfg->m = 2;
      cfg->u_max[0] = 2.0;
      cfg->u_max[1] = 1.57; /* round(pi/2, 2) */
      const double qt[4] = {2.0, 2.0, 40.0, 0.04};
      for (int i = 0; i < 4; i++) diag(cfg->Qt, i, qt[i]);
      break;
    }
    case I2LQR_SYS_BICYCLE6: {
      cfg->n = 6;
      cfg->m = 2;
      cfg->u_max[0] = 1.0;
      cfg->u_max[1] = 0.5;
      const double qt[6] = {2.0, 2.0, 40.0, 0.04, 2.0, 2.0};
      for (int i = 0; i < 6; i++) diag(cfg->Qt, i, qt[i]);
      break;
    }
    case I2LQR_SYS_QUAD12: {
      cfg->n = 12;
      cfg->m = 4;
      for (int a = 0; a < 4; a++) cfg->u_max[a] = 2.0;
      const double qt[4] = {20.0, 10.0, 2.0, 1.0};
      for (int i = 0; i < 12; i++) diag(cfg->Qt, i, qt[i / 3]);
      const double sp[8] = {1.0, 9.81, 0.2, 0.01, 0.01, 0.02, 0.05, 0.0};
      for (int q = 0; q < 8; q++) cfg->sys_par[q] = sp[q];
      break;
    }
    default:
      return fail(I2LQR_ERR_INVALID, "unknown system_id %d", system_id);
  }
  return I2LQR_OK;
}

int i2lqr_create(const i2lqr_config* cfg, i2lqr_handle** out) {
  if (!cfg || !out) return fail(I2LQR_ERR_INVALID, "null argument");
  *out = nullptr;
  if (cfg->struct_size != (int32_t)sizeof(i2lqr_config))
    return fail(I2LQR_ERR_INVALID, "config struct_size %d, library expects %zu", cfg->struct_size,
                sizeof(i2lqr_config));
  int n = 0, m = 0;
  switch (cfg->system_id) {
    case I2LQR_SYS_BICYCLE4: n = 4; m = 2; break;
    case I2LQR_SYS_BICYCLE6: n = 6; m = 2; break;
    case I2LQR_SYS_QUAD12: n = 12; m = 4; break;
    default: return fail(I2LQR_ERR_INVALID, "unknown system_id %d", cfg->system_id);
  }
  if (cfg->n != n || cfg->m != m)
    return fail(I2LQR_ERR_INVALID, "system %d has n=%d m=%d, config says n=%d m=%d", cfg->system_id,
                n, m, cfg->n, cfg->m);
  if (cfg->N < 1 || cfg->N > I2LQR_MAX_HORIZON)
    return fail(I2LQR_ERR_INVALID, "horizon %d outside [1, %d]", cfg->N, I2LQR_MAX_HORIZON);
  if (cfg->dtype != I2LQR_F64 && cfg->dtype != I2LQR_F32)
    return fail(I2LQR_ERR_INVALID, "unknown dtype %d", cfg->dtype);
  if (cfg->layout != I2LQR_LAYOUT_PROBLEM_MAJOR && cfg->layout != I2LQR_LAYOUT_BATCH_MINOR &&
      cfg->layout != I2LQR_LAYOUT_BATCH_TILED)
    return fail(I2LQR_ERR_INVALID, "unknown layout %d", cfg->layout);
  if (!(cfg->dt > 0) || !(cfg->lamb_factor > 1) || cfg->max_iter < 0)
    return fail(I2LQR_ERR_INVALID, "need dt > 0, lamb_factor > 1, max_iter >= 0");
  for (int a = 0; a < m; a++)
    if (!(cfg->u_max[a] > 0)) return fail(I2LQR_ERR_INVALID, "u_max[%d] must be > 0", a);
  if (cfg->layout != I2LQR_LAYOUT_PROBLEM_MAJOR) {
    // the one-problem-per-lane kernels store only the upper triangles of the value function
    for (int i = 0; i < n; i++)
      for (int j = 0; j < i; j++)
        if (cfg->Q[i * I2LQR_MAX_N + j] != cfg->Q[j * I2LQR_MAX_N + i] ||
            cfg->Qt[i * I2LQR_MAX_N + j] != cfg->Qt[j * I2LQR_MAX_N + i])
          return fail(I2LQR_ERR_UNSUPPORTED, "the batch-minor / batch-tiled layouts need "
                      "symmetric Q and Qterminal (use the problem-major layout otherwise)");
    for (int a = 0; a < m; a++)
      for (int b = 0; b < a; b++)
        if (cfg->R[a * I2LQR_MAX_M + b] != cfg->R[b * I2LQR_MAX_M + a])
          return fail(I2LQR_ERR_UNSUPPORTED, "the batch-minor / batch-tiled layouts need a "
                      "symmetric R (use the problem-major layout otherwise)");
  }
  int ndev = 0;
#ifdef I2LQR_DRY_RUN
  const bool dry_run = dry::on();  // (sanitizer build + I2LQR_DRY_RUN=1: launches are recorded, not run)
#else
  constexpr bool dry_run = false;
#endif
  if (!dry_run && (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0))
    return fail(I2LQR_ERR_NODEVICE, "no HIP device visible (this library has no CPU path)");
  i2lqr_handle* h = new (std::nothrow) i2lqr_handle;
  if (!h) return fail(I2LQR_ERR_LAUNCH, "out of host memory");
  h->cfg = *cfg;
  h->geo = device_geometry();  // of the current device: CU count, LDS per CU / per workgroup
  if (h->geo.wave != 64) {
    const int w = h->geo.wave;
    delete h;
    return fail(I2LQR_ERR_UNSUPPORTED, "the kernels are built for 64-lane wavefronts (gfx950), this "
                "device reports %d", w);
  }
  h->ws = nullptr;
  h->ws_bytes = 0;
  h->compact_min_batch = -1;
  h->opt_defer = h->opt_reroll = h->opt_lds_steps = h->opt_fstep = h->opt_group = -1;
  h->opt_merge = h->opt_ckpt = h->opt_spec = h->opt_stagger = h->opt_group_ws = -1;
  h->wave_tail = -1;
  h->opt_first_chunk = -1;
  h->opt_fuse = -1;
  h->opt_final_round = -1;
  h->opt_chunk_step = -1;
  h->opt_pair = -1;
  h->opt_two_x = -1;
  h->ticket = nullptr;
  h->ticket_next.store(0);
  h->ev_ready = h->ev_side_done = nullptr;
  h->side_pending = false;
  constexpr size_t kTicketBytes = i2lqr_handle::kTickets * sizeof(unsigned);
  if (dry_run) {
#ifdef I2LQR_DRY_RUN
    static unsigned fake_tickets[i2lqr_handle::kTickets];  // an address to declare, never written by a kernel
    h->device = 0;
    h->ticket = fake_tickets;
    dry::allow(fake_tickets, sizeof(fake_tickets));
#endif
  } else if (hipGetDevice(&h->device) != hipSuccess ||
      hipMalloc((void**)&h->ticket, kTicketBytes) != hipSuccess ||
      hipMemset(h->ticket, 0, kTicketBytes) != hipSuccess) {
    if (h->ticket) (void)hipFree(h->ticket);
    delete h;
    return fail(I2LQR_ERR_LAUNCH, "could not set up the handle's device state: %s",
                hipGetErrorString(hipGetLastError()));
  }
  const int rc = prepare_dispatch(h);
  if (rc != I2LQR_OK) {
    (void)hipFree(h->ticket);
    delete h;
    return rc;
  }
  {
    std::lock_guard<std::mutex> lock(g_live_mu);
    live_handles().insert(h);
  }
  *out = h;
  g_err[0] = 0;
  return I2LQR_OK;
}

int64_t i2lqr_dry_run(int32_t op, void* p, int64_t n) {
#ifdef I2LQR_DRY_RUN
  switch (op) {
    case 0: return dry::on() ? 1 : 0;
    case 1: if (!dry::on()) break; dry::allow(p, (size_t)n); return 0;
    case 2: if (!dry::on()) break; dry::reset(); return 0;
    case 3: if (!dry::on()) break; return dry::report((char*)p, n);
    default: return fail(I2LQR_ERR_INVALID, "i2lqr_dry_run: unknown operation %d", op);
  }
  return fail(I2LQR_ERR_UNSUPPORTED, "dry-run launches need I2LQR_DRY_RUN=1 in the environment");
#else
  (void)op; (void)p; (void)n;
  return fail(I2LQR_ERR_UNSUPPORTED, "dry-run launches exist in the sanitizer build only (make -C "
              "ilqr_iterative_tasks_amd/csrc asan -> libi2lqr_hip_asan.so, I2LQR_DRY_RUN=1)");
#endif
}

int i2lqr_device_geometry(int32_t* out, int32_t count) {
  if (!out || count < 1) return fail(I2LQR_ERR_INVALID, "null / empty output");
  const DeviceGeometry& g = device_geometry();
  const int32_t v[8] = {g.cus, g.simds_per_cu, (int32_t)g.lds_per_cu, (int32_t)g.max_dyn_lds,
                        (int32_t)g.default_dyn_lds, g.wave, g.faked, g.queried};
  for (int i = 0; i < count && i < 8; i++) out[i] = v[i];
  return I2LQR_OK;
}

int i2lqr_destroy(i2lqr_handle* h) {
  if (!h) return I2LQR_OK;
  {
    std::lock_guard<std::mutex> lock(g_live_mu);
    if (live_handles().erase(h) == 0)
      return fail(I2LQR_ERR_INVALID, "i2lqr_destroy: %p is not a live handle (destroyed twice?)",
                  (void*)h);
  }
  if (h->ticket) (void)hipFree(h->ticket);
  if (h->ev_ready) (void)hipEventDestroy(h->ev_ready);
  if (h->ev_side_done) (void)hipEventDestroy(h->ev_side_done);
  delete h;
  return I2LQR_OK;
}

int64_t i2lqr_workspace_bytes(const i2lqr_handle* h, int64_t B) {
  if (!h || B < 0) return 0;
  if (h->cfg.layout == I2LQR_LAYOUT_PROBLEM_MAJOR) {
    if (h->cfg.system_id == I2LQR_SYS_QUAD12) return quad_workspace_bytes(h->cfg, B);
    // the bicycles: the workspace form of the eight-lane kernel, above kGroupWsBatch problems
    // (or whenever it is pinned); nothing below
    // (only in the range where that form is the automatic choice, or when it is pinned: a
    // problem-major batch of 2^20 problems does not need 5.9 GB of scratch it would never use)
    return ((B > h->geo.scaled(kGroupWsBatch) &&
             (B <= h->geo.scaled(group_ws_top(h->cfg)) || h->opt_group == 8)) || h->opt_group_ws == 1)
               ? group_workspace_bytes(h->cfg, B) : 0;
  }
  const bool tiled = h->cfg.layout == I2LQR_LAYOUT_BATCH_TILED;
  const int N = h->cfg.N;
  const bool f64 = h->cfg.dtype == I2LQR_F64;
  switch (h->cfg.system_id) {
    case I2LQR_SYS_BICYCLE4:
      if (f64) return tiled ? LaneLaunch<double, Bicycle4<double>, true>::ws_bytes(N, B)
                            : LaneLaunch<double, Bicycle4<double>, false>::ws_bytes(N, B);
      return tiled ? LaneLaunch<float, Bicycle4<float>, true>::ws_bytes(N, B)
                   : LaneLaunch<float, Bicycle4<float>, false>::ws_bytes(N, B);
    case I2LQR_SYS_BICYCLE6:
      if (f64) return tiled ? LaneLaunch<double, Bicycle6<double>, true>::ws_bytes(N, B)
                            : LaneLaunch<double, Bicycle6<double>, false>::ws_bytes(N, B);
      return tiled ? LaneLaunch<float, Bicycle6<float>, true>::ws_bytes(N, B)
                   : LaneLaunch<float, Bicycle6<float>, false>::ws_bytes(N, B);
    case I2LQR_SYS_QUAD12:
      if (f64) return tiled ? LaneLaunch<double, Quad12<double>, true>::ws_bytes(N, B)
                            : LaneLaunch<double, Quad12<double>, false>::ws_bytes(N, B);
      return tiled ? LaneLaunch<float, Quad12<float>, true>::ws_bytes(N, B)
                   : LaneLaunch<float, Quad12<float>, false>::ws_bytes(N, B);
    default: return 0;
  }
}

// Batch sizes from which the one-problem-per-lane layouts win over the problem-major kernels.
// Round 4 measured ONE shape (fp64, n=6, N=20: 12289 for both entry points) and applied it to every
// bicycle configuration; round 5 measured twelve (tools/threshold_sweep.py: both bicycles x
// horizons 6 / 20 / 50 x fp64 / fp32, interleaved on one device, every launch on its own copy of the
// batch; profiles/r05_threshold_sweep*.json, table in profiles/README.md).  The crossover moves
// with the horizon and the precision by far more than 25 %: the sixteen-lane kernel holds
// 4 x (wavefronts whose LDS slices fit a CU) x 256 problems per round — 4096 at N = 20 in fp64,
// 1024 at N = 50 — while the lane kernels' launch floor grows only linearly with N, and a solve to
// termination on the lane layouts is chunked (compaction) from 4096 problems only.
// Each entry is the first batch size at which the lane layout was ahead (the last size the
// problem-major kernels won, plus one: their time is a staircase in rounds, the lane kernels'
// is flat).  A horizon between the measured ones takes the nearer one on a log scale.
//   quad12 (fp64): the sixteen-lane kernel 13 M it/s at any size, k_lane_iterate_rows 25 M at
//   8192 and 73 M at 65536.
struct LaneFrom { int64_t iterate, solve; };
// (re-measured with the helper-wavefront form of the lane kernel, which moved the bicycle6 N = 20
// crossover from 12289 to 8193 in fp64 and from 16385 to 12289 in fp32: profiles/
// r05_threshold_sweep5.json ... 9.json)
constexpr LaneFrom kLaneFrom[2][3][2] = {
    // bicycle4:          fp64              fp32
    /* N ~ 6  */ {{{8193, 10241}, {8193, 10241}},
    /* N ~ 20 */  {{8193, 4096}, {8193, 10241}},
    /* N ~ 50 */  {{4097, 4096}, {8193, 5121}}},
    // bicycle6
    /* N ~ 6  */ {{{8193, 6145}, {8193, 20481}},
    /* N ~ 20 */  {{8193, 10241}, {12289, 12289}},
    /* N ~ 50 */  {{3073, 4096}, {8193, 5121}}},
};
inline LaneFrom lane_from(const i2lqr_config& cfg) {
  const int sys = cfg.system_id == I2LQR_SYS_BICYCLE6 ? 1 : 0;
  const int nb = cfg.N <= 10 ? 0 : (cfg.N <= 31 ? 1 : 2);  // sqrt(6 x 20) = 10.95, sqrt(20 x 50) = 31.6
  return kLaneFrom[sys][nb][cfg.dtype == I2LQR_F64 ? 0 : 1];
}
// quad12 (round 5, the same sweep: profiles/r05_threshold_sweep3.json, ..4.json; round 4: 8192 for
// everything, measured at N = 50 fp64 fixed counts only).  The problem-major side is the
// sixteen-lane kernel with its HBM workspace (rounds of 4096 problems); solves stay there longer
// because the lane layouts have no latency tail for this plant.  [N <= 31, N > 31][fp64, fp32]
constexpr LaneFrom kLaneFromQuad[2][2] = {
    /* N ~ 20 */ {{4097, 8193}, {8193, 24577}},
    /* N ~ 50 */ {{4097, 6145}, {8193, 12289}},
};
// with stage weights the problem-major side is the one-problem-per-wavefront kernel: 2048 problems
// already fill the chip's SIMDs (as for the bicycles)
constexpr int64_t kLaneFromWeights = 2048;

int i2lqr_recommended_layout(const i2lqr_config* cfg, int64_t B, int32_t early_exit) {
  if (!cfg) return fail(I2LQR_ERR_INVALID, "null config");
  if (cfg->struct_size != (int32_t)sizeof(i2lqr_config))
    return fail(I2LQR_ERR_INVALID, "config struct_size %d, library expects %zu", cfg->struct_size,
                sizeof(i2lqr_config));
  if (B < 0) return fail(I2LQR_ERR_INVALID, "negative batch %lld", (long long)B);
  // what the lane layouts cannot run stays problem-major: non-symmetric weights (the kernels keep
  // the upper triangles), and for quad12 (row-block kernel) fp32 WITH stage weights
  bool lane_ok = true, weights = false;
  for (int i = 0; i < cfg->n; i++)
    for (int j = 0; j < cfg->n; j++) {
      if (cfg->Q[i * I2LQR_MAX_N + j] != 0.0) weights = true;
      if (cfg->Q[i * I2LQR_MAX_N + j] != cfg->Q[j * I2LQR_MAX_N + i] ||
          cfg->Qt[i * I2LQR_MAX_N + j] != cfg->Qt[j * I2LQR_MAX_N + i]) lane_ok = false;
    }
  for (int a = 0; a < cfg->m; a++)
    for (int b = 0; b < cfg->m; b++) {
      if (cfg->R[a * I2LQR_MAX_M + b] != 0.0) weights = true;
      if (cfg->R[a * I2LQR_MAX_M + b] != cfg->R[b * I2LQR_MAX_M + a]) lane_ok = false;
    }
  int64_t from;
  switch (cfg->system_id) {
    case I2LQR_SYS_BICYCLE4:
    case I2LQR_SYS_BICYCLE6:
      // with stage weights the problem-major side is the one-problem-per-wavefront kernel (the
      // column kernels are built for Q = R = 0): 1024 problems already fill the chip's SIMDs
      from = weights ? kLaneFromWeights
                     : (early_exit ? lane_from(*cfg).solve : lane_from(*cfg).iterate);
      break;
    case I2LQR_SYS_QUAD12: {
      const LaneFrom q = kLaneFromQuad[cfg->N <= 31 ? 0 : 1][cfg->dtype == I2LQR_F64 ? 0 : 1];
      from = weights ? kLaneFromWeights : (early_exit ? q.solve : q.iterate);
      if (cfg->dtype != I2LQR_F64 && weights) lane_ok = false;  // fp32: Q = R = 0 (round 5)
      break;
    }
    default: return fail(I2LQR_ERR_INVALID, "unknown system_id %d", cfg->system_id);
  }
  // measured on the 256-CU chip; on another CU count (a partitioned device) scaled by cus / 256:
  // both sides of the crossover are occupancy effects (i2lqr_geometry.hpp)
  from = device_geometry().scaled_from(from);
  if (!lane_ok || B < from) return I2LQR_LAYOUT_PROBLEM_MAJOR;
  return B % 64 == 0 ? I2LQR_LAYOUT_BATCH_TILED : I2LQR_LAYOUT_BATCH_MINOR;
}

int i2lqr_set_compaction(i2lqr_handle* h, int64_t min_batch) {
  if (!h) return fail(I2LQR_ERR_INVALID, "null handle");
  if (h->cfg.layout == I2LQR_LAYOUT_PROBLEM_MAJOR && min_batch > 0)
    return fail(I2LQR_ERR_UNSUPPORTED, "compaction applies to the batch-minor / batch-tiled layouts");
  h->compact_min_batch = min_batch < 0 ? -1 : min_batch;
  return I2LQR_OK;
}

int i2lqr_set_option(i2lqr_handle* h, const char* name, int64_t value) {
  if (!h || !name) return fail(I2LQR_ERR_INVALID, "null handle or option name");
  if (value < -1 || value > 0x7fffffff) return fail(I2LQR_ERR_INVALID, "option value out of range");
  const int v = (int)value;
  if (!strcmp(name, "defer_states")) h->opt_defer = v < 0 ? -1 : (v != 0);
  else if (!strcmp(name, "reroll_nominal")) h->opt_reroll = v < 0 ? -1 : (v != 0);
  else if (!strcmp(name, "lds_gain_steps")) h->opt_lds_steps = v;
  else if (!strcmp(name, "merge_inputs")) h->opt_merge = v < 0 ? -1 : (v != 0);
  else if (!strcmp(name, "checkpoint_states")) h->opt_ckpt = v < 0 ? -1 : (v != 0);
  else if (!strcmp(name, "stagger")) h->opt_stagger = v;
  else if (!strcmp(name, "per_step_jacobians")) h->opt_fstep = v < 0 ? -1 : (v != 0);
  else if (!strcmp(name, "wave_tail")) h->wave_tail = v < 0 ? -1 : (v > 65536 ? 65536 : v);
  else if (!strcmp(name, "first_chunk")) h->opt_first_chunk = v < 1 ? -1 : (v > 1024 ? 1024 : v);
  else if (!strcmp(name, "helper_wavefront")) h->opt_pair = v < 0 ? -1 : (v != 0);
  else if (!strcmp(name, "state_buffers")) h->opt_two_x = v < 0 ? -1 : (v != 0);
  else if (!strcmp(name, "chunk_step")) h->opt_chunk_step = v < 1 ? -1 : (v > 1024 ? 1024 : v);
  else if (!strcmp(name, "fused_compaction")) h->opt_fuse = v < 0 ? -1 : (v != 0);
  else if (!strcmp(name, "final_round")) h->opt_final_round = v < 0 ? -1 : (v > 20 ? 20 : v);
  else if (!strcmp(name, "speculate")) h->opt_spec = v < 0 ? -1 : (v != 0);
  else if (!strcmp(name, "group_workspace")) h->opt_group_ws = v < 0 ? -1 : (v != 0);
  else if (!strcmp(name, "debug_self_test")) {
    // Debug build only: provoke one index violation on purpose (a Slice of 8 words indexed at 8)
    // and return what the next call would: I2LQR_ERR_LAUNCH with the decoded record.
#ifdef I2LQR_DEBUG
    unsigned long long* w = i2lqr::debug_trap_word();
    double* buf = nullptr;
    HIP_TRY(hipMalloc((void**)&buf, 8 * sizeof(double)));
    hipLaunchKernelGGL(k_debug_self_test, dim3(1), dim3(64), 0, (hipStream_t)0, buf, w);
    const int rc = debug_check(I2LQR_OK, nullptr);
    (void)hipFree(buf);
    return rc;
#else
    return fail(I2LQR_ERR_UNSUPPORTED, "\"debug_self_test\" exists in the index-checked debug build "
                "only (make -C ilqr_iterative_tasks_amd/csrc debug)");
#endif
  }
  else if (!strcmp(name, "group_lanes")) {
    if (v != -1 && v != 8 && v != 16 && v != 64)
      return fail(I2LQR_ERR_INVALID, "\"group_lanes\" is 8, 16, 64 or -1");
    h->opt_group = v;
  }
  else return fail(I2LQR_ERR_INVALID, "unknown option '%s'", name);
  return I2LQR_OK;
}

static const char* kernel_name(const i2lqr_handle* h, int64_t B, bool early_exit) {
  if (!h) return "";
  if (h->cfg.layout != I2LQR_LAYOUT_PROBLEM_MAJOR) {
    if (h->cfg.system_id == I2LQR_SYS_QUAD12) return "k_lane_iterate_rows";
    // the helper-wavefront form (both precisions, with or without stage weights): decided by the
    // launcher's own helpers on scratch arguments (LaneLaunch::names_pair)
    return names_pair_dispatch(const_cast<i2lqr_handle*>(h), B, early_exit ? 1 : 0) == 1
               ? "k_lane_iterate_pair" : "k_lane_iterate";
  }
  switch (select_fused(h, B, early_exit, nullptr)) {
    case K_SPEC: return "k_group_spec";
    case K_SPEC16: return "k_group_spec (sixteen lanes)";
    case K_GROUP: return "k_group_iterate";
    case K_GROUP16: return "k_group_iterate (sixteen lanes)";
    case K_GROUP_WS: return "k_group_iterate (workspace form)";
    case K_QUAD: return "k_quad_iterate";
    case K_WAVE: return "k_iterate";
    default: return "unsupported";  // the launch returns I2LQR_ERR_UNSUPPORTED
  }
}

const char* i2lqr_iterate_kernel(const i2lqr_handle* h, int64_t B) { return kernel_name(h, B, false); }
const char* i2lqr_solve_kernel(const i2lqr_handle* h, int64_t B) { return kernel_name(h, B, true); }

int i2lqr_set_workspace(i2lqr_handle* h, void* workspace, int64_t bytes) {
  if (!h) return fail(I2LQR_ERR_INVALID, "null handle");
  if (bytes < 0 || (bytes > 0 && !workspace) || ((uintptr_t)workspace & 15))
    return fail(I2LQR_ERR_INVALID, "workspace must be a 16-byte aligned device pointer");
  h->ws = workspace;
  h->ws_bytes = bytes;
  return I2LQR_OK;
}

int i2lqr_rollout(i2lqr_handle* h, int64_t B, void* X, void* U, const void* x_term, void* cost,
                  void* stream) {
  if (int rc = check_common(h, B)) return rc;
  if (B == 0) return I2LQR_OK;
  if (!X || !U || !x_term || !cost) return fail(I2LQR_ERR_INVALID, "null buffer");
  return debug_check(dispatch_rollout(h, B, X, U, x_term, cost, stream), stream);
}

int i2lqr_backward(i2lqr_handle* h, int64_t B, const void* X, const void* U, const void* x_term,
                   const void* lamb, const void* obs, void* K, void* k, void* stream) {
  if (int rc = check_common(h, B)) return rc;
  if (B == 0) return I2LQR_OK;
  if (!X || !U || !x_term || !lamb || !K || !k) return fail(I2LQR_ERR_INVALID, "null buffer");
  return debug_check(dispatch_backward(h, B, X, U, x_term, lamb, obs, K, k, stream), stream);
}

int i2lqr_forward(i2lqr_handle* h, int64_t B, const void* X, const void* U, const void* x_term,
                  const void* K, const void* k, void* X_new, void* U_new, void* cost_new,
                  void* stream) {
  if (int rc = check_common(h, B)) return rc;
  if (B == 0) return I2LQR_OK;
  if (!X || !U || !x_term || !K || !k || !X_new || !U_new || !cost_new)
    return fail(I2LQR_ERR_INVALID, "null buffer");
  return debug_check(dispatch_forward(h, B, X, U, x_term, K, k, X_new, U_new, cost_new, stream),
                     stream);
}

int i2lqr_iterate(i2lqr_handle* h, int64_t B, int32_t n_iters, void* X, void* U,
                  const void* x_term, void* lamb, const void* obs, void* cost, void* K, void* k,
                  int32_t* iters, int32_t* status, void* stream) {
  if (int rc = check_common(h, B)) return rc;
  if (n_iters < 0) return fail(I2LQR_ERR_INVALID, "negative n_iters");
  if (B == 0) return I2LQR_OK;
  if (!X || !U || !x_term || !lamb || !cost) return fail(I2LQR_ERR_INVALID, "null buffer");
  if ((K == nullptr) != (k == nullptr))
    return fail(I2LQR_ERR_INVALID, "K and k must both be given or both be NULL");
  return debug_check(dispatch_iterate(h, B, n_iters, 0, X, U, x_term, lamb, obs, cost, K, k, iters,
                                      status, stream), stream);
}

int i2lqr_solve(i2lqr_handle* h, int64_t B, void* X, void* U, const void* x_term, void* lamb,
                const void* obs, void* cost, void* K, void* k, int32_t* iters, int32_t* status,
                void* stream) {
  if (int rc = check_common(h, B)) return rc;
  if (B == 0) return I2LQR_OK;
  if (!X || !U || !x_term || !lamb || !cost) return fail(I2LQR_ERR_INVALID, "null buffer");
  if ((K == nullptr) != (k == nullptr))
    return fail(I2LQR_ERR_INVALID, "K and k must both be given or both be NULL");
  return debug_check(dispatch_iterate(h, B, h->cfg.max_iter, 1, X, U, x_term, lamb, obs, cost, K, k,
                                      iters, status, stream), stream);
}

int i2lqr_solve_chained(i2lqr_handle* h, int64_t chains, int32_t chain_len, void* X, void* U,
                        const void* x_term, void* lamb, const void* obs, void* cost, void* K, void* k,
                        int32_t* iters, int32_t* status, void* stream) {
  if (int rc = check_common(h, chains)) return rc;
  if (chain_len < 1) return fail(I2LQR_ERR_INVALID, "chain_len must be >= 1");
  if (chains * (int64_t)chain_len > (int64_t)0x7fffffff)
    return fail(I2LQR_ERR_INVALID, "%lld chains of %d problems", (long long)chains, chain_len);
  if (chains == 0) return I2LQR_OK;
  if (!X || !U || !x_term || !lamb || !cost) return fail(I2LQR_ERR_INVALID, "null buffer");
  if ((K == nullptr) != (k == nullptr))
    return fail(I2LQR_ERR_INVALID, "K and k must both be given or both be NULL");
  return debug_check(dispatch_solve_chained(h, chains, chain_len, X, U, x_term, lamb, obs, cost, K, k,
                                            iters, status, stream), stream);
}

int i2lqr_relax_cost(i2lqr_handle* h, int64_t B, const void* X, const void* x_term,
                     const int32_t* qfun, int32_t outer_iter, int32_t max_relax_iter,
                     void* cost_it, void* stream) {
  if (int rc = check_common(h, B)) return rc;
  if (B == 0) return I2LQR_OK;
  if (!X || !x_term || !qfun || !cost_it) return fail(I2LQR_ERR_INVALID, "null buffer");
  if (outer_iter < 0 || max_relax_iter < 1)
    return fail(I2LQR_ERR_INVALID, "need outer_iter >= 0 and max_relax_iter >= 1");
  const unsigned grid = (unsigned)((B + 255) / 256);
  const int bm = h->cfg.layout;  // 0 problem-major, 1 batch-minor, 2 batch-tiled
  hipStream_t s = (hipStream_t)stream;
  if (h->cfg.dtype == I2LQR_F64)
    hipLaunchKernelGGL((k_relax_cost<double>), dim3(grid), dim3(256), 0, s, B, h->cfg.n, h->cfg.N,
                       bm, (const double*)X, (const double*)x_term, qfun, outer_iter, max_relax_iter,
                       (double*)cost_it);
  else
    hipLaunchKernelGGL((k_relax_cost<float>), dim3(grid), dim3(256), 0, s, B, h->cfg.n, h->cfg.N,
                       bm, (const float*)X, (const float*)x_term, qfun, outer_iter, max_relax_iter,
                       (float*)cost_it);
  HIP_TRY(hipGetLastError());
  return I2LQR_OK;
}

int64_t i2lqr_argmin_workspace_bytes(int64_t B) {
  // one (value, index) pair per workgroup: 256 for i2lqr_argmin, one per eight problems for the
  // pick that i2lqr_iterate_pick folds into the eight-lane kernels
  const int64_t rows = B > 0 ? (B + 3) / 4 : 0;  // ... one per four on the sixteen-lane form
  return (rows > kArgminBlocks ? rows : (int64_t)kArgminBlocks) * 16;
}

int i2lqr_argmin(i2lqr_handle* h, int64_t B, const void* cost_it, int64_t* best_idx,
                 void* best_cost, void* workspace, int64_t workspace_bytes, void* stream) {
  if (int rc = check_common(h, B)) return rc;
  if ((B > 0 && !cost_it) || !best_idx || !best_cost || !workspace)
    return fail(I2LQR_ERR_INVALID, "null buffer");
  if (workspace_bytes < i2lqr_argmin_workspace_bytes(B))
    return fail(I2LQR_ERR_INVALID, "arg-min workspace of %lld bytes, %lld problems need "
                "i2lqr_argmin_workspace_bytes(B) = %lld", (long long)workspace_bytes, (long long)B,
                (long long)i2lqr_argmin_workspace_bytes(B));
  hipStream_t s = (hipStream_t)stream;
  int64_t want = (B + 255) / 256;
  const int blocks = (int)(want < 1 ? 1 : (want > kArgminBlocks ? kArgminBlocks : want));
  if (B <= kArgminSingle) {  // one workgroup, one launch
    if (h->cfg.dtype == I2LQR_F64)
      hipLaunchKernelGGL((k_argmin_partial<double, true>), dim3(1), dim3(256), 0, s, B,
                         (const double*)cost_it, (MinPair<double>*)nullptr, best_idx,
                         (double*)best_cost);
    else
      hipLaunchKernelGGL((k_argmin_partial<float, true>), dim3(1), dim3(256), 0, s, B,
                         (const float*)cost_it, (MinPair<float>*)nullptr, best_idx,
                         (float*)best_cost);
  } else if (h->cfg.dtype == I2LQR_F64) {
    auto* part = (MinPair<double>*)workspace;
    hipLaunchKernelGGL((k_argmin_partial<double>), dim3(blocks), dim3(256), 0, s, B,
                       (const double*)cost_it, part);
    hipLaunchKernelGGL((k_argmin_final<double>), dim3(1), dim3(256), 0, s, blocks, part, best_idx,
                       (double*)best_cost);
  } else {
    auto* part = (MinPair<float>*)workspace;
    hipLaunchKernelGGL((k_argmin_partial<float>), dim3(blocks), dim3(256), 0, s, B,
                       (const float*)cost_it, part);
    hipLaunchKernelGGL((k_argmin_final<float>), dim3(1), dim3(256), 0, s, blocks, part, best_idx,
                       (float*)best_cost);
  }
  HIP_TRY(hipGetLastError());
  return I2LQR_OK;
}

int i2lqr_iterate_pick(i2lqr_handle* h, int64_t B, int32_t n_iters, void* X, void* U,
                       const void* x_term, void* lamb, const void* obs, void* cost, void* K, void* k,
                       int32_t* iters, int32_t* status, const int32_t* qfun, int32_t outer_iter,
                       int32_t max_relax_iter, void* cost_it, int64_t* best_idx, void* best_cost,
                       void* workspace, int64_t workspace_bytes, void* stream) {
  if (int rc = check_common(h, B)) return rc;
  if (n_iters < 0) return fail(I2LQR_ERR_INVALID, "negative n_iters");
  if (outer_iter < 0 || max_relax_iter < 1)
    return fail(I2LQR_ERR_INVALID, "need outer_iter >= 0 and max_relax_iter >= 1");
  if (best_idx && (!best_cost || !workspace))
    return fail(I2LQR_ERR_INVALID, "best_idx needs best_cost and a workspace of "
                "i2lqr_argmin_workspace_bytes(B)");
  // the fused epilogue writes one (value, index) pair per workgroup of the iterate kernel: a
  // workspace sized for a smaller batch would be written out of bounds on the device
  if (best_idx && workspace_bytes < i2lqr_argmin_workspace_bytes(B))
    return fail(I2LQR_ERR_INVALID, "pick workspace of %lld bytes, %lld problems need "
                "i2lqr_argmin_workspace_bytes(B) = %lld", (long long)workspace_bytes, (long long)B,
                (long long)i2lqr_argmin_workspace_bytes(B));
  if (B == 0)  // nothing to solve; an empty pick is (-1, +inf) as in i2lqr_argmin
    return best_idx ? i2lqr_argmin(h, 0, nullptr, best_idx, best_cost, workspace, workspace_bytes,
                                   stream) : I2LQR_OK;
  if (!X || !U || !x_term || !lamb || !cost || !qfun || !cost_it)
    return fail(I2LQR_ERR_INVALID, "null buffer");
  if ((K == nullptr) != (k == nullptr))
    return fail(I2LQR_ERR_INVALID, "K and k must both be given or both be NULL");
  i2lqr_handle::Epilogue epi{qfun, outer_iter, max_relax_iter, cost_it, workspace, best_idx,
                             best_cost, false};
  t_epi = &epi;
  const int rc = dispatch_iterate(h, B, n_iters, 0, X, U, x_term, lamb, obs, cost, K, k, iters,
                                  status, stream);
  t_epi = nullptr;
  if (rc != I2LQR_OK) return rc;
  if (!epi.fused) {  // kernel families without the epilogue: the same three steps as launches
    if (int r2 = i2lqr_relax_cost(h, B, X, x_term, qfun, outer_iter, max_relax_iter, cost_it, stream))
      return r2;
    if (best_idx)
      if (int r3 = i2lqr_argmin(h, B, cost_it, best_idx, best_cost, workspace, workspace_bytes,
                                stream)) return r3;
  }
  return debug_check(I2LQR_OK, stream);
}

int i2lqr_select_candidates(i2lqr_handle* h, int32_t L, int32_t Tmax, const void* ss,
                            const int32_t* T, const int32_t* qfun, const void* x_guess,
                            int32_t guess_stride, int32_t k, int32_t* idx, void* x_term,
                            int32_t* qf, void* stream) {
  if (!h) return fail(I2LQR_ERR_INVALID, "null handle");
  if (h->cfg.layout != I2LQR_LAYOUT_PROBLEM_MAJOR)
    return fail(I2LQR_ERR_UNSUPPORTED, "controller-round kernels need the problem-major layout");
  if (L < 0 || Tmax < 1 || Tmax > 1024 || k < 1 || k > Tmax || guess_stride < 1)
    return fail(I2LQR_ERR_INVALID, "need L >= 0, 1 <= k <= Tmax <= 1024, guess_stride >= 1");
  if (L == 0) return I2LQR_OK;
  if (!ss || !T || !qfun || !x_guess || !idx || !x_term || !qf)
    return fail(I2LQR_ERR_INVALID, "null buffer");
  hipStream_t s = (hipStream_t)stream;
  if (h->cfg.dtype == I2LQR_F64)
    hipLaunchKernelGGL((k_select_candidates<double>), dim3(L), dim3(128), 0, s, h->cfg.n, Tmax, k,
                       (const double*)ss, T, qfun, (const double*)x_guess, guess_stride, idx,
                       (double*)x_term, qf);
  else
    hipLaunchKernelGGL((k_select_candidates<float>), dim3(L), dim3(128), 0, s, h->cfg.n, Tmax, k,
                       (const float*)ss, T, qfun, (const float*)x_guess, guess_stride, idx,
                       (float*)x_term, qf);
  HIP_TRY(hipGetLastError());
  return I2LQR_OK;
}

int i2lqr_init_candidates(i2lqr_handle* h, int64_t B, const void* x0, double lamb0, void* X,
                          void* U, void* lamb, void* stream) {
  if (int rc = check_common(h, B)) return rc;
  if (h->cfg.layout != I2LQR_LAYOUT_PROBLEM_MAJOR)
    return fail(I2LQR_ERR_UNSUPPORTED, "controller-round kernels need the problem-major layout");
  if (B == 0) return I2LQR_OK;
  if (!x0 || !X || !U || !lamb) return fail(I2LQR_ERR_INVALID, "null buffer");
  hipStream_t s = (hipStream_t)stream;
  const int n = h->cfg.n, m = h->cfg.m, N = h->cfg.N;
  if (h->cfg.dtype == I2LQR_F64)
    hipLaunchKernelGGL((k_init_candidates<double>), dim3((unsigned)B), dim3(64), 0, s, B, n, m, N,
                       (const double*)x0, lamb0, (double*)X, (double*)U, (double*)lamb);
  else
    hipLaunchKernelGGL((k_init_candidates<float>), dim3((unsigned)B), dim3(64), 0, s, B, n, m, N,
                       (const float*)x0, (float)lamb0, (float*)X, (float*)U, (float*)lamb);
  HIP_TRY(hipGetLastError());
  return I2LQR_OK;
}

int i2lqr_pick_best(i2lqr_handle* h, int32_t L, int32_t k, const void* cost_it, const void* X,
                    const void* U, int32_t* best, void* x_pred, void* u_pred, void* stream) {
  if (!h) return fail(I2LQR_ERR_INVALID, "null handle");
  if (L < 1 || k < 1) return fail(I2LQR_ERR_INVALID, "need L >= 1 and k >= 1");
  if (!cost_it || !best) return fail(I2LQR_ERR_INVALID, "null buffer");
  // X, U, x_pred, u_pred all NULL: the pick alone (a sharded round picks on the gathered costs;
  // the winner's trajectory is on the rank that solved it) — any layout
  const bool index_only = !X && !U && !x_pred && !u_pred;
  if (!index_only && (!X || !U || !x_pred || !u_pred))
    return fail(I2LQR_ERR_INVALID, "X, U, x_pred, u_pred must all be given or all be NULL");
  if (!index_only && h->cfg.layout != I2LQR_LAYOUT_PROBLEM_MAJOR)
    return fail(I2LQR_ERR_UNSUPPORTED, "controller-round kernels need the problem-major layout");
  hipStream_t s = (hipStream_t)stream;
  const int n = h->cfg.n, m = h->cfg.m, N = h->cfg.N;
  if (h->cfg.dtype == I2LQR_F64)
    hipLaunchKernelGGL((k_pick_best<double>), dim3(1), dim3(64), 0, s, L, k, n, m, N,
                       (const double*)cost_it, (const double*)X, (const double*)U, best,
                       (double*)x_pred, (double*)u_pred);
  else
    hipLaunchKernelGGL((k_pick_best<float>), dim3(1), dim3(64), 0, s, L, k, n, m, N,
                       (const float*)cost_it, (const float*)X, (const float*)U, best,
                       (float*)x_pred, (float*)u_pred);
  HIP_TRY(hipGetLastError());
  return I2LQR_OK;
}

int i2lqr_comm_available(void) {
  const RcclApi& api = rccl_api();
  if (!api.ok) return fail(I2LQR_ERR_UNSUPPORTED, "librccl could not be loaded: %s", api.why);
  return I2LQR_OK;
}

int i2lqr_comm_unique_id(void* id) {
  if (!id) return fail(I2LQR_ERR_INVALID, "null id buffer");
  const RcclApi& api = rccl_api();
  if (!api.ok) return fail(I2LQR_ERR_UNSUPPORTED, "librccl could not be loaded: %s", api.why);
  static_assert(sizeof(ncclUniqueId) == I2LQR_COMM_ID_BYTES, "unique id size");
  ncclUniqueId uid;
  RCCL_TRY(api, api.GetUniqueId(&uid));
  std::memcpy(id, &uid, sizeof(uid));
  return I2LQR_OK;
}

int i2lqr_comm_create(const void* id, int32_t world, int32_t rank, void** comm) {
  if (!id || !comm) return fail(I2LQR_ERR_INVALID, "null argument");
  *comm = nullptr;
  if (world < 1 || rank < 0 || rank >= world)
    return fail(I2LQR_ERR_INVALID, "need 0 <= rank < world (got rank %d, world %d)", rank, world);
  const RcclApi& api = rccl_api();
  if (!api.ok) return fail(I2LQR_ERR_UNSUPPORTED, "librccl could not be loaded: %s", api.why);
  ncclUniqueId uid;
  std::memcpy(&uid, id, sizeof(uid));
  ncclComm_t c = nullptr;
  RCCL_TRY(api, api.CommInitRank(&c, world, uid, rank));  // binds the calling thread's device
  *comm = (void*)c;
  return I2LQR_OK;
}

int i2lqr_comm_destroy(void* comm) {
  if (!comm) return I2LQR_OK;
  const RcclApi& api = rccl_api();
  if (!api.ok) return fail(I2LQR_ERR_UNSUPPORTED, "librccl could not be loaded");
  RCCL_TRY(api, api.CommDestroy((ncclComm_t)comm));
  return I2LQR_OK;
}

int i2lqr_comm_abort(void* comm) {
  if (!comm) return I2LQR_OK;
  const RcclApi& api = rccl_api();
  if (!api.ok) return fail(I2LQR_ERR_UNSUPPORTED, "librccl could not be loaded");
  RCCL_TRY(api, api.CommAbort((ncclComm_t)comm));
  return I2LQR_OK;
}

int i2lqr_comm_info(void* comm, int32_t* world, int32_t* rank) {
  if (!comm || !world || !rank) return fail(I2LQR_ERR_INVALID, "null argument");
  const RcclApi& api = rccl_api();
  if (!api.ok) return fail(I2LQR_ERR_UNSUPPORTED, "librccl could not be loaded");
  int w = 0, r = 0;
  RCCL_TRY(api, api.CommCount((ncclComm_t)comm, &w));
  RCCL_TRY(api, api.CommUserRank((ncclComm_t)comm, &r));
  *world = w;
  *rank = r;
  return I2LQR_OK;
}

int i2lqr_allgather_costs(i2lqr_handle* h, void* comm, const void* cost_local, void* cost_all,
                          int64_t n_local, void* stream) {
  if (int rc = check_common(h, n_local)) return rc;
  if (!comm) return fail(I2LQR_ERR_INVALID, "null communicator");
  if (n_local == 0) return I2LQR_OK;
  if (!cost_local || !cost_all) return fail(I2LQR_ERR_INVALID, "null buffer");
  const RcclApi& api = rccl_api();
  if (!api.ok) return fail(I2LQR_ERR_UNSUPPORTED, "librccl could not be loaded");
  const ncclDataType_t dt = h->cfg.dtype == I2LQR_F64 ? ncclDouble : ncclFloat;
  RCCL_TRY(api, api.AllGather(cost_local, cost_all, (size_t)n_local, dt, (ncclComm_t)comm,
                              (hipStream_t)stream));
  return I2LQR_OK;
}

int i2lqr_allgather_round(i2lqr_handle* h, void* comm, const void* cost_local, void* cost_all,
                          int64_t n_local, const void* pack_local, void* pack_all,
                          int64_t pack_count, void* stream) {
  if (int rc = check_common(h, n_local)) return rc;
  if (!comm) return fail(I2LQR_ERR_INVALID, "null communicator");
  if (pack_count < 0) return fail(I2LQR_ERR_INVALID, "negative pack_count");
  if ((n_local > 0 && (!cost_local || !cost_all)) || (pack_count > 0 && (!pack_local || !pack_all)))
    return fail(I2LQR_ERR_INVALID, "null buffer");
  if (n_local == 0 && pack_count == 0) return I2LQR_OK;
  const RcclApi& api = rccl_api();
  if (!api.ok) return fail(I2LQR_ERR_UNSUPPORTED, "librccl could not be loaded");
  const ncclDataType_t dt = h->cfg.dtype == I2LQR_F64 ? ncclDouble : ncclFloat;
  // both all-gathers as ONE grouped operation (one launch on the communicator's stream)
  const bool grouped = api.GroupStart && api.GroupEnd;
  if (grouped) RCCL_TRY(api, api.GroupStart());
  ncclResult_t r1 = ncclSuccess, r2 = ncclSuccess;
  if (n_local > 0)
    r1 = api.AllGather(cost_local, cost_all, (size_t)n_local, dt, (ncclComm_t)comm,
                       (hipStream_t)stream);
  if (r1 == ncclSuccess && pack_count > 0)
    r2 = api.AllGather(pack_local, pack_all, (size_t)pack_count, dt, (ncclComm_t)comm,
                       (hipStream_t)stream);
  if (grouped) {  // the group is closed whatever happened inside it
    const ncclResult_t r3 = api.GroupEnd();
    if (r1 == ncclSuccess && r2 == ncclSuccess) RCCL_TRY(api, r3);
  }
  RCCL_TRY(api, r1);
  RCCL_TRY(api, r2);
  return I2LQR_OK;
}

int i2lqr_pack_problem(i2lqr_handle* h, int64_t B, const void* X, const void* U, const int64_t* idx,
                       void* pack, void* stream) {
  if (int rc = check_common(h, B)) return rc;
  if (B < 1) return fail(I2LQR_ERR_INVALID, "need at least one problem");
  if (!X || !U || !idx || !pack) return fail(I2LQR_ERR_INVALID, "null buffer");
  if (h->cfg.layout == I2LQR_LAYOUT_BATCH_TILED && (B & 63))
    return fail(I2LQR_ERR_INVALID, "the batch-tiled layout needs a batch that is a multiple of 64");
  hipStream_t s = (hipStream_t)stream;
  const int n = h->cfg.n, m = h->cfg.m, N = h->cfg.N;
  if (h->cfg.dtype == I2LQR_F64)
    hipLaunchKernelGGL((k_pack_problem<double>), dim3(1), dim3(256), 0, s, B, n, m, N,
                       (int)h->cfg.layout, (const double*)X, (const double*)U, idx, (double*)pack);
  else
    hipLaunchKernelGGL((k_pack_problem<float>), dim3(1), dim3(256), 0, s, B, n, m, N,
                       (int)h->cfg.layout, (const float*)X, (const float*)U, idx, (float*)pack);
  HIP_TRY(hipGetLastError());
  return I2LQR_OK;
}

int i2lqr_round_winner(i2lqr_handle* h, int32_t world, int64_t width, int64_t total,
                       int64_t pack_count, const int64_t* best_padded, const void* pack_all,
                       void* winner, int64_t* best_global, void* stream) {
  if (!h) return fail(I2LQR_ERR_INVALID, "null handle");
  if (world < 1 || width < 1 || total < 0 || total > (int64_t)world * width || pack_count < 0)
    return fail(I2LQR_ERR_INVALID, "need world >= 1, width >= 1, 0 <= total <= world * width, "
                "pack_count >= 0");
  if (!best_padded || !best_global || (pack_count > 0 && (!pack_all || !winner)))
    return fail(I2LQR_ERR_INVALID, "null buffer");
  hipStream_t s = (hipStream_t)stream;
  if (h->cfg.dtype == I2LQR_F64)
    hipLaunchKernelGGL((k_round_winner<double>), dim3(1), dim3(256), 0, s, world, width, total,
                       pack_count, best_padded, (const double*)pack_all, (double*)winner,
                       best_global);
  else
    hipLaunchKernelGGL((k_round_winner<float>), dim3(1), dim3(256), 0, s, world, width, total,
                       pack_count, best_padded, (const float*)pack_all, (float*)winner,
                       best_global);
  HIP_TRY(hipGetLastError());
  return I2LQR_OK;
}

int i2lqr_broadcast_winner(i2lqr_handle* h, void* comm, void* buf, int64_t count, int32_t root,
                           void* stream) {
  if (int rc = check_common(h, count)) return rc;
  if (!comm) return fail(I2LQR_ERR_INVALID, "null communicator");
  if (root < 0) return fail(I2LQR_ERR_INVALID, "negative root rank");
  if (count == 0) return I2LQR_OK;
  if (!buf) return fail(I2LQR_ERR_INVALID, "null buffer");
  const RcclApi& api = rccl_api();
  if (!api.ok) return fail(I2LQR_ERR_UNSUPPORTED, "librccl could not be loaded");
  const ncclDataType_t dt = h->cfg.dtype == I2LQR_F64 ? ncclDouble : ncclFloat;
  RCCL_TRY(api, api.Broadcast(buf, buf, (size_t)count, dt, root, (ncclComm_t)comm,
                              (hipStream_t)stream));
  return I2LQR_OK;
}

int i2lqr_round_pick(i2lqr_handle* h, int32_t world, int64_t width, int64_t total,
                     int64_t pack_count, const void* cost_all, const void* pack_all,
                     void* best_cost, void* winner, int64_t* best_global, void* workspace,
                     int64_t workspace_bytes, void* stream) {
  if (!h) return fail(I2LQR_ERR_INVALID, "null handle");
  if (world < 1 || width < 1 || total < 0 || total > (int64_t)world * width || pack_count < 0)
    return fail(I2LQR_ERR_INVALID, "need world >= 1, width >= 1, 0 <= total <= world * width, "
                "pack_count >= 0");
  const int64_t G = (int64_t)world * width;
  if (G > (int64_t)0x7fffffff) return fail(I2LQR_ERR_INVALID, "%lld gathered costs", (long long)G);
  if (!cost_all || !best_cost || !best_global || (pack_count > 0 && (!pack_all || !winner)))
    return fail(I2LQR_ERR_INVALID, "null buffer");
  hipStream_t s = (hipStream_t)stream;
  if (G <= kArgminSingle) {
    if (h->cfg.dtype == I2LQR_F64)
      hipLaunchKernelGGL((k_round_pick<double>), dim3(1), dim3(256), 0, s, world, width, total,
                         pack_count, (const double*)cost_all, (const double*)pack_all,
                         (double*)best_cost, (double*)winner, best_global);
    else
      hipLaunchKernelGGL((k_round_pick<float>), dim3(1), dim3(256), 0, s, world, width, total,
                         pack_count, (const float*)cost_all, (const float*)pack_all,
                         (float*)best_cost, (float*)winner, best_global);
    HIP_TRY(hipGetLastError());
    return I2LQR_OK;
  }
  // large gathers: the two-level arg-min, its index parked behind the partial minima
  const int64_t need = i2lqr_argmin_workspace_bytes(G) + 16;
  if (!workspace || workspace_bytes < need)
    return fail(I2LQR_ERR_INVALID, "round-pick workspace of %lld bytes, %lld gathered costs need "
                "i2lqr_argmin_workspace_bytes + 16 = %lld", (long long)workspace_bytes, (long long)G,
                (long long)need);
  int64_t* best_padded = (int64_t*)((char*)workspace + i2lqr_argmin_workspace_bytes(G));
  if (int rc = i2lqr_argmin(h, G, cost_all, best_padded, best_cost, workspace,
                            i2lqr_argmin_workspace_bytes(G), stream)) return rc;
  return i2lqr_round_winner(h, world, width, total, pack_count, best_padded, pack_all, winner,
                            best_global, stream);
}

int i2lqr_sharded_round_flat(i2lqr_handle* h, void* comm, const i2lqr_round* r, void* side_stream,
                             void* stream) {
  if (!h) return fail(I2LQR_ERR_INVALID, "null handle");
  if (!r) return fail(I2LQR_ERR_INVALID, "null round");
  if (r->struct_size != (int32_t)sizeof(i2lqr_round))
    return fail(I2LQR_ERR_INVALID, "round struct_size %d, library expects %zu", r->struct_size,
                sizeof(i2lqr_round));
  if (int rc = check_common(h, r->B)) return rc;
  if (r->world < 1 || r->rank < 0 || r->rank >= r->world)
    return fail(I2LQR_ERR_INVALID, "need 0 <= rank < world (got rank %d, world %d)", r->rank, r->world);
  if (r->total < 1)
    return fail(I2LQR_ERR_INVALID, "a round needs at least one candidate over all ranks (total %lld)",
                (long long)r->total);
  if (!comm && r->world != 1 && !r->loopback)
    return fail(I2LQR_ERR_INVALID, "a world of %d ranks needs a communicator", r->world);
  // the contiguous split every rank derives from (total, world): dist.shard_range
  const int64_t base = r->total / r->world, rem = r->total - base * r->world;
  const int64_t mine = base + (r->rank < rem ? 1 : 0);
  const int64_t width = base + (rem ? 1 : 0);
  if (r->B != mine)
    return fail(I2LQR_ERR_INVALID, "rank %d of %d owns %lld of %lld candidates, the round says %lld",
                r->rank, r->world, (long long)mine, (long long)r->total, (long long)r->B);
  if (width * r->world > (int64_t)0x7fffffff)
    return fail(I2LQR_ERR_INVALID, "%lld gathered costs", (long long)(width * r->world));
  const int n = h->cfg.n, m = h->cfg.m, N = h->cfg.N;
  const int64_t P = (int64_t)m * N + (int64_t)n * (N + 1);
  if (!r->pack_local || !r->cost_all || !r->pack_all || !r->best_cost || !r->winner ||
      !r->best_global || (r->B > 0 && (!r->cost_it || !r->local_best || !r->local_best_cost)))
    return fail(I2LQR_ERR_INVALID, "null buffer");
  if (r->B != width && !r->cost_padded)
    return fail(I2LQR_ERR_INVALID, "a ragged split (%lld of width %lld) needs cost_padded",
                (long long)r->B, (long long)width);
  if (width * r->world > kArgminSingle &&
      (!r->side_ws || r->side_ws_bytes < i2lqr_argmin_workspace_bytes(width * r->world) + 16))
    return fail(I2LQR_ERR_INVALID, "side workspace of %lld bytes, %lld gathered costs need %lld",
                (long long)r->side_ws_bytes, (long long)(width * r->world),
                (long long)(i2lqr_argmin_workspace_bytes(width * r->world) + 16));
  if (h->cfg.layout == I2LQR_LAYOUT_BATCH_TILED && (r->B & 63))
    return fail(I2LQR_ERR_INVALID, "the batch-tiled layout needs a batch that is a multiple of 64");
  hipStream_t s = (hipStream_t)stream;
  hipStream_t ss = side_stream ? (hipStream_t)side_stream : s;
  const bool two = ss != s;
  if (two && !h->ev_ready) {
    HIP_TRY(hipEventCreateWithFlags(&h->ev_ready, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&h->ev_side_done, hipEventDisableTiming));
  }
  if (r->guard_previous && h->side_pending) HIP_TRY(hipStreamWaitEvent(s, h->ev_side_done, 0));
  // -- the shard (launch stream) -------------------------------------------------------------------
  if (r->B > 0) {
    if (r->n_iters >= 0) {
      if (int rc = i2lqr_iterate_pick(h, r->B, r->n_iters, r->X, r->U, r->x_term, r->lamb, r->obs,
                                      r->cost, r->K, r->k, r->iters, r->status, r->qfun,
                                      r->outer_iter, r->max_relax_iter, r->cost_it, r->local_best,
                                      r->local_best_cost, r->pick_ws, r->pick_ws_bytes, stream))
        return rc;
    } else {
      if (!r->qfun) return fail(I2LQR_ERR_INVALID, "null buffer");
      if (int rc = i2lqr_solve(h, r->B, r->X, r->U, r->x_term, r->lamb, r->obs, r->cost, r->K, r->k,
                               r->iters, r->status, stream)) return rc;
      if (int rc = i2lqr_relax_cost(h, r->B, r->X, r->x_term, r->qfun, r->outer_iter,
                                    r->max_relax_iter, r->cost_it, stream)) return rc;
      if (int rc = i2lqr_argmin(h, r->B, r->cost_it, r->local_best, r->local_best_cost, r->pick_ws,
                                r->pick_ws_bytes, stream)) return rc;
    }
  }
  if (two) {
    HIP_TRY(hipEventRecord(h->ev_ready, s));
    HIP_TRY(hipStreamWaitEvent(ss, h->ev_ready, 0));
  }
  // -- the exchange (side stream) ------------------------------------------------------------------
  const bool ragged = r->B != width;
  {
    int64_t padblocks = ragged ? (width + 255) / 256 : 1;
    if (padblocks > 64) padblocks = 64;
    if (h->cfg.dtype == I2LQR_F64)
      hipLaunchKernelGGL((k_round_prepare<double>), dim3((unsigned)padblocks), dim3(256), 0, ss, r->B,
                         n, m, N, (int)h->cfg.layout, (const double*)r->X, (const double*)r->U,
                         r->local_best, (double*)r->pack_local, width, (const double*)r->cost_it,
                         ragged ? (double*)r->cost_padded : (double*)nullptr);
    else
      hipLaunchKernelGGL((k_round_prepare<float>), dim3((unsigned)padblocks), dim3(256), 0, ss, r->B,
                         n, m, N, (int)h->cfg.layout, (const float*)r->X, (const float*)r->U,
                         r->local_best, (float*)r->pack_local, width, (const float*)r->cost_it,
                         ragged ? (float*)r->cost_padded : (float*)nullptr);
    HIP_TRY(hipGetLastError());
  }
  const void* src = ragged ? r->cost_padded : r->cost_it;
  if (comm && !r->loopback) {
    if (int rc = i2lqr_allgather_round(h, comm, src, r->cost_all, width, r->pack_local, r->pack_all,
                                       P, (void*)ss)) return rc;
  } else {  // a world of one without RCCL, or one process playing rank after rank: own slots only
    const size_t item = h->cfg.dtype == I2LQR_F64 ? 8 : 4;
    char* ca = (char*)r->cost_all + (size_t)r->rank * (size_t)width * item;
    char* pa = (char*)r->pack_all + (size_t)r->rank * (size_t)P * item;
    HIP_TRY(hipMemcpyAsync(ca, src, (size_t)width * item, hipMemcpyDeviceToDevice, ss));
    HIP_TRY(hipMemcpyAsync(pa, r->pack_local, (size_t)P * item, hipMemcpyDeviceToDevice, ss));
  }
  if (int rc = i2lqr_round_pick(h, r->world, width, r->total, P, r->cost_all, r->pack_all,
                                r->best_cost, r->winner, r->best_global, r->side_ws,
                                r->side_ws_bytes, (void*)ss)) return rc;
  if (two) {
    HIP_TRY(hipEventRecord(h->ev_side_done, ss));
    h->side_pending = true;
  }
  return debug_check(I2LQR_OK, (void*)ss);
}

}  // extern "C"

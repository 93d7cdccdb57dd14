// Wave-cooperative iLQR kernels for gfx950 (MI355X): LANES lanes of one 64-wide wavefront own one
// iLQR problem (LANES = 64: one problem per wavefront).  The whole problem state — trajectory,
// gains, per-step transcendental cache, the small A/B/Q blocks of the Riccati step — lives in
// that wave's slice of LDS; HBM is touched once on entry and once on exit with time-contiguous,
// coalesced records.  No MFMA: every contraction is n <= 12 wide.
//
// Workgroup = one wavefront (64 threads): there is no s_barrier anywhere; lanes of a wave
// synchronise their LDS traffic with wave_sync() (compiler-level fence, the hardware executes a
// wave's DS instructions in order).
//
// Reference being replaced (paths relative to the reference root):
//   ilqr()          control/iterative_ilqr.py:7-85
//   backward_pass() control/iterative_ilqr.py:88-130  (+ control/ilqr_helper.py:9-150)
//   forward_pass()  control/iterative_ilqr.py:133-160
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "i2lqr_debug.hpp"
#include "i2lqr_systems.hpp"

// A wavefront alone on its SIMD pays ~100 cycles for every TAKEN branch (instruction-fetch bubble;
// tools/ubench_issue.hip): the serial horizon loops take two horizon steps per loop iteration
// (written out by hand: the compiler does not unroll loops that contain wave barriers).

namespace i2lqr {

enum : int { FLAG_HAS_Q = 1, FLAG_HAS_R = 2 };

// Device copy of i2lqr_config, typed and sized for one system; passed by value as a kernel
// argument (kernarg segment: uniform, served by the scalar cache).
template <class T, int n, int m> struct DevCfg {
  int N, max_iter, flags;
  int fast_barrier;  // host: |2 ctrl_q2 u_max[a]| < 600 for every input (see LaneWorker::backward)
  T dt, eps, lamb_factor, max_lamb;
  T ctrl_q1, ctrl_q2, obs_q1, obs_q2, safety_margin;
  T u_max[m];
  T ctrl_c[m];  // exp(-2 ctrl_q2 u_max[a]), rounded from the host's double
  T xtarget[n];
  T Q[n * n], Qt[n * n], R[m * m];
  T sys_par[8];
  // Products / quotients of the constants above that every lane would otherwise form itself and
  // keep in vector registers across the horizon loops (host-computed in T, same roundings):
  // q1 q2 and q1 q2^2 of the two barrier families, and the plant's derived constants
  // (Quad12: {1/mass, arm/Ix, arm/Iy, ctau/Iz, (Iy-Iz)/Ix, (Iz-Ix)/Iy, (Ix-Iy)/Iz, mass g,
  // plant_const(0..5)}).
  T ctrl_q12, ctrl_q122, obs_q12, obs_q122;
  T pd[16];
  unsigned long long* trap;  // debug build: the handle's violation record (i2lqr_debug.hpp); null otherwise
};

// (value, index) pair of the arg-min kernels and of the fused pick epilogue
template <class T> struct MinPair { T v; int64_t i; };

template <class T> __device__ __forceinline__ bool better(T v, int64_t i, T bv, int64_t bi) {
  // NaN never wins; ties resolve to the lower index (Python list.index(min(list)))
  if (v != v) return false;
  if (bi < 0) return true;
  return v < bv || (v == bv && i < bi);
}

// flag on a work set's status word: the problem's results are already in the caller's arrays
constexpr int kStatusDelivered = 0x100;

template <class T> struct IterArgs {
  int64_t B;
  int n_iters;    // iterations to run (iterate) / max iterations (solve)
  int early_exit; // 1: reference exits (solve); 0: fixed count (iterate)
  T* X;           // [B][n][N+1] in/out (X[:, :, 0] = x0 on entry)
  T* U;           // [B][m][N]   in/out
  const T* x_term; // [B][n]
  T* lamb;        // [B] in/out
  const T* obs;   // [B][6] or null
  T* cost;        // [B] out
  T* K;           // [B][m][n][N] out or null
  T* k;           // [B][m][N] out or null
  int32_t* iters;  // [B] out or null
  int32_t* status; // [B] out or null
  unsigned long long* dbg;  // diagnostic builds only: [B][8] phase cycle sums; null otherwise
  // SETIO kernels (tail of the chunked solve of the one-problem-per-lane layouts): the problems are
  // columns 0..*count-1 of a batch-minor, time-major work set with row stride `set_stride`
  // (i2lqr_lane.hpp); the launch does nothing unless *count <= count_max; the iteration counter
  // continues from iters[] and stops at max_total.
  const int32_t* count;
  int count_max, max_total;
  int64_t set_stride;
  // SETIO, optional (orig non-null): a finished problem goes straight to the caller's arrays —
  // problem orig[prob] of out_B, time-major rows like the work set's, batch-tiled or batch-minor —
  // and is marked kStatusDelivered in the work set's status: the compaction that follows has
  // nothing left to move for it (k_lane_compact).
  const int32_t* orig = nullptr;
  T* out_X = nullptr;
  T* out_U = nullptr;
  T* out_K = nullptr;   // null: the caller did not ask for gains
  T* out_k = nullptr;
  T* out_lamb = nullptr;
  T* out_cost = nullptr;
  int32_t* out_iters = nullptr;
  int32_t* out_status = nullptr;
  int64_t out_B = 0;
  int out_tiled = 0;
  // Optional epilogue of the eight-lane kernels (i2lqr_iterate_pick): the relaxed terminal cost of
  // every candidate (utils/base.py:427-437) from the x_N the kernel still holds, and the flat
  // first-index arg-min over them (:462-465) by a last-workgroup-done reduction — one launch per
  // control round instead of four.  qfun null: no epilogue; pick_part null: costs only.
  const int32_t* qfun = nullptr;       // [B] cost-to-go in steps (I2LQR_QF_NONE: empty slot)
  int outer_iter = 0, max_relax_iter = 0;
  T* cost_it = nullptr;                // [B] out
  MinPair<T>* pick_part = nullptr;     // [gridDim.x] partial minima (caller's arg-min workspace)
  unsigned* pick_ticket = nullptr;     // the handle's ticket word (wraps to 0 by itself)
  int64_t* best_idx = nullptr;         // [1] out
  T* best_cost = nullptr;              // [1] out
  // k_group_spec<.., CHAIN>: B chains of chain_len problems each, stored chain after chain
  int chain_len = 1;
};

// relaxed terminal cost of one candidate, utils/base.py:427-437: ss = ||x_N - x_term||_2^2 summed in
// double in component order (k_relax_cost and the fused epilogue share this: bit-identical)
__device__ __forceinline__ double relax_cost_value(double ss, int32_t qf, int N, int outer_iter,
                                                   int max_relax_iter) {
  const double nrm = sqrt(ss);
  double scale = 1.0;
  for (int q = 0; q < outer_iter; q++) scale *= 10.0;
  double out = INFINITY;
  if (qf == 0x7fffffff) return out;  // I2LQR_QF_NONE: an empty candidate slot (i2lqr_select_candidates)
  for (int i = 1; i <= max_relax_iter; i++) {
    if (nrm <= 80.0 * i / scale) { out = (double)qf + N + 100 * i; break; }
    if (nrm > 80.0 * max_relax_iter / scale) break;
  }
  return out;
}

// The pick part of the epilogue.  `cand`: this lane carries candidate `prob` with cost `v`.
// Wavefront minimum -> part[workgroup]; the workgroup that draws the last ticket reduces the
// partial minima and writes the result.  better() is a total order on (value, index) with NaN
// excluded, so the result does not depend on the order of the reduction: it is the flat arg-min
// with first-index tie-break of k_argmin_partial / k_argmin_final.  Called by ONE wavefront per
// workgroup, all 64 lanes.
template <class T>
__device__ __forceinline__ void pick_epilogue(const IterArgs<T>& a, bool cand, int64_t prob, T v) {
  T bv = cand ? v : T(0);
  int64_t bi = (cand && v == v) ? prob : -1;
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) {
    const T ov = __shfl_xor(bv, s);
    const int64_t oi = __shfl_xor(bi, s);
    if (oi >= 0 && better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
  }
  const int lane = threadIdx.x & 63;
  unsigned ticket = 0;
  if (lane == 0) {
    __hip_atomic_store(&a.pick_part[blockIdx.x].v, bv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&a.pick_part[blockIdx.x].i, bi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    // wraps to 0 at gridDim.x - 1: the word is ready for the next launch without a reset
    ticket = atomicInc(a.pick_ticket, gridDim.x - 1);
  }
  ticket = __builtin_amdgcn_readfirstlane(ticket);
  if (ticket != gridDim.x - 1) return;
  __threadfence();
  bv = T(0);
  bi = -1;
  for (unsigned p = lane; p < gridDim.x; p += 64) {
    const T ov = __hip_atomic_load(&a.pick_part[p].v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int64_t oi = __hip_atomic_load(&a.pick_part[p].i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (oi >= 0 && better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
  }
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) {
    const T ov = __shfl_xor(bv, s);
    const int64_t oi = __shfl_xor(bi, s);
    if (oi >= 0 && better(ov, oi, bv, bi)) { bv = ov; bi = oi; }
  }
  if (lane == 0) {
    *a.best_idx = bi;
    *a.best_cost = bi >= 0 ? bv : (T)INFINITY;
  }
}

// Diagnostic build only (-DI2LQR_STAMPS, tools/stamp_build.sh): per-phase cycle shares of one
// wave, accumulated in scalar registers and written to a debug buffer no other code reads.
#ifdef I2LQR_STAMPS
#define STAMP_DECL unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t0 = 0, st_t1 = 0
#define STAMP_BEGIN()                                                            \
  do {                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                           \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_t0)::"memory"); \
    __builtin_amdgcn_sched_barrier(0);                                           \
  } while (0)
#define STAMP_END(slot)                                                          \
  do {                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                           \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_t1)::"memory"); \
    __builtin_amdgcn_sched_barrier(0);                                           \
    st_acc[slot] += st_t1 - st_t0;                                               \
    st_t0 = st_t1;                                                               \
  } while (0)
#else
#define STAMP_DECL
#define STAMP_BEGIN()
#define STAMP_END(slot)
#endif

template <int Begin, int End, class F> __device__ __forceinline__ void static_for_i(F&& f) {
  if constexpr (Begin < End) {
    f(std::integral_constant<int, Begin>{});
    static_for_i<Begin + 1, End>(f);
  }
}

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// a[i] for a lane-dependent i without sending the array to scratch
template <class T, int L> __device__ __forceinline__ T pick(const T (&a)[L], int i) {
  T v = a[0];
#pragma unroll
  for (int q = 1; q < L; q++) v = (i == q) ? a[q] : v;
  return v;
}

// np.clip(v, lo, hi) = min(max(v, lo), hi) with NaN passing through (utils: iterative_ilqr.py:36,
// :145).  v_max / v_min + a NaN select: 5 instructions in fp64 against 12 for two compare-selects.
template <class T> __device__ __forceinline__ T clip(T v, T lo, T hi);
template <> __device__ __forceinline__ double clip<double>(double v, double lo, double hi) {
  const double r = __builtin_fmin(__builtin_fmax(v, lo), hi);
  return v != v ? v : r;
}
template <> __device__ __forceinline__ float clip<float>(float v, float lo, float hi) {
  const float r = __builtin_fminf(__builtin_fmaxf(v, lo), hi);
  return v != v ? v : r;
}

// LDS layout of one problem, in words of T.  Trajectories and gains are TIME-major in LDS
// (X[t][n], U[t][m], Kk[t][m][n+1] with k as the last column) so one horizon step touches one
// contiguous span; the HBM records are component-major with time contiguous (the reference's
// NumPy layout) and are transposed on the way in/out.
template <class Sys> struct Layout {
  static constexpr int n = Sys::n, m = Sys::m, W = n + m, NT = Sys::NTRIG;
  int N;
  int X0, X1, U0, U1, Kk, trg, lu, luu, ob, Va, F, T1, H, g, Qt, Fs, total;
  // fstep: one F = [A | B] per horizon step (written by prep(), parallel over t) instead of one
  // matrix refreshed inside the serial backward recursion
  __host__ __device__ explicit Layout(int N_, bool fstep = false) : N(N_) {
    int o = 0;
    X0 = o; o += n * (N + 1);
    X1 = o; o += n * (N + 1);
    U0 = o; o += m * N;
    U1 = o; o += m * N;
    Kk = o; o += m * (n + 1) * N;
    trg = o; o += NT * N;
    lu = o; o += m * N;
    luu = o; o += m * N;
    ob = o; o += 5 * (N + 1);
    Va = o; o += n * (n + 1);
    F = o; o += n * W;
    T1 = o; o += W * (n + 1);
    H = o; o += W * W;
    g = o; o += W;
    Qt = o; o += n * n;
    Fs = o; o += fstep ? n * W * N : 0;
    total = (o + 1) & ~1;  // keep every problem slice 16-byte aligned for fp64
  }
};

// ---------------------------------------------------------------------------------------------
// The per-problem worker.  All LANES lanes of a problem execute every method together.
// ---------------------------------------------------------------------------------------------
template <class T, class Sys, int LANES, bool HASQR, bool FSTEP = false> struct Worker {
  static constexpr int n = Sys::n, m = Sys::m, W = n + m, NT = Sys::NTRIG;
  using Cfg = DevCfg<T, n, m>;
  const Cfg& c;
  const Layout<Sys> L;
  const Slice<T> S;  // this problem's LDS slice
  const int sl;      // lane index inside the problem's lane group
  const int N;
#ifdef I2LQR_STAMPS
  mutable unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t0 = 0, st_t1 = 0;
#endif

  __device__ Worker(const Cfg& c_, T* smem, int lane)
      : c(c_), L(c_.N, FSTEP),
        S(make_slice(smem + (lane / LANES) * Layout<Sys>(c_.N, FSTEP).total,
                     Layout<Sys>(c_.N, FSTEP).total, c_.trap, TAG_WAVE_LDS)),
        sl(lane % LANES),
        N(c_.N) {}

  // d^T M d with NumPy's association (d.T @ M) @ d: control/iterative_ilqr.py:43-48, :151-159
  template <int D> __device__ __forceinline__ T quad_form(const T* M, const T (&d)[D]) const {
    T acc = T(0);
#pragma unroll
    for (int j = 0; j < D; j++) {
      T col = T(0);
#pragma unroll
      for (int i = 0; i < D; i++) col += d[i] * M[i * D + j];
      acc += col * d[j];
    }
    return acc;
  }

  __device__ __forceinline__ T stage_cost(const T (&x)[n], const T* ref, const T (&u)[m]) const {
    T l = T(0);
    if constexpr (HASQR) {
      T d[n];
#pragma unroll
      for (int i = 0; i < n; i++) d[i] = x[i] - ref[i];
      l += quad_form<n>(c.Q, d);
      l += quad_form<m>(c.R, u);
    }
    return l;
  }

  // Q_terminal is staged in LDS (stage_consts): the n^2 uniform weights would otherwise occupy
  // 2 n^2 SGPRs for the whole kernel and spill
  __device__ __forceinline__ T terminal_cost(const T (&x)[n], const T (&xT)[n]) const {
    T d[n];
#pragma unroll
    for (int i = 0; i < n; i++) d[i] = x[i] - xT[i];
    return quad_form<n>(S + L.Qt, d);
  }

  // once per kernel: constants that live in LDS
  __device__ __forceinline__ void stage_consts() const {
    for (int e = sl; e < n * n; e += LANES) S[L.Qt + e] = c.Qt[e];
    if constexpr (FSTEP) {  // constant pattern of every step's F; prep() fills in the varying entries
      for (int e = sl; e < n * W * N; e += LANES) {
        const int r = e % (n * W);
        S[L.Fs + e] = Sys::jac_const(c, r / W, r % W);
      }
    }
  }

  // -- HBM <-> LDS ---------------------------------------------------------------------------
  // record [comp][len] (time contiguous) <-> LDS [t][comp]; global side coalesced
  __device__ __forceinline__ void load_rec(const T* g, int ldsoff, int comps, int len) const {
    for (int e = sl; e < comps * len; e += LANES) {
      const int cc = e / len, t = e - cc * len;
      S[ldsoff + t * comps + cc] = g[e];
    }
  }
  __device__ __forceinline__ void store_rec(T* g, int ldsoff, int comps, int len) const {
    for (int e = sl; e < comps * len; e += LANES) {
      const int cc = e / len, t = e - cc * len;
      g[e] = S[ldsoff + t * comps + cc];
    }
  }
  // gains: global K[m][n][N], k[m][N]  <->  LDS Kk[t][m][n+1]
  __device__ __forceinline__ void store_gains(T* gK, T* gk) const {
    for (int e = sl; e < m * n * N; e += LANES) {
      const int a = e / (n * N), r = e - a * (n * N), j = r / N, t = r - j * N;
      gK[e] = S[L.Kk + (t * m + a) * (n + 1) + j];
    }
    for (int e = sl; e < m * N; e += LANES) {
      const int a = e / N, t = e - a * N;
      gk[e] = S[L.Kk + (t * m + a) * (n + 1) + n];
    }
  }
  __device__ __forceinline__ void load_gains(const T* gK, const T* gk) const {
    for (int e = sl; e < m * n * N; e += LANES) {
      const int a = e / (n * N), r = e - a * (n * N), j = r / N, t = r - j * N;
      S[L.Kk + (t * m + a) * (n + 1) + j] = gK[e];
    }
    for (int e = sl; e < m * N; e += LANES) {
      const int a = e / N, t = e - a * N;
      S[L.Kk + (t * m + a) * (n + 1) + n] = gk[e];
    }
  }

  // publish an n-vector held (redundantly) in registers: lane 0 stores it contiguously
  template <int D> __device__ __forceinline__ void publish(int off, const T (&v)[D]) const {
    if (sl == 0) {
#pragma unroll
      for (int i = 0; i < D; i++) S[off + i] = v[i];
    }
  }

  // -- nominal rollout + cost: control/iterative_ilqr.py:32-48 -------------------------------
  // Every lane of the problem runs the (serial) recursion redundantly; lane 0 publishes.
  __device__ __forceinline__ T rollout(int Xo, int Uo, const T (&xT)[n]) const {
    T x[n], u[m], xn[n];
#pragma unroll
    for (int i = 0; i < n; i++) x[i] = S[Xo + i];
    T cost = T(0);
    for (int t = 0; t < N; t++) {
#pragma unroll
      for (int a = 0; a < m; a++) u[a] = clip(S[Uo + t * m + a], -c.u_max[a], c.u_max[a]);
      publish<m>(Uo + t * m, u);
      Sys::step(c, x, u, xn);
      publish<n>(Xo + (t + 1) * n, xn);
      cost = cost + stage_cost(x, c.xtarget, u);
#pragma unroll
      for (int i = 0; i < n; i++) x[i] = xn[i];
    }
    cost = cost + terminal_cost(x, xT);
    wave_sync();
    return cost;
  }

  // nominal cost of a trajectory that is already rolled out in LDS (needed only when Q != 0:
  // the nominal stage cost is measured to xtarget, the forward one to x_terminal)
  __device__ __forceinline__ T nominal_cost(int Xo, int Uo, const T (&xT)[n]) const {
    T x[n], u[m];
    T cost = T(0);
    for (int t = 0; t < N; t++) {
#pragma unroll
      for (int i = 0; i < n; i++) x[i] = S[Xo + t * n + i];
#pragma unroll
      for (int a = 0; a < m; a++) u[a] = S[Uo + t * m + a];
      cost = cost + stage_cost(x, c.xtarget, u);
    }
#pragma unroll
    for (int i = 0; i < n; i++) x[i] = S[Xo + N * n + i];
    return cost + terminal_cost(x, xT);
  }

  // -- per-step caches, parallel over t: trig of x_{t+1}, input barrier of u_t, obstacle barrier
  //    of x_t (t = 0..N; index N is the terminal term of get_cost_final) -----------------------
  __device__ __forceinline__ void prep(int Xo, int Uo, const T (&ob)[6]) const {
    for (int t = sl; t <= N; t += LANES) {
      if (t < N) {
        T xe[n], tr[NT];
#pragma unroll
        for (int i = 0; i < n; i++) xe[i] = S[Xo + (t + 1) * n + i];
        Sys::trig(xe, tr);
        T u[m];
#pragma unroll
        for (int a = 0; a < m; a++) u[a] = S[Uo + t * m + a];
        if constexpr (FSTEP) {
          // state-dependent entries of F_t = [A | B] at (x_{t+1}, u_t): control/iterative_ilqr.py:92-99
          T jv[Sys::NVAR];
          Sys::jac_var(c, xe, u, tr, jv);
          static_for_i<0, Sys::NVAR>([&](auto q_) {
            constexpr int q = decltype(q_)::value;
            S[L.Fs + t * (n * W) + Sys::var_idx_c(q)] = jv[q];
          });
        } else {
#pragma unroll
          for (int q = 0; q < NT; q++) S[L.trg + t * NT + q] = tr[q];
        }
        // add_control_constraint(): control/ilqr_helper.py:83-103, one symmetric box per input
#pragma unroll
        for (int a = 0; a < m; a++) {
          const T e_hi = t_exp(c.ctrl_q2 * (u[a] - c.u_max[a]));
          const T e_lo = t_exp(c.ctrl_q2 * (-c.u_max[a] - u[a]));
          T lu = T(0);
          if constexpr (HASQR) {
#pragma unroll
            for (int b = 0; b < m; b++) lu += T(2) * c.R[a * m + b] * u[b];
          }
          lu += c.ctrl_q12 * e_hi - c.ctrl_q12 * e_lo;
          const T luu = c.ctrl_q122 * e_hi +
                        c.ctrl_q122 * e_lo;
          S[L.lu + t * m + a] = lu;
          S[L.luu + t * m + a] = luu;
        }
      }
      // obstacle barrier: control/ilqr_helper.py:32-51 (stage) / :121-147 (terminal, index N)
      T o0 = T(0), o1 = T(0), o2 = T(0), o3 = T(0), o4 = T(0);
      if (ob[5] >= T(0)) {
        const T px = S[Xo + t * n + 0], py = S[Xo + t * n + 1];
        const int opt = (int)ob[5];
        T dz = px - ob[0], dy = py - ob[1];
        if (opt == 1) dy = py - (ob[1] + T(t) * ob[4]);
        if (opt == 2) dz = px - (ob[0] - T(t) * ob[4]);
        const T pa = T(1) / (ob[2] * ob[2]), pb = T(1) / (ob[3] * ob[3]);
        const T h = T(1) + c.safety_margin - (dz * pa * dz + dy * pb * dy);
        const T hd0 = T(-2) * pa * dz, hd1 = T(-2) * pb * dy;
        const T e = t_exp(c.obs_q2 * h);
        const T c1 = c.obs_q12 * e, c2 = c.obs_q122 * e;
        o0 = c1 * hd0;
        o1 = c1 * hd1;
        o2 = c2 * (hd0 * hd0);
        o3 = c2 * (hd0 * hd1);
        o4 = c2 * (hd1 * hd1);
      }
      S[L.ob + t * 5 + 0] = o0;
      S[L.ob + t * 5 + 1] = o1;
      S[L.ob + t * 5 + 2] = o2;
      S[L.ob + t * 5 + 3] = o3;
      S[L.ob + t * 5 + 4] = o4;
    }
    wave_sync();
  }

  // Regularised inverse of Q_uu: control/iterative_ilqr.py:118-123
  //   w, V = eig(Quu); w[w<0] = 0; w += lamb; inv = V diag(1/w) V^T.
  // m == 2: t_quu_inverse2 (i2lqr_systems.hpp).
  // m > 2 runs cyclic Jacobi on the symmetrised matrix (no reference counterpart).
  __device__ __forceinline__ void quu_inverse(const T (&Quu)[m * m], T lamb,
                                              T (&inv)[m * m]) const {
    if constexpr (m == 2) {
      t_quu_inverse2(Quu, lamb, inv);
    } else {
      bool unused = false;
      t_quu_inverse_m<T, m, true>(Quu, lamb, inv, &unused);
    }
  }

  // -- backward pass: control/iterative_ilqr.py:88-130 ----------------------------------------
  // Needs prep() on the same trajectory.  Leaves the gains in LDS (Kk).  Three LDS round trips
  // per horizon step; every lane's role in each phase is fixed, so its indices are hoisted.
  // GENERAL = false (m == 2 plants): Quu is inverted in its positive-definite form only, the loop
  // has no branch, and the return value says whether some Quu was not positive definite — the
  // caller then repeats the pass with GENERAL = true.  Where Quu is positive definite both
  // compute the same numbers (t_quu_inverse2_pd / t_quu_inverse2).
  template <bool GENERAL = true>
  __device__ __forceinline__ bool backward(int Xo, int Uo, const T (&xT)[n], T lamb) const {
    bool bad = false;
    constexpr int NA = n + 1;                               // width of [Vxx | Vx], [K | k]
    constexpr int P1N = W * NA, P2N = W * W, P4N = n * NA;  // elements per phase
    constexpr int P1P = (P1N + LANES - 1) / LANES, P2P = (P2N + LANES - 1) / LANES,
                  P4P = (P4N + LANES - 1) / LANES, GP = (W + LANES - 1) / LANES;
    // constant pattern of F = [A | B]; the state-dependent entries are refreshed every step
    if constexpr (!FSTEP)
      for (int e = sl; e < n * W; e += LANES) S[L.F + e] = Sys::jac_const(c, e / W, e % W);
    constexpr int FW = n * W;                       // words of one F
    const int Fbase = FSTEP ? L.Fs : L.F;           // step t's F sits at Fbase + (FSTEP ? t FW : 0)
    // terminal value function, get_cost_final(): control/ilqr_helper.py:106-150
    for (int e = sl; e < P4N; e += LANES) {
      const int i = e / NA, j = e - i * NA;
      T v;
      if (j < n) {
        v = T(2) * S[L.Qt + i * n + j];
        if (i < 2 && j < 2) v += S[L.ob + N * 5 + 2 + i + j];
      } else {
        v = T(0);
#pragma unroll
        for (int r = 0; r < n; r++) v += T(2) * S[L.Qt + i * n + r] * (S[Xo + N * n + r] - xT[r]);
        if (i < 2) v += S[L.ob + N * 5 + i];
      }
      S[L.Va + e] = v;
    }
    // Fixed lane roles.  Every phase is written branch-free: each lane's LDS addresses (and the
    // weights of optional terms) are computed once here, so that inside the horizon loop all
    // reads of a phase are issued unconditionally up front and waited for once; lanes without
    // an element in a phase read a valid dummy address and only skip the final store.
    int p1_f[P1P], p1_v[P1P];    // column a of F, column j of [Vxx|Vx]
    bool p1_on[P1P];
#pragma unroll
    for (int r = 0; r < P1P; r++) {
      const int e0 = sl + r * LANES;
      p1_on[r] = e0 < P1N;
      const int e = p1_on[r] ? e0 : 0, a = e / NA, j = e - a * NA;
      p1_f[r] = Fbase + a;
      p1_v[r] = L.Va + j;
    }
    // P2: H[a][b] = lconst + [extra term] + T1[a][:] . F[:, b]
    int p2_t[P2P], p2_f[P2P], p2_x[P2P], p2_xs[P2P];
    bool p2_on[P2P], p2_xon[P2P];
    T p2_const[P2P];
#pragma unroll
    for (int r = 0; r < P2P; r++) {
      const int e0 = sl + r * LANES;
      const int e = e0 < P2N ? e0 : 0, a = e / W, b = e - a * W;
      p2_on[r] = e0 < P2N && !(a < n && b >= n);  // Qxu is never used by the reference
      p2_t[r] = L.T1 + a * NA;
      p2_f[r] = Fbase + b;
      // optional term: obstacle block (a, b < 2; control/ilqr_helper.py:51) or the input barrier
      // on the diagonal of l_uu (:28)
      p2_xon[r] = (a < 2 && b < 2) || (a >= n && a == b);
      p2_x[r] = (a >= n && a == b) ? L.luu + (a - n) : L.ob + 2 + ((a < 2 && b < 2) ? a + b : 0);
      p2_xs[r] = (a >= n && a == b) ? m : 5;
      T lc = T(0);
      if constexpr (HASQR) {
        if (a < n && b < n) lc = T(2) * c.Q[a * n + b];       // l_xx = 2Q, :30
        if (a >= n && b >= n) lc = T(2) * c.R[(a - n) * m + (b - n)];  // l_uu = 2R, :28
      }
      p2_const[r] = lc;
    }
    // g[a] = l[a] + T1[a][n]:  l_x obstacle part (a < 2), l_u (a >= n)
    int g_x[GP], g_xs[GP], g_t[GP];
    bool g_on[GP], g_xon[GP];
#pragma unroll
    for (int r = 0; r < GP; r++) {
      const int a0 = sl + r * LANES;
      g_on[r] = a0 < W;
      const int a = g_on[r] ? a0 : 0;
      g_xon[r] = a < 2 || a >= n;
      g_x[r] = (a >= n) ? L.lu + (a - n) : L.ob + (a < 2 ? a : 0);
      g_xs[r] = (a >= n) ? m : 5;
      g_t[r] = L.T1 + a * NA + n;
    }
    // P34: element (i, j) of [Vxx | Vx]
    int p4_gi[P4P], p4_gj[P4P], p4_gjs[P4P], p4_q[P4P];
    bool p4_on[P4P];
#pragma unroll
    for (int r = 0; r < P4P; r++) {
      const int e0 = sl + r * LANES;
      p4_on[r] = e0 < P4N;
      const int e = p4_on[r] ? e0 : 0, i = e / NA, j = e - i * NA;
      p4_gi[r] = L.H + n * W + i;                            // Qux[:, i], stride W
      p4_gj[r] = (j < n) ? L.H + n * W + j : L.g + n;        // Qux[:, j] or Qu
      p4_gjs[r] = (j < n) ? W : 1;
      p4_q[r] = (j < n) ? L.H + i * W + j : L.g + i;         // Qxx[i][j] or Qx[i]
    }
    // P0: the state-dependent entries of F = [A | B] at (x_{t+1}, u_t)
    // (control/iterative_ilqr.py:92-99).  Issued for step t-1 inside the last phase of step t:
    // nothing there reads F, so the refresh costs no LDS round trip of its own.
    T rf_xe[n], rf_u[m], rf_tr[NT];
    auto refresh_load = [&](int t) {  // the LDS reads of the refresh, issued early
#pragma unroll
      for (int i = 0; i < n; i++) rf_xe[i] = S[Xo + (t + 1) * n + i];
#pragma unroll
      for (int a = 0; a < m; a++) rf_u[a] = S[Uo + t * m + a];
#pragma unroll
      for (int q = 0; q < NT; q++) rf_tr[q] = S[L.trg + t * NT + q];
    };
    auto refresh_store = [&]() {
      // every lane holds all NVAR values; lane 0 stores them at their compile-time positions
      // (a lane-indexed select over the array would be demoted to scratch memory by the compiler)
      T jv[Sys::NVAR];
      Sys::jac_var(c, rf_xe, rf_u, rf_tr, jv);
      if (sl == 0) {
        static_for_i<0, Sys::NVAR>([&](auto q_) {
          constexpr int q = decltype(q_)::value;
          S[L.F + Sys::var_idx_c(q)] = jv[q];
        });
      }
    };
    wave_sync();  // F's constant pattern is in place before the first refresh writes into it
    if constexpr (!FSTEP) {
      refresh_load(N - 1);
      refresh_store();
      wave_sync();
    }

    auto step = [&](const int t) __attribute__((always_inline)) {
      STAMP_BEGIN();
      // P1: T1 = F^T [Vxx | Vx]   ((n+m) x (n+1)); f.T @ V of control/iterative_ilqr.py:112-116
      const int fo = FSTEP ? t * FW : 0;
#pragma unroll
      for (int r = 0; r < P1P; r++) {
        T acc = T(0);
#pragma unroll
        for (int i = 0; i < n; i++) acc = t_fma(S[p1_f[r] + fo + i * W], S[p1_v[r] + i * NA], acc);
        if (p1_on[r]) S[L.T1 + sl + r * LANES] = acc;
      }
      wave_sync();
      STAMP_END(1);
      // P2: H = L + T1[:, :n] F  ((n+m) x (n+m): Qxx | . ; Qux | Quu),  g = l + T1[:, n].
      // All LDS reads of the phase (H and g operands) are issued before the first store.
      T g_extra[GP], g_t1n[GP];
#pragma unroll
      for (int r = 0; r < GP; r++) {
        g_extra[r] = S[g_x[r] + t * g_xs[r]];
        g_t1n[r] = S[g_t[r]];
      }
      T h_val[P2P];
#pragma unroll
      for (int r = 0; r < P2P; r++) {
        const T extra = S[p2_x[r] + t * p2_xs[r]];
        T acc = T(0);
#pragma unroll
        for (int i = 0; i < n; i++) acc = t_fma(S[p2_t[r] + i], S[p2_f[r] + fo + i * W], acc);
        h_val[r] = (p2_const[r] + (p2_xon[r] ? extra : T(0))) + acc;
      }
#pragma unroll
      for (int r = 0; r < P2P; r++)
        if (p2_on[r]) S[L.H + sl + r * LANES] = h_val[r];
#pragma unroll
      for (int r = 0; r < GP; r++) {
        T l = g_xon[r] ? g_extra[r] : T(0);
        if constexpr (HASQR) {
          // l_x = 2Q dX[:, t]: control/ilqr_helper.py:29
          const int a = sl + r * LANES;
          if (a < n) {
#pragma unroll
            for (int q = 0; q < n; q++)
              l += T(2) * c.Q[a * n + q] * (S[Xo + t * n + q] - c.xtarget[q]);
          }
        }
        if (g_on[r]) S[L.g + sl + r * LANES] = l + g_t1n[r];
      }
      wave_sync();
      STAMP_END(2);
      // P3+P4 fused: every lane inverts Quu (m x m, redundantly) and forms the two gain columns
      // it needs itself:  [K | k] = -Quu_inv [Qux | Qu]  (control/iterative_ilqr.py:118-126),
      //   [Vxx | Vx] = [Qxx | Qx] - (K^T Quu) [K | k]  with the UNregularised Quu (:128-129).
      T Quu[m * m], Qinv[m * m];
#pragma unroll
      for (int a = 0; a < m; a++)
#pragma unroll
        for (int b = 0; b < m; b++) Quu[a * m + b] = S[L.H + (n + a) * W + (n + b)];
      T Gi[P4P][m], Gj[P4P][m], Qv[P4P];
#pragma unroll
      for (int r = 0; r < P4P; r++) {
#pragma unroll
        for (int b = 0; b < m; b++) {
          Gi[r][b] = S[p4_gi[r] + b * W];
          Gj[r][b] = S[p4_gj[r] + b * p4_gjs[r]];
        }
        Qv[r] = S[p4_q[r]];
      }
      if constexpr (!FSTEP) refresh_load(t > 0 ? t - 1 : 0);  // reads of the next step's refresh
      if constexpr (!GENERAL && m == 2) t_quu_inverse2_pd(Quu, lamb, Qinv, &bad);
      else quu_inverse(Quu, lamb, Qinv);
      STAMP_END(3);
#pragma unroll
      for (int r = 0; r < P4P; r++) {
        const int e = sl + r * LANES;
        T Ki[m], Kj[m];
#pragma unroll
        for (int a = 0; a < m; a++) {
          T ai = T(0), aj = T(0);
#pragma unroll
          for (int b = 0; b < m; b++) {
            ai = t_fma(Qinv[a * m + b], Gi[r][b], ai);
            aj = t_fma(Qinv[a * m + b], Gj[r][b], aj);
          }
          Ki[a] = -ai;
          Kj[a] = -aj;
        }
        if (e < NA) {  // lanes of the first row (i == 0) publish column j = e of [K | k]
#pragma unroll
          for (int a = 0; a < m; a++) S[L.Kk + (t * m + a) * NA + e] = Kj[a];
        }
        T acc = T(0);
#pragma unroll
        for (int b = 0; b < m; b++) {
          T ktq = T(0);
#pragma unroll
          for (int a = 0; a < m; a++) ktq = t_fma(Ki[a], Quu[a * m + b], ktq);
          acc = t_fma(ktq, Kj[b], acc);
        }
        if (p4_on[r]) S[L.Va + e] = Qv[r] - acc;
      }
      STAMP_END(4);
      // (at t == 0 the refresh rewrites step 0's entries: harmless, F is not read again)
      if constexpr (!FSTEP) refresh_store();
      wave_sync();
      STAMP_END(5);
    };
    // two horizon steps per loop iteration (see the note on taken branches at the top of this file)
    int t = N - 1;
    for (; t >= 1; t -= 2) {
      step(t);
      step(t - 1);
    }
    if (t == 0) step(0);
    return bad;
  }

  // -- forward pass: control/iterative_ilqr.py:133-160 ----------------------------------------
  // (Xo, Uo) nominal -> (Xn, Un) candidate; returns cost_new (stage cost measured to x_terminal).
  // GENERAL = false: short sincos kernel only, *bad set if an angle left its range (the caller
  // repeats the pass with GENERAL = true; see t_sincos_fast).
  template <bool GENERAL = true>
  __device__ __forceinline__ T forward(int Xo, int Uo, int Xn, int Un, const T (&xT)[n],
                                       bool* bad = nullptr) const {
    bool bad_local = false;
    bool* const badp = bad ? bad : &bad_local;
    T x[n], u[m], xn[n];
#pragma unroll
    for (int i = 0; i < n; i++) x[i] = S[Xo + i];
    publish<n>(Xn, x);
    T cost = T(0);
    // The nominal state / input / gains of a step are consumed at its very start (the feedback
    // law); the reads for step t+1 are issued right after, into the same registers, so their LDS
    // latency hides under the rest of the serial step (trig, dynamics, publish).
    T xo[n], uo[m], kk[m][n + 1];
    auto load_step = [&](int t) {
#pragma unroll
      for (int j = 0; j < n; j++) xo[j] = S[Xo + t * n + j];
#pragma unroll
      for (int a = 0; a < m; a++) {
        uo[a] = S[Uo + t * m + a];
#pragma unroll
        for (int j = 0; j <= n; j++) kk[a][j] = S[L.Kk + (t * m + a) * (n + 1) + j];
      }
    };
    load_step(0);
    auto step = [&](const int t) __attribute__((always_inline)) {
#pragma unroll
      for (int a = 0; a < m; a++) {
        T acc = T(0);
#pragma unroll
        for (int j = 0; j < n; j++) acc = t_fma(kk[a][j], x[j] - xo[j], acc);
        u[a] = clip(uo[a] + kk[a][n] + acc, -c.u_max[a], c.u_max[a]);
      }
      load_step(t + 1 < N ? t + 1 : t);
      publish<m>(Un + t * m, u);
      T tr[NT];
      Sys::template trig_g<GENERAL>(x, tr, badp);
      Sys::step_tr(c, x, u, tr, xn);
      publish<n>(Xn + (t + 1) * n, xn);
      cost = cost + stage_cost(x, xT, u);
#pragma unroll
      for (int i = 0; i < n; i++) x[i] = xn[i];
    };
    int t = 0;
    for (; t + 1 < N; t += 2) {
      step(t);
      step(t + 1);
    }
    if (t < N) step(t);
    cost = cost + terminal_cost(x, xT);
    wave_sync();
    return cost;
  }
};

// ---------------------------------------------------------------------------------------------
// Kernels.  Grid: ceil(B / (64 / LANES)) workgroups of one wavefront each.
// ---------------------------------------------------------------------------------------------
template <class T> __device__ __forceinline__ bool t_isfinite(T v) {
  return (v - v) == T(0);
}

template <class T, class Sys, int LANES, bool HASQR, bool FSTEP = false, bool SETIO = false>
__global__ __launch_bounds__(64) void k_iterate(const DevCfg<T, Sys::n, Sys::m> c,
                                                const IterArgs<T> a) {
  constexpr int n = Sys::n, m = Sys::m;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int lane = threadIdx.x;
  const int64_t prob = (int64_t)blockIdx.x * (64 / LANES) + lane / LANES;
  if constexpr (SETIO) {
    const int64_t live = *a.count;
    if (live > a.count_max || prob >= live) return;
  } else {
    if (prob >= a.B) return;
  }
  Worker<T, Sys, LANES, HASQR, FSTEP> w(c, reinterpret_cast<T*>(smem_raw), lane);
  const int N = c.N;
  const auto& L = w.L;
  const auto S = w.S;
  const int64_t Bs = SETIO ? a.set_stride : 0;

  // entry: x0, U, x_term, lamb, obs  (HBM -> LDS/registers)
  T xT[n], ob[6];
  if constexpr (SETIO) {
    // a column of the work set: row e of X / U is exactly word e of the time-major LDS copy
    if (w.sl < n) S[L.X0 + w.sl] = a.X[w.sl * Bs + prob];
    for (int e = w.sl; e < m * N; e += LANES) S[L.U0 + e] = a.U[e * Bs + prob];
#pragma unroll
    for (int i = 0; i < n; i++) xT[i] = a.x_term[i * Bs + prob];
#pragma unroll
    for (int q = 0; q < 6; q++) ob[q] = a.obs ? a.obs[q * Bs + prob] : T(q == 5 ? -1 : 1);
  } else {
    // problem-major records, coalesced along the record
    const T* gX = a.X + prob * (int64_t)(n * (N + 1));
    if (w.sl < n) S[L.X0 + w.sl] = gX[w.sl * (N + 1)];
    w.load_rec(a.U + prob * (int64_t)(m * N), L.U0, m, N);
#pragma unroll
    for (int i = 0; i < n; i++) xT[i] = a.x_term[prob * n + i];
#pragma unroll
    for (int q = 0; q < 6; q++) ob[q] = a.obs ? a.obs[prob * 6 + q] : T(q == 5 ? -1 : 1);
  }
  w.stage_consts();
  T lamb = a.lamb[prob];
  const int it0 = SETIO ? a.iters[prob] : 0;             // iterations of the earlier chunks
  const int it_cap = SETIO ? a.max_total - it0 : a.n_iters;
  wave_sync();

  // The nominal rollout of iteration i+1 is bit-identical to the forward rollout of an accepted
  // iteration i (same inputs, same code) and unchanged after a rejected one: it runs once.
  int cur = 0;  // which of the two trajectory buffers holds the nominal
  T cost = w.rollout(L.X0, L.U0, xT);
  int it = 0, status = a.early_exit ? 2 /*MAX_ITER*/ : 0 /*RUNNING*/;
  T cost_ret = cost;
  bool fresh = true;  // the nominal trajectory changed since the last prep()
  while (it < it_cap) {
    const int Xo = cur ? L.X1 : L.X0, Uo = cur ? L.U1 : L.U0;
    const int Xn = cur ? L.X0 : L.X1, Un = cur ? L.U0 : L.U1;
#ifdef I2LQR_STAMPS
    STAMP_DECL;
    STAMP_BEGIN();
#endif
    // the per-step caches depend on the nominal trajectory only: still valid after a rejected step
    if (fresh) w.prep(Xo, Uo, ob);
#ifdef I2LQR_STAMPS
    STAMP_END(0);
#endif
    // optimistic, branch-free passes first; the general forms only if a lane asked for them
    if (__builtin_expect(__any(w.template backward<false>(Xo, Uo, xT, lamb)), 0))
      w.template backward<true>(Xo, Uo, xT, lamb);
#ifdef I2LQR_STAMPS
    STAMP_BEGIN();
#endif
    bool big = false;
    T cost_new = w.template forward<false>(Xo, Uo, Xn, Un, xT, &big);
    if (__builtin_expect(__any(big), 0)) cost_new = w.template forward<true>(Xo, Uo, Xn, Un, xT);
#ifdef I2LQR_STAMPS
    STAMP_END(6);
    if (a.dbg && w.sl == 0) {
      for (int q = 0; q < 8; q++) a.dbg[prob * 8 + q] += st_acc[q] + w.st_acc[q];
      for (int q = 0; q < 8; q++) w.st_acc[q] = 0;
    }
#endif
    it++;
    // accept / reject with the lamb schedule: control/iterative_ilqr.py:74-84
    fresh = cost_new < cost;
    if (fresh) {
      cur ^= 1;
      lamb /= c.lamb_factor;
      const bool conv = t_abs((cost_new - cost) / cost) < c.eps;
      cost_ret = cost_new;
      // next nominal cost: stage terms are measured to xtarget, not x_terminal, when Q != 0
      cost = HASQR ? w.nominal_cost(Xn, Un, xT) : cost_new;
      if (conv) {
        if (a.early_exit) { status = 1; break; }
        if (status == 0) status = 1;
      }
    } else {
      lamb *= c.lamb_factor;
      cost_ret = cost;
      if (lamb > c.max_lamb) {
        if (a.early_exit) { status = 3; break; }
        if (status == 0) status = 3;
      }
    }
  }
  if (!t_isfinite(cost_ret)) status = 4;

  // exit: X, U, gains, scalars (LDS -> HBM)
  const int Xo = cur ? L.X1 : L.X0, Uo = cur ? L.U1 : L.U0;
  if constexpr (SETIO) {
    for (int e = w.sl; e < n * (N + 1); e += LANES) a.X[e * Bs + prob] = S[Xo + e];
    for (int e = w.sl; e < m * N; e += LANES) a.U[e * Bs + prob] = S[Uo + e];
    if (a.K) {  // K rows (t m + a) n + j, k rows t m + a  <-  LDS Kk[t][m][n+1]
      for (int e = w.sl; e < m * N * (n + 1); e += LANES) {
        const int r = e / (n + 1), j = e - r * (n + 1);
        if (j < n) a.K[((int64_t)r * n + j) * Bs + prob] = S[L.Kk + e];
        else a.k[(int64_t)r * Bs + prob] = S[L.Kk + e];
      }
    }
  } else {
    w.store_rec(a.X + prob * (int64_t)(n * (N + 1)), Xo, n, N + 1);
    w.store_rec(a.U + prob * (int64_t)(m * N), Uo, m, N);
    if (a.K) w.store_gains(a.K + prob * (int64_t)(m * n * N), a.k + prob * (int64_t)(m * N));
  }
  if (w.sl == 0) {
    a.lamb[prob] = lamb;
    a.cost[prob] = cost_ret;
    if (a.iters) a.iters[prob] = it0 + it;
    if (a.status) a.status[prob] = status;
  }
}

// rollout only: control/iterative_ilqr.py:32-48
template <class T, class Sys, int LANES, bool HASQR>
__global__ __launch_bounds__(64) void k_rollout(const DevCfg<T, Sys::n, Sys::m> c, int64_t B,
                                                T* X, T* U, const T* x_term, T* cost) {
  constexpr int n = Sys::n, m = Sys::m;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int lane = threadIdx.x;
  const int64_t prob = (int64_t)blockIdx.x * (64 / LANES) + lane / LANES;
  if (prob >= B) return;
  Worker<T, Sys, LANES, HASQR> w(c, reinterpret_cast<T*>(smem_raw), lane);
  const int N = c.N;
  const T* gX = X + prob * (int64_t)(n * (N + 1));
  if (w.sl < n) w.S[w.L.X0 + w.sl] = gX[w.sl * (N + 1)];
  w.load_rec(U + prob * (int64_t)(m * N), w.L.U0, m, N);
  w.stage_consts();
  T xT[n];
#pragma unroll
  for (int i = 0; i < n; i++) xT[i] = x_term[prob * n + i];
  wave_sync();
  const T cst = w.rollout(w.L.X0, w.L.U0, xT);
  w.store_rec(X + prob * (int64_t)(n * (N + 1)), w.L.X0, n, N + 1);
  w.store_rec(U + prob * (int64_t)(m * N), w.L.U0, m, N);
  if (w.sl == 0) cost[prob] = cst;
}

// backward only: control/iterative_ilqr.py:88-130
template <class T, class Sys, int LANES, bool HASQR>
__global__ __launch_bounds__(64) void k_backward(const DevCfg<T, Sys::n, Sys::m> c, int64_t B,
                                                 const T* X, const T* U, const T* x_term,
                                                 const T* lamb, const T* obs, T* K, T* k) {
  constexpr int n = Sys::n, m = Sys::m;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int lane = threadIdx.x;
  const int64_t prob = (int64_t)blockIdx.x * (64 / LANES) + lane / LANES;
  if (prob >= B) return;
  Worker<T, Sys, LANES, HASQR> w(c, reinterpret_cast<T*>(smem_raw), lane);
  const int N = c.N;
  w.load_rec(X + prob * (int64_t)(n * (N + 1)), w.L.X0, n, N + 1);
  w.load_rec(U + prob * (int64_t)(m * N), w.L.U0, m, N);
  w.stage_consts();
  T xT[n], ob[6];
#pragma unroll
  for (int i = 0; i < n; i++) xT[i] = x_term[prob * n + i];
#pragma unroll
  for (int q = 0; q < 6; q++) ob[q] = obs ? obs[prob * 6 + q] : T(q == 5 ? -1 : 1);
  wave_sync();
  w.prep(w.L.X0, w.L.U0, ob);
  w.backward(w.L.X0, w.L.U0, xT, lamb[prob]);
  w.store_gains(K + prob * (int64_t)(m * n * N), k + prob * (int64_t)(m * N));
}

// forward only: control/iterative_ilqr.py:133-160
template <class T, class Sys, int LANES, bool HASQR>
__global__ __launch_bounds__(64) void k_forward(const DevCfg<T, Sys::n, Sys::m> c, int64_t B,
                                                const T* X, const T* U, const T* x_term,
                                                const T* K, const T* k, T* Xn, T* Un,
                                                T* cost_new) {
  constexpr int n = Sys::n, m = Sys::m;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int lane = threadIdx.x;
  const int64_t prob = (int64_t)blockIdx.x * (64 / LANES) + lane / LANES;
  if (prob >= B) return;
  Worker<T, Sys, LANES, HASQR> w(c, reinterpret_cast<T*>(smem_raw), lane);
  const int N = c.N;
  w.load_rec(X + prob * (int64_t)(n * (N + 1)), w.L.X0, n, N + 1);
  w.load_rec(U + prob * (int64_t)(m * N), w.L.U0, m, N);
  w.load_gains(K + prob * (int64_t)(m * n * N), k + prob * (int64_t)(m * N));
  w.stage_consts();
  T xT[n];
#pragma unroll
  for (int i = 0; i < n; i++) xT[i] = x_term[prob * n + i];
  wave_sync();
  const T cst = w.forward(w.L.X0, w.L.U0, w.L.X1, w.L.U1, xT);
  w.store_rec(Xn + prob * (int64_t)(n * (N + 1)), w.L.X1, n, N + 1);
  w.store_rec(Un + prob * (int64_t)(m * N), w.L.U1, m, N);
  if (w.sl == 0) cost_new[prob] = cst;
}

// relaxed terminal cost: utils/base.py:427-437.  One lane per candidate.
template <class T>
__global__ void k_relax_cost(int64_t B, int n, int N, int batch_minor, const T* X,
                             const T* x_term, const int32_t* qfun, int outer_iter,
                             int max_relax_iter, T* cost_it) {
  const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double ss = 0.0;
  for (int i = 0; i < n; i++) {
    // batch_minor: 0 problem-major, 1 batch-minor, 2 batch-tiled (tiles of 64); the lane
    // layouts are time-major: state row of (t, i) is t n + i
    double xN, xt;
    if (batch_minor == 2) {
      const int64_t tile = b >> 6, l = b & 63;
      xN = (double)X[(tile * (int64_t)(n * (N + 1)) + (int64_t)N * n + i) * 64 + l];
      xt = (double)x_term[(tile * n + i) * 64 + l];
    } else if (batch_minor == 1) {
      xN = (double)X[((int64_t)N * n + i) * B + b];
      xt = (double)x_term[(int64_t)i * B + b];
    } else {
      xN = (double)X[b * (int64_t)(n * (N + 1)) + i * (N + 1) + N];
      xt = (double)x_term[b * n + i];
    }
    const double d = xN - xt;
    ss += d * d;
  }
  const double out = relax_cost_value(ss, qfun[b], N, outer_iter, max_relax_iter);
  cost_it[b] = (T)out;
}

}  // namespace i2lqr

// Eight-lanes-per-problem iLQR kernel for gfx950 (MI355X): eight problems share one wavefront.
//
// The one-problem-per-wavefront kernel (i2lqr_wave.hpp) gives each of 64 lanes ONE element of the
// Riccati step's small products, so every multiply-add comes with two LDS reads and the wavefront
// — alone on its SIMD, one instruction per 4 cycles whatever the instruction — is bound by its
// instruction count.  Here a lane owns a whole COLUMN of the blocks instead:
//   lane j (0..n)  of a group holds column j of [Vxx | Vx]   (n values, registers)
//   lane b (0..n-1) forms column b of H = L + (F^T [Vxx|Vx])[:, :n] F, lane n forms g = l + F^T Vx
// so that per horizon step
//   T1[:, j] = F^T Va[:, j]          is lane-local (compile-time sparsity of F = [A | B]),
//   H[:, b]  = sum_i F[i][b] T1[:, i]  needs other lanes' T1 columns only for the <= 3 rows i != b
//                                    in which column b of F is non-zero,
//   [K | k][:, j] = -Quu^-1 H[n:, j]  is lane-local once Quu^-1 is known (every lane forms the
//                                    m x m block from T1 itself),
//   Va'[:, j] = H[:n, j] - K^T (Quu [K|k][:, j])  needs the K columns of the other lanes.
// Two LDS exchanges per step (T1 columns, gain columns) instead of three, each multiply-add fed
// from registers, about a third of the instructions per step — and eight problems per wavefront,
// so batches between 1024 and ~16384 problems no longer leave most of the chip idle.
// The serial forward rollout runs redundantly on the eight lanes of a group, like the 64 lanes of
// the one-problem-per-wavefront kernel.
//
// Requirements on the plant (checked at compile time from Sys::pat): n + m <= 8, and every column b
// of F = [A | B] is non-zero only in rows 0 and 1 (state-dependent or constant), in row b (the
// identity of A) and in at most one further row r(b) holding the constant dt — the structure of
// the kinematic bicycles (systems/kinetic_bicycle.py:30-52).  Q = R = 0 (the reference's defaults,
// utils/base.py:243-246); other weights take the one-problem-per-wavefront kernel.
//
// Reference being replaced: control/iterative_ilqr.py:7-160, control/ilqr_helper.py:9-150 (see
// i2lqr_wave.hpp for the per-phase citations).  Same algorithm; the association of
// K^T Quu K differs (K^T (Quu K) instead of (K^T Quu) K), i.e. results agree with the other kernels
// to round-off, not bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "i2lqr_systems.hpp"
#include "i2lqr_wave.hpp"

namespace i2lqr {

constexpr int kGroup = 8;                 // lanes per problem
constexpr int kGroupsPerWave = 64 / kGroup;

// Compile-time description of the columns of F = [A | B] (see the header comment).
template <class Sys> struct GroupPattern {
  static constexpr int n = Sys::n, m = Sys::m, W = n + m, NV = Sys::NVAR;
  // the further constant-dt row of column b, or -1
  static constexpr int dt_row(int b) {
    for (int i = 2; i < n; i++)
      if (i != b && Sys::pat(i, b) == 2) return i;
    return -1;
  }
  static constexpr bool column_ok(int b) {
    int extra = 0;
    for (int i = 0; i < n; i++) {
      const int code = Sys::pat(i, b);
      if (i < 2) {
        if (code == 2) return false;               // a constant dt in rows 0 / 1 is not encoded
        if (i == b && code != 1) return false;
        continue;
      }
      if (i == b) {
        if (code != 1) return false;               // identity of A
      } else if (code == 2) {
        extra++;
      } else if (code != 0) {
        return false;                              // a state-dependent entry below row 1
      }
    }
    return extra <= 1;
  }
  static constexpr bool ok() {
    if (W > kGroup || n + 1 > kGroup || n < 2) return false;
    for (int b = 0; b < W; b++)
      if (!column_ok(b)) return false;
    return true;
  }
};

// LDS layout in words of T.  Per problem:
//   XU0, XU1  two trajectory buffers, time-major records {x[n], u[m]} of W words (t = 0..N)
//   Kk        gains, [t][a][KW]: K[a][0..n-1], k[a] at column n, padding
//   R         per-step record (t = 0..N), RW words:
//               jv[NV]  state-dependent entries of F_t at (x_{t+1}, u_t)
//               0, 1    constants (targets of the per-lane coefficient reads)
//               lu[m], luu[m]  input-barrier gradient / curvature      (control/ilqr_helper.py:83-103)
//               ob[5]   obstacle-barrier gradient (2) and Gauss-Newton block (3)   (:32-51)
//   TR0, TR1  sin / cos values the plant step evaluated at x_t (t = 0..N), one array per trajectory
//             buffer: the rollouts compute them anyway, prep() reads them back instead of
//             evaluating sincos a second time
// Per wavefront (after the eight problem slices):
//   T1c       exchange buffers for the T1 columns, kT1Stride words per problem, [row pair][column][2]
//             words: the eight columns of a row pair are eight consecutive 16-byte slots, so the
//             eight lanes of a problem store their columns without a bank conflict (column-major
//             put them on two slots: 4-way) and lanes that fetch different columns of one row pair
//             do not collide.  The stride (640 bytes) alternates the problems between the two
//             halves of the 256-byte bank row and the columns of problems 2, 3, 6, 7 are rotated by
//             one slot (t1_word): with that, none of the 16-lane groups a ds_read_b128 is served in
//             has two lanes on one slot for this kernel's read patterns.
//   Qt[n][n]
#ifndef I2LQR_GROUP_UNROLL
#define I2LQR_GROUP_UNROLL 4
#endif
template <class Sys, int G = kGroup> struct GLayout {
  static constexpr int n = Sys::n, m = Sys::m, W = n + m, NV = Sys::NVAR, NT = Sys::NTRIG;
  static constexpr int PW = 64 / G;            // problems per wavefront
  static constexpr int KW = (n + 1 + 1) & ~1;  // gain row [K[a][0..n-1], k[a]] padded to 16 bytes (fp64)
  static constexpr int R_JV = 0, R_ZERO = NV, R_ONE = NV + 1, R_LU = NV + 2, R_LUU = NV + 2 + m,
                       R_OB = NV + 2 + 2 * m, RW = (NV + 2 + 2 * m + 5 + 1) & ~1;
  int N;
  static constexpr int kT1Stride = 80;
  int XU0, XU1, Kk, R, TR0, TR1, total;
  // ws (k_group_iterate<.., WS = true>): the records and the gains live in a caller-provided HBM
  // workspace (ws_words() words per problem, L2-resident at the batch sizes this form is for),
  // LDS keeps the trajectories, the sin / cos caches and ONE step's gain exchange (Kk = m KW
  // words): 4 KB per problem at n=6, N=20 instead of 9.5 KB — four wavefronts per CU instead of two.
  __host__ __device__ explicit GLayout(int N_, bool ws = false) : N(N_) {
    int o = 0;
    XU0 = o; o += W * (N + 1); o = (o + 3) & ~3;
    XU1 = o; o += W * (N + 1); o = (o + 3) & ~3;
    Kk = o; o += ws ? m * KW : m * KW * N; o = (o + 3) & ~3;
    R = o; o += ws ? 0 : RW * (N + 1);
    TR0 = o; o += NT * (N + 1); o = (o + 3) & ~3;
    TR1 = o; o += NT * (N + 1); o = (o + 3) & ~3;
    // keep consecutive problem slices on different LDS banks for group-uniform 16-byte reads:
    // slice stride = 4 * odd words
    o = (o + 3) & ~3;
    if (((o / 4) & 1) == 0) o += 4;
    total = o;
  }
  __host__ __device__ int t1_base() const { return PW * total; }
  // (the sixteen-lane form exchanges columns by DPP row broadcasts: no T1 buffers)
  __host__ __device__ int qt_base() const { return t1_base() + (G == kGroup ? PW * kT1Stride : 0); }
  // control words of the helper wavefronts (k_group_iterate<.., H > 1>): int[8] nominal buffer of
  // each problem, int flags
  __host__ __device__ int ctl_base() const { return (qt_base() + n * n + 3) & ~3; }
  __host__ __device__ int wave_words() const { return ctl_base() + 16; }
  // HBM workspace of the ws form, words per problem: records [N+1][RW], gains [N][m][KW]
  __host__ __device__ int ws_rec() const { return 0; }
  __host__ __device__ int ws_gain() const { return RW * (N + 1); }
  __host__ __device__ int ws_words() const { return (RW * (N + 1) + m * KW * N + 15) & ~15; }
};


// ---------------------------------------------------------------------------------------------
// DPP row broadcasts: the exchange primitive of the sixteen-lane form (G = 16: one problem per
// 16-lane DPP row).  acc += bcast_L(src) * coef, where bcast_L is the value lane L of the row
// holds: v_fmac_f64_dpp / v_fmac_f32_dpp with row_newbcast — the one DPP mode the fp64 ALU of
// gfx90a+ has — costs what a plain multiply-add costs (tools/ubench_dpp.hip: 5.3 cycles
// independent, 8.3 dependent for a wavefront alone on its SIMD) and replaces an LDS write -> wait
// -> read round trip (108 cycles + 5 per 16-byte read).
// Hazard: the hardware needs two wait states between a VALU write of a register and a DPP read of
// it, and inline asm is invisible to the compiler's hazard recogniser.  One statement per
// instruction (so that the scheduler can fill the latency of the step's dependent chain with
// them); dpp_ready() is the fence a producer passes its values through (s_nop 1 tied to the
// registers), and tests/test_isa_hygiene.py checks on the compiled ISA that no *_dpp instruction
// reads a register one of the two instructions in front of it wrote.
// ---------------------------------------------------------------------------------------------
template <int L> __device__ __forceinline__ void bcast_fmac1(double& a, double s, double c) {
  asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(s), "v"(c), "n"(L));
}
template <int L> __device__ __forceinline__ void bcast_fmac1(float& a, float s, float c) {
  asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(s), "v"(c), "n"(L));
}
template <int L, class T, int R>
__device__ __forceinline__ void bcast_fmac(T (&a)[R], const T (&s)[R], T c) {
#pragma unroll
  for (int r = 0; r < R; r++) bcast_fmac1<L>(a[r], s[r], c);
}
// value of lane L of the 16-lane row (v_mov_b64_dpp / v_mov_b32_dpp row_newbcast: emitted by the
// compiler itself, which also keeps its hazards)
template <int L> __device__ __forceinline__ double bcast_mov(double v) {
  long long x = __builtin_bit_cast(long long, v);
  x = __builtin_amdgcn_mov_dpp(x, 0x150 + L, 0xf, 0xf, false);
  return __builtin_bit_cast(double, x);
}
template <int L> __device__ __forceinline__ float bcast_mov(float v) {
  int x = __builtin_bit_cast(int, v);
  x = __builtin_amdgcn_mov_dpp(x, 0x150 + L, 0xf, 0xf, false);
  return __builtin_bit_cast(float, x);
}
template <class T, int R> __device__ __forceinline__ void dpp_ready(T (&v)[R]) {
  if constexpr (R == 8)
    asm("s_nop 1" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
  else if constexpr (R == 6)
    asm("s_nop 1" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]));
  else static_assert(R == 6 || R == 8, "dpp_ready: n + m of the bicycles");
}

template <class T, class Sys, bool WS = false, int G = kGroup> struct GroupWorker {
  static constexpr int n = Sys::n, m = Sys::m, W = n + m, NV = Sys::NVAR, NT = Sys::NTRIG;
  static constexpr int NA = n + 1;
  using Cfg = DevCfg<T, n, m>;
  using GL = GLayout<Sys, G>;
  static_assert(G == kGroup || G == 16, "eight lanes per problem, or one 16-lane DPP row");
  static_assert(G == kGroup || !WS, "the workspace form is built for eight lanes per problem");
  using GP = GroupPattern<Sys>;
  static_assert(GP::ok(), "plant does not have the column structure this kernel is written for");
  const Cfg& c;
  const GL L;
  const Slice<T> S;  // this problem's LDS slice
  const T* const Qt; // the wavefront's copy of Q_terminal
  const int g;       // lane inside the group = column index
  const int N;
  int oR, oKk;       // where the records and the gains sit in the slice (the layout's by default;
                     // the speculative kernel gives each wavefront its own)
  T* T1c;            // this problem's exchange buffer
  int rho;           // its column rotation (see GLayout)
  T* Wp = nullptr;   // WS: this problem's HBM workspace (records, gains)

  // per-lane column description (constant over the kernel)
  int off_c0, off_c1;        // record offsets of F[0][g], F[1][g]
  int off_l0, off_l1;        // record offsets of the cost terms of rows 0, 1
  int off_lu[m];             // ... of rows n..n+m-1
  T own, cdt;                // coefficient of the lane's own T1 column / of column r(g)
  int rsrc;                  // r(g)
#ifdef I2LQR_STAMPS
  mutable unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t0 = 0, st_t1 = 0;
#endif

  __device__ GroupWorker(const Cfg& c_, T* smem, int lane)
      : GroupWorker(c_, smem + (lane / G) * GL(c_.N, WS).total,
                    smem + GL(c_.N, WS).qt_base(), lane % G,
                    GL(c_.N, WS).total) {
    const int p = lane / G;
    if constexpr (G == kGroup) T1c = smem + L.t1_base() + p * GL::kT1Stride;
    rho = (p >> 1) & 1;
  }
  // record t / gains of step t (LDS slice, or the HBM workspace in the WS form); gain_x: where the
  // lanes exchange the gain columns of step t (the gains themselves in LDS; one buffer in WS)
  __device__ __forceinline__ T* rec(int t) const {
    if constexpr (WS) return Wp + L.ws_rec() + t * GL::RW;
    else return S + oR + t * GL::RW;
  }
  __device__ __forceinline__ T* gain(int t) const {
    if constexpr (WS) return Wp + L.ws_gain() + t * (m * GL::KW);
    else return S + oKk + t * (m * GL::KW);
  }
  __device__ __forceinline__ T* gain_x(int t) const {
    if constexpr (WS) return S + oKk;
    else return S + oKk + t * (m * GL::KW);
  }
  // WS: stores to the workspace by one lane are read by the other lanes of the wavefront
  __device__ __forceinline__ void ws_publish() const {
    if constexpr (WS) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_s_waitcnt(0);
      wave_sync();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
  }
  // word of (column, row) in the problem's exchange buffer
  __device__ __forceinline__ int t1_word(int col, int row) const {
    return (row >> 1) * (2 * kGroup) + ((col + rho) & (kGroup - 1)) * 2 + (row & 1);
  }

  // slice: this problem's LDS slice of slice_words words; qt: Q_terminal in LDS
  __device__ GroupWorker(const Cfg& c_, T* slice, const T* qt, int g_, int slice_words)
      : c(c_), L(c_.N, WS), S(make_slice(slice, slice_words, c_.trap, TAG_GROUP_LDS)), Qt(qt),
        g(g_), N(c_.N) {
    oR = L.R; oKk = L.Kk;
    T1c = nullptr;  // set by the caller (with rho) before the first pass
    rho = 0;
    off_c0 = GL::R_ZERO; off_c1 = GL::R_ZERO; off_l0 = GL::R_ZERO; off_l1 = GL::R_ZERO;
#pragma unroll
    for (int a = 0; a < m; a++) off_lu[a] = GL::R_ZERO;
    own = T(0); cdt = T(0); rsrc = 0;
    static_for_i<0, W>([&](auto b_) {
      constexpr int b = decltype(b_)::value;
      if (g == b && b < n) {   // lanes 0..n-1: column b of H
        constexpr int c0 = Sys::pat(0, b), c1 = Sys::pat(1, b);
        off_c0 = c0 == 0 ? GL::R_ZERO : (c0 == 1 ? GL::R_ONE : GL::R_JV + (c0 - 3));
        off_c1 = c1 == 0 ? GL::R_ZERO : (c1 == 1 ? GL::R_ONE : GL::R_JV + (c1 - 3));
        own = (b >= 2) ? T(1) : T(0);
        constexpr int r = GP::dt_row(b);
        cdt = r >= 0 ? c.dt : T(0);
        rsrc = r >= 0 ? r : 0;
        if (b < 2) {  // obstacle block l_xx[a][b], a, b < 2: ob[2 + a + b]
          off_l0 = GL::R_OB + 2 + b;
          off_l1 = GL::R_OB + 3 + b;
        }
      }
    });
    if (g == n) {  // lane n: g = l + T1[:, n]
      own = T(1);
      off_l0 = GL::R_OB + 0;
      off_l1 = GL::R_OB + 1;
#pragma unroll
      for (int a = 0; a < m; a++) off_lu[a] = GL::R_LU + a;
    }
    if constexpr (G == 16) {
      // Sixteen-lane form: the lane's own T1 column enters its H column with coefficient ONE on
      // every lane (columns 0 and 1 included: their identity entry is the "own" term here, not a
      // broadcast of the column to itself with a coefficient 1 read from the record), so the step
      // starts from h = t1 without a multiply.  Lanes past the gradient column carry zeros.
      own = T(1);
      if (g == 0) off_c0 = GL::R_ZERO;
      if (g == 1) off_c1 = GL::R_ZERO;
    }
  }

  __device__ __forceinline__ T terminal_cost(const T (&x)[n], const T (&xT)[n]) const {
    T d[n];
#pragma unroll
    for (int i = 0; i < n; i++) d[i] = x[i] - xT[i];
    T acc = T(0);
#pragma unroll
    for (int j = 0; j < n; j++) {
      T col = T(0);
#pragma unroll
      for (int i = 0; i < n; i++) col += d[i] * Qt[i * n + j];
      acc += col * d[j];
    }
    return acc;
  }

  // acc += F[i][a] * v from the compile-time pattern (exact for ones, skipped for zeros)
  template <int i, int a> __device__ __forceinline__ void f_acc(T& acc, bool& first, T v,
                                                                const T (&jv)[NV]) const {
    constexpr int code = Sys::pat(i, a);
    if constexpr (code == 0) {
      return;
    } else if constexpr (code == 1) {
      acc = first ? v : acc + v;
      first = false;
    } else {
      const T f = code == 2 ? c.dt : jv[code >= 3 ? code - 3 : 0];
      acc = first ? f * v : t_fma(f, v, acc);
      first = false;
    }
  }

  // -- nominal rollout + cost (control/iterative_ilqr.py:32-48), all lanes of the group redundantly
  __device__ __forceinline__ T rollout(int XUo, int TRo, const T (&xT)[n]) const {
    T x[n], u[m], xn[n], tr[NT];
#pragma unroll
    for (int i = 0; i < n; i++) x[i] = S[XUo + i];
    for (int t = 0; t < N; t++) {
#pragma unroll
      for (int a = 0; a < m; a++) u[a] = clip(S[XUo + t * W + n + a], -c.u_max[a], c.u_max[a]);
#pragma unroll
      for (int a = 0; a < m; a++) S[XUo + t * W + n + a] = u[a];
      Sys::trig(x, tr);
#pragma unroll
      for (int q = 0; q < NT; q++) S[TRo + t * NT + q] = tr[q];
      Sys::step_tr(c, x, u, tr, xn);
#pragma unroll
      for (int i = 0; i < n; i++) S[XUo + (t + 1) * W + i] = xn[i];
#pragma unroll
      for (int i = 0; i < n; i++) x[i] = xn[i];
    }
    Sys::trig(x, tr);  // at x_N: the Jacobian of the last step is evaluated there
#pragma unroll
    for (int q = 0; q < NT; q++) S[TRo + N * NT + q] = tr[q];
    const T cost = terminal_cost(x, xT);  // Q = R = 0: only the terminal term
    wave_sync();
    return cost;
  }

  // -- per-step records, lanes of the group take the horizon steps in turn --------------------
  // pa, pb = 1 / width^2, 1 / height^2 of the obstacle: the same for every step, computed once by
  // the kernel.  The sin / cos of x_{t+1} come from the trajectory's cache (TRo).  Input barrier:
  //   l_u = q1 q2 (e_hi - e_lo),  l_uu = q1 q2^2 (e_hi + e_lo),
  //   e_hi = exp(q2 (u - u_max)),  e_lo = exp(q2 (-u_max - u));
  // the inputs of a rolled-out trajectory lie in [-u_max, u_max], so e_hi e_lo = exp(-2 q2 u_max)
  // is a constant and (fp64, |2 q2 u_max| < 600) e_lo = ctrl_c / e_hi: one short exp without range
  // handling and one reciprocal instead of two general exps (a few ulp apart; i2lqr_lane.hpp).
  // Every lane of the group takes the records g, g + 8, ... in turn; there is NO divergent control
  // flow: a lane past the end of the horizon recomputes the last record (and stores the same
  // values again), a problem without obstacle computes the barrier of a dummy ellipse and
  // stores zeros.
  // first / stride: which records this lane takes (g, 8 in the plain kernel; the speculative
  // kernel spreads them over the lanes of all its wavefronts)
  __device__ __forceinline__ void prep(int XUo, int TRo, const T (&ob)[6], T pa, T pb, int first,
                                       int stride) const {
    const bool has_ob = ob[5] >= T(0);
    const int opt = has_ob ? (int)ob[5] : 0;
    const T spd_y = opt == 1 ? ob[4] : T(0), spd_x = opt == 2 ? ob[4] : T(0);
    const int rounds = (N + stride) / stride;  // ceil((N + 1) / stride)
    for (int r = 0; r < rounds; r++) {
      const int t0 = first + r * stride;
      const int t = t0 < N ? t0 : N;          // record index (obstacle term of x_t)
      const int ts = t0 < N ? t0 : N - 1;     // step index (Jacobian entries, input barrier)
      T* Rs = rec(ts);
      {
        T xe[n], tr[NT], u[m], jv[NV];
#pragma unroll
        for (int i = 0; i < n; i++) xe[i] = S[XUo + (ts + 1) * W + i];
#pragma unroll
        for (int a = 0; a < m; a++) u[a] = S[XUo + ts * W + n + a];
#pragma unroll
        for (int q = 0; q < NT; q++) tr[q] = S[TRo + (ts + 1) * NT + q];
        Sys::jac_var(c, xe, u, tr, jv);  // at (x_{t+1}, u_t): control/iterative_ilqr.py:92-99
#pragma unroll
        for (int q = 0; q < NV; q++) Rs[GL::R_JV + q] = jv[q];
#pragma unroll
        for (int a = 0; a < m; a++) {  // add_control_constraint(): control/ilqr_helper.py:83-103
          T e_hi, e_lo;
          if (sizeof(T) == 8 && c.fast_barrier) {
            e_hi = t_exp_bounded(c.ctrl_q2 * (u[a] - c.u_max[a]));
            e_lo = c.ctrl_c[a] * t_rcp(e_hi);
          } else {
            e_hi = t_exp(c.ctrl_q2 * (u[a] - c.u_max[a]));
            e_lo = t_exp(c.ctrl_q2 * (-c.u_max[a] - u[a]));
          }
          Rs[GL::R_LU + a] = c.ctrl_q12 * e_hi - c.ctrl_q12 * e_lo;
          Rs[GL::R_LUU + a] = c.ctrl_q122 * e_hi +
                              c.ctrl_q122 * e_lo;
        }
      }
      T* R = rec(t);
      R[GL::R_ZERO] = T(0);
      R[GL::R_ONE] = T(1);
      // obstacle barrier: control/ilqr_helper.py:32-51 (stage) / :121-147 (terminal, index N);
      // the centre moves by spd per horizon index without dt (:37-43)
      const T px = S[XUo + t * W + 0], py = S[XUo + t * W + 1];
      const T dz = opt == 2 ? px - (ob[0] - T(t) * spd_x) : px - ob[0];
      const T dy = opt == 1 ? py - (ob[1] + T(t) * spd_y) : py - ob[1];
      const T h = T(1) + c.safety_margin - (dz * pa * dz + dy * pb * dy);
      const T hd0 = T(-2) * pa * dz, hd1 = T(-2) * pb * dy;
      const T e = t_exp(c.obs_q2 * h);
      const T c1 = c.obs_q12 * e, c2 = c.obs_q122 * e;
      R[GL::R_OB + 0] = has_ob ? c1 * hd0 : T(0);
      R[GL::R_OB + 1] = has_ob ? c1 * hd1 : T(0);
      R[GL::R_OB + 2] = has_ob ? c2 * (hd0 * hd0) : T(0);
      R[GL::R_OB + 3] = has_ob ? c2 * (hd0 * hd1) : T(0);
      R[GL::R_OB + 4] = has_ob ? c2 * (hd1 * hd1) : T(0);
    }
    wave_sync();
    ws_publish();
  }

  // regularised inverse of Q_uu, m == 2 (control/iterative_ilqr.py:118-123)
  __device__ __forceinline__ void quu_inverse(const T (&Quu)[m * m], T lamb, T (&inv)[m * m]) const {
    static_assert(m == 2, "the eight-lane kernel is written for m == 2 plants");
    t_quu_inverse2(Quu, lamb, inv);
  }

  // -- backward pass: control/iterative_ilqr.py:88-130.  Needs prep() on the same trajectory;
  //    leaves the gains in LDS (Kk).
  //    GENERAL = false: Quu is inverted in its positive-definite form only and the loop has no
  //    branch; returns true if some Quu was not positive definite — the caller then repeats the
  //    pass with GENERAL = true (eigenvalue clamping of control/iterative_ilqr.py:118-123 behind a
  //    branch).  Where Quu is positive definite both compute the same numbers.
  //    commit = false (a problem that has already terminated while others of its wavefront still
  //    run): the gains in LDS are left as the problem's last iteration wrote them.
  template <bool GENERAL>
  __device__ __forceinline__ bool backward(int XUo, const T (&xT)[n], T lamb, bool commit) const {
    if constexpr (G == 16) return backward_row<GENERAL>(XUo, xT, lamb, commit);
    bool bad = false;
    // terminal value function, get_cost_final(): control/ilqr_helper.py:106-150.
    // va[i] = column g of [Vxx | Vx]
    T va[n];
    {
      const T* Rn = rec(N);
      T dx[n];
#pragma unroll
      for (int i = 0; i < n; i++) dx[i] = S[XUo + N * W + i] - xT[i];
#pragma unroll
      for (int i = 0; i < n; i++) {
        T vxx = T(0), vx = T(0);
#pragma unroll
        for (int r = 0; r < n; r++) {
          const T q = T(2) * Qt[i * n + r];
          vxx = (g == r) ? q : vxx;
          vx += q * dx[r];
        }
        va[i] = (g == n) ? vx : vxx;
      }
      // obstacle terms of rows 0, 1 (per-lane record offsets, zero for the other lanes)
      va[0] += Rn[off_l0];
      va[1] += Rn[off_l1];
    }
    const int gcol = g < GL::KW ? g : GL::KW - 1;  // lanes past the gain row write its padding word
    // The record of a step (uniform and per-lane words) is loaded one step ahead, behind the gain
    // exchange of the previous step: its LDS latency hides under that step's value update.
    // WS: the record comes from the HBM workspace (L2): a whole step ahead, into the other of two
    // register sets.
    struct Rec { T jv[NV], luu[m], c0, c1, l0, l1, lrow[m]; };
    Rec ra, rb;
    auto load_record = [&](int t, Rec& r) __attribute__((always_inline)) {
      const T* R = rec(t);
#pragma unroll
      for (int q = 0; q < NV; q++) r.jv[q] = R[GL::R_JV + q];
#pragma unroll
      for (int a = 0; a < m; a++) r.luu[a] = R[GL::R_LUU + a];
      r.c0 = R[off_c0];
      r.c1 = R[off_c1];
      r.l0 = R[off_l0];
      r.l1 = R[off_l1];
#pragma unroll
      for (int a = 0; a < m; a++) r.lrow[a] = R[off_lu[a]];
    };
    load_record(N - 1, ra);
    auto step = [&](const int t, Rec& rc, Rec& rn) __attribute__((always_inline)) {
      if constexpr (WS) load_record(t > 0 ? t - 1 : 0, rn);
      const T (&jv)[NV] = rc.jv;
      const T (&luu)[m] = rc.luu;
      const T (&lrow)[m] = rc.lrow;
      const T c0 = rc.c0, c1 = rc.c1, l0 = rc.l0, l1 = rc.l1;
      STAMP_BEGIN();
      // P1: own column of T1 = F^T [Vxx | Vx]  (f.T @ V of control/iterative_ilqr.py:112-116)
      T t1[W];
      static_for_i<0, W>([&](auto a_) {
        constexpr int a = decltype(a_)::value;
        T acc = T(0);
        bool first = true;
        static_for_i<0, n>([&](auto i_) {
          constexpr int i = decltype(i_)::value;
          f_acc<i, a>(acc, first, va[i], jv);
        });
        t1[a] = acc;
      });
#pragma unroll
      for (int a = 0; a < W; a++) T1c[t1_word(g, a)] = t1[a];
      wave_sync();
      STAMP_END(1);
      // P2: column g of H = L + T1[:, :n] F (lanes < n), g = l + T1[:, n] (lane n)
      T h[W];
      {
        T s0[W], s1[W], sr[W];
#pragma unroll
        for (int a = 0; a < W; a++) {
          s0[a] = T1c[t1_word(0, a)];
          s1[a] = T1c[t1_word(1, a)];
          sr[a] = T1c[t1_word(rsrc, a)];
        }
#pragma unroll
        for (int a = 0; a < W; a++) {
          T acc = c0 * s0[a];
          acc = t_fma(c1, s1[a], acc);
          acc = t_fma(own, t1[a], acc);
          acc = t_fma(cdt, sr[a], acc);
          h[a] = acc;
        }
        h[0] += l0;
        h[1] += l1;
#pragma unroll
        for (int a = 0; a < m; a++) h[n + a] += lrow[a];
      }
      STAMP_END(2);
      // Quu (every lane, from the T1 columns its F columns touch) and its regularised inverse
      T Quu[m * m], Qinv[m * m];
      static_for_i<0, m>([&](auto a_) {
        constexpr int a = decltype(a_)::value;
        static_for_i<0, m>([&](auto b_) {
          constexpr int b = decltype(b_)::value;
          T acc = T(0);
          bool first = true;
          static_for_i<0, n>([&](auto i_) {
            constexpr int i = decltype(i_)::value;
            if constexpr (Sys::pat(i, n + b) != 0) f_acc<i, n + b>(acc, first, T1c[t1_word(i, n + a)], jv);
          });
          Quu[a * m + b] = (a == b ? luu[a] : T(0)) + acc;
        });
      });
      if constexpr (GENERAL) quu_inverse(Quu, lamb, Qinv);
      else t_quu_inverse2_pd(Quu, lamb, Qinv, &bad);
      STAMP_END(3);
      // own column of [K | k] = -Quu_inv [Qux | Qu]: control/iterative_ilqr.py:118-126
      T kc[m];
#pragma unroll
      for (int a = 0; a < m; a++) {
        T acc = T(0);
#pragma unroll
        for (int b = 0; b < m; b++) acc = t_fma(Qinv[a * m + b], h[n + b], acc);
        kc[a] = -acc;
      }
      T* Kt = gain_x(t);
      if constexpr (WS) {
        // the exchange buffer is one step's: always written; the stored gains (workspace) keep
        // those of the problem's last executed iteration
        T* Kg = gain(t);
#pragma unroll
        for (int a = 0; a < m; a++) {
          Kt[a * GL::KW + gcol] = kc[a];
          if (commit) Kg[a * GL::KW + gcol] = kc[a];
        }
      } else {
#pragma unroll
        for (int a = 0; a < m; a++)
          if (commit) Kt[a * GL::KW + gcol] = kc[a];
      }
      wave_sync();
      STAMP_END(4);
      // value update with the UNregularised Quu: control/iterative_ilqr.py:128-129
      //   Va'[:, g] = H[:n, g] - K^T (Quu [K|k][:, g])
      T kr[m][n];
#pragma unroll
      for (int a = 0; a < m; a++)
#pragma unroll
        for (int i = 0; i < n; i++) kr[a][i] = Kt[a * GL::KW + i];
      if constexpr (!WS) load_record(t > 0 ? t - 1 : 0, rn);
      T qk[m];
#pragma unroll
      for (int a = 0; a < m; a++) {
        T acc = T(0);
#pragma unroll
        for (int b = 0; b < m; b++) acc = t_fma(Quu[a * m + b], kc[b], acc);
        qk[a] = acc;
      }
#pragma unroll
      for (int i = 0; i < n; i++) {
        T acc = T(0);
#pragma unroll
        for (int a = 0; a < m; a++) acc = t_fma(kr[a][i], qk[a], acc);
        va[i] = h[i] - acc;
      }
      STAMP_END(5);
    };
    // four horizon steps per loop iteration (I2LQR_GROUP_UNROLL; then two, then one): a taken branch
    // costs a lone wavefront ~100 cycles — 0.200 -> 0.196 ms per 10 iterations at 1024 problems
    // against two steps per iteration; ten steps per iteration gave half of that back
    int t = N - 1;
    if constexpr (I2LQR_GROUP_UNROLL >= 4) {
      for (; t >= 3; t -= 4) {
        step(t, ra, rb);
        step(t - 1, rb, ra);
        step(t - 2, ra, rb);
        step(t - 3, rb, ra);
      }
    }
    for (; t >= 1; t -= 2) {
      step(t, ra, rb);
      step(t - 1, rb, ra);
    }
    if (t == 0) step(0, ra, rb);
    ws_publish();  // WS: the gains are read back by all lanes of the group in the forward pass
    return bad;
  }

  // -- backward pass, sixteen lanes per problem (G = 16): the same column ownership — lane j < n
  //    holds column j of [Vxx | Vx] and forms column j of H, lane n the gradient column, lanes
  //    above n idle along — but NO LDS exchange in the step: a problem is one 16-lane DPP row, and
  //    every foreign operand of the step is a T1 entry of a lane whose index is known at compile
  //    time, so it arrives as the broadcast operand of a multiply-add (bcast_fmac):
  //      H[a][g]  = own T1[a][g] + F[0][g] T1[a][0] + F[1][g] T1[a][1] + dt T1[a][r(g)]   (P2)
  //      Quu[a][b] = l_uu + sum_k B[k][b] T1[n+a][k]
  //      Va'[i][g] = H[i][g] - sum_a K[a][i] (Quu [K|k][:, g])[a],  K[:, i] = lane i's gain column
  //    so the gain columns are not exchanged through LDS either: every lane stores its column for
  //    the forward pass and nothing waits for the store.  130 instructions per step at the issue
  //    rate of a lone wavefront (683 cycles), against 150 and two LDS round trips (2 x 108 cycles
  //    + 20 16-byte reads; 1293 cycles) with eight lanes per problem.
  template <bool GENERAL>
  __device__ __forceinline__ bool backward_row(int XUo, const T (&xT)[n], T lamb, bool commit) const {
    static_assert(GL::KW > n + 1, "the padding word of a gain row takes the stores of idle lanes");
    bool bad = false;
    T va[n];
    {
      const T* Rn = rec(N);
      T dx[n];
#pragma unroll
      for (int i = 0; i < n; i++) dx[i] = S[XUo + N * W + i] - xT[i];
      // column g of 2 Qt by a per-lane LDS address (lanes past the columns read column 0 and drop
      // it) instead of a chain of selects over the whole matrix; the gradient 2 Qt (x_N - x_T) in
      // the same operation order as the other kernel families
      const int gq = g < n ? g : 0;
#pragma unroll
      for (int i = 0; i < n; i++) {
        T vx = T(0);
#pragma unroll
        for (int r = 0; r < n; r++) vx += (T(2) * Qt[i * n + r]) * dx[r];
        const T vxx = T(2) * Qt[i * n + gq];
        va[i] = (g == n) ? vx : (g < n ? vxx : T(0));
      }
      va[0] += Rn[off_l0];
      va[1] += Rn[off_l1];
    }
    // lanes past the gain row and problems that no longer commit store into the row's padding word
    const int gcol = (commit && g <= n) ? g : GL::KW - 1;
    // dt entry of column g of A in row r: per-lane coefficient of the broadcast from lane r
    T cdr[n];
#pragma unroll
    for (int r = 0; r < n; r++) cdr[r] = (rsrc == r) ? cdt : T(0);
    struct Rec { T jv[NV], luu[m], c0, c1, l0, l1, lrow[m]; };
    Rec ra, rb;
    auto load_record = [&](int t, Rec& r) __attribute__((always_inline)) {
      const T* R = rec(t);
#pragma unroll
      for (int q = 0; q < NV; q++) r.jv[q] = R[GL::R_JV + q];
#pragma unroll
      for (int a = 0; a < m; a++) r.luu[a] = R[GL::R_LUU + a];
      r.c0 = R[off_c0];
      r.c1 = R[off_c1];
      r.l0 = R[off_l0];
      r.l1 = R[off_l1];
#pragma unroll
      for (int a = 0; a < m; a++) r.lrow[a] = R[off_lu[a]];
    };
    // F[k][n + b] as a register operand
    auto f_entry = [&](auto k_, auto b_, const T (&jv)[NV]) __attribute__((always_inline)) {
      constexpr int code = Sys::pat(decltype(k_)::value, n + decltype(b_)::value);
      if constexpr (code == 1) return T(1);
      else if constexpr (code == 2) return c.dt;
      else return jv[code >= 3 ? code - 3 : 0];
    };
    load_record(N - 1, ra);
    auto step = [&](const int t, Rec& rc, Rec& rn) __attribute__((always_inline)) {
      const T (&jv)[NV] = rc.jv;
      // P1: own column of T1 = F^T [Vxx | Vx]
      T t1[W];
      static_for_i<0, W>([&](auto a_) {
        constexpr int a = decltype(a_)::value;
        T acc = T(0);
        bool first = true;
        static_for_i<0, n>([&](auto i_) {
          constexpr int i = decltype(i_)::value;
          f_acc<i, a>(acc, first, va[i], jv);
        });
        t1[a] = acc;
      });
      dpp_ready(t1);
      load_record(t > 0 ? t - 1 : 0, rn);  // lands under the broadcast block
      // Source order = the order that keeps the serial chain short: Quu and the input rows of the
      // H column first (the inverse and the gain column wait for them), the state rows of the
      // column — 4 n independent multiply-adds nothing waits for until the value update — behind
      // them, where they fill the latency of the chain Quu -> inverse -> gain column.
      // Quu (every lane) = l_uu + sum_k B[k][b] T1[n+a][k]
      T Quu[m * m], Qinv[m * m];
      {
        T t1u[m];
#pragma unroll
        for (int a = 0; a < m; a++) t1u[a] = t1[n + a];
        static_for_i<0, m>([&](auto b_) {
          constexpr int b = decltype(b_)::value;
          T col[m];
#pragma unroll
          for (int a = 0; a < m; a++) col[a] = (a == b) ? rc.luu[a] : T(0);
          static_for_i<0, n>([&](auto k_) {
            constexpr int k = decltype(k_)::value;
            if constexpr (Sys::pat(k, n + b) != 0) bcast_fmac<k>(col, t1u, f_entry(k_, b_, jv));
          });
#pragma unroll
          for (int a = 0; a < m; a++) Quu[a * m + b] = col[a];
        });
      }
      // P2: column g of H (lanes < n) / the gradient column (lane n), rows [r0, r0 + R)
      auto h_rows = [&](auto r0_, auto& hr) __attribute__((always_inline)) {
        constexpr int r0 = decltype(r0_)::value;
        constexpr int R = sizeof(hr) / sizeof(T);
        T src[R];
#pragma unroll
        for (int a = 0; a < R; a++) { src[a] = t1[r0 + a]; hr[a] = t1[r0 + a]; }  // own = 1
        bcast_fmac<0>(hr, src, rc.c0);
        bcast_fmac<1>(hr, src, rc.c1);
        static_for_i<2, n>([&](auto r_) {
          constexpr int r = decltype(r_)::value;
          constexpr bool used = [] {
            for (int b = 0; b < n; b++)
              if (GP::dt_row(b) == r) return true;
            return false;
          }();
          if constexpr (used) bcast_fmac<r>(hr, src, cdr[r]);
        });
      };
      T hu[m], hx[n];
      h_rows(std::integral_constant<int, n>{}, hu);
#pragma unroll
      for (int a = 0; a < m; a++) hu[a] += rc.lrow[a];
      if constexpr (GENERAL) quu_inverse(Quu, lamb, Qinv);
      else t_quu_inverse2_pd(Quu, lamb, Qinv, &bad);
      h_rows(std::integral_constant<int, 0>{}, hx);
      hx[0] += rc.l0;
      hx[1] += rc.l1;
      // own column of [K | k] = -Quu_inv [Qux | Qu]: control/iterative_ilqr.py:118-126
      // (pk = Quu_inv H[n:, g] = -kc: the sign rides in the operands that follow)
      T pk[m], kc[m];
#pragma unroll
      for (int a = 0; a < m; a++) {
        T acc = T(0);
#pragma unroll
        for (int b = 0; b < m; b++) acc = t_fma(Qinv[a * m + b], hu[b], acc);
        pk[a] = acc;
        kc[a] = -acc;
      }
      T* Kt = gain_x(t);
#pragma unroll
      for (int a = 0; a < m; a++) Kt[a * GL::KW + gcol] = kc[a];
      // value update with the UNregularised Quu: control/iterative_ilqr.py:128-129
      //   Va'[i][g] = H[i][g] - sum_a K[a][i] (Quu kc_g)[a],  K[a][i] = -pk[a] of LANE i: the
      //   broadcast operand of the multiply-add — the gain columns are not exchanged through LDS
      T qk[m];
#pragma unroll
      for (int a = 0; a < m; a++) {
        T acc = T(0);
#pragma unroll
        for (int b = 0; b < m; b++) acc = t_fma(Quu[a * m + b], kc[b], acc);
        qk[a] = acc;
      }
      T vn[n];
#pragma unroll
      for (int i = 0; i < n; i++) vn[i] = hx[i];
      static_for_i<0, n>([&](auto i_) {
        constexpr int i = decltype(i_)::value;
#pragma unroll
        for (int a = 0; a < m; a++) bcast_fmac1<i>(vn[i], pk[a], qk[a]);
      });
#pragma unroll
      for (int i = 0; i < n; i++) va[i] = vn[i];
    };
    STAMP_BEGIN();  // (whole pass: per-phase stamps inside the step serialise what they measure)
    int t = N - 1;
    if constexpr (I2LQR_GROUP_UNROLL >= 4) {
      for (; t >= 3; t -= 4) {
        step(t, ra, rb);
        step(t - 1, rb, ra);
        step(t - 2, ra, rb);
        step(t - 3, rb, ra);
      }
    }
    for (; t >= 1; t -= 2) {
      step(t, ra, rb);
      step(t - 1, rb, ra);
    }
    if (t == 0) step(0, ra, rb);
    STAMP_END(1);
    wave_sync();  // the forward pass reads the gain columns the other lanes stored
    return bad;
  }

  // -- forward pass, sixteen lanes per problem, plants whose heading is known a step ahead
  //    (Sys::kHeadingAhead).  The rollout is serial and every lane of a problem runs all of it;
  //    what the two halves of the 16-lane row can share is the work INSIDE a step:
  //      * the feedback law and the clip of input a run on half a (lanes 0-7: input 0, lanes
  //        8-15: input 1; each lane reads only its input's gain row), the two inputs are
  //        exchanged by row broadcasts;
  //      * sin / cos — 45 of the 117 instructions of a step — are evaluated ONCE PER TWO STEPS:
  //        after step t the lower half holds the heading of x_{t+1}, the upper half that of
  //        x_{t+2} (= theta_{t+1} + delta_{t+1} dt: it does not depend on the inputs of step
  //        t+1), one evaluation serves both and each half stores its pair into the trajectory's
  //        sin / cos cache itself.
  //    Same operations on the same operands as forward<false>: bit-identical states and inputs
  //    (~75 instructions per step against 117).
  __device__ __forceinline__ T forward_row(int XUo, int XUn, int TRn, const T (&xT)[n],
                                           bool* bad) const {
    static_assert(G == 16 && m == 2, "two inputs on the two halves of a 16-lane row");
    const int half = g >> 3;
    const T umax = half ? c.u_max[1] : c.u_max[0];
    T x[n], u[m], xn[n], tr[NT];
#pragma unroll
    for (int i = 0; i < n; i++) x[i] = S[XUo + i];
#pragma unroll
    for (int i = 0; i < n; i++) S[XUn + i] = x[i];
    Sys::template trig_g<false>(x, tr, bad);
#pragma unroll
    for (int q = 0; q < NT; q++) S[TRn + q] = tr[q];
    // Running bases, bumped once per pair of steps; every access of a step is a base plus a
    // compile-time offset (a per-step address is otherwise a vector add per array and step).
    // The look-ahead loads of the last step read record N of the nominal buffer (its input slot
    // is unused) and one gain row past the last step's (the record array follows in the slice):
    // in bounds, never consumed.
    static_assert(!WS, "gain rows and records are neighbours in the LDS slice");
    auto pXo = S + XUo;                        // nominal record of step t
    auto pXn = S + XUn;                        // candidate record of step t
    auto pG = S + (oKk + half * GL::KW);       // this half's gain row of step t
    auto pTR = S + (TRn + half * NT);          // this half's sin / cos slot: record t + half
    T xo[n], uo, kk[NA];
    auto load_step = [&](auto k_) __attribute__((always_inline)) {
      constexpr int k = decltype(k_)::value;
#pragma unroll
      for (int j = 0; j < n; j++) xo[j] = pXo[k * W + j];
      uo = pXo[k * W + n + half];
#pragma unroll
      for (int j = 0; j < NA; j++) kk[j] = pG[k * (m * GL::KW) + j];
    };
    load_step(std::integral_constant<int, 0>{});
    auto step = [&](auto k_, const T (&trc)[NT]) __attribute__((always_inline)) {
      constexpr int k = decltype(k_)::value;   // step t + k of the pair that starts at t
      T acc = T(0);
#pragma unroll
      for (int j = 0; j < n; j++) acc = t_fma(kk[j], x[j] - xo[j], acc);
      const T ua = clip(uo + kk[n] + acc, -umax, umax);
      load_step(std::integral_constant<int, k + 1>{});
      pXn[k * W + n + half] = ua;
      u[0] = bcast_mov<0>(ua);
      u[1] = bcast_mov<8>(ua);
      Sys::step_tr(c, x, u, trc, xn);
#pragma unroll
      for (int i = 0; i < n; i++) pXn[(k + 1) * W + i] = xn[i];
#pragma unroll
      for (int i = 0; i < n; i++) x[i] = xn[i];
    };
    int t = 0;
    for (; t + 1 < N; t += 2) {
      step(std::integral_constant<int, 0>{}, tr);  // x = x_{t+1}
      T sc[NT], tr1[NT];
      Sys::trig_heading_fast(half ? Sys::next_heading(c, x) : Sys::heading(x), sc, bad);
#pragma unroll
      for (int q = 0; q < NT; q++) pTR[NT + q] = sc[q];   // record t + 1 + half
#pragma unroll
      for (int q = 0; q < NT; q++) {
        tr1[q] = bcast_mov<0>(sc[q]);
        tr[q] = bcast_mov<8>(sc[q]);   // of x_{t+2}: the next pair's first step
      }
      step(std::integral_constant<int, 1>{}, tr1);  // x = x_{t+2}
      pXo = pXo + 2 * W;
      pXn = pXn + 2 * W;
      pG = pG + 2 * (m * GL::KW);
      pTR = pTR + 2 * NT;
    }
    if (t < N) {  // odd horizon: the last step, and the sin / cos of x_N for the records
      step(std::integral_constant<int, 0>{}, tr);
      Sys::template trig_g<false>(x, tr, bad);
#pragma unroll
      for (int q = 0; q < NT; q++) S[TRn + N * NT + q] = tr[q];
    }
    const T cost = terminal_cost(x, xT);
    wave_sync();
    return cost;
  }

  // -- forward pass: control/iterative_ilqr.py:133-160; all lanes of the group redundantly ------
  //    GENERAL = false: short sincos kernel only, *bad set if an angle left its range (the caller
  //    repeats the pass with GENERAL = true); see t_sincos_fast.
  template <bool GENERAL>
  __device__ __forceinline__ T forward(int XUo, int XUn, int TRn, const T (&xT)[n],
                                       bool* bad) const {
    if constexpr (G == 16 && !GENERAL && Sys::kHeadingAhead && m == 2)
      return forward_row(XUo, XUn, TRn, xT, bad);
    T x[n], u[m], xn[n], tr[NT];
#pragma unroll
    for (int i = 0; i < n; i++) x[i] = S[XUo + i];
#pragma unroll
    for (int i = 0; i < n; i++) S[XUn + i] = x[i];
    // WS: the gains come from the HBM workspace (L2): two steps ahead, two register sets in turn
    T xo[n], uo[m];
    struct Gains { T kk[m][NA]; };
    Gains ga, gb;
    auto load_gains = [&](int t, Gains& q) __attribute__((always_inline)) {
      const T* Kg = gain(t);
#pragma unroll
      for (int a = 0; a < m; a++)
#pragma unroll
        for (int j = 0; j < NA; j++) q.kk[a][j] = Kg[a * GL::KW + j];
    };
    auto load_step = [&](int t, Gains& q) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < n; j++) xo[j] = S[XUo + t * W + j];
#pragma unroll
      for (int a = 0; a < m; a++) uo[a] = S[XUo + t * W + n + a];
      if constexpr (!WS) load_gains(t, q);
    };
    if constexpr (WS) {
      load_gains(0, ga);
      load_gains(N >= 2 ? 1 : 0, gb);
    }
    load_step(0, ga);
    auto step = [&](const int t, Gains& q, Gains& qn) __attribute__((always_inline)) {
#pragma unroll
      for (int a = 0; a < m; a++) {
        T acc = T(0);
#pragma unroll
        for (int j = 0; j < n; j++) acc = t_fma(q.kk[a][j], x[j] - xo[j], acc);
        u[a] = clip(uo[a] + q.kk[a][n] + acc, -c.u_max[a], c.u_max[a]);
      }
      if constexpr (WS) {
        load_gains(t + 2 < N ? t + 2 : N - 1, q);  // the set just consumed
        load_step(t + 1 < N ? t + 1 : t, qn);
      } else {
        load_step(t + 1 < N ? t + 1 : t, q);       // one set, reloaded in place
      }
#pragma unroll
      for (int a = 0; a < m; a++) S[XUn + t * W + n + a] = u[a];
      Sys::template trig_g<GENERAL>(x, tr, bad);
#pragma unroll
      for (int q = 0; q < NT; q++) S[TRn + t * NT + q] = tr[q];
      Sys::step_tr(c, x, u, tr, xn);
#pragma unroll
      for (int i = 0; i < n; i++) S[XUn + (t + 1) * W + i] = xn[i];
#pragma unroll
      for (int i = 0; i < n; i++) x[i] = xn[i];
    };
    int t = 0;
    if constexpr (WS) {
      for (; t + 1 < N; t += 2) {
        step(t, ga, gb);
        step(t + 1, gb, ga);
      }
      if (t < N) step(t, ga, gb);
    } else {
      if constexpr (I2LQR_GROUP_UNROLL >= 4) {
        for (; t + 3 < N; t += 4) {
          step(t, ga, ga);
          step(t + 1, ga, ga);
          step(t + 2, ga, ga);
          step(t + 3, ga, ga);
        }
      }
      for (; t + 1 < N; t += 2) {
        step(t, ga, ga);
        step(t + 1, ga, ga);
      }
      if (t < N) step(t, ga, ga);
    }
    Sys::template trig_g<GENERAL>(x, tr, bad);
#pragma unroll
    for (int q = 0; q < NT; q++) S[TRn + N * NT + q] = tr[q];
    const T cost = terminal_cost(x, xT);
    wave_sync();
    return cost;
  }
};

// e / d for the index arithmetic of the entry / exit copies (0 <= e < 2^22, 1 <= d <= 2^12): a
// float reciprocal and one correction instead of the ~30-instruction integer division by a
// run-time divisor (the exit copy of K alone does 30 of them per lane).
__device__ __forceinline__ int idx_div(int e, int d, float rcp_d) {
  int q = (int)((float)e * rcp_d);
  const int r = e - q * d;
  q += (r >= d) ? 1 : 0;
  q -= (r < 0) ? 1 : 0;
  return q;
}

// Grid: ceil(B / 8) workgroups of H wavefronts; dynamic LDS = GLayout::wave_words() * sizeof(T).
// H > 1: wavefronts 1..H-1 are helpers for the one phase of an iteration that is parallel over the
// horizon — the per-step records (prep): 21 records of eight problems are three rounds for one
// wavefront and one round for three.  They sleep at a workgroup barrier the rest of the time (a
// batch of 1024 problems leaves seven of eight SIMDs idle anyway).  Same values whoever computes
// a record: bit-identical to H = 1.
// WS: records and gains in the HBM workspace `ws` (GLayout::ws_words() words per problem, sized for
// whole wavefronts: ceil(B / 8) * 8 problems), four wavefronts per CU instead of two — the form
// for more than 4096 problems on the problem-major layout.  Same arithmetic, bit-identical.
template <class T, class Sys, int H = 1, bool WS = false, int G = kGroup>
__global__ __launch_bounds__(64 * H) void k_group_iterate(const DevCfg<T, Sys::n, Sys::m> c,
                                                          const IterArgs<T> a, T* ws = nullptr) {
  static_assert(!(WS && H > 1), "the workspace form runs without helper wavefronts");
  constexpr int n = Sys::n, m = Sys::m, W = n + m;
  using GL = GLayout<Sys, G>;
  extern __shared__ __align__(16) unsigned char gsmem_raw[];
  T* smem = reinterpret_cast<T*>(gsmem_raw);
  const int lane = threadIdx.x & 63, hv = threadIdx.x >> 6;
#ifdef I2LQR_STAMPS
  unsigned long long st_k0, st_k1, st_k2, st_k3;  // kernel start, loop start, loop end, kernel end
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_k0)::"memory");
#endif
  const int64_t prob0 = (int64_t)blockIdx.x * GL::PW + lane / G;
  // groups past the end of the batch work on a copy of the last problem and store nothing, so
  // that every lane of the wavefront runs the same control flow
  const bool real = prob0 < a.B;
  const int64_t prob = real ? prob0 : a.B - 1;
  GroupWorker<T, Sys, WS, G> w(c, smem, lane);
  const int N = c.N, g = w.g;
  const float rN = 1.0f / (float)N, rN1 = 1.0f / (float)(N + 1), rnN = 1.0f / (float)(n * N);
  const GL& L = w.L;
  const auto S = w.S;
  if constexpr (WS) w.Wp = ws + prob0 * (int64_t)L.ws_words();

  // entry: x0, U, x_term, lamb, obs (HBM, problem-major records) -> LDS / registers.  Only the
  // main wavefront stores: rollout() clips the inputs IN these LDS words right away, and a late
  // helper store of the caller's raw value would undo the clip (inputs outside [-u_max, u_max] are
  // legal: the reference clips them, control/iterative_ilqr.py:33-41).
  if (H == 1 || hv == 0) {
    const T* gX = a.X + prob * (int64_t)(n * (N + 1));
    if (g < n) S[L.XU0 + g] = gX[g * (N + 1)];
    const T* gU = a.U + prob * (int64_t)(m * N);
    for (int e = g; e < m * N; e += G) {
      const int aa = idx_div(e, N, rN), t = e - aa * N;
      S[L.XU0 + t * W + n + aa] = gU[e];
    }
    for (int e = lane; e < n * n; e += 64) smem[L.qt_base() + e] = c.Qt[e];
  }
  T xT[n], ob[6];
#pragma unroll
  for (int i = 0; i < n; i++) xT[i] = a.x_term[prob * n + i];
#pragma unroll
  for (int q = 0; q < 6; q++) ob[q] = a.obs ? a.obs[prob * 6 + q] : T(q == 5 ? -1 : 1);
  T lamb = a.lamb[prob];
  wave_sync();

  int cur = 0;
  const T ob_pa = T(1) / (ob[2] * ob[2]), ob_pb = T(1) / (ob[3] * ob[3]);
  int* const ctl = reinterpret_cast<int*>(smem + L.ctl_base());
  if constexpr (H > 1) {
    if (hv > 0) {  // helper: its share of the records whenever the main wavefront asks for them
      for (;;) {
        __syncthreads();  // B1: the control words of this iteration are written
        const int flags = ctl[8];
        if (!(flags & 1)) break;
        if (flags & 2) {
          const int pc = ctl[lane / G];
          w.prep(pc ? L.XU1 : L.XU0, pc ? L.TR1 : L.TR0, ob, ob_pa, ob_pb, hv * G + g,
                 H * G);
        }
        __syncthreads();  // B2: the records are complete
      }
      return;
    }
  }
  T cost = w.rollout(L.XU0, L.TR0, xT);
  int it = 0, status = a.early_exit ? 2 /*MAX_ITER*/ : 0 /*RUNNING*/;
  T cost_ret = cost;
  bool fresh = true, active = a.n_iters > 0;
  // the problems of a wavefront stop at different iterations (early exits): the loop runs while any
  // of them is active; finished ones keep computing on their (unchanged) state and commit nothing
#ifdef I2LQR_STAMPS
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_k1)::"memory");
#endif
  while (__any(active)) {
    const int XUo = cur ? L.XU1 : L.XU0, XUn = cur ? L.XU0 : L.XU1;
    const int TRo = cur ? L.TR1 : L.TR0, TRn = cur ? L.TR0 : L.TR1;
    // the per-step records depend on the nominal trajectory only: still valid after a rejected step
#ifdef I2LQR_STAMPS
    {
      auto& st_t0 = w.st_t0; auto& st_t1 = w.st_t1; auto& st_acc = w.st_acc;
      STAMP_BEGIN();
      if (__any(fresh)) w.prep(XUo, TRo, ob, ob_pa, ob_pb, g, G);
      STAMP_END(0);
    }
#else
    if constexpr (H > 1) {
      const bool do_prep = __any(fresh);
      if (g == 0) ctl[lane / G] = cur;
      if (lane == 0) ctl[8] = 1 | (do_prep ? 2 : 0);
      __syncthreads();  // B1
      if (do_prep) w.prep(XUo, TRo, ob, ob_pa, ob_pb, g, H * G);
      __syncthreads();  // B2
    } else {
      if (__any(fresh)) w.prep(XUo, TRo, ob, ob_pa, ob_pb, g, G);
    }
#endif
    // optimistic, branch-free passes first; the general forms only if a lane asked for them
    if (__builtin_expect(__any(w.template backward<false>(XUo, xT, lamb, active)), 0))
      w.template backward<true>(XUo, xT, lamb, active);
    T cost_new;
    {
#ifdef I2LQR_STAMPS
      auto& st_t0 = w.st_t0; auto& st_t1 = w.st_t1; auto& st_acc = w.st_acc;
      STAMP_BEGIN();
#endif
      bool big = false;
      cost_new = w.template forward<false>(XUo, XUn, TRn, xT, &big);
      if (__builtin_expect(__any(big), 0)) cost_new = w.template forward<true>(XUo, XUn, TRn, xT, &big);
#ifdef I2LQR_STAMPS
      STAMP_END(6);
#endif
    }
    {
      // accept / reject with the lamb schedule: control/iterative_ilqr.py:74-84 — as selects, not
      // branches: the problems of a wavefront decide differently, so both arms would run anyway,
      // and every taken branch costs a wavefront alone on its SIMD ~100 cycles (the two fp64
      // divisions are straight-line code; their results are dropped where they do not apply)
      const bool acc = active && cost_new < cost, rej = active && !(cost_new < cost);
      const T lamb_dn = lamb / c.lamb_factor, lamb_up = lamb * c.lamb_factor;
      const bool conv = t_abs((cost_new - cost) / cost) < c.eps;
      it += active ? 1 : 0;
      fresh = acc;
      cur ^= acc ? 1 : 0;
      lamb = acc ? lamb_dn : (rej ? lamb_up : lamb);
      cost_ret = acc ? cost_new : (rej ? cost : cost_ret);
      cost = acc ? cost_new : cost;
      const bool stop_conv = acc && conv, stop_lamb = rej && lamb > c.max_lamb;
      const int ended = stop_conv ? 1 : 3;
      if (a.early_exit) {
        status = (stop_conv || stop_lamb) ? ended : status;
        active = active && !(stop_conv || stop_lamb);
      } else {
        status = (status == 0 && (stop_conv || stop_lamb)) ? ended : status;
      }
      active = active && it < a.n_iters;
    }
  }
#ifdef I2LQR_STAMPS
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_k2)::"memory");
#endif
  if constexpr (H > 1) {
    if (lane == 0) ctl[8] = 0;  // release the helpers
    __syncthreads();            // B1 of an iteration that does not happen
  }
  if (!t_isfinite(cost_ret)) status = 4;

  // exit: X, U, gains, scalars (LDS -> HBM, problem-major records with time contiguous)
  if (real) {
    const int XUo = cur ? L.XU1 : L.XU0;
    T* gX = a.X + prob * (int64_t)(n * (N + 1));
    for (int e = g; e < n * (N + 1); e += G) {
      const int i = idx_div(e, N + 1, rN1), t = e - i * (N + 1);
      gX[e] = S[XUo + t * W + i];
    }
    T* gU = a.U + prob * (int64_t)(m * N);
    for (int e = g; e < m * N; e += G) {
      const int aa = idx_div(e, N, rN), t = e - aa * N;
      gU[e] = S[XUo + t * W + n + aa];
    }
    if (a.K) {
      T* gK = a.K + prob * (int64_t)(m * n * N);
      for (int e = g; e < m * n * N; e += G) {
        const int aa = idx_div(e, n * N, rnN), r = e - aa * (n * N);
        const int j = idx_div(r, N, rN), t = r - j * N;
        gK[e] = w.gain(t)[aa * GL::KW + j];
      }
      T* gk = a.k + prob * (int64_t)(m * N);
      for (int e = g; e < m * N; e += G) {
        const int aa = idx_div(e, N, rN), t = e - aa * N;
        gk[e] = w.gain(t)[aa * GL::KW + n];
      }
    }
    if (g == 0) {
      a.lamb[prob] = lamb;
      a.cost[prob] = cost_ret;
      if (a.iters) a.iters[prob] = it;
      if (a.status) a.status[prob] = status;
    }
#ifdef I2LQR_STAMPS
    // slots 0 (records), 1 (backward), 6 (forward): sums over the iterations; 2: entry (loads +
    // nominal rollout), 3: the whole iteration loop, 4: exit stores
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_k3)::"memory");
    w.st_acc[2] = st_k1 - st_k0;
    w.st_acc[3] = st_k2 - st_k1;
    w.st_acc[4] = st_k3 - st_k2;
    if (a.dbg && g == 0)
      for (int q = 0; q < 8; q++) a.dbg[prob * 8 + q] = w.st_acc[q];
#endif
  }
  // epilogue (i2lqr_iterate_pick): relaxed terminal cost of utils/base.py:427-437 from the x_N
  // still in LDS — the words just stored to X, so i2lqr_relax_cost on the returned X gives the
  // same bits — and the flat pick of :462-465 over the launch
  if (a.qfun) {
    const bool cand = real && g == 0;
    T ci = T(0);
    if (cand) {
      const int XUo = cur ? L.XU1 : L.XU0;
      double ss = 0.0;
#pragma unroll
      for (int i = 0; i < n; i++) {
        const double d = (double)S[XUo + N * W + i] - (double)xT[i];
        ss += d * d;
      }
      ci = (T)relax_cost_value(ss, a.qfun[prob], N, a.outer_iter, a.max_relax_iter);
      a.cost_it[prob] = ci;
    }
    if (a.pick_part) pick_epilogue(a, cand, prob, ci);
  }
}

// ---------------------------------------------------------------------------------------------
// Speculative form: solves to termination of small batches (one workgroup of V wavefronts per
// eight problems and per CU) and the tail of the chunked solves (SETIO).
//
// An iLQR iteration ends in accept (new nominal, lamb / 10) or reject (same nominal, lamb * 10):
// control/iterative_ilqr.py:74-84.  After a reject the next iteration runs the backward and forward
// pass on the SAME nominal trajectory with lamb * 10 — work that does not depend on the outcome of
// the current iteration except through "was it a reject".  With the eight-lane kernel a batch of
// 1024 problems occupies an eighth of the chip's SIMDs; the speculative kernel puts V wavefronts
// on each group of eight problems, wavefront v running the iteration that follows v rejects
// (lamb * 10^v).  After every round the chain of outcomes is resolved in order: the iterations up
// to and including the first accept (or all V if none accepts) count, the rest is discarded.  The
// iterations a problem executes, their order and their arithmetic are exactly those of the
// sequential kernel — results are bit-identical — but a run of r rejects followed by one accept costs one
// round instead of r + 1.  Rejects are frequent (the lamb schedule probes until a step is
// accepted, converged problems reject until lamb overflows, and the stragglers that decide how
// long a solve lasts alternate accept / reject): ~1.8 iterations per round at V = 3 on the
// benchmark workload, 1.7 at V = 2.
//
// LDS per problem: V + 1 trajectory buffers (nominal + one candidate per wavefront; a per-problem
// table says which is which, an accepted candidate becomes the nominal by swapping two table
// entries), the records (shared, written by all wavefronts' lanes together), and per wavefront
// its gains and exchange buffer.  Workgroup barriers: after the record phase and after the
// forward pass.
// ---------------------------------------------------------------------------------------------
template <class Sys, int V, int G = kGroup> struct GSpecLayout {
  static constexpr int n = Sys::n, m = Sys::m, W = n + m, NT = Sys::NTRIG;
  static constexpr int PW = 64 / G;  // problems per workgroup
  using GL = GLayout<Sys, G>;
  int N;
  int traj_words, R, var0, var_words, Kk_in_var, T1c_in_var, total;
  __host__ __device__ explicit GSpecLayout(int N_) : N(N_) {
    const int xu = (W * (N + 1) + 3) & ~3, tr = (NT * (N + 1) + 3) & ~3;
    traj_words = xu + tr;  // buffer b: XU at b * traj_words, TR right behind it (+ xu)
    int o = (V + 1) * traj_words;
    R = o; o += GL::RW * (N + 1); o = (o + 3) & ~3;
    var0 = o;
    Kk_in_var = 0;
    T1c_in_var = (m * GL::KW * N + 3) & ~3;
    // eight lanes: the T1 exchange buffer [row pair][column][2] behind the gains; sixteen lanes
    // (DPP broadcasts, no exchange buffer): one gain row of slack, which the look-ahead load of
    // the forward pass's last step reads (GroupWorker::forward_row)
    var_words = G == kGroup ? (T1c_in_var + kGroup * 2 * ((W + 1) / 2) + 3) & ~3
                            : (T1c_in_var + m * GL::KW + 3) & ~3;
    o += V * var_words;
    if (((o / 4) & 1) == 0) o += 4;
    total = o;
  }
  __host__ __device__ int xu_off(int b) const { return b * traj_words; }
  __host__ __device__ int tr_off(int b) const { return b * traj_words + ((W * (N + 1) + 3) & ~3); }
  // + Q_terminal + the V x 8 candidate costs of a round + the rollout cost
  __host__ __device__ int group_words() const { return PW * total + n * n + (V + 1) * PW; }
};

// SETIO: the problems are the first *a.count columns of a batch-minor, time-major work set (the
// tail of the chunked solve of the one-problem-per-lane layouts, see IterArgs and k_iterate): the
// launch does nothing unless *count <= count_max, the iteration counters continue from iters[] and
// stop at max_total.
// CHAIN (round 6; the controller's chained regularisation, utils/base.py:393, :414-426): a.B CHAINS
// of a.chain_len problems each, stored chain after chain; a workgroup solves the problems of its
// chains one after the other, the final lamb of problem c being the initial lamb of problem c + 1
// (lamb[] is read for the first problem of a chain only and written for all of them) — the 8
// dependent solves of a lap's candidate list in ONE launch instead of eight.
template <class T, class Sys, int V, bool SETIO = false, int G = kGroup, bool CHAIN = false>
__global__ __launch_bounds__(64 * V) void k_group_spec(const DevCfg<T, Sys::n, Sys::m> c,
                                                       const IterArgs<T> a) {
  static_assert(!(CHAIN && SETIO), "chains are problem-major batches");
  constexpr int n = Sys::n, m = Sys::m, W = n + m;
  using GL = GLayout<Sys, G>;
  using SL = GSpecLayout<Sys, V, G>;
  constexpr int PW = SL::PW;
  extern __shared__ __align__(16) unsigned char gsmem_raw[];
  T* smem = reinterpret_cast<T*>(gsmem_raw);
  const int v = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int p = lane / G, g = lane % G;
  const int64_t prob0 = (int64_t)blockIdx.x * PW + p;  // (CHAIN: the chain's index)
  int64_t live = a.B;
  if constexpr (SETIO) {
    live = *a.count;
    if (live > a.count_max || (int64_t)blockIdx.x * PW >= live) return;  // block-uniform
  }
  const int64_t Bs = SETIO ? a.set_stride : 0;
  const bool real = prob0 < live;
  const int clen = CHAIN ? a.chain_len : 1;
  T lamb_carry = T(0);
 for (int cstep = 0; cstep < clen; cstep++) {  // (one pass unless CHAIN)
  const int64_t prob = CHAIN ? (real ? prob0 : live - 1) * clen + cstep : (real ? prob0 : live - 1);
  const SL SLay(c.N);
  const int N = c.N;
  T* const S = smem + p * SLay.total;
  T* const QtL = smem + PW * SLay.total;
  T* const CN = QtL + n * n;  // [V + 1][8]: candidate cost of wavefront v / rollout cost at row V
  GroupWorker<T, Sys, false, G> w(c, S, QtL, g, SLay.total);
  w.oR = SLay.R;
  w.oKk = SLay.var0 + v * SLay.var_words + SLay.Kk_in_var;
  if constexpr (G == kGroup) w.T1c = S + SLay.var0 + v * SLay.var_words + SLay.T1c_in_var;

  // entry (wavefront 0): x0, U into buffer 0, Q_terminal; nominal rollout
  if (v == 0) {
    if constexpr (SETIO) {  // work-set rows: x_t[i] is row t n + i of X, u_t[a] row t m + a of U
      if (g < n) S[SLay.xu_off(0) + g] = a.X[g * Bs + prob];
      for (int e = g; e < m * N; e += G) {
        const int t = e / m, aa = e - t * m;
        S[SLay.xu_off(0) + t * W + n + aa] = a.U[e * Bs + prob];
      }
    } else {
      const T* gX = a.X + prob * (int64_t)(n * (N + 1));
      if (g < n) S[SLay.xu_off(0) + g] = gX[g * (N + 1)];
      const T* gU = a.U + prob * (int64_t)(m * N);
      for (int e = g; e < m * N; e += G) {
        const int aa = e / N, t = e - aa * N;
        S[SLay.xu_off(0) + t * W + n + aa] = gU[e];
      }
    }
    for (int e = lane; e < n * n; e += 64) QtL[e] = c.Qt[e];
  }
  T xT[n], ob[6];
#pragma unroll
  for (int i = 0; i < n; i++) xT[i] = SETIO ? a.x_term[i * Bs + prob] : a.x_term[prob * n + i];
#pragma unroll
  for (int q = 0; q < 6; q++)
    ob[q] = a.obs ? (SETIO ? a.obs[q * Bs + prob] : a.obs[prob * 6 + q]) : T(q == 5 ? -1 : 1);
  T lamb = (CHAIN && cstep > 0) ? lamb_carry : a.lamb[prob];
  const int it0 = SETIO ? a.iters[prob] : 0;              // iterations of the earlier chunks
  const int it_cap = SETIO ? a.max_total - it0 : a.n_iters;
  const T ob_pa = T(1) / (ob[2] * ob[2]), ob_pb = T(1) / (ob[3] * ob[3]);
  __syncthreads();
  if (v == 0) {
    const T c0 = w.rollout(SLay.xu_off(0), SLay.tr_off(0), xT);
    if (g == 0) CN[V * PW + p] = c0;
  }
  __syncthreads();
  T cost = CN[V * PW + p];

  // which buffer is the nominal, which is wavefront k's candidate (identical in every wavefront)
  int nb = 0, cb[V];
#pragma unroll
  for (int k = 0; k < V; k++) cb[k] = k + 1;
  int it = 0, status = a.early_exit ? 2 /*MAX_ITER*/ : 0 /*RUNNING*/, gsel = 0;
  T cost_ret = cost;
  bool fresh = true, active = it_cap > 0;
  while (__any(active)) {
    if (__any(fresh)) {
      w.prep(SLay.xu_off(nb), SLay.tr_off(nb), ob, ob_pa, ob_pb, v * G + g, V * G);
      __syncthreads();
    }
    // this wavefront's iteration: the one that follows v rejects
    T lamb_v = lamb;
    for (int k = 0; k < v; k++) lamb_v *= c.lamb_factor;
    int mycb = cb[0];
#pragma unroll
    for (int k = 1; k < V; k++) mycb = (v == k) ? cb[k] : mycb;
    const int XUo = SLay.xu_off(nb), XUn = SLay.xu_off(mycb), TRn = SLay.tr_off(mycb);
    if (__builtin_expect(__any(w.template backward<false>(XUo, xT, lamb_v, active)), 0))
      w.template backward<true>(XUo, xT, lamb_v, active);
    bool big = false;
    T cost_new = w.template forward<false>(XUo, XUn, TRn, xT, &big);
    if (__builtin_expect(__any(big), 0)) cost_new = w.template forward<true>(XUo, XUn, TRn, xT, &big);
    if (g == 0) CN[v * PW + p] = cost_new;
    __syncthreads();
    // resolve the chain: control/iterative_ilqr.py:74-84 for iteration it, it + 1, ...
    fresh = false;
    bool chain = active;
#pragma unroll
    for (int k = 0; k < V; k++) {
      if (chain) {
        const T cn = CN[k * PW + p];
        it++;
        gsel = k;
        if (cn < cost) {
          const int tb = nb; nb = cb[k]; cb[k] = tb;  // the candidate becomes the nominal
          lamb /= c.lamb_factor;
          const bool conv = t_abs((cn - cost) / cost) < c.eps;
          cost_ret = cn;
          cost = cn;
          fresh = true;
          chain = false;  // the later wavefronts worked on the old nominal
          if (conv) {
            if (a.early_exit) { status = 1; active = false; }
            if (status == 0) status = 1;
          }
        } else {
          lamb *= c.lamb_factor;
          cost_ret = cost;
          if (lamb > c.max_lamb) {
            if (a.early_exit) { status = 3; active = false; chain = false; }
            if (status == 0) status = 3;
          }
        }
        if (it >= it_cap) { active = false; chain = false; }
      }
    }
    __syncthreads();  // every wavefront has read the costs before the next round overwrites them
  }
  if (!t_isfinite(cost_ret)) status = 4;

  // exit (wavefront 0): X, U from the nominal buffer, the gains of the last executed iteration
  if (real && v == 0 && SETIO) {
    // to the work set (column prob, row stride Bs), or straight to the caller's arrays (a.orig)
    const bool deliver = a.orig != nullptr;
    const int64_t o = deliver ? (int64_t)a.orig[prob] : prob;
    const int64_t ot = (o >> 6) * 64, ol = o & 63;
    auto at = [&](int rows, int64_t row) -> int64_t {
      if (!deliver) return row * Bs + prob;
      return a.out_tiled ? (ot * rows + row * 64 + ol) : (row * a.out_B + o);
    };
    T* const dX = deliver ? a.out_X : a.X;
    T* const dU = deliver ? a.out_U : a.U;
    T* const dK = deliver ? a.out_K : a.K;
    T* const dk = deliver ? a.out_k : a.k;
    const int XUo = SLay.xu_off(nb);
    for (int e = g; e < n * (N + 1); e += G) {
      const int t = e / n, i = e - t * n;
      dX[at(n * (N + 1), e)] = S[XUo + t * W + i];
    }
    for (int e = g; e < m * N; e += G) {
      const int t = e / m, aa = e - t * m;
      dU[at(m * N, e)] = S[XUo + t * W + n + aa];
    }
    if (dK) {  // K rows (t m + a) n + j, k rows t m + a
      const int oK = SLay.var0 + gsel * SLay.var_words + SLay.Kk_in_var;
      for (int e = g; e < m * N * (n + 1); e += G) {
        const int r = e / (n + 1), j = e - r * (n + 1);
        const T val = S[oK + r * GL::KW + j];
        if (j < n) dK[at(m * n * N, (int64_t)r * n + j)] = val;
        else dk[at(m * N, r)] = val;
      }
    }
    if (g == 0) {
      if (deliver) {
        a.out_lamb[o] = lamb;
        a.out_cost[o] = cost_ret;
        if (a.out_iters) a.out_iters[o] = it0 + it;
        if (a.out_status) a.out_status[o] = status;
        a.status[prob] = status | kStatusDelivered;
      } else {
        a.lamb[prob] = lamb;
        a.cost[prob] = cost_ret;
        if (a.iters) a.iters[prob] = it0 + it;
        if (a.status) a.status[prob] = status;
      }
    }
  }
  if (real && v == 0 && !SETIO) {
    const int XUo = SLay.xu_off(nb);
    T* gX = a.X + prob * (int64_t)(n * (N + 1));
    for (int e = g; e < n * (N + 1); e += G) {
      const int i = e / (N + 1), t = e - i * (N + 1);
      gX[e] = S[XUo + t * W + i];
    }
    T* gU = a.U + prob * (int64_t)(m * N);
    for (int e = g; e < m * N; e += G) {
      const int aa = e / N, t = e - aa * N;
      gU[e] = S[XUo + t * W + n + aa];
    }
    if (a.K) {
      const int oK = SLay.var0 + gsel * SLay.var_words + SLay.Kk_in_var;
      T* gK = a.K + prob * (int64_t)(m * n * N);
      for (int e = g; e < m * n * N; e += G) {
        const int aa = e / (n * N), r = e - aa * (n * N), j = r / N, t = r - j * N;
        gK[e] = S[oK + (t * m + aa) * GL::KW + j];
      }
      T* gk = a.k + prob * (int64_t)(m * N);
      for (int e = g; e < m * N; e += G) {
        const int aa = e / N, t = e - aa * N;
        gk[e] = S[oK + (t * m + aa) * GL::KW + n];
      }
    }
    if (g == 0) {
      a.lamb[prob] = lamb;
      a.cost[prob] = cost_ret;
      if (a.iters) a.iters[prob] = it;
      if (a.status) a.status[prob] = status;
    }
  }
  lamb_carry = lamb;  // CHAIN: the next problem of the chain starts from it
 }
}

}  // namespace i2lqr

// Sixteen-lanes-per-problem iLQR kernel for gfx950 (MI355X): the throughput path of the
// quadrotor-sized plant (n = 12, m = 4, N = 50: BASELINE.json configs[4]).
//
// The column-per-lane decomposition of i2lqr_group.hpp with n + m = 16 lanes per problem and four
// problems per wavefront:
//   lane j (0..n)   holds column j of [Vxx | Vx] and forms T1[:, j] = F^T Va[:, j] from the
//                   compile-time sparsity of F = [A | B] (Sys::pat),
//   lane b (0..n-1) forms column b of H = L + T1[:, :n] F: its own T1 column (still in registers)
//                   times F[b][b] plus a sum over the <= NZ other rows in which column b of F is
//                   non-zero: per-lane (source column, coefficient) lists, uniform code; lane n
//                   runs the same code with coefficients (1; 0, ..) and forms g = l + T1[:, n],
//   Quu             every lane, from rows n.. of the exchanged T1 columns that the B columns of
//                   F touch (compile-time pattern), LDL^T-factored in every lane,
//   [K | k][:, j]   = -(Quu + lamb I)^-1 H[n:, j] by a lane-local solve; Va'[:, j] needs the other
//                   lanes' K columns.
// What does not fit next to it in LDS lives in a caller-provided HBM workspace
// (i2lqr_workspace_bytes): the per-step records (Jacobian entries, barrier terms: written by the
// record phase, read one step ahead by the backward pass), the gains (written by the backward
// pass, read one step ahead by the forward pass) and the candidate trajectory with its sin / cos
// values (written by the forward pass; copied into LDS if the step is accepted).  LDS keeps the
// nominal trajectory, the candidate inputs and one exchange buffer: 10 KB per problem, 40 KB per
// wavefront — four wavefronts per CU, one per SIMD (the one-problem-per-wavefront kernel needs
// 48 KB per problem: three wavefronts per CU, each lane one element of the products).
//
// Q = R = 0 (the defaults); other weights take the one-problem-per-wavefront kernel.
// Reference being replaced: control/iterative_ilqr.py:7-160, control/ilqr_helper.py:9-150 in the
// build-defined quad12 plant (i2lqr_systems.hpp); same algorithm as the other kernels, results
// agree to round-off.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "i2lqr_systems.hpp"
#include "i2lqr_wave.hpp"

namespace i2lqr {

constexpr int kQG = 16;              // lanes per problem (= one DPP row)
constexpr int kQPW = 64 / kQG;       // problems per wavefront

// Exchange buffer of the T1 columns, [row pair][column][2] words: the sixteen columns of a row pair
// fill one 256-byte LDS bank row, 16 bytes each, so lanes that fetch DIFFERENT columns with
// ds_read_b128 hit different banks (column-major, 128-byte columns, put every column on one of two
// bank groups: 65 % of the LDS cycles of the kernel were bank conflicts).  `colw` = 2 * (column
// rotated by the problem's bank phase, see QuadWorker).
__device__ __forceinline__ int ex_word(int colw, int row) {
  return (row >> 1) * (2 * kQG) + colw + (row & 1);
}

// value of lane `L` of each 16-lane row, in every lane of that row
template <int L> __device__ __forceinline__ double row_bcast(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x150 + L, 0xf, 0xf, false);  // row_newbcast:L
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x150 + L, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <int L> __device__ __forceinline__ float row_bcast(float v) {
  int w = __float_as_int(v);
  w = __builtin_amdgcn_update_dpp(w, w, 0x150 + L, 0xf, 0xf, false);
  return __int_as_float(w);
}

// Compile-time column lists of F = [A | B]
template <class Sys> struct QPattern {
  static constexpr int n = Sys::n, m = Sys::m, W = n + m, NV = Sys::NVAR;
  // The diagonal entry F[b][b] of a state column multiplies the lane's OWN T1 column, which is
  // still in its registers: it is kept out of the lists of columns fetched through LDS.
  static constexpr bool has_own(int b) { return b < n && Sys::pat(b, b) != 0; }
  static constexpr int own_code(int b) { return has_own(b) ? Sys::pat(b, b) : 0; }
  static constexpr bool listed(int i, int b) { return Sys::pat(i, b) != 0 && !(has_own(b) && i == b); }
  static constexpr int nnz(int b) {
    int c = 0;
    for (int i = 0; i < n; i++) c += listed(i, b);
    return c;
  }
  static constexpr int NZ = [] {  // over the state columns: the lanes that form H columns
    int mx = 0;
    for (int b = 0; b < n; b++) mx = nnz(b) > mx ? nnz(b) : mx;
    return mx;
  }();
  // does some B column of F touch row i?  (the T1 columns Quu is formed from)
  static constexpr bool in_b(int i) {
    for (int b = n; b < W; b++)
      if (Sys::pat(i, b) != 0) return true;
    return false;
  }
  static constexpr int src(int b, int s) {  // s-th listed row of column b (0 past the end)
    int c = 0;
    for (int i = 0; i < n; i++)
      if (listed(i, b)) {
        if (c == s) return i;
        c++;
      }
    return 0;
  }
  static constexpr int code(int b, int s) { return s < nnz(b) ? Sys::pat(src(b, s), b) : 0; }
};

// Layouts in words of T.
template <class Sys> struct QLayout {
  static constexpr int n = Sys::n, m = Sys::m, W = n + m, NV = Sys::NVAR, NT = Sys::NTRIG,
                       NC = Sys::NCONST;
  // record (HBM): jv[NV], 0, 1, dt, plant constants, lu[m], luu[m], ob[5]
  static constexpr int R_ZERO = NV, R_ONE = NV + 1, R_DT = NV + 2, R_PC = NV + 3,
                       R_LU = NV + 3 + NC, R_LUU = R_LU + m, R_OB = R_LUU + m,
                       RW = (R_OB + 5 + 1) & ~1;
  static constexpr int GW = 16;            // gain row: K[a][0..n-1], k[a] at column n, padding
  static constexpr int XCW = n + NT;       // candidate record: x[n], sin / cos values at x
  static constexpr int rec_off(int code) {
    return code == 0 ? R_ZERO : code == 1 ? R_ONE : code == 2 ? R_DT
         : code >= 100 ? R_PC + (code - 100) : code - 3;
  }
  int N;
  // LDS per problem
  int XU, UC, EX, lds_total;
  // HBM workspace per problem
  int REC, GK, XC0, XC1, ws_total;
  __host__ __device__ explicit QLayout(int N_) : N(N_) {
    int o = 0;
    XU = o; o += W * (N + 1);
    UC = o; o += m * N;
    o = (o + 1) & ~1;
    EX = o; o += kQG * W;
    lds_total = (o + 1) & ~1;
    int w = 0;
    REC = w; w += RW * (N + 1);
    GK = w; w += m * GW * N;
    XC0 = w; w += XCW * (N + 1);
    XC1 = w; w += XCW * (N + 1);
    ws_total = (w + 15) & ~15;
  }
};

template <class T, class Sys> struct QuadWorker {
  static constexpr int n = Sys::n, m = Sys::m, W = n + m, NV = Sys::NVAR, NT = Sys::NTRIG,
                       NC = Sys::NCONST, NA = n + 1;
  static_assert(W == kQG, "the sixteen-lane kernel needs n + m == 16");
  using Cfg = DevCfg<T, n, m>;
  using QL = QLayout<Sys>;
  using QP = QPattern<Sys>;
  static constexpr int NZ = QP::NZ;
  const Cfg& c;
  const QL L;
  const Slice<T> S;   // this problem's LDS slice
  const Slice<T> Wp;  // this problem's HBM workspace
  const int g;   // lane inside the group = column index
  const int N;
  T pc[NC];      // plant constants
  // per-lane column description
  int off_c[NZ];  // record offsets of the coefficients of column g's listed rows
  int srcw[NZ];   // exchange-buffer column words (ex_word) of the T1 columns they multiply
  int off_own;    // record offset of the coefficient of the lane's own T1 column
  int ownw;       // exchange-buffer column word of the lane's own T1 column
  int off_l0, off_l1;  // record offsets of the cost terms of rows 0, 1 (lanes 0, 1: obstacle block;
                       // lane n: obstacle gradient)
  int off_lu[m];       // ... of rows n..n+m-1 (lane n: input-barrier gradient)
  int rotw;            // 2 * column rotation of this problem's exchange buffer
#ifdef I2LQR_STAMPS
  mutable unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t0 = 0, st_t1 = 0;
#endif

  __device__ QuadWorker(const Cfg& c_, T* smem, T* ws, int lane, int64_t prob)
      : c(c_), L(c_.N),
        S(make_slice(smem + (lane / kQG) * QLayout<Sys>(c_.N).lds_total,
                     QLayout<Sys>(c_.N).lds_total, c_.trap, TAG_QUAD_LDS)),
        Wp(make_slice(ws + prob * (int64_t)QLayout<Sys>(c_.N).ws_total,
                      QLayout<Sys>(c_.N).ws_total, c_.trap, TAG_QUAD_WS)),
        g(lane % kQG), N(c_.N) {
#pragma unroll
    for (int q = 0; q < NC; q++) pc[q] = Sys::plant_const(c, q);
    // Bank phase: a ds_read_b128 is served in lane groups that hold eight lanes of one problem
    // and eight of its neighbour; the columns are rotated so that problem p's column c sits in
    // 16-byte slot (c + 12 p) mod 16 of the bank row whatever the slice size is (the rotation that
    // leaves the fewest shared slots for this plant's column lists).
    const int slot0 = (int)(((unsigned)(uintptr_t)(T*)(S + L.EX) >> 4) & 15u);
    const int rot = (12 * (lane / kQG) - slot0) & 15;
#pragma unroll
    for (int s = 0; s < NZ; s++) { off_c[s] = QL::R_ZERO; srcw[s] = 2 * (rot & 15); }
    off_l0 = QL::R_ZERO;
    off_l1 = QL::R_ZERO;
#pragma unroll
    for (int a = 0; a < m; a++) off_lu[a] = QL::R_ZERO;
    off_own = QL::R_ZERO;
    ownw = 2 * ((g + rot) & 15);
    rotw = 2 * rot;
    if (g == n) {  // lane n: g = l + T1[:, n]
      off_own = QL::R_ONE;
      off_l0 = QL::R_OB + 0;
      off_l1 = QL::R_OB + 1;
#pragma unroll
      for (int a = 0; a < m; a++) off_lu[a] = QL::R_LU + a;
    }
    static_for_i<0, n>([&](auto b_) {
      constexpr int b = decltype(b_)::value;
      if (g == b) {
        off_own = QL::rec_off(QP::own_code(b));
        static_for_i<0, NZ>([&](auto s_) {
          constexpr int s = decltype(s_)::value;
          off_c[s] = QL::rec_off(QP::code(b, s));
          srcw[s] = 2 * ((QP::src(b, s) + rot) & 15);
        });
        if (b < 2) {  // obstacle block l_xx[a][b], a, b < 2: ob[2 + a + b]
          off_l0 = QL::R_OB + 2 + b;
          off_l1 = QL::R_OB + 3 + b;
        }
      }
    });
  }

  // Q_terminal: n^2 uniform words of the kernel argument (there is no LDS left to stage them in).
  // (Hiding the pointer from the compiler so that the loads stay at their uses, and pacing the LDS
  // reads of the H columns with scheduling barriers, both measured slower: 23.7 / 23.0 ms against
  // 21.3 ms per 4-iteration launch at 65536 problems.)
  __device__ __forceinline__ const T* qt_ptr() const { return c.Qt; }
  __device__ __forceinline__ T terminal_cost(const T (&x)[n], const T (&xT)[n]) const {
    const T* Qt = qt_ptr();
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

  // acc += F[i][a] * v from the compile-time pattern
  template <int i, int a> __device__ __forceinline__ void f_acc(T& acc, bool& first, T v,
                                                                const T (&jv)[NV]) const {
    constexpr int code = Sys::pat(i, a);
    if constexpr (code == 0) {
      return;
    } else if constexpr (code == 1) {
      acc = first ? v : acc + v;
      first = false;
    } else {
      const T f = code == 2 ? c.dt : (code >= 100 ? pc[code >= 100 ? code - 100 : 0]
                                                  : jv[code >= 3 && code < 100 ? code - 3 : 0]);
      acc = first ? f * v : t_fma(f, v, acc);
      first = false;
    }
  }

  // -- rollout of the inputs in `Us` (LDS, stride us) from the x_0 in XU: states and clipped
  //    inputs into XU, candidate records {x_t, trig(x_t)} into the HBM buffer XCo.  Used for the
  //    nominal rollout at entry (control/iterative_ilqr.py:32-48).
  __device__ __forceinline__ T rollout(int XCo, const T (&xT)[n]) const {
    T x[n], u[m], xn[n], tr[NT];
#pragma unroll
    for (int i = 0; i < n; i++) x[i] = S[L.XU + i];
    T* XC = Wp + XCo;
    // the three angles on three lanes, as in the forward pass (same values as Sys::trig)
    auto bc = [](auto lane_, T v) { return row_bcast<decltype(lane_)::value>(v); };
    bool unused = false;
    for (int t = 0; t < N; t++) {
#pragma unroll
      for (int a = 0; a < m; a++) u[a] = clip(S[L.XU + t * W + n + a], -c.u_max[a], c.u_max[a]);
#pragma unroll
      for (int a = 0; a < m; a++) S[L.XU + t * W + n + a] = u[a];
      Sys::template trig_row<true>(x, tr, &unused, g, bc);
      if (g == 0) {
#pragma unroll
        for (int q = 0; q < NT; q++) XC[t * QL::XCW + n + q] = tr[q];
      }
      Sys::step_tr(c, x, u, tr, xn);
#pragma unroll
      for (int i = 0; i < n; i++) S[L.XU + (t + 1) * W + i] = xn[i];
#pragma unroll
      for (int i = 0; i < n; i++) x[i] = xn[i];
    }
    Sys::template trig_row<true>(x, tr, &unused, g, bc);
    if (g == 0) {
#pragma unroll
      for (int q = 0; q < NT; q++) XC[N * QL::XCW + n + q] = tr[q];
    }
    const T cost = terminal_cost(x, xT);
    wave_sync();
    return cost;
  }

  // -- per-step records (HBM), the lanes of the group take the steps g, g + 16, ... in turn; no
  //    divergent control flow (see GroupWorker::prep).  TRo: the HBM buffer that holds the sin /
  //    cos values of the nominal trajectory.
  __device__ __forceinline__ void prep(int TRo, const T (&ob)[6], T pa, T pb) const {
    const bool has_ob = ob[5] >= T(0);
    const int opt = has_ob ? (int)ob[5] : 0;
    const T spd_y = opt == 1 ? ob[4] : T(0), spd_x = opt == 2 ? ob[4] : T(0);
    const int rounds = (N + kQG) / kQG;  // ceil((N + 1) / 16)
    const T* XC = Wp + TRo;
    for (int r = 0; r < rounds; r++) {
      const int t0 = g + r * kQG;
      const int t = t0 < N ? t0 : N;        // record index (obstacle term of x_t)
      const int ts = t0 < N ? t0 : N - 1;   // step index (Jacobian entries, input barrier)
      T* Rs = Wp + L.REC + ts * QL::RW;
      {
        T xe[n], tr[NT], u[m], jv[NV];
#pragma unroll
        for (int i = 0; i < n; i++) xe[i] = S[L.XU + (ts + 1) * W + i];
#pragma unroll
        for (int a = 0; a < m; a++) u[a] = S[L.XU + ts * W + n + a];
#pragma unroll
        for (int q = 0; q < NT; q++) tr[q] = XC[(ts + 1) * QL::XCW + n + q];
        Sys::jac_var(c, xe, u, tr, jv);  // at (x_{t+1}, u_t): control/iterative_ilqr.py:92-99
#pragma unroll
        for (int q = 0; q < NV; q++) Rs[q] = jv[q];
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
          Rs[QL::R_LU + a] = c.ctrl_q12 * e_hi - c.ctrl_q12 * e_lo;
          Rs[QL::R_LUU + a] = c.ctrl_q122 * e_hi +
                              c.ctrl_q122 * e_lo;
        }
      }
      T* R = Wp + L.REC + t * QL::RW;
      R[QL::R_ZERO] = T(0);
      R[QL::R_ONE] = T(1);
      R[QL::R_DT] = c.dt;
#pragma unroll
      for (int q = 0; q < NC; q++) R[QL::R_PC + q] = pc[q];
      // obstacle barrier: control/ilqr_helper.py:32-51 (stage) / :121-147 (terminal, index N)
      const T px = S[L.XU + t * W + 0], py = S[L.XU + t * W + 1];
      const T dz = opt == 2 ? px - (ob[0] - T(t) * spd_x) : px - ob[0];
      const T dy = opt == 1 ? py - (ob[1] + T(t) * spd_y) : py - ob[1];
      const T h = T(1) + c.safety_margin - (dz * pa * dz + dy * pb * dy);
      const T hd0 = T(-2) * pa * dz, hd1 = T(-2) * pb * dy;
      const T e = t_exp(c.obs_q2 * h);
      const T c1 = c.obs_q12 * e, c2 = c.obs_q122 * e;
      R[QL::R_OB + 0] = has_ob ? c1 * hd0 : T(0);
      R[QL::R_OB + 1] = has_ob ? c1 * hd1 : T(0);
      R[QL::R_OB + 2] = has_ob ? c2 * (hd0 * hd0) : T(0);
      R[QL::R_OB + 3] = has_ob ? c2 * (hd0 * hd1) : T(0);
      R[QL::R_OB + 4] = has_ob ? c2 * (hd1 * hd1) : T(0);
    }
    // the records are read back by other lanes of this wavefront only: make the stores visible
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0) expcnt(0) lgkmcnt(0)
    wave_sync();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }

  // -- backward pass: control/iterative_ilqr.py:88-130.  Leaves the gains in the HBM workspace.
  //    GENERAL: see GroupWorker::backward.
  //    commit: the problem is still running — a problem that has terminated keeps computing beside
  //    its wavefront neighbours but must not overwrite the gains of its last executed iteration.
  template <bool GENERAL>
  __device__ __forceinline__ bool backward(const T (&xT)[n], T lamb, bool commit) const {
    bool bad = false;
    // terminal value function, get_cost_final(): control/ilqr_helper.py:106-150
    T va[n];
    {
      const T* Rn = Wp + L.REC + N * QL::RW;
      const T* Qt = qt_ptr();
      T dx[n];
#pragma unroll
      for (int i = 0; i < n; i++) dx[i] = S[L.XU + N * W + i] - xT[i];
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
      va[0] += Rn[off_l0];
      va[1] += Rn[off_l1];
    }
    T* const EX = S + L.EX;
    // the record of a step is loaded one step ahead (HBM / L2 latency under the previous step)
    T jv[NV], luu[m], cf[NZ], cf_own, l0, l1, lrow[m];
    auto load_record = [&](int t) __attribute__((always_inline)) {
      const T* R = Wp + L.REC + t * QL::RW;
#pragma unroll
      for (int q = 0; q < NV; q++) jv[q] = R[q];
#pragma unroll
      for (int a = 0; a < m; a++) luu[a] = R[QL::R_LUU + a];
#pragma unroll
      for (int s = 0; s < NZ; s++) cf[s] = R[off_c[s]];
      cf_own = R[off_own];
      l0 = R[off_l0];
      l1 = R[off_l1];
#pragma unroll
      for (int a = 0; a < m; a++) lrow[a] = R[off_lu[a]];
    };
    load_record(N - 1);
    auto step = [&](const int t) __attribute__((always_inline)) {
      STAMP_BEGIN();
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
#pragma unroll
      for (int a = 0; a < W; a++) EX[ex_word(ownw, a)] = t1[a];
      wave_sync();
      STAMP_END(1);
      // P2: column g of H = L + T1[:, :n] F (lanes < n), g = l + T1[:, n] (lane n)
      T h[W];
      {
        T colA[W], colB[W];
        auto load_col = [&](int s_, T (&dst)[W]) __attribute__((always_inline)) {
          const T* col = EX + srcw[s_];
#pragma unroll
          for (int a = 0; a < W; a++) dst[a] = col[ex_word(0, a)];
        };
        load_col(0, colA);
#pragma unroll
        for (int a = 0; a < W; a++) h[a] = cf_own * t1[a];  // own column: no LDS round trip
        static_for_i<0, NZ>([&](auto s_) {
          constexpr int s2 = decltype(s_)::value;
          T (&cur)[W] = (s2 & 1) ? colB : colA;
          T (&nxt)[W] = (s2 & 1) ? colA : colB;
          if constexpr (s2 + 1 < NZ) load_col(s2 + 1, nxt);
#pragma unroll
          for (int a = 0; a < W; a++) h[a] = t_fma(cf[s2], cur[a], h[a]);
        });
      }
      h[0] += l0;
      h[1] += l1;
#pragma unroll
      for (int a = 0; a < m; a++) h[n + a] += lrow[a];
      // Quu (every lane): l_uu + B^T T1[n.., :n]^T, from the T1 columns the B columns of F touch
      T Quu[m * m];
      {
        T tb[n][m];  // T1[n + a][i] for the rows i of B that are not structurally zero
        static_for_i<0, n>([&](auto i_) {
          constexpr int i = decltype(i_)::value;
          if constexpr (QP::in_b(i)) {
            const T* col = EX + ((rotw + 2 * i) & 31);
#pragma unroll
            for (int a = 0; a < m; a++) tb[i][a] = col[ex_word(0, n + a)];
          }
        });
        static_for_i<0, m>([&](auto a_) {
          constexpr int a = decltype(a_)::value;
          static_for_i<0, m>([&](auto b_) {
            constexpr int b = decltype(b_)::value;
            T acc = T(0);
            bool first = true;
            static_for_i<0, n>([&](auto i_) {
              constexpr int i = decltype(i_)::value;
              if constexpr (Sys::pat(i, n + b) != 0) f_acc<i, n + b>(acc, first, tb[i][a], jv);
            });
            Quu[a * m + b] = (a == b ? luu[a] : T(0)) + acc;
          });
        });
      }
      STAMP_END(4);
      // every value of this step's record has been consumed: fetch the next one now, its HBM / L2
      // latency hides under the factorisation, the gain exchange and the value update
      load_record(t > 0 ? t - 1 : 0);
      // own column of [K | k] = -(regularised Quu)^-1 [Qux | Qu]: control/iterative_ilqr.py:118-126
      T kc[m];
      if constexpr (GENERAL) {
        T Qinv[m * m];
        if constexpr (m == 2) t_quu_inverse2(Quu, lamb, Qinv);
        else t_quu_inverse_m<T, m, true>(Quu, lamb, Qinv, &bad);
#pragma unroll
        for (int a = 0; a < m; a++) {
          T acc = T(0);
#pragma unroll
          for (int b = 0; b < m; b++) acc = t_fma(Qinv[a * m + b], h[n + b], acc);
          kc[a] = -acc;
        }
      } else {
        T Lf[m * m], ir[m], hu[m], sol[m];
        t_quu_factor_pd<T, m>(Quu, lamb, Lf, ir, &bad);
#pragma unroll
        for (int b = 0; b < m; b++) hu[b] = h[n + b];
        t_quu_solve<T, m>(Lf, ir, hu, sol);
#pragma unroll
        for (int a = 0; a < m; a++) kc[a] = -sol[a];
      }
      STAMP_END(5);
      wave_sync();  // every lane is done reading the T1 columns: EX now carries the gain columns
      T* Gt = Wp + L.GK + t * (m * QL::GW);
#pragma unroll
      for (int a = 0; a < m; a++) {
        EX[a * QL::GW + g] = kc[a];
        if (commit) Gt[a * QL::GW + g] = kc[a];
      }
      wave_sync();
      // value update with the UNregularised Quu: Va'[:, g] = [H | g][:n, g] - K^T (Quu [K|k][:, g])
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
        for (int a = 0; a < m; a++) acc = t_fma(EX[a * QL::GW + i], qk[a], acc);
        va[i] = h[i] - acc;
      }
      wave_sync();  // EX is free for the next step's T1 columns
      STAMP_END(6);
    };
    int t = N - 1;
    for (; t >= 1; t -= 2) {
      step(t);
      step(t - 1);
    }
    if (t == 0) step(0);
    // the gains are read back by all lanes of the group in the forward pass
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    wave_sync();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return bad;
  }

  // -- forward pass: control/iterative_ilqr.py:133-160; all lanes of the group redundantly.
  //    Candidate inputs into LDS (UC), candidate records {x_t, trig(x_t)} into the HBM buffer XCn.
  template <bool GENERAL>
  __device__ __forceinline__ T forward(int XCn, const T (&xT)[n], bool* bad) const {
    T x[n], u[m], xn[n], tr[NT];
#pragma unroll
    for (int i = 0; i < n; i++) x[i] = S[L.XU + i];
    T* XC = Wp + XCn;
    // The feedback law u'_a = clip(u_a + k_a + K_a (x' - x)) of the m inputs runs on m lanes at
    // once (lane q takes input q & (m - 1): one gain row, n multiply-adds) and the sin / cos of
    // the three angles on three lanes; the results travel along the 16-lane row by DPP
    // broadcasts.  Same arithmetic as the all-lanes-redundant form, a third of the instructions.
    static_assert((m & (m - 1)) == 0, "m must be a power of two");
    const int ga = g & (m - 1);
    T umax_l = c.u_max[0];
#pragma unroll
    for (int a = 1; a < m; a++) umax_l = (ga == a) ? c.u_max[a] : umax_l;
    auto bc = [](auto lane_, T v) { return row_bcast<decltype(lane_)::value>(v); };
    T kr[NA];  // the lane's gain row [K_a | k_a]
    auto load_gains = [&](int t) __attribute__((always_inline)) {
      const T* Gt = Wp + L.GK + t * (m * QL::GW) + ga * QL::GW;
#pragma unroll
      for (int j = 0; j < NA; j++) kr[j] = Gt[j];
    };
    load_gains(0);
    auto step = [&](const int t) __attribute__((always_inline)) {
      T acc = T(0);
#pragma unroll
      for (int j = 0; j < n; j++) acc = t_fma(kr[j], x[j] - S[L.XU + t * W + j], acc);
      const T ul = clip(S[L.XU + t * W + n + ga] + kr[n] + acc, -umax_l, umax_l);
      load_gains(t + 1 < N ? t + 1 : t);
      static_for_i<0, m>([&](auto a_) {
        constexpr int a = decltype(a_)::value;
        u[a] = row_bcast<a>(ul);
      });
#pragma unroll
      for (int a = 0; a < m; a++) S[L.UC + t * m + a] = u[a];
      Sys::template trig_row<GENERAL>(x, tr, bad, g, bc);
      if (g == 0) {
#pragma unroll
        for (int i = 0; i < n; i++) XC[t * QL::XCW + i] = x[i];
#pragma unroll
        for (int q = 0; q < NT; q++) XC[t * QL::XCW + n + q] = tr[q];
      }
      Sys::step_tr(c, x, u, tr, xn);
#pragma unroll
      for (int i = 0; i < n; i++) x[i] = xn[i];
    };
    int t = 0;
    for (; t + 1 < N; t += 2) {
      step(t);
      step(t + 1);
    }
    if (t < N) step(t);
    Sys::template trig_row<GENERAL>(x, tr, bad, g, bc);
    if (g == 0) {
#pragma unroll
      for (int i = 0; i < n; i++) XC[N * QL::XCW + i] = x[i];
#pragma unroll
      for (int q = 0; q < NT; q++) XC[N * QL::XCW + n + q] = tr[q];
    }
    const T cost = terminal_cost(x, xT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
    wave_sync();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    return cost;
  }

  // an accepted step: the candidate becomes the nominal (states from the HBM record, inputs from UC)
  __device__ __forceinline__ void adopt(int XCn, bool acc) const {
    const T* XC = Wp + XCn;
    constexpr int CH = 8;  // loads in flight per lane (a dependent load-store loop pays the HBM
                           // / L2 latency once per element)
    const int total = (N + 1) * n;
    for (int base = 0; base < total; base += CH * kQG) {
      T v[CH];
#pragma unroll
      for (int q = 0; q < CH; q++) {
        const int e = base + q * kQG + g, ec = e < total ? e : total - 1;
        const int t = ec / n, i = ec - t * n;
        v[q] = XC[t * QL::XCW + i];
      }
#pragma unroll
      for (int q = 0; q < CH; q++) {
        const int e = base + q * kQG + g;
        const int t = e / n, i = e - t * n;
        if (acc && e < total) S[L.XU + t * W + i] = v[q];
      }
    }
    for (int e = g; e < N * m; e += kQG) {
      const int t = e / m, a = e - t * m;
      const T v = S[L.UC + e];
      if (acc) S[L.XU + t * W + n + a] = v;
    }
    wave_sync();
  }
};

// Grid: ceil(B / 4) workgroups of one wavefront; dynamic LDS = 4 * QLayout::lds_total words.
// ws: HBM workspace of B * QLayout::ws_total words.
template <class T, class Sys>
__global__ __launch_bounds__(64) void k_quad_iterate(const DevCfg<T, Sys::n, Sys::m> c,
                                                     const IterArgs<T> a, T* ws) {
  constexpr int n = Sys::n, m = Sys::m, W = n + m;
  using QL = QLayout<Sys>;
  extern __shared__ __align__(16) unsigned char qsmem_raw[];
  T* smem = reinterpret_cast<T*>(qsmem_raw);
  const int lane = threadIdx.x;
  const int64_t prob0 = (int64_t)blockIdx.x * kQPW + lane / kQG;
  // groups past the end of the batch work on a copy of the last problem — in the workspace slot
  // of their own index, which exists (the workspace is sized for whole wavefronts) — and store
  // nothing to the caller's arrays
  const bool real = prob0 < a.B;
  const int64_t prob = real ? prob0 : a.B - 1;
  QuadWorker<T, Sys> w(c, smem, ws, lane, prob0);
  const int N = c.N, g = w.g;
  const QL& L = w.L;
  const auto S = w.S;

  {
    const T* gX = a.X + prob * (int64_t)(n * (N + 1));
    if (g < n) S[L.XU + g] = gX[g * (N + 1)];
    const T* gU = a.U + prob * (int64_t)(m * N);
    for (int e = g; e < m * N; e += kQG) {
      const int aa = e / N, t = e - aa * N;
      S[L.XU + t * W + n + aa] = gU[e];
    }
  }
  T xT[n], ob[6];
#pragma unroll
  for (int i = 0; i < n; i++) xT[i] = a.x_term[prob * n + i];
#pragma unroll
  for (int q = 0; q < 6; q++) ob[q] = a.obs ? a.obs[prob * 6 + q] : T(q == 5 ? -1 : 1);
  T lamb = a.lamb[prob];
  const T ob_pa = T(1) / (ob[2] * ob[2]), ob_pb = T(1) / (ob[3] * ob[3]);
  wave_sync();

#ifdef I2LQR_STAMPS
  auto& st_acc = w.st_acc; auto& st_t0 = w.st_t0; auto& st_t1 = w.st_t1;
  STAMP_BEGIN();
#endif
  int cur = 0;  // which HBM candidate buffer holds the sin / cos values of the nominal
  T cost = w.rollout(L.XC0, xT);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_waitcnt(0);
  wave_sync();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  int it = 0, status = a.early_exit ? 2 /*MAX_ITER*/ : 0 /*RUNNING*/;
  T cost_ret = cost;
  bool fresh = true, active = a.n_iters > 0;
#ifdef I2LQR_STAMPS
  STAMP_END(7);
#endif
  while (__any(active)) {
    const int XCo = cur ? L.XC1 : L.XC0, XCn = cur ? L.XC0 : L.XC1;
#ifdef I2LQR_STAMPS
    STAMP_BEGIN();
#endif
    if (__any(fresh)) w.prep(XCo, ob, ob_pa, ob_pb);
#ifdef I2LQR_STAMPS
    STAMP_END(0);
#endif
    if (__builtin_expect(__any(w.template backward<false>(xT, lamb, active)), 0))
      w.template backward<true>(xT, lamb, active);
#ifdef I2LQR_STAMPS
    STAMP_BEGIN();
#endif
    bool big = false;
    T cost_new = w.template forward<false>(XCn, xT, &big);
    if (__builtin_expect(__any(big), 0)) cost_new = w.template forward<true>(XCn, xT, &big);
#ifdef I2LQR_STAMPS
    STAMP_END(2);
#endif
    bool accepted = false;
    if (active) {
      it++;
      // accept / reject with the lamb schedule: control/iterative_ilqr.py:74-84
      accepted = cost_new < cost;
      fresh = accepted;
      if (accepted) {
        cur ^= 1;
        lamb /= c.lamb_factor;
        const bool conv = t_abs((cost_new - cost) / cost) < c.eps;
        cost_ret = cost_new;
        cost = cost_new;
        if (conv) {
          if (a.early_exit) { status = 1; active = false; }
          if (status == 0) status = 1;
        }
      } else {
        lamb *= c.lamb_factor;
        cost_ret = cost;
        if (lamb > c.max_lamb) {
          if (a.early_exit) { status = 3; active = false; }
          if (status == 0) status = 3;
        }
      }
      if (it >= a.n_iters) active = false;
    } else {
      fresh = false;
    }
    if (__any(accepted)) w.adopt(XCn, accepted);
#ifdef I2LQR_STAMPS
    STAMP_END(3);
#endif
  }
  if (!t_isfinite(cost_ret)) status = 4;
#ifdef I2LQR_STAMPS
  STAMP_BEGIN();
#endif
  if (real) {
    T* gX = a.X + prob * (int64_t)(n * (N + 1));
    for (int e = g; e < n * (N + 1); e += kQG) {
      const int i = e / (N + 1), t = e - i * (N + 1);
      gX[e] = S[L.XU + t * W + i];
    }
    T* gU = a.U + prob * (int64_t)(m * N);
    for (int e = g; e < m * N; e += kQG) {
      const int aa = e / N, t = e - aa * N;
      gU[e] = S[L.XU + t * W + n + aa];
    }
    if (a.K) {  // gains of the last backward pass: workspace [t][a][16] -> K[m][n][N], k[m][N]
      // Eight gathers in flight per lane: a load-store loop that waits for every element pays the
      // L2 / HBM latency 150 times per problem.
      const T* GK = w.Wp + L.GK;
      T* gK = a.K + prob * (int64_t)(m * n * N);
      constexpr int CH = 8;
      const int totK = m * n * N;
      for (int base = 0; base < totK; base += CH * kQG) {
        T v[CH];
#pragma unroll
        for (int q = 0; q < CH; q++) {
          const int e0 = base + q * kQG + g, e = e0 < totK ? e0 : totK - 1;
          const int aa = e / (n * N), r = e - aa * (n * N), j = r / N, t = r - j * N;
          v[q] = GK[(t * m + aa) * QL::GW + j];
        }
#pragma unroll
        for (int q = 0; q < CH; q++) {
          const int e = base + q * kQG + g;
          if (e < totK) gK[e] = v[q];
        }
      }
      T* gk = a.k + prob * (int64_t)(m * N);
      const int totk = m * N;
      for (int base = 0; base < totk; base += CH * kQG) {
        T v[CH];
#pragma unroll
        for (int q = 0; q < CH; q++) {
          const int e0 = base + q * kQG + g, e = e0 < totk ? e0 : totk - 1;
          const int aa = e / N, t = e - aa * N;
          v[q] = GK[(t * m + aa) * QL::GW + n];
        }
#pragma unroll
        for (int q = 0; q < CH; q++) {
          const int e = base + q * kQG + g;
          if (e < totk) gk[e] = v[q];
        }
      }
    }
    if (g == 0) {
      a.lamb[prob] = lamb;
      a.cost[prob] = cost_ret;
      if (a.iters) a.iters[prob] = it;
      if (a.status) a.status[prob] = status;
    }
  }
#ifdef I2LQR_STAMPS
  __builtin_amdgcn_s_waitcnt(0);
  STAMP_END(7);  // slot 7: entry (loads + nominal rollout) + exit (stores), per launch
  if (a.dbg && g == 0 && real)
    for (int q = 0; q < 8; q++) a.dbg[prob * 8 + q] = st_acc[q];
#endif
}

}  // namespace i2lqr

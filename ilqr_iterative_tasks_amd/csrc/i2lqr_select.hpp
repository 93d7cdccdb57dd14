// GPU-side candidate selection and pick of the i2LQR controller round: with
// these two kernels the three outer rounds of iLqr.calc_input (utils/base.py:384-478) chain on the
// device — select -> solve -> relaxed cost -> pick -> next round's guess — with one host
// read-back per control step instead of three.  Sizes are tiny (<= 16 candidates, <= 128 safe-set
// columns): one workgroup each, no tuning needed; they exist to remove host round trips.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace i2lqr {

constexpr int32_t kQfNone = 0x7fffffff;  // I2LQR_QF_NONE: "no candidate in this slot"

// k nearest safe-set columns in the 1-norm for each of L laps, and the gather of the candidates:
// replaces iLqr.select_close_ss (utils/base.py:332-341: argsort of the column-wise 1-norm, first
// k) and the candidate set-up x_terminal = ss[id][:, j], cost_terminal = Qfun[id][j] (:411-412).
//   ss     [L][n][Tmax]  safe-set states, component-major, time contiguous, padded to Tmax
//   T      [L]           valid columns per lap
//   qfun   [L][Tmax]     cost-to-go in steps
//   xg     x_guess, element i at xg[i * xg_stride]
// out: idx[L][k], x_term[L*k][n] (problem-major), qf[L*k].  Ties resolve to the lower column
// (a stable argsort; numpy's default sort is not stable, exact ties do not occur in practice).
// A lap with fewer than k valid columns (T[lap] < k; the reference's argsort()[0:k] then simply
// returns fewer candidates) fills its surplus slots with idx = -1, qf = I2LQR_QF_NONE and a copy of
// the lap's last valid state (zeros if the lap is empty): i2lqr_relax_cost turns that sentinel into
// cost +inf, so a surplus slot can never be picked.
// One workgroup of 128 threads per lap; Tmax <= 1024.
template <class T>
__global__ __launch_bounds__(128) void k_select_candidates(int n, int Tmax, int k, const T* ss,
                                                           const int32_t* Tl, const int32_t* qfun,
                                                           const T* xg, int xg_stride,
                                                           int32_t* idx, T* x_term, int32_t* qf) {
  __shared__ double dist[1024];
  __shared__ double red_v[128];
  __shared__ int red_i[128];
  const int lap = blockIdx.x, tid = threadIdx.x;
  const int Tn = Tl[lap] < 0 ? 0 : (Tl[lap] > Tmax ? Tmax : Tl[lap]);
  const T* S = ss + (int64_t)lap * n * Tmax;
  for (int j = tid; j < Tmax; j += 128) {
    double d = NAN;  // padding columns and columns already taken: never picked (NaN compares false)
    if (j < Tn) {
      d = 0.0;
      for (int i = 0; i < n; i++) d += fabs((double)S[i * Tmax + j] - (double)xg[i * xg_stride]);
      if (!(d == d)) d = INFINITY;  // NaN sorts last
    }
    dist[j] = d;
  }
  __syncthreads();
  for (int r = 0; r < k; r++) {
    double bv = INFINITY;
    int bi = 0x7fffffff;
    for (int j = tid; j < Tmax; j += 128) {
      const double d = dist[j];
      if (d < bv || (d == bv && j < bi)) { bv = d; bi = j; }
    }
    red_v[tid] = bv;
    red_i[tid] = bi;
    __syncthreads();
    for (int s = 64; s > 0; s >>= 1) {
      if (tid < s) {
        const double ov = red_v[tid + s];
        const int oi = red_i[tid + s];
        if (ov < red_v[tid] || (ov == red_v[tid] && oi < red_i[tid])) {
          red_v[tid] = ov;
          red_i[tid] = oi;
        }
      }
      __syncthreads();
    }
    const bool surplus = red_i[0] >= Tn;  // the lap has no r-th nearest column
    const int j = surplus ? (Tn > 0 ? Tn - 1 : 0) : red_i[0];
    if (tid == 0) {
      idx[lap * k + r] = surplus ? -1 : j;
      qf[lap * k + r] = surplus ? kQfNone : qfun[(int64_t)lap * Tmax + j];
      if (!surplus) dist[j] = NAN;  // remove from the next round
    }
    if (tid < n) x_term[(int64_t)(lap * k + r) * n + tid] = Tn > 0 ? S[tid * Tmax + j] : T(0);
    __syncthreads();
  }
}

// The pick of utils/base.py:462-469: `cost_list.index(min(cost_list))` on a list of L lists
// (Python compares lists lexicographically), then the first minimum inside that lap's list; the
// winner's trajectory becomes u_pred / x_pred.  cost_it[L][k], X[L*k][n][N+1], U[L*k][m][N]
// (problem-major).  out: best[2] = {lap position, candidate position}, x_pred[n][N+1],
// u_pred[m][N].  One workgroup.
template <class T>
__global__ __launch_bounds__(64) void k_pick_best(int L, int k, int n, int m, int N,
                                                  const T* cost_it, const T* X, const T* U,
                                                  int32_t* best, T* x_pred, T* u_pred) {
  __shared__ int s_best[2];
  if (threadIdx.x == 0) {
    // Python list ordering: first differing element decides; NaN compares false both ways,
    // which Python's list comparison treats as "not less" -> keep the earlier list
    int bl = 0;
    for (int l = 1; l < L; l++) {
      bool less = false;
      for (int c = 0; c < k; c++) {
        const T a = cost_it[l * k + c], b = cost_it[bl * k + c];
        if (a == b) continue;
        less = a < b;
        break;
      }
      if (less) bl = l;
    }
    int bc = 0;
    for (int c = 1; c < k; c++)
      if (cost_it[bl * k + c] < cost_it[bl * k + bc]) bc = c;
    s_best[0] = bl;
    s_best[1] = bc;
    best[0] = bl;
    best[1] = bc;
  }
  __syncthreads();
  if (!X) return;  // index only (sharded rounds: the winner's trajectory lives on its owner's rank)
  const int64_t w = (int64_t)s_best[0] * k + s_best[1];
  for (int e = threadIdx.x; e < n * (N + 1); e += 64) x_pred[e] = X[w * n * (N + 1) + e];
  for (int e = threadIdx.x; e < m * N; e += 64) u_pred[e] = U[w * m * N + e];
}

// pack[0 : m N] = U[m][N], pack[m N : m N + n (N+1)] = X[n][N+1] of problem idx[0] (clamped to 0 from
// below: "nothing can win" still packs a defined trajectory), in the reference's orientation
// (component-major, time contiguous) whatever the layout: layout 0 problem-major [B][c][T],
// 1 batch-minor [T][c][B], 2 batch-tiled [B/64][T][c][64].  One workgroup.
template <class T>
__global__ __launch_bounds__(256) void k_pack_problem(int64_t B, int n, int m, int N, int layout,
                                                      const T* X, const T* U, const int64_t* idx,
                                                      T* pack) {
  int64_t b = idx[0];
  if (b < 0) b = 0;
  if (b >= B) b = B - 1;
  auto fetch = [&](const T* A, int comps, int T_, int c, int t) -> T {
    if (layout == 0) return A[(b * comps + c) * T_ + t];
    if (layout == 1) return A[((int64_t)t * comps + c) * B + b];
    return A[(((b >> 6) * T_ + t) * comps + c) * 64 + (b & 63)];
  };
  const int nu = m * N, nx = n * (N + 1);
  for (int e = threadIdx.x; e < nu; e += blockDim.x) pack[e] = fetch(U, m, N, e / N, e % N);
  for (int e = threadIdx.x; e < nx; e += blockDim.x)
    pack[nu + e] = fetch(X, n, N + 1, e / (N + 1), e % (N + 1));
}

// Winner of a sharded round whose hand-off rode in the all-gather (i2lqr_allgather_round): every
// rank contributed `width` costs (its shard, padded with +inf) and the pack of its LOCAL winner;
// best_padded[0] is the flat arg-min over the world x width gathered costs.  The global winner is
// its owner's local winner (the arg-min of a union is the arg-min of one of its parts), so the
// owner's pack IS the winner's trajectory: copy it out and translate the padded index into the
// index of the unpadded, contiguously sharded batch (shard r = [r base + min(r, rem), ...),
// dist.shard_range).  best_global = {index (-1 if nothing can win), owner rank}.  One workgroup.
template <class T>
__global__ __launch_bounds__(256) void k_round_winner(int world, int64_t width, int64_t total,
                                                      int64_t pack_count,
                                                      const int64_t* best_padded, const T* pack_all,
                                                      T* winner, int64_t* best_global) {
  const int64_t p = best_padded[0];
  const int64_t q = p < 0 ? 0 : p;  // nothing can win: the first rank's pack, index -1
  const int64_t owner = q / width, loc = q - owner * width;
  const int64_t base = total / world, rem = total - base * world;
  const int64_t lo = owner * base + (owner < rem ? owner : rem);
  if (threadIdx.x == 0) {
    best_global[0] = p < 0 ? -1 : lo + loc;
    best_global[1] = owner;
  }
  for (int64_t e = threadIdx.x; e < pack_count; e += blockDim.x)
    winner[e] = pack_all[owner * pack_count + e];
}

// broadcast x0 into X[:, :, 0] and zero / reset the per-candidate in/out state for one round:
// uvar = 0, xvar[:, 0] = x (utils/base.py:405-408), lamb = lamb0 (:393)
template <class T>
__global__ void k_init_candidates(int64_t B, int n, int m, int N, const T* x0, T lamb0, T* X, T* U,
                                  T* lamb) {
  const int64_t b = blockIdx.x;
  for (int e = threadIdx.x; e < n * (N + 1); e += blockDim.x) {
    const int i = e / (N + 1), t = e - i * (N + 1);
    X[b * n * (N + 1) + e] = (t == 0) ? x0[i] : T(0);
  }
  for (int e = threadIdx.x; e < m * N; e += blockDim.x) U[b * m * N + e] = T(0);
  if (threadIdx.x == 0) lamb[b] = lamb0;
}

}  // namespace i2lqr

// One-problem-per-LANE iLQR kernels for gfx950 (MI355X): the throughput path for large batches.
//
// Every lane of a wavefront owns one iLQR problem; the Riccati step's small blocks (V_xx, the
// A/B Jacobians, Q_xx / Q_ux / Q_uu) live in that lane's registers with every loop fully
// unrolled at compile time, so the sparsity pattern of [A | B] folds into the instruction stream
// and all 64 lanes do useful arithmetic on every VALU instruction.  The trajectory and the
// gains stream through HBM in the BATCH-MINOR, TIME-MAJOR layout
//     X[N+1][n][B]  U[N][m][B]  K[N][m][n][B]  k[N][m][B]  x_term[n][B]  lamb[B]  obs[6][B]
// so a wavefront's access to one (t, component) is one fully coalesced 512-byte (fp64) row and
// the words of one horizon step are adjacent rows.
// This is the streaming form of the algorithm: per iteration and problem it moves X, U, K, k once
// in each direction (SURVEY.md §8(d) algorithmic bytes).
//
// Reference being replaced: control/iterative_ilqr.py:7-160, control/ilqr_helper.py:9-150,
// systems/kinetic_bicycle.py:10-52 (see i2lqr_wave.hpp for the per-phase citations; the
// arithmetic and its order are the same).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "i2lqr_systems.hpp"
#include "i2lqr_wave.hpp"

// This file encodes s_waitcnt fields by hand (gfx9 layout) and is written for gfx950; another
// target's encoding (gfx10+ lgkmcnt width, gfx12 split counters) would turn the literal into a
// silent race: refuse to compile device code for anything else.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__)
#error "i2lqr_lane.hpp hard-codes the gfx9 s_waitcnt encoding: build with --offload-arch=gfx950"
#endif

namespace i2lqr {

constexpr bool kGfx9Waitcnt = true;  // see the target guard above


template <int Begin, int End, class F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (Begin < End) {
    f(std::integral_constant<int, Begin>{});
    static_for<Begin + 1, End>(f);
  }
}

// Scheduling fence between the phases of the unrolled row-block Riccati step: the phases are
// thousands of independent multiply-adds, and a scheduler free to interleave them across phase
// boundaries stretches every live range (registers are the scarce resource at one wavefront per
// SIMD with 90 doubles of value function resident).
#ifndef I2LQR_PHASE_FENCE
#define I2LQR_PHASE_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

template <class T> struct LaneSet {   // one set of per-problem arrays, batch-minor, row stride B
  T* X; T* U; T* x_term; T* obs; T* lamb; T* cost; T* K; T* k;
  int32_t* iters; int32_t* status; int32_t* orig;
  int64_t B;
};

// Compaction folded into the EXIT of a chunk of the chunked solve (round 6; before: a kernel of its
// own, k_lane_compact, between every two chunks).  A wavefront that has finished its chunk packs
// its still-running problems into the next work set — ONE atomic add per wavefront claims their
// slots — and scatters the problems that terminated to the caller's arrays, while the other
// wavefronts of the launch are still iterating.
template <class T> struct LaneCompact {
  int on;             // 0: the kernel's exit is the plain one
  int src_is_user;    // the chunk ran on the caller's arrays: terminated problems are in place
  int user_tiled;     // the caller's arrays are batch-tiled (else batch-minor)
  const int32_t* orig;  // [chunk's columns] index of each problem in the caller's arrays (work sets)
  LaneSet<T> dst;     // work set that receives the survivors; X null: the last chunk, nothing survives
  int32_t* count_out; // slot counter of dst (zeroed by the host before the first chunk)
  LaneSet<T> usr;     // the caller's arrays (K / k / iters / status may be null)
};

template <class T> struct LaneArgs {
  int64_t B;                 // row stride (capacity) of every array; also the batch unless `count`
  int n_iters, early_exit;
  T* X; T* U; const T* x_term; T* lamb; const T* obs; T* cost; T* K; T* k;
  int32_t* iters; int32_t* status;
  // workspace (batch-minor): candidate inputs, gains if K == null
  T* wsU; T* wsK; T* wsk;
  T* wsX;  // second state buffer (k_lane_iterate_pair; launches of <= 256 workgroups), or null
  // chunked solve (solve_compacting): the live batch size is read from device memory, the
  // iteration counter continues from iters[b], and a problem that is still running when the
  // chunk ends gets status RUNNING unless it has reached max_total iterations
  const int32_t* count;
  int resume, max_total;
  // chunks the DEVICE chooses between (two kernels enqueued, one of them runs): the launch is a
  // no-op unless count_lo < *count <= count_hi
  int count_lo, count_hi;
  int two_max;  // k_lane_iterate_pair: the second state buffer is used up to this many live problems
  int lds_grow;  // k_lane_iterate_pair: the launcher may keep more gain steps in LDS than lds_steps
  unsigned long long* dbg;  // diagnostic builds only: [B/64][8] phase cycle sums
  int defer;      // forward pass stores no states; an accepted step re-rolls them
  int lds_steps;  // horizon steps whose gains stay in LDS (dynamic LDS = 64 lds_steps m (n+1) words)
  int reroll;     // forward pass re-rolls the nominal states instead of reading them (fp64, big B)
  int merge;      // deferred mode: accepted candidate inputs are merged into ONE input buffer
  int ckpt;       // only every kSeg-th state lives in HBM between the passes (see backward<.., CK>)
  int stagger;    // fused kernels: every second half-thousand of workgroups starts this many x ~8000 cycles late
  LaneCompact<T> cp;  // chunked solve: compaction at the kernel's exit (cp.on)
};

// Exit of a chunk with the compaction folded in (LaneCompact).  Called by every lane that took part
// in the chunk, with the arrays as the KERNEL sees them (re-based to the wavefront: element (row,
// lane) at p[row * Bs + bl]) and the lane's final lamb / cost / iteration count / status.
//   terminated (status != 0)  chunk on a work set: X, U, gains, lamb, cost, iters, status go to the
//                             caller's arrays at orig[b]; chunk on the caller's arrays: in place.
//   running (status == 0)     x_0, the inputs, x_term, obs, lamb, the iteration count and the
//                             original index go to slot base + rank of the next work set, base from
//                             ONE atomic add per wavefront (every chunk starts by rolling the states
//                             out again, so a survivor carries its inputs and x_0 only).
// The order of the slots is arbitrary (every problem is independent: results do not depend on it).
template <class T>
__device__ __forceinline__ void lane_exit_compact(const LaneCompact<T>& cp, const int n, const int m,
                                                  const int N, const int64_t b, const int64_t Bs,
                                                  const unsigned bl, const T* X, const T* U,
                                                  const T* xt, const T* ob, const T* gK, const T* gk,
                                                  const T lamb, const T cost, const int iters,
                                                  const int status, unsigned long long* trap) {
  (void)trap;
  const int rx = n * (N + 1), ru = m * N, rK = m * n * N;
  const LaneSet<T>& usr = cp.usr;
  auto uaddr = [&](int rows, int row, int64_t p) -> int64_t {
    I2LQR_DBG_CHECK(trap, TAG_COMPACT, row, rows);
    I2LQR_DBG_CHECK(trap, TAG_COMPACT, p, usr.B);
    if (cp.user_tiled) return ((p >> 6) * rows + row) * 64 + (p & 63);
    return (int64_t)row * usr.B + p;
  };
  // rows are moved 32 at a time: 32 independent loads in flight, then 32 stores.  The exit runs
  // with one or two wavefronts per SIMD and every wavefront of a chunk reaches it at about the
  // same time, so what it costs is round trips to memory, not bytes (eight rows per round trip:
  // +41 us on the first chunk of 65536 problems, as much as the k_lane_compact launch it replaced).
  auto move_rows = [&](int rows, auto&& dst_at, auto&& src_at) __attribute__((always_inline)) {
    constexpr int W = 32;
    for (int r = 0; r < rows; r += W) {
      T v[W];
#pragma unroll
      for (int q = 0; q < W; q++)
        if (r + q < rows) v[q] = src_at(r + q);
#pragma unroll
      for (int q = 0; q < W; q++)
        if (r + q < rows) dst_at(r + q, v[q]);
    }
  };
  const bool run = status == 0;
  const unsigned long long mask = __ballot(run);
  if (!run) {
    if (cp.src_is_user || (status & kStatusDelivered)) return;  // in place / the tail kernel's
    const int64_t o = cp.orig[b];
    move_rows(rx, [&](int r, T v) { usr.X[uaddr(rx, r, o)] = v; },
              [&](int r) { return X[(int64_t)r * Bs + bl]; });
    move_rows(ru, [&](int r, T v) { usr.U[uaddr(ru, r, o)] = v; },
              [&](int r) { return U[(int64_t)r * Bs + bl]; });
    if (usr.K) {
      move_rows(rK, [&](int r, T v) { usr.K[uaddr(rK, r, o)] = v; },
                [&](int r) { return gK[(int64_t)r * Bs + bl]; });
      move_rows(ru, [&](int r, T v) { usr.k[uaddr(ru, r, o)] = v; },
                [&](int r) { return gk[(int64_t)r * Bs + bl]; });
    }
    usr.lamb[o] = lamb;
    usr.cost[o] = cost;
    if (usr.iters) usr.iters[o] = iters;
    if (usr.status) usr.status[o] = status;
    return;
  }
  const LaneSet<T>& dst = cp.dst;
  if (!dst.X) return;  // (cannot happen: the last chunk leaves no problem running)
  // the running lanes of this wavefront claim consecutive slots with one atomic add
  const unsigned lane = threadIdx.x & 63;
  const int rank = __popcll(mask & ((1ull << lane) - 1ull));
  int base = 0;
  if (rank == 0) base = atomicAdd(cp.count_out, __popcll(mask));
  base = __builtin_amdgcn_readfirstlane(base);  // (the first active lane here IS the lane of rank 0)
  const int64_t j = (int64_t)base + rank;
  I2LQR_DBG_CHECK(trap, TAG_COMPACT, j, dst.B);
  move_rows(n, [&](int r, T v) { dst.X[(int64_t)r * dst.B + j] = v; },
            [&](int r) { return X[(int64_t)r * Bs + bl]; });
  move_rows(ru, [&](int r, T v) { dst.U[(int64_t)r * dst.B + j] = v; },
            [&](int r) { return U[(int64_t)r * Bs + bl]; });
  move_rows(n, [&](int r, T v) { dst.x_term[(int64_t)r * dst.B + j] = v; },
            [&](int r) { return xt[(int64_t)r * Bs + bl]; });
  if (ob)
    move_rows(6, [&](int r, T v) { dst.obs[(int64_t)r * dst.B + j] = v; },
              [&](int r) { return ob[(int64_t)r * Bs + bl]; });
  dst.lamb[j] = lamb;
  dst.iters[j] = iters;
  dst.status[j] = 0;  // RUNNING (a tail launch may finish it before the next chunk sees it)
  dst.orig[j] = cp.src_is_user ? (int32_t)b : cp.orig[b];
}

// State checkpointing (fp64, large batches: the kernel sits on the HBM roof).  Between the passes of
// an iteration only the states x_0, x_4, x_8, ... are kept in HBM; the backward pass re-rolls the
// states of one segment of kSeg horizon steps at a time from the segment's checkpoint into LDS
// (bit-identical to what a full rollout stores: same code, same inputs) and reads them from
// there.  X traffic per iteration drops from n (N+1) read + n N written to a quarter of that, for
// one more plant step per horizon step; the caller's X is completed by one full re-roll at exit.
constexpr int kSeg = 4;

// Barrier of a main / helper pair of wavefronts (k_lane_iterate_pair) that orders their LDS traffic
// ONLY.  __syncthreads() is a workgroup-scope release + acquire on every address space: its
// s_waitcnt vmcnt(0) would drain the main wavefront's gain stores to HBM — and the helper's
// prefetch loads — at every horizon step.  What the two wavefronts exchange lives in LDS: the LDS
// operations of a wavefront complete in order, lgkmcnt(0) lands them, s_barrier makes the other
// wavefront wait for that point.
__device__ __forceinline__ void pair_barrier() {
  static_assert(kGfx9Waitcnt, "s_waitcnt literal below is the gfx9 field layout");
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// words of T the workspace needs for B problems
template <class Sys> __host__ __device__ inline int64_t lane_workspace_words(int N, int64_t B) {
  constexpr int n = Sys::n, m = Sys::m;
  return B * (int64_t)(m * N + m * n * N + m * N);
}

// Batch-minor ("rows of B") vs batch-tiled ("tiles of 64 problems, rows of 64") addressing.  In
// the tiled layout all rows of a wavefront's 64 problems are contiguous (one ~0.7 MB region at
// n=6, N=20, fp64): DRAM-page and TLB friendly at very large B.  Each array of `rows` rows is
// re-based to the wavefront's tile, after which the kernels index it as a batch of 64.
template <bool TILED> struct LaneView {
  int64_t Bs;     // row stride
  unsigned bl;    // this lane's index inside the wavefront's 64 columns of a row
  int64_t tile;   // wavefront index (workgroups are one wavefront): uniform, lives in SGPRs
  __device__ LaneView(int64_t B)
      : Bs(TILED ? 64 : B), bl(threadIdx.x & 63), tile(blockIdx.x) {}  // (& 63: two-wavefront workgroups)
  // Re-base an array of `rows` rows to this wavefront's 64 problems.  The result is WAVE-UNIFORM
  // (scalar registers); element (row, lane) is p[row * Bs + bl], which the compiler addresses as
  // scalar base + per-lane 32-bit offset + immediate row offset.
  template <class P> __device__ __forceinline__ P* rebase(P* p, int rows) const {
    if (!p) return p;
    return TILED ? p + tile * (int64_t)rows * 64 : p + tile * 64;
  }
};

#ifndef I2LQR_ROWS_NT
#define I2LQR_ROWS_NT 1
#endif
constexpr bool kRowsNT = I2LQR_ROWS_NT != 0;
#ifndef I2LQR_ROWS_NT_ALL
#define I2LQR_ROWS_NT_ALL 0
#endif
constexpr bool kRowsNTAll = I2LQR_ROWS_NT_ALL != 0;  // experiment: every row access of those kernels

#ifndef I2LQR_DEEP_PREFETCH
#define I2LQR_DEEP_PREFETCH 1
#endif
#ifndef I2LQR_FWD_DEPTH
#define I2LQR_FWD_DEPTH 2  // forward_rows: horizon steps between a step's loads and their use
#endif
#ifndef I2LQR_WARM_INPUTS
#define I2LQR_WARM_INPUTS 1  // k_lane_iterate_rows: LDS-direct warm-up loads of the next step's inputs
#endif
#ifndef I2LQR_PAIR_FWD_DEPTH
// k_lane_iterate_pair: prefetch distance of the forward pass in horizon steps (tools/ab_bench.py at
// 16384 problems, fp64: 1: 357, 2: 374, 3: 374, 4: 369 M it/s; the register sets of 3 and 4 spill)
#define I2LQR_PAIR_FWD_DEPTH 2
#endif
#ifndef I2LQR_DEEP64
#define I2LQR_DEEP64 0  // experiment: the two-step prefetch distance of the fp32 kernels in fp64 too
#endif
#ifndef I2LQR_F64_WAVES
#define I2LQR_F64_WAVES 1
#endif
#ifndef I2LQR_F32_WAVES
#define I2LQR_F32_WAVES 1
#endif
template <class T, class Sys, bool HASQR, bool TILED> struct LaneWorker {
  static constexpr int n = Sys::n, m = Sys::m, W = n + m, NT = Sys::NTRIG, NV = Sys::NVAR,
                       NC = Sys::NCONST;
  using Cfg = DevCfg<T, n, m>;
  const Cfg& c;
  const int N;
  T pc[NC > 0 ? NC : 1];  // plant-constant entries of F (pattern codes >= 100); wave-uniform
  const int64_t Bs_;    // row stride of the batch-minor layout (the tiled one is 64 at compile time)
  const unsigned lane;  // this lane's column inside the wavefront's re-based rows
  // Gains of the first horizon steps stay in LDS: the backward pass produces them last and the
  // forward pass consumes them first, so they never need to travel through HBM (flushed once at
  // kernel exit for the caller).  Step 0 is special: x_0 is given, so the candidate starts ON the
  // nominal and K_0 (x_0 - x_0) = 0 whatever K_0 is — the forward pass needs k_0 only (m words
  // instead of m (n + 1)); K_0 is written to HBM by the backward passes that can be the launch's
  // last (k0_out) and never read back.  Steps 1..lds_steps keep [K | k]: word (t, q) of lane l
  // sits at lds[((t - 1) m (n+1) + q) * 64 + l], k_0[a] behind them: consecutive lanes on
  // consecutive words, conflict-free.
  // (typed as LDS: through a generic pointer some of these accesses were flat_load / flat_store)
  typedef __attribute__((address_space(3))) T lds_t;
  lds_t* lds = nullptr;
  bool has_lds = false;  // false (the one-pass kernels): every gain goes through HBM.  (Not the
                         // pointer's nullness: the dynamic LDS of a kernel starts at LDS address 0.)
  int lds_steps = 0;
#ifdef I2LQR_STAMPS
  mutable unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t0 = 0, st_t1 = 0;
#endif

  __device__ LaneWorker(const Cfg& c_, int64_t Bs, unsigned lane_)
      : c(c_), N(c_.N), Bs_(Bs), lane(lane_) {
#pragma unroll
    for (int q = 0; q < NC; q++) pc[q] = uniform(Sys::plant_const(c, q));
  }
  // hides a value's origin from the optimiser (no instruction)
  static __device__ __forceinline__ void opaque(double& v) { asm volatile("" : "+v"(v)); }
  static __device__ __forceinline__ void opaque(float& v) { asm volatile("" : "+v"(v)); }
  // a value every lane computes identically, moved to scalar registers (a v_fma_f64 takes one
  // scalar operand: the constant costs no vector registers in the unrolled Riccati step)
  static __device__ __forceinline__ double uniform(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)),
                            __builtin_amdgcn_readfirstlane(__double2loint(v)));
  }
  static __device__ __forceinline__ float uniform(float v) {
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
  }
  // Element (row, this lane) of a re-based (wave-uniform) array.  Rows are TIME-major:
  //   X: t n + i    U, k: t m + a    K: (t m + a) n + j
  // so the words of one horizon step are adjacent rows: one scalar base per step, immediate
  // offsets per word, no per-access vector address arithmetic.
  __device__ __forceinline__ int64_t stride() const { return TILED ? (int64_t)64 : Bs_; }
  template <class P> __device__ __forceinline__ P& at(P* p, int row) const {
    return (p + (int64_t)row * stride())[lane];
  }
  // Row groups (kernels of the row-block plants): CNT consecutive rows starting at `row0` are
  // addressed as ONE scalar base per group of eight rows (formed per use: two scalar adds, hidden
  // from the optimiser so that it is not split into per-row loop invariants — 52 gain rows per
  // step otherwise become 52 hoisted address registers, spilled to scratch and reloaded one by
  // one) + the lane's offset + an immediate row offset (global_* immediates reach 4 KiB: eight
  // 512-byte rows of the tiled layout).  f(integral_constant<row index>, word reference).
  template <class P> __device__ __forceinline__ P* rows(P* p, int row) const {
    P* q = p + (int64_t)row * stride();
    asm volatile("" : "+s"(q));
    return q;
  }
  // The hidden base has lost its address space with its origin: it is cast back to GLOBAL here.
  // Left generic, every access was a flat_load / flat_store — 64-bit vector addresses instead of
  // scalar base + immediate, and counted in BOTH vmcnt and lgkmcnt: every wait for an LDS read then
  // also waited for the 52 gain stores in flight to HBM.  f takes (index, const T&) to read the
  // word or (index, T&) to assign it.
  // NT: the words are touched once per pass (the gains of the row-block plants: written by the
  // backward pass, read by the forward pass, 4 GB apart at 65536 problems) — non-temporal accesses,
  // so that they do not push the state and input rows the prefetches brought in out of the caches.
  // Round 6, same-process A/B (profiles/r06_ab_quad12_nontemporal.json): quad12 65536 problems fp64
  // 76.8 -> 78.9 M it/s, fp32 92.7 -> 99.8, 16384 problems 31.1 -> 32.1, stage weights 19.2 -> 20.0;
  // on EVERY row access of the kernel (-DI2LQR_ROWS_NT_ALL=1): 79.6 / 97.7 / 31.1 / 20.2 — not
  // ahead everywhere, so only the gains.  Same values either way (cache hints).
  template <int CNT, bool NT = kRowsNTAll, class P, class F>
  __device__ __forceinline__ void for_rows(P* p, int row0, F&& f) const {
    I2LQR_DBG_CHECK(c.trap, TAG_LANE_ROW_K, row0 + CNT - 1, m * n * N + n);  // (largest array)
    using V = std::remove_const_t<P>;
    typedef __attribute__((address_space(1))) P GP;
    static_for<0, (CNT + 7) / 8>([&](auto g_) {
      constexpr int g = decltype(g_)::value;
      GP* q = (GP*)rows(p, row0 + 8 * g);
      static_for<0, (CNT - 8 * g < 8 ? CNT - 8 * g : 8)>([&](auto r_) {
        constexpr int r = decltype(r_)::value;
        using IC = std::integral_constant<int, 8 * g + r>;
        GP* word = q + (int64_t)r * stride() + lane;
        if constexpr (std::is_invocable_v<F&, IC, const V&>) {
          V v;
          if constexpr (NT) v = __builtin_nontemporal_load(word);
          else v = *word;
          f(IC{}, v);
        } else {
          V v;
          f(IC{}, v);
          typedef __attribute__((address_space(1))) V GV;
          if constexpr (NT) __builtin_nontemporal_store(v, const_cast<GV*>(word));
          else *const_cast<GV*>(word) = v;  // (this branch is only taken with a non-const P)
        }
      });
    });
  }
  // Warm-up of CNT consecutive rows (this lane's word of each): a 4-byte LDS-direct load per row
  // into a 256-byte sink nobody reads.  No vector register is written, so nothing stays allocated
  // while the loads are in flight — the rows are in the L2 when the real loads ask for them.
  unsigned* sink = nullptr;  // set by the kernel (an LDS object of its own)
  template <int CNT, class P>
  __device__ __forceinline__ void warm_rows(const P* p, int row0) const {
    typedef __attribute__((address_space(1))) const void gptr;
    typedef __attribute__((address_space(3))) void lptr;
    static_for<0, (CNT + 7) / 8>([&](auto g_) {
      constexpr int g = decltype(g_)::value;
      const P* q = rows(p, row0 + 8 * g);
      static_for<0, (CNT - 8 * g < 8 ? CNT - 8 * g : 8)>([&](auto r_) {
        constexpr int r = decltype(r_)::value;
        if constexpr (TILED)  // rows 512 bytes apart: the instruction's immediate offset
          __builtin_amdgcn_global_load_lds((gptr*)&q[lane], (lptr*)sink, 4, r * 64 * (int)sizeof(P), 0);
        else
          __builtin_amdgcn_global_load_lds((gptr*)&(q + (int64_t)r * stride())[lane], (lptr*)sink, 4,
                                           0, 0);
      });
    });
  }
  // (debug build: the indices are checked against the array's extent, i2lqr_debug.hpp)
  __device__ __forceinline__ int rx(int i, int t) const {
    I2LQR_DBG_CHECK(c.trap, TAG_LANE_ROW_X, t * n + i, n * (N + 1));
    I2LQR_DBG_CHECK(c.trap, TAG_LANE_ROW_X, i, n);
    return t * n + i;
  }
  __device__ __forceinline__ int ru(int a, int t) const {
    I2LQR_DBG_CHECK(c.trap, TAG_LANE_ROW_U, t * m + a, m * N);
    I2LQR_DBG_CHECK(c.trap, TAG_LANE_ROW_U, a, m);
    return t * m + a;
  }
  __device__ __forceinline__ int rK(int a, int j, int t) const {
    I2LQR_DBG_CHECK(c.trap, TAG_LANE_ROW_K, (t * m + a) * n + j, m * n * N);
    I2LQR_DBG_CHECK(c.trap, TAG_LANE_ROW_K, j, n);
    I2LQR_DBG_CHECK(c.trap, TAG_LANE_ROW_K, a, m);
    return (t * m + a) * n + j;
  }
  __device__ __forceinline__ lds_t& lds_gain(int t, int q) const {  // 1 <= t <= lds_steps
    I2LQR_DBG_CHECK(c.trap, TAG_LANE_LDS, t - 1, lds_steps);
    I2LQR_DBG_CHECK(c.trap, TAG_LANE_LDS, q, m * (n + 1));
    return lds[((t - 1) * (m * (n + 1)) + q) * 64 + (threadIdx.x & 63)];
  }
  __device__ __forceinline__ lds_t& lds_k0(int a) const {
    I2LQR_DBG_CHECK(c.trap, TAG_LANE_LDS, a, m);
    return lds[(lds_steps * (m * (n + 1)) + a) * 64 + (threadIdx.x & 63)];
  }
  // write the LDS-resident gains of this lane to HBM (kernel exit; K_0 is there already)
  __device__ __forceinline__ void flush_gains(T* gK, T* gk) const {
#pragma unroll
    for (int a = 0; a < m; a++) at(gk, ru(a, 0)) = lds_k0(a);
    for (int t = 1; t <= lds_steps && t < N; t++) {
#pragma unroll
      for (int a = 0; a < m; a++) {
#pragma unroll
        for (int j = 0; j < n; j++)
          at(gK, rK(a, j, t)) = lds_gain(t, a * (n + 1) + j);
        at(gk, ru(a, t)) = lds_gain(t, a * (n + 1) + n);
      }
    }
  }

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
  __device__ __forceinline__ T terminal_cost(const T (&x)[n], const T (&xT)[n]) const {
    T d[n];
#pragma unroll
    for (int i = 0; i < n; i++) d[i] = x[i] - xT[i];
    return quad_form<n>(c.Qt, d);
  }

  // entry (i, a) of F = [A | B] from the compile-time pattern
  template <int i, int a> __device__ __forceinline__ T f_entry(const T (&jv)[NV]) const {
    constexpr int code = Sys::pat(i, a);
    if constexpr (code == 1) return T(1);
    else if constexpr (code == 2) return c.dt;
    else if constexpr (code >= 100) return pc[code - 100];
    else return jv[code - 3];
  }
  // acc += F[i][a] * v  (skipped at compile time for structural zeros; exact for ones)
  template <int i, int a> __device__ __forceinline__ void f_acc(T& acc, bool& first, T v,
                                                                const T (&jv)[NV]) const {
    constexpr int code = Sys::pat(i, a);
    if constexpr (code == 0) {
      return;
    } else if constexpr (code == 1) {
      acc = first ? v : acc + v;
      first = false;
    } else {
      const T f = f_entry<i, a>(jv);
      acc = first ? f * v : t_fma(f, v, acc);
      first = false;
    }
  }

  // -- nominal rollout + cost (control/iterative_ilqr.py:32-48) --------------------------------
  // ckx: only the checkpoint states (t % kSeg == 0) are stored (state checkpointing, below)
  __device__ __forceinline__ T rollout(T* X, T* U, const T (&xT)[n], bool ckx = false) const {
    T x[n], u[m], xn[n], tr[NT];
#pragma unroll
    for (int i = 0; i < n; i++) x[i] = at(X, rx(i, 0));
    T cost = T(0);
    for (int t = 0; t < N; t++) {
#pragma unroll
      for (int a = 0; a < m; a++) {
        u[a] = clip(at(U, ru(a, t)), -c.u_max[a], c.u_max[a]);
        at(U, ru(a, t)) = u[a];
      }
      Sys::trig(x, tr);
      Sys::step_tr(c, x, u, tr, xn);
      if (!ckx || ((t + 1) & (kSeg - 1)) == 0) {
#pragma unroll
        for (int i = 0; i < n; i++) at(X, rx(i, t + 1)) = xn[i];
      }
      cost = cost + stage_cost(x, c.xtarget, u);
#pragma unroll
      for (int i = 0; i < n; i++) x[i] = xn[i];
    }
    cost = cost + terminal_cost(x, xT);
    return cost;
  }

  // Re-roll the nominal states X[:, 1..N] from the (already clipped) nominal inputs: restores the
  // trajectory after a rejected step, whose candidate was written over it in place.  Bit-identical
  // to what rollout() / an accepted forward() stored (same code, same inputs).
  // Blocks of RB horizon steps: the inputs of block b+1 are loaded while block b computes (two
  // register sets take turns), and the states of a block are stored one block LATE, right after the
  // next block's inputs have arrived.  On gfx950 loads and stores share vmcnt and a wait with both
  // kinds pending drains the counter: this way everything pending at the one wait per block was
  // issued a whole block (RB steps) earlier.
  // MERGE (deferred mode): the inputs of the lanes that ACCEPTED come from the candidate buffer
  // Un, the others keep theirs; every lane writes its current inputs back to U, so that U stays
  // ONE buffer with full 64-lane rows for the whole wavefront whatever the lanes decided (per-lane
  // buffer swapping split every input row of a wavefront with mixed decisions over two buffers:
  // twice the lines per access).
  template <bool MERGE = false>
  __device__ __forceinline__ void restore_states_blocked(T* X, T* U, const T* Un = nullptr,
                                                         bool acc = false) const {
    constexpr int RB = 4;
    T x[n], xn[n], tr[NT];
#pragma unroll
    for (int i = 0; i < n; i++) x[i] = at(X, rx(i, 0));
    T xs[RB][n];  // states computed by the previous block
    auto load_block = [&](const int b, T (&ul)[RB][m]) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < RB; j++)
        if (b * RB + j < N) {
#pragma unroll
          for (int a = 0; a < m; a++) {
            ul[j][a] = at(U, ru(a, b * RB + j));
            if constexpr (MERGE) {
              const T un = at(Un, ru(a, b * RB + j));
              ul[j][a] = acc ? un : ul[j][a];
            }
          }
        }
    };
    auto block = [&](const int b, T (&ucur)[RB][m], T (&unext)[RB][m])
        __attribute__((always_inline)) {
      T u[RB][m];
#pragma unroll
      for (int j = 0; j < RB; j++)
#pragma unroll
        for (int a = 0; a < m; a++) u[j][a] = ucur[j][a];
      if ((b + 1) * RB < N) load_block(b + 1, unext);
      if (b > 0) {  // the previous block was complete: RB states
#pragma unroll
        for (int j = 0; j < RB; j++)
#pragma unroll
          for (int i = 0; i < n; i++) at(X, rx(i, (b - 1) * RB + j + 1)) = xs[j][i];
      }
      if constexpr (MERGE) {
#pragma unroll
        for (int j = 0; j < RB; j++)
          if (b * RB + j < N) {
#pragma unroll
            for (int a = 0; a < m; a++) at(U, ru(a, b * RB + j)) = u[j][a];
          }
      }
#pragma unroll
      for (int j = 0; j < RB; j++)
        if (b * RB + j < N) {
          Sys::trig(x, tr);
          Sys::step_tr(c, x, u[j], tr, xn);
#pragma unroll
          for (int i = 0; i < n; i++) {
            xs[j][i] = xn[i];
            x[i] = xn[i];
          }
        }
    };
    T ua[RB][m], ub[RB][m];
    load_block(0, ua);
    const int nb = (N + RB - 1) / RB;
    int b = 0;
    for (; b + 1 < nb; b += 2) {
      block(b, ua, ub);
      block(b + 1, ub, ua);
    }
    if (b < nb) block(b, ua, ub);
#pragma unroll
    for (int j = 0; j < RB; j++)  // the last block's states
      if ((nb - 1) * RB + j < N) {
#pragma unroll
        for (int i = 0; i < n; i++) at(X, rx(i, (nb - 1) * RB + j + 1)) = xs[j][i];
      }
  }

  // CKX: only the checkpoint states (t % kSeg == 0) are stored
  template <bool MERGE = false, bool CKX = false>
  __device__ __forceinline__ void restore_states_paired(T* X, T* U, const T* Un = nullptr,
                                                        bool acc = false) const {
    T x[n], xn[n], tr[NT];
#pragma unroll
    for (int i = 0; i < n; i++) x[i] = at(X, rx(i, 0));
    // two input register sets take turns, each re-loaded for step t+2 right after step t read it
    auto load_u = [&](const int t, T (&ul)[m]) __attribute__((always_inline)) {
#pragma unroll
      for (int a = 0; a < m; a++) {
        ul[a] = at(U, ru(a, t));
        if constexpr (MERGE) {
          const T un = at(Un, ru(a, t));
          ul[a] = acc ? un : ul[a];
        }
      }
    };
    auto body = [&](const int t, T (&ul)[m]) __attribute__((always_inline)) {
      T u[m];
#pragma unroll
      for (int a = 0; a < m; a++) u[a] = ul[a];
      if (t + 2 < N) load_u(t + 2, ul);
      if constexpr (MERGE) {
#pragma unroll
        for (int a = 0; a < m; a++) at(U, ru(a, t)) = u[a];
      }
      Sys::trig(x, tr);
      Sys::step_tr(c, x, u, tr, xn);
      const bool keep = !CKX || ((t + 1) % kSeg == 0);
#pragma unroll
      for (int i = 0; i < n; i++) {
        if (keep) at(X, rx(i, t + 1)) = xn[i];
        x[i] = xn[i];
      }
    };
    T ua[m], ub[m];
    load_u(0, ua);
    if (N >= 2) load_u(1, ub);
    int t = 0;
    for (; t + 1 < N; t += 2) {
      body(t, ua);
      body(t + 1, ub);
    }
    if (t < N) body(t, ua);
  }

  // fp32 (DEEP) takes the blocked form (+3-4 %); in fp64 its 40 extra live registers cost more
  // AGPR traffic than the waits it saves (-1.5 %)
  __device__ __forceinline__ void restore_states(T* X, T* U) const {
    if constexpr (DEEP) restore_states_blocked(X, U);
    else restore_states_paired(X, U);
  }
  // deferred mode: merge the accepted candidates into U (one buffer, full rows) and re-roll
  template <bool CKX = false>
  __device__ __forceinline__ void merge_and_restore(T* X, T* U, const T* Un, bool acc) const {
    if constexpr (DEEP) restore_states_blocked<true>(X, U, Un, acc);
    else restore_states_paired<true, CKX>(X, U, Un, acc);
  }

  // obstacle barrier terms at (px, py), horizon index t: control/ilqr_helper.py:32-51, :121-147
  // pa, pb = 1 / width^2, 1 / height^2: the same for every horizon step, computed once by the caller
  __device__ __forceinline__ void obstacle(const T (&ob)[6], T pa, T pb, T px, T py, int t,
                                           T (&o)[5]) const {
#pragma unroll
    for (int q = 0; q < 5; q++) o[q] = T(0);
    if (ob[5] >= T(0)) {
      const int opt = (int)ob[5];
      T dz = px - ob[0], dy = py - ob[1];
      if (opt == 1) dy = py - (ob[1] + T(t) * ob[4]);
      if (opt == 2) dz = px - (ob[0] - T(t) * ob[4]);
      const T h = T(1) + c.safety_margin - (dz * pa * dz + dy * pb * dy);
      const T hd0 = T(-2) * pa * dz, hd1 = T(-2) * pb * dy;
      const T e = t_exp(c.obs_q2 * h);
      const T c1 = c.obs_q12 * e, c2 = c.obs_q122 * e;
      o[0] = c1 * hd0;
      o[1] = c1 * hd1;
      o[2] = c2 * (hd0 * hd0);
      o[3] = c2 * (hd0 * hd1);
      o[4] = c2 * (hd1 * hd1);
    }
  }

  // GENERAL = false: the same terms without a branch — computed for every lane (a lane without an
  // obstacle carries finite placeholders) and selected
  template <bool GENERAL>
  __device__ __forceinline__ void obstacle_sel(const T (&ob)[6], T pa, T pb, T px, T py, int t,
                                               T (&o)[5]) const {
    if constexpr (GENERAL) {
      obstacle(ob, pa, pb, px, py, t, o);
    } else {
      const bool has = ob[5] >= T(0);
      const int opt = has ? (int)ob[5] : 0;
      const T dy = opt == 1 ? py - (ob[1] + T(t) * ob[4]) : py - ob[1];
      const T dz = opt == 2 ? px - (ob[0] - T(t) * ob[4]) : px - ob[0];
      const T h = T(1) + c.safety_margin - (dz * pa * dz + dy * pb * dy);
      const T hd0 = T(-2) * pa * dz, hd1 = T(-2) * pb * dy;
      const T e = t_exp(c.obs_q2 * h);
      const T c1 = c.obs_q12 * e, c2 = c.obs_q122 * e;
      o[0] = has ? c1 * hd0 : T(0);
      o[1] = has ? c1 * hd1 : T(0);
      o[2] = has ? c2 * (hd0 * hd0) : T(0);
      o[3] = has ? c2 * (hd0 * hd1) : T(0);
      o[4] = has ? c2 * (hd1 * hd1) : T(0);
    }
  }

  // regularised inverse of Q_uu (m == 2 closed form, control/iterative_ilqr.py:118-123)
  // m > 2: LDL^T inverse of Quu + lamb I where Quu is positive definite, clamped eigenvalues behind
  // a wave-uniform unlikely branch otherwise (t_quu_inverse_m)
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
  // Reads the nominal (X, U), writes the gains to gK[N][m][n][B], gk[N][m][B].
  // SYM: V_xx, Q_xx and Q_uu are symmetric in exact arithmetic (the reference never re-symmetrises
  // them, so its copies differ by round-off, ~1e-16 relative).  The lane kernels keep only the
  // upper triangles: a quarter fewer multiply-adds and ~30 fewer live doubles per lane, which is
  // what lets the unrolled step stay in registers.  Cost weights must be symmetric (checked in
  // i2lqr_create for these layouts).  The one-problem-per-wavefront kernels keep the full blocks.
  static constexpr bool SYM = true;
  static constexpr bool DEEP = (sizeof(T) == 4 || I2LQR_DEEP64) && I2LQR_DEEP_PREFETCH;

  // CK: checkpointed states (kSeg above); seg = (kSeg + 1) n 64 words of LDS for this wavefront.
  // ROLE (k_lane_iterate_pair: a HELPER wavefront beside the main one, where the launch leaves SIMDs
  // idle): 0 = the whole pass on this wavefront; 2 = the helper's half — the part of every step that
  // depends on the nominal trajectory only (loads of x_{t+1}, x_t, u_t; sin / cos; Jacobian entries;
  // the input barriers' exponentials; the obstacle term: 30 % of an iteration's cycles at 16384
  // problems), written as a RECORD of kRec words per lane into one of two LDS slots; 1 = the main
  // wavefront's half — the Riccati step on the record the helper left.  One workgroup barrier per
  // horizon step: helper  R(N-1) | R(N-2) | ... ,  main  | S(N-1) | S(N-2) ...  — the helper is a step
  // ahead and the shorter of the two, so the main wavefront does not wait.  Same operations on the
  // same operands as ROLE 0 (the record travels through LDS unchanged): bit-identical.
  static constexpr int kRec = NV + 5 + 2 * m + (HASQR ? n : 0);  // jv, obstacle terms, l_u, l_uu (+ 2 Q dx_t)
  lds_t* rec = nullptr;                        // [2][kRec][64], set by the kernel
  template <bool FASTBAR = false, bool CK = false, int ROLE = 0>
  __device__ __forceinline__ void backward(const T* X, const T* U, const T (&xT)[n],
                                           const T (&ob)[6], T lamb, T* gK, T* gk, bool k0_out,
                                           lds_t* seg = nullptr) const {
    static_assert(!(CK && DEEP), "checkpointed states are built for the fp64 kernels");
    static_assert(ROLE == 0 || (!CK && Sys::NBLK == 0),
                  "the helper form is built for the plain pass of the bicycles");
    if constexpr (Sys::NBLK > 0) {
      static_assert(!CK, "the row-block form has no checkpointed variant");
      __shared__ T lds_gains[kGainWords];
      backward_blocked<FASTBAR, true>(X, U, xT, ob, lamb, gK, gk, lds_gains);
      return;
    }
    const T ob_pa = T(1) / (ob[2] * ob[2]), ob_pb = T(1) / (ob[3] * ob[3]);
    const unsigned l64 = threadIdx.x & 63;
    // state k of the current segment (k = 0: the checkpoint), component i: conflict-free LDS words
    auto seg_at = [&](int k, int i) -> lds_t& {
      I2LQR_DBG_CHECK(c.trap, TAG_LANE_LDS, k * n + i, (kSeg + 1) * n);
      return seg[(k * n + i) * 64 + l64];
    };
    T useg[kSeg][m];  // inputs of the current segment
    // re-roll segment sg (steps sg kSeg .. sg kSeg + len - 1) from its checkpoint into LDS
    auto roll_segment = [&](const int sg, const int len) __attribute__((always_inline)) {
      T x[n], xn[n], tr[NT];
      const int t0 = sg * kSeg;
#pragma unroll
      for (int i = 0; i < n; i++) x[i] = at(X, rx(i, t0));
#pragma unroll
      for (int k = 0; k < kSeg; k++)
        if (k < len) {
#pragma unroll
          for (int a = 0; a < m; a++) useg[k][a] = at(U, ru(a, t0 + k));
        }
#pragma unroll
      for (int i = 0; i < n; i++) seg_at(0, i) = x[i];
#pragma unroll
      for (int k = 0; k < kSeg; k++)
        if (k < len) {
          Sys::trig(x, tr);
          Sys::step_tr(c, x, useg[k], tr, xn);
#pragma unroll
          for (int i = 0; i < n; i++) {
            seg_at(k + 1, i) = xn[i];
            x[i] = xn[i];
          }
        }
    };
    const int last_sg = (N - 1) / kSeg, last_len = N - last_sg * kSeg;
    if constexpr (CK) roll_segment(last_sg, last_len);
    T Va[n][n + 1];  // [Vxx | Vx]; with SYM only Va[i][j >= i] and the last column are live
    if constexpr (ROLE != 2) {
      // get_cost_final(): control/ilqr_helper.py:106-150
      T xN[n], o[5];
#pragma unroll
      for (int i = 0; i < n; i++) xN[i] = CK ? seg_at(last_len, i) : at(X, rx(i, N));
      obstacle(ob, ob_pa, ob_pb, xN[0], xN[1], N, o);
#pragma unroll
      for (int i = 0; i < n; i++) {
        T vx = T(0);
#pragma unroll
        for (int r = 0; r < n; r++) {
          Va[i][r] = T(2) * c.Qt[i * n + r];
          vx += T(2) * c.Qt[i * n + r] * (xN[r] - xT[r]);
        }
        Va[i][n] = vx;
      }
      Va[0][0] += o[2]; Va[0][1] += o[3]; Va[1][0] += o[3]; Va[1][1] += o[4];
      Va[0][n] += o[0]; Va[1][n] += o[1];
    }
    // Software pipeline: the trajectory inputs of a step are consumed at its start (Jacobian
    // entries, barrier terms); the loads of step t-1 are then issued into the same registers, so
    // the HBM latency hides under the Riccati arithmetic of step t.  X[:, t] is read ONCE: it is
    // x_t of step t (obstacle / stage terms) and the evaluation state x_{t+1} of step t-1.
    // DEEP (fp32): two buffers take turns, each re-loaded for step t-2 right after step t has
    // consumed it — twice the distance between a load and its use.  fp32 moves half the bytes of
    // fp64 per step, so its steps are too short for a one-step distance under load (a wavefront
    // alone on its SIMD cannot hide the rest); fp64 sits on the HBM roof either way.
    constexpr int D = DEEP ? 2 : 1;  // prefetch distance in horizon steps
    T xe[n], xp[n], u[m];  // x_{t+1}, x_t, u_t
    if constexpr (!CK && ROLE != 1) {
#pragma unroll
      for (int i = 0; i < n; i++) xe[i] = at(X, rx(i, N));
#pragma unroll
      for (int i = 0; i < n; i++) xp[i] = at(X, rx(i, N - 1));
#pragma unroll
      for (int a = 0; a < m; a++) u[a] = at(U, ru(a, N - 1));
    }
    auto body = [&](const int t, T (&xp)[n], T (&u)[m]) __attribute__((always_inline)) {
      T jv[NV], o[5], tr[NT];
      T lu[m], luu[m];
      T lxq[n];  // 2Q dX[:, t]: control/ilqr_helper.py:29
      STAMP_BEGIN();
      if constexpr (ROLE != 1) {
      Sys::trig(xe, tr);  // the same values the rollout used for the dynamics of step t+1
      Sys::jac_var(c, xe, u, tr, jv);
      obstacle(ob, ob_pa, ob_pb, xp[0], xp[1], t, o);
      // input barrier, add_control_constraint(): control/ilqr_helper.py:83-103
      //   l_u = q1 q2 (e_hi - e_lo),  l_uu = q1 q2^2 (e_hi + e_lo),
      //   e_hi = exp(q2 (u - u_max)),  e_lo = exp(q2 (-u_max - u)).
      // FASTBAR (fused fp64 kernels, whose inputs are clipped to [-u_max, u_max] by the rollout /
      // forward pass, and configurations with |2 q2 u_max| < 600): e_hi e_lo = exp(-2 q2 u_max) is
      // a constant, so e_lo = ctrl_c / e_hi — one short exp (no range handling: the argument is
      // bounded) and one reciprocal per input instead of two general exps; a few ulp apart.
#pragma unroll
      for (int a = 0; a < m; a++) {
        T e_hi, e_lo;
        if (FASTBAR && sizeof(T) == 8 && c.fast_barrier) {
          e_hi = t_exp_bounded(c.ctrl_q2 * (u[a] - c.u_max[a]));
          e_lo = c.ctrl_c[a] * t_rcp(e_hi);
        } else {
          e_hi = t_exp(c.ctrl_q2 * (u[a] - c.u_max[a]));
          e_lo = t_exp(c.ctrl_q2 * (-c.u_max[a] - u[a]));
        }
        T l = T(0);
        if constexpr (HASQR) {
#pragma unroll
          for (int bb = 0; bb < m; bb++) l += T(2) * c.R[a * m + bb] * u[bb];
        }
        lu[a] = l + (c.ctrl_q12 * e_hi - c.ctrl_q12 * e_lo);
        luu[a] = c.ctrl_q122 * e_hi +
                 c.ctrl_q122 * e_lo;
      }
#pragma unroll
      for (int a = 0; a < n; a++) {
        T l = T(0);
        if constexpr (HASQR) {
#pragma unroll
          for (int r = 0; r < n; r++) l += T(2) * c.Q[a * n + r] * (xp[r] - c.xtarget[r]);
        }
        lxq[a] = l;
      }
      if constexpr (!CK) {
#pragma unroll
        for (int i = 0; i < n; i++) xe[i] = xp[i];
        if (t >= D) {
#pragma unroll
          for (int i = 0; i < n; i++) xp[i] = at(X, rx(i, t - D));
#pragma unroll
          for (int a = 0; a < m; a++) u[a] = at(U, ru(a, t - D));
        }
      }
      }  // ROLE != 1
      if constexpr (ROLE == 2) {  // the helper's step ends here: record -> LDS slot t & 1, barrier
        lds_t* const r = rec + (size_t)(t & 1) * kRec * 64 + (threadIdx.x & 63);
#pragma unroll
        for (int q = 0; q < NV; q++) r[q * 64] = jv[q];
#pragma unroll
        for (int q = 0; q < 5; q++) r[(NV + q) * 64] = o[q];
#pragma unroll
        for (int a = 0; a < m; a++) {
          r[(NV + 5 + a) * 64] = lu[a];
          r[(NV + 5 + m + a) * 64] = luu[a];
        }
        if constexpr (HASQR) {
#pragma unroll
          for (int a = 0; a < n; a++) r[(NV + 5 + 2 * m + a) * 64] = lxq[a];
        }
        pair_barrier();
        return;
      }
      if constexpr (ROLE == 1) {  // the main wavefront's step starts here: the helper's record
        pair_barrier();
        const lds_t* const r = rec + (size_t)(t & 1) * kRec * 64 + (threadIdx.x & 63);
#pragma unroll
        for (int q = 0; q < NV; q++) jv[q] = r[q * 64];
#pragma unroll
        for (int q = 0; q < 5; q++) o[q] = r[(NV + q) * 64];
#pragma unroll
        for (int a = 0; a < m; a++) {
          lu[a] = r[(NV + 5 + a) * 64];
          luu[a] = r[(NV + 5 + m + a) * 64];
        }
#pragma unroll
        for (int a = 0; a < n; a++) lxq[a] = HASQR ? T(r[(NV + 5 + 2 * m + a) * 64]) : T(0);
      }

      STAMP_END(0);
      // Row by row: T1[a][:] = (F^T [Vxx|Vx])[a][:], then H[a][:] = L[a][:] + T1[a][:n] F
      T Qa[n][n + 1];   // [Qxx | Qx]
      T G[m][n + 1];    // [Qux | Qu]
      T Quu[m * m];
      static_for<0, W>([&](auto a_) {
        constexpr int a = decltype(a_)::value;
        T t1[n + 1];
#pragma unroll
        for (int j = 0; j <= n; j++) {
          T acc = T(0);
          bool first = true;
          static_for<0, n>([&](auto i_) {
            constexpr int i = decltype(i_)::value;
            const T vij = (!SYM || j == n || i <= j) ? Va[i][j] : Va[j][i];
            f_acc<i, a>(acc, first, vij, jv);
          });
          t1[j] = acc;
        }
        static_for<0, W>([&](auto b_) {
          constexpr int bcol = decltype(b_)::value;
          // Qxu is never used by the reference; with SYM neither are the lower triangles
          if constexpr (!(a < n && bcol >= n) && !(SYM && a < n && bcol < a) &&
                        !(SYM && a >= n && bcol >= n && bcol < a)) {
            T acc = T(0);
            bool first = true;
            static_for<0, n>([&](auto i_) {
              constexpr int i = decltype(i_)::value;
              f_acc<i, bcol>(acc, first, t1[i], jv);
            });
            T l = T(0);
            if constexpr (a < n) {
              if constexpr (HASQR) l = T(2) * c.Q[a * n + bcol];
              if constexpr (a < 2 && bcol < 2) l += o[2 + a + bcol];
              Qa[a][bcol] = l + acc;
            } else if constexpr (bcol < n) {
              G[a - n][bcol] = acc;
            } else {
              if constexpr (HASQR) l = T(2) * c.R[(a - n) * m + (bcol - n)];
              if constexpr (a == bcol) l += luu[a - n];
              Quu[(a - n) * m + (bcol - n)] = l + acc;
            }
          }
        });
        if constexpr (a < n) {
          T l = lxq[a];
          if constexpr (a < 2) l += o[a];
          Qa[a][n] = l + t1[n];
        } else {
          G[a - n][n] = lu[a - n] + t1[n];
        }
      });

      STAMP_END(1);
      // gains [K | k] = -Quu_inv [Qux | Qu]: control/iterative_ilqr.py:118-126
      T Qinv[m * m], Kk[m][n + 1];
      if constexpr (SYM) {
#pragma unroll
        for (int a = 1; a < m; a++)
#pragma unroll
          for (int bb = 0; bb < a; bb++) Quu[a * m + bb] = Quu[bb * m + a];
      }
      quu_inverse(Quu, lamb, Qinv);
#pragma unroll
      for (int a = 0; a < m; a++)
#pragma unroll
        for (int j = 0; j <= n; j++) {
          T acc = T(0);
#pragma unroll
          for (int bb = 0; bb < m; bb++) acc = t_fma(Qinv[a * m + bb], G[bb][j], acc);
          Kk[a][j] = -acc;
        }
      if (t == 0 && has_lds) {
#pragma unroll
        for (int a = 0; a < m; a++) lds_k0(a) = Kk[a][n];
        if (k0_out) {
#pragma unroll
          for (int a = 0; a < m; a++)
#pragma unroll
            for (int j = 0; j < n; j++) at(gK, rK(a, j, 0)) = Kk[a][j];
        }
      } else if (t >= 1 && t <= lds_steps) {
#pragma unroll
        for (int a = 0; a < m; a++)
#pragma unroll
          for (int j = 0; j <= n; j++) lds_gain(t, a * (n + 1) + j) = Kk[a][j];
      } else {
#pragma unroll
        for (int a = 0; a < m; a++) {
#pragma unroll
          for (int j = 0; j < n; j++) at(gK, rK(a, j, t)) = Kk[a][j];
          at(gk, ru(a, t)) = Kk[a][n];
        }
      }
      STAMP_END(2);
      // value update with the UNregularised Quu: control/iterative_ilqr.py:128-129
#pragma unroll
      for (int i = 0; i < n; i++) {
        T ktq[m];
#pragma unroll
        for (int bb = 0; bb < m; bb++) {
          T acc = T(0);
#pragma unroll
          for (int a = 0; a < m; a++) acc = t_fma(Kk[a][i], Quu[a * m + bb], acc);
          ktq[bb] = acc;
        }
#pragma unroll
        for (int j = (SYM ? i : 0); j <= n; j++) {
          T acc = T(0);
#pragma unroll
          for (int bb = 0; bb < m; bb++) acc = t_fma(ktq[bb], Kk[bb][j], acc);
          Va[i][j] = Qa[i][j] - acc;
        }
      }
      STAMP_END(3);
    };
    if constexpr (DEEP) {
      T xq[n], uq[m];  // the second buffer: x_{N-2}, u_{N-2}
      if (N >= 2 && ROLE != 1) {
#pragma unroll
        for (int i = 0; i < n; i++) xq[i] = at(X, rx(i, N - 2));
#pragma unroll
        for (int a = 0; a < m; a++) uq[a] = at(U, ru(a, N - 2));
      }
      int t = N - 1;
      for (; t >= 1; t -= 2) {
        body(t, xp, u);
        body(t - 1, xq, uq);
      }
      if (t == 0) body(0, xp, u);
    } else if constexpr (CK) {
      for (int sg = last_sg; sg >= 0; sg--) {
        const int len = sg == last_sg ? last_len : kSeg;
        if (sg != last_sg) roll_segment(sg, len);
        static_for<0, kSeg>([&](auto kk_) {
          constexpr int k = kSeg - 1 - decltype(kk_)::value;  // steps of the segment, last first
          if (k < len) {
#pragma unroll
            for (int i = 0; i < n; i++) {
              xe[i] = seg_at(k + 1, i);
              xp[i] = seg_at(k, i);
            }
#pragma unroll
            for (int a = 0; a < m; a++) u[a] = useg[k][a];
            body(sg * kSeg + k, xp, u);
          }
        });
      }
    } else {
      for (int t = N - 1; t >= 0; t--) body(t, xp, u);
    }
  }

  // -- backward pass, row-block form (plants with Sys::NBLK > 0: quad12) ------------------------
  // Same recursion (control/iterative_ilqr.py:88-130), organised for a plant whose unrolled step
  // does not fit a lane's registers otherwise (n = 12: [Vxx | Vx] and [Qxx | Qx] are 90 doubles
  // EACH, next to 52 of [Qux | Qu] and 37 Jacobian entries; a lane has 256).  With
  // A = M_0 M_1 .. M_{NBLK-1}, M_k = I + (rows of block k of E = A - I) (see Quad12::blk) and
  // Kc = -Quu_reg^-1 B^T Vxx (the gains before the state Jacobian is applied):
  //   G    = B^T [Vxx | Vx]                                      m x (n+1), column by column
  //   Quu  = l_uu + G[:, :n] B,   Qu = l_u + G[:, n]
  //   [Kc | k] = -Quu_reg^-1 [G[:, :n] | Qu]                     columns -> LDS
  //   [W | w]  = [Vxx | Vx] - Kc^T (Quu [Kc | k])                in place over V, row by row of Quu Kc
  //   K    = Kc A = Kc M_0 M_1 ...                               row by row from LDS, to HBM
  //   Vxx' = l_xx + A^T W A = .. M_1^T (M_0^T W M_0) M_1 ..      in place, upper triangle
  //   Vx'  = l_x + A^T w    = .. M_1^T (M_0^T w)                 in place
  // which is the reference's Qux = B^T Vxx A, K = -Quu_reg^-1 Qux, Vxx' = Qxx - K^T Quu K with the
  // products associated differently (K^T Quu K = A^T (Kc^T Quu Kc) A): agreement with the other
  // kernel families and the oracle is to round-off, not bit for bit.  Nothing of size n x n lives
  // next to V, and the gains are stored and dead before the state blocks are applied.
  // LDS words per wavefront of backward_blocked: the step's [Kc | k] and, parked there between
  // their producer and their (late) consumers instead of occupying registers through the sweep:
  // the obstacle terms o[5], l_u[m], the sin / cos values, the Sys::NJX state entries and u[m]
  // that the second evaluation of the Jacobian needs
  static constexpr int kParkO = m * (n + 1), kParkLu = kParkO + 5, kParkTr = kParkLu + m,
                       kParkJx = kParkTr + NT, kParkU = kParkJx + Sys::NJX,
                       kStepSlots = kParkU + m;
  // ... and, for the whole pass, the five obstacle parameters the hot loop reads every step (centre,
  // speed, 1 / width^2, 1 / height^2): per-lane loop invariants that the register allocator
  // otherwise sends to SCRATCH — three of them were reloaded at the top of every horizon step, each
  // behind an s_waitcnt vmcnt(0) that also drained the previous step's 52 gain stores
  static constexpr int kParkOb = kStepSlots;
  static constexpr int kGainWords = (kStepSlots + 5) * 64;
  static constexpr bool e_nz(int i, int j) {
    return i == j ? Sys::pat(i, j) != 1 : Sys::pat(i, j) != 0;
  }
  static constexpr bool blocks_valid() {  // E[i][j] != 0 only where blk(j) <= blk(i)
    for (int i = 0; i < n; i++)
      for (int j = 0; j < n; j++)
        if (e_nz(i, j) && Sys::blk(j) > Sys::blk(i)) return false;
    return true;
  }
  // does some row of block b reach column cc?
  static constexpr bool in_s(int b, int cc) {
    for (int r = 0; r < n; r++)
      if (Sys::blk(r) == b && e_nz(r, cc)) return true;
    return false;
  }
  // acc += E[i][j] * v
  template <int i, int j> __device__ __forceinline__ void e_acc(T& acc, T v, const T (&jv)[NV]) const {
    if constexpr (!e_nz(i, j)) {
      return;
    } else if constexpr (i == j) {  // a varying diagonal entry of A: E = A - 1
      static_assert(Sys::pat(i, j) >= 3 && Sys::pat(i, j) < 100, "diagonal of A: one or varying");
      acc = t_fma(jv[Sys::pat(i, j) - 3] - T(1), v, acc);
    } else if constexpr (Sys::pat(i, j) == 1) {
      acc = acc + v;
    } else {
      acc = t_fma(f_entry<i, j>(jv), v, acc);
    }
  }
  // is row i of B = F[:, n:] structurally non-zero?
  static constexpr bool b_row(int i) {
    for (int a = 0; a < m; a++)
      if (Sys::pat(i, n + a) != 0) return true;
    return false;
  }
  // g <- g M_b (one row of the gains)
  template <int b>
  __device__ __forceinline__ void block_gain_row(T (&g)[n], const T (&jv)[NV]) const {
    T go[n];
#pragma unroll
    for (int i = 0; i < n; i++) go[i] = g[i];
    static_for<0, n>([&](auto c_) {
      constexpr int cc = decltype(c_)::value;
      if constexpr (in_s(b, cc)) {
        T acc = go[cc];
        static_for<0, n>([&](auto r_) {
          constexpr int r = decltype(r_)::value;
          if constexpr (Sys::blk(r) == b) e_acc<r, cc>(acc, go[r], jv);
        });
        g[cc] = acc;
      }
    });
  }
  // V <- M_b^T V M_b (upper triangle), vx <- M_b^T vx.  With rho_r = row r of E and
  // z_r = V[r][:] + sum_r' (V[r][r'] / 2) rho_r' over the rows r, r' of the block:
  //   M_b^T V M_b = V + sum_r (rho_r z_r^T + z_r rho_r^T)
  // — a symmetric rank-2 update per row; the only temporaries are the z_r.
  template <int b>
  __device__ __forceinline__ void block_value(T (&V)[n][n], T (&vx)[n], const T (&jv)[NV]) const {
    T z[n][n];  // rows of the block only
    static_for<0, n>([&](auto r_) {
      constexpr int r = decltype(r_)::value;
      if constexpr (Sys::blk(r) == b) {
        static_for<0, n>([&](auto c_) {
          constexpr int cc = decltype(c_)::value;
          T acc = r <= cc ? V[r][cc] : V[cc][r];
          if constexpr (in_s(b, cc)) {
            static_for<0, n>([&](auto q_) {
              constexpr int q = decltype(q_)::value;
              if constexpr (Sys::blk(q) == b && e_nz(q, cc))
                e_acc<q, cc>(acc, T(0.5) * (r <= q ? V[r][q] : V[q][r]), jv);
            });
          }
          z[r][cc] = acc;
        });
      }
    });
    T vo[n];
#pragma unroll
    for (int i = 0; i < n; i++) vo[i] = vx[i];
    static_for<0, n>([&](auto x_) {
      constexpr int x = decltype(x_)::value;
      static_for<x, n>([&](auto c_) {
        constexpr int cc = decltype(c_)::value;
        if constexpr (in_s(b, x) || in_s(b, cc)) {
          T acc = V[x][cc];
          static_for<0, n>([&](auto r_) {
            constexpr int r = decltype(r_)::value;
            if constexpr (Sys::blk(r) == b) {
              e_acc<r, x>(acc, z[r][cc], jv);
              e_acc<r, cc>(acc, z[r][x], jv);
            }
          });
          V[x][cc] = acc;
        }
      });
      if constexpr (in_s(b, x)) {
        T acc = vo[x];
        static_for<0, n>([&](auto r_) {
          constexpr int r = decltype(r_)::value;
          if constexpr (Sys::blk(r) == b) e_acc<r, x>(acc, vo[r], jv);
        });
        vx[x] = acc;
      }
    });
  }

  // Every gain goes to HBM (no LDS-resident steps: one step of this plant's gains is 26 KB).
  // GENERAL = false: the hot form — no branch inside the horizon loop (short sin / cos kernels,
  // the positive-definite Quu inverse, the one-exp input barrier, the obstacle term computed for
  // every lane and selected); returns true if some lane needed a general form (argument of a
  // sin / cos out of range, Quu not positive definite), in which case the caller repeats the pass
  // with GENERAL = true (same protocol as the other kernel families: the general forms' cold code
  // — library sincos, Jacobi sweeps — stays out of the loop the register allocator has to fit).
  template <bool FASTBAR, bool GENERAL = true>
  __device__ __forceinline__ bool backward_blocked(const T* X, const T* U, const T (&xT)[n],
                                                   const T (&ob)[6], T lamb, T* gK, T* gk,
                                                   T* lds_gains) const {
    bool bad = false;
    // the step's [Kc | k] in this wavefront's LDS slice (kGainWords words), typed as LDS (ds_*
    // with immediate offsets).  Stores index with the lane, loads with a copy of it whose origin
    // is hidden: the optimiser cannot prove that a load reads what a store wrote and forward the
    // value — i.e. keep the 52 doubles in registers after all — but orders them as it must.
    typedef __attribute__((address_space(3))) T lds_word;
    lds_word* const kcs = (lds_word*)lds_gains;
    const unsigned l64 = threadIdx.x & 63;
    unsigned lrd = l64;
    asm volatile("" : "+v"(lrd));
    static_assert(blocks_valid(), "Sys::blk does not factor A = I + E into row blocks");
    const T ob_pa = T(1) / (ob[2] * ob[2]), ob_pb = T(1) / (ob[3] * ob[3]);
    if constexpr (!GENERAL && sizeof(T) == 8) {
      kcs[(kParkOb + 0) * 64 + l64] = ob[0];
      kcs[(kParkOb + 1) * 64 + l64] = ob[1];
      kcs[(kParkOb + 2) * 64 + l64] = ob[4];
      kcs[(kParkOb + 3) * 64 + l64] = ob_pa;
      kcs[(kParkOb + 4) * 64 + l64] = ob_pb;
    }
    T V[n][n], vx[n];  // Vxx (upper triangle live), Vx
    {
      // get_cost_final(): control/ilqr_helper.py:106-150
      T xN[n], o[5];
      for_rows<n>(X, rx(0, N), [&](auto i_, const T& w) { xN[decltype(i_)::value] = w; });
      obstacle_sel<GENERAL>(ob, ob_pa, ob_pb, xN[0], xN[1], N, o);
#pragma unroll
      for (int i = 0; i < n; i++) {
        T acc = T(0);
#pragma unroll
        for (int r = 0; r < n; r++) {
          V[i][r] = T(2) * c.Qt[i * n + r];
          acc += T(2) * c.Qt[i * n + r] * (xN[r] - xT[r]);
        }
        vx[i] = acc;
      }
      V[0][0] += o[2]; V[0][1] += o[3]; V[1][1] += o[4];
      vx[0] += o[0]; vx[1] += o[1];
    }
    // Register budget of the step (a lane has 256 doubles, 90 of them hold [Vxx | Vx]): the inputs
    // of step t-1 are loaded late in step t (before the state blocks, ~7000 cycles ahead of their
    // first use) instead of at its top, and only what the step's first phases need stays live
    // through them: of x_t the position (obstacle term; all of it with stage weights), of the
    // Jacobian the B entries — the 25 varying A entries are formed again from the kept sin / cos
    // values, rates and thrust right before the phases that use them.
    constexpr int NP = HASQR ? n : 2;
    T xe[n], xp[NP], u[m];  // x_{t+1}, (leading entries of) x_t, u_t
    for_rows<n>(X, rx(0, N), [&](auto i_, const T& w) { xe[decltype(i_)::value] = w; });
    for_rows<NP>(X, rx(0, N - 1), [&](auto i_, const T& w) { xp[decltype(i_)::value] = w; });
    for_rows<m>(U, ru(0, N - 1), [&](auto a_, const T& w) { u[decltype(a_)::value] = w; });
    STAMP_BEGIN();
    for (int t = N - 1; t >= 0; t--) {
      T jv[NV], o[5], tr[NT];
      STAMP_END(6);
      if constexpr (GENERAL) Sys::trig(xe, tr);
      else Sys::trig_s(xe, tr, &bad);
      {
        T jb[NV];
        Sys::jac_var(c, xe, u, tr, jb);
#pragma unroll
        for (int q = 0; q < NV; q++) jv[q] = jb[q];  // only the B entries are used before `refresh`
      }

      T lu[m], luu[m];  // input barrier: control/ilqr_helper.py:83-103 (see backward())
      T ehi[m];
      if constexpr (!GENERAL && sizeof(T) == 8) {
        // hot form: the obstacle term without a branch (obstacle_sel) and the step's m + 1
        // exponentials evaluated together, their literals shared in scalar registers (t_exp_d)
        const bool has = ob[5] >= T(0);
        const int opt = has ? (int)ob[5] : 0;
        asm volatile("" : "+v"(lrd));  // (this step's reads: not the previous step's values)
        const T ob0 = kcs[(kParkOb + 0) * 64 + lrd], ob1 = kcs[(kParkOb + 1) * 64 + lrd],
                ob4 = kcs[(kParkOb + 2) * 64 + lrd], pa = kcs[(kParkOb + 3) * 64 + lrd],
                pb = kcs[(kParkOb + 4) * 64 + lrd];
        const T dy = opt == 1 ? xp[1] - (ob1 + T(t) * ob4) : xp[1] - ob1;
        const T dz = opt == 2 ? xp[0] - (ob0 - T(t) * ob4) : xp[0] - ob0;
        const T h = T(1) + c.safety_margin - (dz * pa * dz + dy * pb * dy);
        const T hd0 = T(-2) * pa * dz, hd1 = T(-2) * pb * dy;
        T ea[m + 1], ee[m + 1];
        ea[0] = c.obs_q2 * h;
#pragma unroll
        for (int a = 0; a < m; a++) ea[1 + a] = c.ctrl_q2 * (u[a] - c.u_max[a]);
        t_exp_d<true, m + 1, m>(ea, ee);
        const T c1 = c.obs_q12 * ee[0], c2 = c.obs_q122 * ee[0];
        o[0] = has ? c1 * hd0 : T(0);
        o[1] = has ? c1 * hd1 : T(0);
        o[2] = has ? c2 * (hd0 * hd0) : T(0);
        o[3] = has ? c2 * (hd0 * hd1) : T(0);
        o[4] = has ? c2 * (hd1 * hd1) : T(0);
#pragma unroll
        for (int a = 0; a < m; a++) ehi[a] = ee[1 + a];
      } else {
        obstacle_sel<GENERAL>(ob, ob_pa, ob_pb, xp[0], xp[1], t, o);
      }
#pragma unroll
      for (int a = 0; a < m; a++) {
        T e_hi, e_lo;
        if constexpr (!GENERAL && sizeof(T) == 8) {
          e_hi = ehi[a];
          e_lo = c.ctrl_c[a] * t_rcp(e_hi);
        } else if (!GENERAL || (FASTBAR && sizeof(T) == 8 && c.fast_barrier)) {
          e_hi = t_exp_bounded(c.ctrl_q2 * (u[a] - c.u_max[a]));
          e_lo = c.ctrl_c[a] * t_rcp(e_hi);
        } else {
          e_hi = t_exp(c.ctrl_q2 * (u[a] - c.u_max[a]));
          e_lo = t_exp(c.ctrl_q2 * (-c.u_max[a] - u[a]));
        }
        T l = T(0);
        if constexpr (HASQR) {
#pragma unroll
          for (int bb = 0; bb < m; bb++) l += T(2) * c.R[a * m + bb] * u[bb];
        }
        lu[a] = l + (c.ctrl_q12 * e_hi - c.ctrl_q12 * e_lo);
        luu[a] = c.ctrl_q122 * e_hi +
                 c.ctrl_q122 * e_lo;
      }
      T lxq[n];  // 2Q dX[:, t]: control/ilqr_helper.py:29
#pragma unroll
      for (int a = 0; a < n; a++) {
        T l = T(0);
        if constexpr (HASQR) {
#pragma unroll
          for (int r = 0; r < n; r++) l += T(2) * c.Q[a * n + r] * (xp[r] - c.xtarget[r]);
        }
        lxq[a] = l;
      }
      // park what only the end of the step reads (see kStepSlots)
#pragma unroll
      for (int q = 0; q < 5; q++) kcs[(kParkO + q) * 64 + l64] = o[q];
#pragma unroll
      for (int a = 0; a < m; a++) kcs[(kParkLu + a) * 64 + l64] = lu[a];
#pragma unroll
      for (int q = 0; q < NT; q++) kcs[(kParkTr + q) * 64 + l64] = tr[q];
#pragma unroll
      for (int q = 0; q < Sys::NJX; q++) kcs[(kParkJx + q) * 64 + l64] = xe[Sys::jx(q)];
#pragma unroll
      for (int a = 0; a < m; a++) kcs[(kParkU + a) * 64 + l64] = u[a];
      // ---- gains.  [Kc | k] never lives in registers as a whole (52 doubles next to the 90 of
      // [Vxx | Vx]): its columns go to this wavefront's LDS slice as they are formed and come back
      // in the order the two consumers want them — word (a, j) of lane l at kcs[(a (n+1) + j) 64 + l],
      // every lane reads only what it wrote (no barrier), consecutive lanes on consecutive words.
      I2LQR_PHASE_FENCE();
      STAMP_END(0);
      if constexpr (!GENERAL && I2LQR_WARM_INPUTS) {
        // the rows the end of this step loads for step t-1, on their way to the L2 a whole gain
        // sweep earlier: under a full chip the loads themselves came back after the next step's
        // top had waited 2000 cycles for them and its first phase another 2400
        const int tl = t >= 1 ? t : 1;
        warm_rows<n>(X, rx(0, tl));
        warm_rows<NP>(X, rx(0, tl - 1));
        warm_rows<m>(U, ru(0, tl - 1));
      }
      auto gcol = [&](auto j_, T (&g)[m]) __attribute__((always_inline)) {  // column j of B^T [Vxx | Vx]
        constexpr int j = decltype(j_)::value;
        static_for<0, m>([&](auto a_) {
          constexpr int a = decltype(a_)::value;
          T acc = T(0);
          bool first = true;
          static_for<0, n>([&](auto i_) {
            constexpr int i = decltype(i_)::value;
            T vij;
            if constexpr (j == n) vij = vx[i];
            else vij = i <= j ? V[i][j] : V[j][i];
            f_acc<i, n + a>(acc, first, vij, jv);
          });
          g[a] = acc;
        });
      };
      // Quu = l_uu + (B^T Vxx) B, from the columns of B^T Vxx that the rows of B reach
      T Quu[m * m];
      {
        T Gb[n][m];  // columns i of B^T Vxx with a non-zero row i of B
        static_for<0, n>([&](auto i_) {
          constexpr int i = decltype(i_)::value;
          if constexpr (b_row(i)) gcol(i_, Gb[i]);
        });
        static_for<0, m>([&](auto a_) {
          constexpr int a = decltype(a_)::value;
          static_for<a, m>([&](auto b_) {
            constexpr int bb = decltype(b_)::value;
            T acc = T(0);
            bool first = true;
            static_for<0, n>([&](auto i_) {
              constexpr int i = decltype(i_)::value;
              if constexpr (b_row(i)) f_acc<i, n + bb>(acc, first, Gb[i][a], jv);
            });
            T l = T(0);
            if constexpr (HASQR) l = T(2) * c.R[a * m + bb];
            if constexpr (a == bb) l += luu[a];
            Quu[a * m + bb] = l + acc;
            Quu[bb * m + a] = l + acc;
          });
        });
      }
      // [Kc | k] = -Quu_reg^-1 [B^T Vxx | Qu], column by column: control/iterative_ilqr.py:118-126
      T Qinv[m * m];
      if constexpr (GENERAL) quu_inverse(Quu, lamb, Qinv);
      else t_quu_inverse_m<T, m, false>(Quu, lamb, Qinv, &bad);
      static_for<0, n + 1>([&](auto j_) {
        constexpr int j = decltype(j_)::value;
        T g[m];
        gcol(j_, g);
        if constexpr (j == n) {
#pragma unroll
          for (int a = 0; a < m; a++) g[a] = kcs[(kParkLu + a) * 64 + lrd] + g[a];  // Qu = l_u + B^T Vx
        }
#pragma unroll
        for (int a = 0; a < m; a++) {
          T acc = T(0);
#pragma unroll
          for (int bb = 0; bb < m; bb++) acc = t_fma(Qinv[a * m + bb], g[bb], acc);
          I2LQR_DBG_CHECK(c.trap, TAG_LANE_LDS, (a * (n + 1) + j) * 64 + l64, kGainWords);
          kcs[(a * (n + 1) + j) * 64 + l64] = -acc;
        }
      });
      // [W | w] = [Vxx | Vx] - Kc^T (Quu [Kc | k]), the UNregularised Quu
      // (control/iterative_ilqr.py:128-129 before the state Jacobian is applied):
      // [W | w][i][j] -= sum_b Y[b][i] [Kc | k][b][j] with Y = Quu Kc.  ONE sweep, row by row
      // (round 5): the four rows of [Kc | k] come back from LDS once and stay in registers (52
      // doubles), the column Y[:, i] is formed when row i is updated (4 doubles live), every word
      // of [W | w] is touched ONCE per step.  Rounds 3-4 swept twice (two rows of Y per sweep, Kc read
      // from LDS twice): every word of the part of V that lives in accumulation registers — a lane
      // has 128 doubles of vector registers, 90 of them are V — then travelled to the vector
      // registers and back twice per step.  Same operations on every word in the same order (b = 0,
      // 1, 2, 3; Y accumulated over a = 0 .. 3 from zero): bit-identical.  361 -> 268 accumulation-
      // register moves and 127 -> 103 LDS operations per step, 72.7-73.4 -> 75.8-76.8 M it/s at 65536
      // problems (tools/ab_bench.py, same process).  Keeping [Kc | k] in registers from its
      // formation instead of reloading it (no LDS reads at all) spills: 62.5 M it/s.
      I2LQR_PHASE_FENCE();
      {
        T kr[m][n + 1];
        asm volatile("" : "+v"(lrd));  // this step's words (no forwarding from the stores above)
#pragma unroll
        for (int a = 0; a < m; a++)
#pragma unroll
          for (int j = 0; j <= n; j++) kr[a][j] = kcs[(a * (n + 1) + j) * 64 + lrd];
        I2LQR_PHASE_FENCE();
        static_for<0, n>([&](auto i_) {
          constexpr int i = decltype(i_)::value;
          T y[m];
#pragma unroll
          for (int bb = 0; bb < m; bb++) {
            T acc = T(0);
#pragma unroll
            for (int a = 0; a < m; a++) acc = t_fma(Quu[bb * m + a], kr[a][i], acc);
            y[bb] = acc;
          }
#pragma unroll
          for (int j = i; j < n; j++) {
#pragma unroll
            for (int bb = 0; bb < m; bb++) V[i][j] = t_fma(-y[bb], kr[bb][j], V[i][j]);
          }
#pragma unroll
          for (int bb = 0; bb < m; bb++) vx[i] = t_fma(-y[bb], kr[bb][n], vx[i]);
        });
      }
      // K = Kc A row by row, then straight to HBM
      I2LQR_PHASE_FENCE();
      STAMP_END(1);
      // inputs of step t-1: x_t in full (its evaluation state), x_{t-1}, u_{t-1}.  Issued first in
      // this phase and LANDED (an explicit vmcnt(0), below) before the first of the step's 52 gain
      // stores: the compiler treats pending loads and stores as unordered among each other, so a
      // wait for these loads behind the stores was a wait for all of them to be written —
      // s_waitcnt vmcnt(0) in the middle of the store sequence.  With the rows warm in the L2
      // (warm_rows) they arrive while the Jacobian and the first gain row are formed, and nothing
      // in the loop waits on vector memory after that: the stores drain behind the state blocks
      // and the next step's first phase.  (Step 0 loads the rows of step 1 again: no branch.)
      {
        const int tl = t >= 1 ? t : 1;
        for_rows<n>(X, rx(0, tl), [&](auto i_, const T& w) { xe[decltype(i_)::value] = w; });
        for_rows<NP>(X, rx(0, tl - 1), [&](auto i_, const T& w) { xp[decltype(i_)::value] = w; });
        for_rows<m>(U, ru(0, tl - 1), [&](auto a_, const T& w) { u[decltype(a_)::value] = w; });
      }
      I2LQR_PHASE_FENCE();
      {  // the A entries of the Jacobian, formed again (hidden from value numbering: the compiler
         // would otherwise keep the first evaluation's 25 doubles live through the phases above)
        T xr[n], ur[m], trr[NT];
        asm volatile("" : "+v"(lrd));
#pragma unroll
        for (int i = 0; i < n; i++) xr[i] = T(0);
#pragma unroll
        for (int q = 0; q < Sys::NJX; q++) xr[Sys::jx(q)] = kcs[(kParkJx + q) * 64 + lrd];
#pragma unroll
        for (int a = 0; a < m; a++) ur[a] = kcs[(kParkU + a) * 64 + lrd];
#pragma unroll
        for (int q = 0; q < NT; q++) trr[q] = kcs[(kParkTr + q) * 64 + lrd];
        Sys::jac_var(c, xr, ur, trr, jv);
      }
      static_for<0, m>([&](auto a_) {
        constexpr int a = decltype(a_)::value;
        T g[n];
        asm volatile("" : "+v"(lrd));
#pragma unroll
        for (int j = 0; j < n; j++) g[j] = kcs[(a * (n + 1) + j) * 64 + lrd];
        static_for<0, Sys::NBLK>([&](auto b_) { block_gain_row<decltype(b_)::value>(g, jv); });
        if constexpr (a == 0) {
          I2LQR_PHASE_FENCE();
          // vmcnt(0) in the gfx9 s_waitcnt encoding (vmcnt = bits 3:0 and 15:14, expcnt 6:4 and
          // lgkmcnt 11:8 left at their maxima): the inputs of step t-1 have landed
          static_assert(kGfx9Waitcnt, "s_waitcnt literal below is the gfx9 field layout");
          __builtin_amdgcn_s_waitcnt(0x0F70);
          I2LQR_PHASE_FENCE();
        }
        for_rows<n, kRowsNT>(gK, rK(a, 0, t), [&](auto j_, T& w) { w = g[decltype(j_)::value]; });
      });
      for_rows<m, kRowsNT>(gk, ru(0, t), [&](auto a_, T& w) {
        w = kcs[(decltype(a_)::value * (n + 1) + n) * 64 + lrd];
      });
      STAMP_END(2);
      // [Vxx' | Vx'] = [l_xx | l_x] + A^T [W A | w], block after block
      static_for<0, Sys::NBLK>([&](auto b_) {
        I2LQR_PHASE_FENCE();
        block_value<decltype(b_)::value>(V, vx, jv);
      });
#pragma unroll
      for (int i = 0; i < n; i++) {
        if constexpr (HASQR) {
#pragma unroll
          for (int j = i; j < n; j++) V[i][j] = T(2) * c.Q[i * n + j] + V[i][j];
        }
        vx[i] = lxq[i] + vx[i];
      }
      {
        asm volatile("" : "+v"(lrd));
        T op[5];
#pragma unroll
        for (int q = 0; q < 5; q++) op[q] = kcs[(kParkO + q) * 64 + lrd];
        V[0][0] += op[2]; V[0][1] += op[3]; V[1][1] += op[4];
        vx[0] += op[0]; vx[1] += op[1];
      }
      STAMP_END(3);
    }
    return bad;
  }

  // -- forward pass: control/iterative_ilqr.py:133-160 ----------------------------------------
  // REROLL: the nominal states x_t that the feedback law needs are re-rolled from the nominal
  // inputs next to the candidate (bit-identical to the stored X: same code, same inputs) instead
  // of being read back — n (N+1) words less HBM traffic per iteration for one more dynamics step
  // per horizon step.  Pays where the kernel sits on the HBM roof (fp64).
  // WRITEX = false: the candidate states are not stored at all (the nominal ones stay intact); the
  // caller re-rolls them from the candidate inputs if the step is accepted (restore_states()).
  // D: horizon steps between the loads of a step's inputs and their use (D register sets that take
  // turns).  A step is ~700 cycles of arithmetic and a load from HBM / the far L2 takes 1500-2000:
  // with D = 1 the pass runs at the memory latency, not at the issue rate.
  template <bool REROLL, bool WRITEX = true, int D = (DEEP ? 2 : 1)>
  __device__ __forceinline__ T forward(const T* X, const T* U, const T* gK, const T* gk, T* Xn,
                                       T* Un, const T (&xT)[n]) const {
    T x[n], u[m], xn[n], tr[NT];
    T xo[n];
#pragma unroll
    for (int i = 0; i < n; i++) {
      x[i] = at(X, rx(i, 0));
      xo[i] = x[i];
      if (WRITEX && Xn != X) at(Xn, rx(i, 0)) = x[i];
    }
    T cost = T(0);
    // The nominal state / input / gains of a step are consumed at its very start (the feedback
    // law); the loads for step t+D are issued right after, into the same registers, so the HBM
    // latency hides under the rest of the serial step(s).
    auto load_step = [&](int t, T (&xl)[n], T (&ul)[m], T (&kl)[m][n + 1])
        __attribute__((always_inline)) {
      if constexpr (!REROLL) {
#pragma unroll
        for (int j = 0; j < n; j++) xl[j] = at(X, rx(j, t));
      }
#pragma unroll
      for (int a = 0; a < m; a++) ul[a] = at(U, ru(a, t));
      if (t == 0 && has_lds) {  // x_0 is common to the nominal and the candidate: K_0 multiplies zeros
#pragma unroll
        for (int a = 0; a < m; a++) {
#pragma unroll
          for (int j = 0; j < n; j++) kl[a][j] = T(0);
          kl[a][n] = lds_k0(a);
        }
      } else if (t >= 1 && t <= lds_steps) {
#pragma unroll
        for (int a = 0; a < m; a++)
#pragma unroll
          for (int j = 0; j <= n; j++) kl[a][j] = lds_gain(t, a * (n + 1) + j);
      } else {
#pragma unroll
        for (int a = 0; a < m; a++) {
#pragma unroll
          for (int j = 0; j < n; j++) kl[a][j] = at(gK, rK(a, j, t));
          kl[a][n] = at(gk, ru(a, t));
        }
      }
    };
    // one horizon step on the register set (xl, ul, kl); with REROLL the nominal state is the
    // loop-carried xo, otherwise it is the loaded xl
    auto body = [&](const int t, T (&xl)[n], T (&ul)[m], T (&kl)[m][n + 1])
        __attribute__((always_inline)) {
      T uon[m];
#pragma unroll
      for (int a = 0; a < m; a++) {
        T acc = T(0);
#pragma unroll
        for (int j = 0; j < n; j++)
          acc = t_fma(kl[a][j], x[j] - (REROLL ? xo[j] : xl[j]), acc);
        u[a] = clip(ul[a] + kl[a][n] + acc, -c.u_max[a], c.u_max[a]);
        uon[a] = ul[a];
      }
      // all memory operations of the step are issued here, right after the wait for its inputs:
      // the next wait drains the shared load / store counter, and by then they are a step old
      if (t + D < N) load_step(t + D, xl, ul, kl);
#pragma unroll
      for (int a = 0; a < m; a++) at(Un, ru(a, t)) = u[a];
      if constexpr (REROLL) {  // nominal state of step t+1
        Sys::trig(xo, tr);
        Sys::step_tr(c, xo, uon, tr, xn);
#pragma unroll
        for (int i = 0; i < n; i++) xo[i] = xn[i];
      }
      Sys::trig(x, tr);
      Sys::step_tr(c, x, u, tr, xn);
      if constexpr (WRITEX) {
#pragma unroll
        for (int i = 0; i < n; i++) at(Xn, rx(i, t + 1)) = xn[i];
      }
      cost = cost + stage_cost(x, xT, u);
#pragma unroll
      for (int i = 0; i < n; i++) x[i] = xn[i];
    };
    T xl[D][n], ul[D][m], kl[D][m][n + 1];
#pragma unroll
    for (int i = 0; i < n; i++) xl[0][i] = xo[i];
#pragma unroll
    for (int d = 0; d < D; d++)
      if (d < N) load_step(d, xl[d], ul[d], kl[d]);
    if constexpr (D > 1) {
      int t = 0;
      for (; t + D <= N; t += D) {
#pragma unroll
        for (int d = 0; d < D; d++) body(t + d, xl[d], ul[d], kl[d]);
      }
#pragma unroll
      for (int d = 0; d < D - 1; d++)
        if (t + d < N) body(t + d, xl[d], ul[d], kl[d]);
    } else {
      for (int t = 0; t < N; t++) body(t, xl[0], ul[0], kl[0]);
    }
    cost = cost + terminal_cost(x, xT);
    return cost;
  }

  // -- forward pass of the row-block plants: control/iterative_ilqr.py:133-160 ------------------
  // The deferred, re-rolling form of forward(): the nominal states the feedback law needs are
  // re-rolled from the nominal inputs next to the candidate (bit-identical to the stored X: same
  // code, same inputs), the candidate inputs go to Un, no candidate state is stored (an accepted
  // step re-rolls them: restore_rows()).  Same arithmetic, word for word, as forward<true, false>;
  // the step's 56 words are loaded two steps ahead through row groups.
  template <bool GENERAL>
  __device__ __forceinline__ T forward_rows(const T* X, const T* U, const T* gK, const T* gk, T* Un,
                                            const T (&xT)[n], bool* bad) const {
    T x[n], u[m], xn[n], tr[NT];
    T xo[n];  // nominal state (re-rolled)
    // Nominal inputs and gains of a step: two register sets take turns, each re-loaded for step
    // t + 2 right after step t has consumed it.  A step is ~550 instructions (~1 us) — less than
    // an HBM round trip under load (~3 us with every wavefront streaming): with a one-step
    // distance every step waited ~8000 cycles for its 56 words.
    struct Buf { T ul[m], kl[m][n + 1]; };
    constexpr int D = I2LQR_FWD_DEPTH;  // register sets = prefetch distance in horizon steps
    Buf q[D];
    for_rows<n>(X, rx(0, 0), [&](auto i_, const T& w) { x[decltype(i_)::value] = w; });
#pragma unroll
    for (int i = 0; i < n; i++) xo[i] = x[i];
    auto load_step = [&](const int t, Buf& b) __attribute__((always_inline)) {
      for_rows<m>(U, ru(0, t), [&](auto a_, const T& w) { b.ul[decltype(a_)::value] = w; });
      for_rows<m * n, kRowsNT>(gK, rK(0, 0, t), [&](auto e_, const T& w) {
        constexpr int e = decltype(e_)::value;
        b.kl[e / n][e % n] = w;
      });
      for_rows<m, kRowsNT>(gk, ru(0, t), [&](auto a_, const T& w) { b.kl[decltype(a_)::value][n] = w; });
    };
    // step 0: x_0 is common to the nominal and the candidate, K_0 multiplies zeros and is not read
    for_rows<m>(U, ru(0, 0), [&](auto a_, const T& w) { q[0].ul[decltype(a_)::value] = w; });
    for_rows<m>(gk, ru(0, 0), [&](auto a_, const T& w) { q[0].kl[decltype(a_)::value][n] = w; });
#pragma unroll
    for (int a = 0; a < m; a++)
#pragma unroll
      for (int j = 0; j < n; j++) q[0].kl[a][j] = T(0);
    static_for<1, D>([&](auto d_) {
      constexpr int d = decltype(d_)::value;
      load_step(d < N ? d : N - 1, q[d]);
    });
    T cost = T(0);
    auto body = [&](const int t, Buf& b) __attribute__((always_inline)) {
      T uo[m];
#pragma unroll
      for (int a = 0; a < m; a++) {
        T acc = T(0);
#pragma unroll
        for (int j = 0; j < n; j++) acc = t_fma(b.kl[a][j], x[j] - xo[j], acc);
        u[a] = clip(b.ul[a] + b.kl[a][n] + acc, -c.u_max[a], c.u_max[a]);
        uo[a] = b.ul[a];
      }
      load_step(t + D < N ? t + D : N - 1, b);  // (the last steps load a row again: no branch)
      for_rows<m>(Un, ru(0, t), [&](auto a_, T& w) { w = u[decltype(a_)::value]; });
      // nominal state of step t + 1, re-rolled from the nominal inputs (the pass streams the gains
      // at the HBM rate and has issue slots to spare: n fewer rows to read per step)
      if constexpr (GENERAL) Sys::trig(xo, tr);
      else Sys::trig_s(xo, tr, bad);
      Sys::step_tr(c, xo, uo, tr, xn);
#pragma unroll
      for (int i = 0; i < n; i++) xo[i] = xn[i];
      if constexpr (GENERAL) Sys::trig(x, tr);
      else Sys::trig_s(x, tr, bad);
      Sys::step_tr(c, x, u, tr, xn);
      cost = cost + stage_cost(x, xT, u);
#pragma unroll
      for (int i = 0; i < n; i++) x[i] = xn[i];
    };
    int t = 0;
    for (; t + D <= N; t += D)
      static_for<0, D>([&](auto d_) { body(t + decltype(d_)::value, q[decltype(d_)::value]); });
    static_for<0, D - 1>([&](auto d_) {
      constexpr int d = decltype(d_)::value;
      if (t + d < N) body(t + d, q[d]);
    });
    cost = cost + terminal_cost(x, xT);
    return cost;
  }

  // Re-roll X[:, 1..N] from the inputs (restore_states_paired through row groups).  MERGE: the
  // lanes that accepted take their inputs from the candidate buffer Un, every lane writes its
  // current inputs back to U (one buffer with full rows for the whole wavefront).
  // Running it twice gives the same result (the repeat with GENERAL = true relies on it).
  template <bool MERGE, bool GENERAL>
  __device__ __forceinline__ bool restore_rows(T* X, T* U, const T* Un = nullptr,
                                               bool acc = false) const {
    bool bad = false;
    T x[n], xn[n], tr[NT], u[m], ul[m], uc[m];
    for_rows<n>(X, rx(0, 0), [&](auto i_, const T& w) { x[decltype(i_)::value] = w; });
    auto load_u = [&](const int t) __attribute__((always_inline)) {
      for_rows<m>(U, ru(0, t), [&](auto a_, const T& w) { ul[decltype(a_)::value] = w; });
      if constexpr (MERGE)
        for_rows<m>(Un, ru(0, t), [&](auto a_, const T& w) { uc[decltype(a_)::value] = w; });
    };
    load_u(0);
    for (int t = 0; t < N; t++) {
#pragma unroll
      for (int a = 0; a < m; a++) u[a] = (MERGE && acc) ? uc[a] : ul[a];
      if (t + 1 < N) load_u(t + 1);
      if constexpr (MERGE)
        for_rows<m>(U, ru(0, t), [&](auto a_, T& w) { w = u[decltype(a_)::value]; });
      if constexpr (GENERAL) Sys::trig(x, tr);
      else Sys::trig_s(x, tr, &bad);
      Sys::step_tr(c, x, u, tr, xn);
      for_rows<n>(X, rx(0, t + 1), [&](auto i_, T& w) { w = xn[decltype(i_)::value]; });
#pragma unroll
      for (int i = 0; i < n; i++) x[i] = xn[i];
    }
    return bad;
  }

  // stage cost of a stored trajectory measured to xtarget (only needed when Q != 0, where the
  // nominal cost of the next iteration differs from the accepted forward cost)
  __device__ __forceinline__ T nominal_cost(const T* X, const T* U, const T (&xT)[n]) const {
    T x[n], u[m];
    T cost = T(0);
    for (int t = 0; t < N; t++) {
#pragma unroll
      for (int i = 0; i < n; i++) x[i] = at(X, rx(i, t));
#pragma unroll
      for (int a = 0; a < m; a++) u[a] = at(U, ru(a, t));
      cost = cost + stage_cost(x, c.xtarget, u);
    }
#pragma unroll
    for (int i = 0; i < n; i++) x[i] = at(X, rx(i, N));
    return cost + terminal_cost(x, xT);
  }
};

// ---------------------------------------------------------------------------------------------
// Kernels: 64-thread workgroups (one wavefront), one problem per lane.
// ---------------------------------------------------------------------------------------------
// One wave per SIMD in both precisions: the unrolled Riccati step needs ~410 registers in fp64
// (256 VGPRs + AGPR spill space) and ~370 in fp32 — held to 256 (two waves per SIMD) fp32 spilled
// 107 registers to scratch and ran 16-20 % slower.
template <class T, class Sys, bool HASQR, bool TILED>
__global__ __launch_bounds__(64, (sizeof(T) == 4 ? I2LQR_F32_WAVES : I2LQR_F64_WAVES)) void k_lane_iterate(
    const DevCfg<T, Sys::n, Sys::m> c, const LaneArgs<T> a) {
  constexpr int n = Sys::n, m = Sys::m;
  const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
  const int64_t live = a.count ? (int64_t)*a.count : a.B;
  if (b >= live) return;
  if (a.count && (live <= a.count_lo || live > a.count_hi)) return;  // the other kernel's chunk
  const int N = c.N;
  const LaneView<TILED> v(a.B);
  if (a.resume && a.status[b] != 0) {  // finished since the compaction (tail kernel)
    // fused compaction: what a tail kernel finished without delivering it goes to the caller now
    if (a.cp.on && true)
      lane_exit_compact(a.cp, n, m, N, b, v.Bs, v.bl, v.rebase(a.X, n * (N + 1)),
                        v.rebase(a.U, m * N), v.rebase(a.x_term, n), v.rebase(a.obs, 6),
                        v.rebase(a.K ? a.K : a.wsK, m * n * N), v.rebase(a.K ? a.k : a.wsk, m * N),
                        a.lamb[b], a.cost[b], a.iters[b], a.status[b], c.trap);
    return;
  }
  LaneWorker<T, Sys, HASQR, TILED> w(c, v.Bs, v.bl);
  extern __shared__ __align__(16) unsigned char lane_smem[];
  typedef __attribute__((address_space(3))) T lds_t;
  w.lds = (lds_t*)lane_smem;
  w.has_lds = true;
  w.lds_steps = a.lds_steps;
  const T* gxt = v.rebase(a.x_term, n);
  const T* gob = v.rebase(a.obs, 6);
  T xT[n], ob[6];
#pragma unroll
  for (int i = 0; i < n; i++) xT[i] = gxt[(int64_t)i * v.Bs + v.bl];
#pragma unroll
  for (int q = 0; q < 6; q++) ob[q] = gob ? gob[(int64_t)q * v.Bs + v.bl] : T(q == 5 ? -1 : 1);
  T lamb = a.lamb[b];
  T* gK = v.rebase(a.K ? a.K : a.wsK, m * n * N);
  T* gk = v.rebase(a.K ? a.k : a.wsk, m * N);
  // States live in ONE buffer (the caller's X); only the inputs (m N words) are double-buffered
  // per lane.  With X double-buffered per lane too, divergent accept/reject decisions split every
  // row access of a wavefront over two buffers: +38 % HBM traffic per iteration (rocprofv3
  // FETCH_SIZE / WRITE_SIZE, tools/pmc_iters.sh).  Either the forward pass stores no states and
  // accepted steps re-roll them (a.defer), or it writes the candidate states over the nominal
  // ones in place (each x_t is loaded one step ahead of being overwritten) and rejected steps
  // re-roll the nominal ones; both re-rolls are bit-identical to what the rollout stored.
  T* const X = v.rebase(a.X, n * (N + 1));
  T* const U0 = v.rebase(a.U, m * N);
  T *Uc = U0, *Un = v.rebase(a.wsU, m * N);

  // The nominal rollout of iteration i+1 is bit-identical to the forward rollout of an accepted
  // iteration i (same inputs, same code), and unchanged after a rejected one: roll out once.
  // checkpointed states (fp64 only; host: deferred + merged + re-rolling forward pass, Q = R = 0)
  constexpr bool kCanCkpt = sizeof(T) == 8 && !HASQR && Sys::NBLK == 0 && !I2LQR_DEEP64;
  const bool ckpt = kCanCkpt && a.ckpt;
  if (a.stagger > 0 && ((blockIdx.x >> 9) & 1)) {  // see k_lane_iterate_rows
    for (int q = 0; q < a.stagger; q++) __builtin_amdgcn_s_sleep(127);
  }
  T cost = w.rollout(X, Uc, xT, ckpt);
  const int it0 = a.resume ? a.iters[b] : 0;  // iterations of earlier chunks
  int it = 0, status = a.early_exit ? 2 : 0;
  T cost_ret = cost;
  lds_t* const seg = (lds_t*)lane_smem + (size_t)a.lds_steps * 64 * m * (n + 1) + 64 * m;
  while (it < a.n_iters && it0 + it < a.max_total) {
    // K_0 goes to HBM from the passes that can be this launch's last one for the problem
    const bool k0_out = a.early_exit || it + 1 >= a.n_iters || it0 + it + 1 >= a.max_total;
    if constexpr (kCanCkpt) {
      if (ckpt) w.template backward<true, true>(X, Uc, xT, ob, lamb, gK, gk, k0_out, seg);
      else w.template backward<true>(X, Uc, xT, ob, lamb, gK, gk, k0_out);
    } else {
      w.template backward<true>(X, Uc, xT, ob, lamb, gK, gk, k0_out);
    }
#ifdef I2LQR_STAMPS
    {
      auto& st_t0 = w.st_t0; auto& st_t1 = w.st_t1; auto& st_acc = w.st_acc;
      STAMP_BEGIN();
    }
#endif
    T cost_new;
    if (a.defer) {
      cost_new = a.reroll ? w.template forward<true, false>(X, Uc, gK, gk, X, Un, xT)
                          : w.template forward<false, false>(X, Uc, gK, gk, X, Un, xT);
    } else {
      cost_new = a.reroll ? w.template forward<true>(X, Uc, gK, gk, X, Un, xT)
                          : w.template forward<false>(X, Uc, gK, gk, X, Un, xT);
    }
#ifdef I2LQR_STAMPS
    {
      auto& st_t0 = w.st_t0; auto& st_t1 = w.st_t1; auto& st_acc = w.st_acc;
      STAMP_END(4);
    }
#endif
    it++;
    const bool accepted = cost_new < cost;
    // X must hold the states of each lane's CURRENT inputs: deferred mode owes them to the lanes
    // that accepted, in-place mode to the lanes that rejected.  If any lane of the wavefront needs
    // it, ALL of them re-roll and store (the others rewrite what is already there, bit for bit):
    // full 64-lane rows instead of masked partial ones, which cost a read-modify-write in HBM.
    if (a.defer && a.merge) {
      // the current inputs stay in ONE buffer for the whole wavefront: accepted candidates are
      // merged into it during the re-roll (Uc == U0, Un == workspace throughout)
      if (__all(accepted)) {
        // every lane of the wavefront accepted: the two input buffers change roles for the whole
        // wavefront (still ONE buffer with full rows, no copy) and the states are re-rolled
        T* tp = Uc; Uc = Un; Un = tp;
        if constexpr (kCanCkpt) {
          if (ckpt) w.template restore_states_paired<false, true>(X, Uc);
          else w.restore_states(X, Uc);
        } else {
          w.restore_states(X, Uc);
        }
      } else if (__any(accepted)) {
        if constexpr (kCanCkpt) {
          if (ckpt) w.template merge_and_restore<true>(X, Uc, Un, accepted);
          else w.merge_and_restore(X, Uc, Un, accepted);
        } else {
          w.merge_and_restore(X, Uc, Un, accepted);
        }
      }
    } else {
      if (accepted) {
        T* tp = Uc; Uc = Un; Un = tp;
      }
      if (__any(a.defer ? accepted : !accepted)) w.restore_states(X, Uc);
    }
    if (accepted) {  // control/iterative_ilqr.py:74-80
      lamb /= c.lamb_factor;
      const bool conv = t_abs((cost_new - cost) / cost) < c.eps;
      cost_ret = cost_new;
      // next iteration's nominal cost: stage terms are measured to xtarget, not x_terminal
      cost = HASQR ? w.nominal_cost(X, Uc, xT) : cost_new;
      if (conv) {
        if (a.early_exit) { status = 1; break; }
        if (status == 0) status = 1;
      }
    } else {  // control/iterative_ilqr.py:81-84
      lamb *= c.lamb_factor;
      cost_ret = cost;
      if (lamb > c.max_lamb) {
        if (a.early_exit) { status = 3; break; }
        if (status == 0) status = 3;
      }
    }
#ifdef I2LQR_STAMPS
    {
      auto& st_t0 = w.st_t0; auto& st_t1 = w.st_t1; auto& st_acc = w.st_acc;
      STAMP_END(5);
    }
#endif
  }
  // a chunk that ran out before the problem terminated: RUNNING unless the iteration cap is hit
  if (a.early_exit && status == 2 && it0 + it < a.max_total) status = 0;
  if (!t_isfinite(cost_ret) && (status != 0 || !a.early_exit)) status = 4;
#ifdef I2LQR_STAMPS
  if (a.dbg && threadIdx.x == 0)
    for (int q = 0; q < 8; q++) a.dbg[blockIdx.x * 8 + q] = w.st_acc[q];
#endif
  if constexpr (kCanCkpt) {
    if (ckpt) w.restore_states(X, Uc);  // the caller's X in full: one re-roll per launch
  }
  w.flush_gains(gK, gk);
  if (Uc != U0) {  // the accepted inputs sit in the workspace: copy them out
    for (int e = 0; e < m * N; e++) U0[(int64_t)e * v.Bs + v.bl] = Uc[(int64_t)e * v.Bs + v.bl];
  }
  a.lamb[b] = lamb;
  a.cost[b] = cost_ret;
  if (a.iters) a.iters[b] = it0 + it;
  if (a.status) a.status[b] = status;
  if (a.cp.on)
    lane_exit_compact(a.cp, n, m, N, b, v.Bs, v.bl, X, U0, gxt, gob, gK, gk, lamb, cost_ret, it0 + it,
                      status, c.trap);
}

// k_lane_iterate with a HELPER wavefront (round 5; VERDICT r4 #4: 8 k - 32 k problems).  Up to 32768
// problems the one-problem-per-lane launch is one or two wavefronts per CU — issue-bound on ITS
// SIMD while the CU's other SIMDs idle (launch floor 0.53 ms per 10 iterations up to 16384
// problems), and masked lanes do not issue faster.  What CAN run elsewhere is the part of the
// backward pass that depends on the nominal trajectory only: workgroups of TWO wavefronts, the second
// one computing every step's record a step ahead of the first (LaneWorker::backward<.., ROLE>).
// fp64 and fp32 (HASQR: stage weights, the record grows by l_x), states not checkpointed; same
// arguments, results bit-identical to k_lane_iterate.
template <class T, class Sys, bool HASQR, bool TILED>
__global__ __launch_bounds__(128, 1) void k_lane_iterate_pair(
    const DevCfg<T, Sys::n, Sys::m> c, const LaneArgs<T> a) {
  constexpr int n = Sys::n, m = Sys::m;
  const int role = threadIdx.x >> 6;  // 0: main wavefront, 1: helper
  const unsigned l64 = threadIdx.x & 63;
  const int64_t b = (int64_t)blockIdx.x * 64 + l64;
  const int64_t live = a.count ? (int64_t)*a.count : a.B;
  if (b >= live) return;
  if (a.count && (live <= a.count_lo || live > a.count_hi)) return;  // the other kernel's chunk
  const int N = c.N;
  const LaneView<TILED> v(a.B);
  if (a.resume && a.status[b] != 0) {  // finished since the compaction (tail kernel)
    // fused compaction: what a tail kernel finished without delivering it goes to the caller now
    if (a.cp.on && role == 0)
      lane_exit_compact(a.cp, n, m, N, b, v.Bs, v.bl, v.rebase(a.X, n * (N + 1)),
                        v.rebase(a.U, m * N), v.rebase(a.x_term, n), v.rebase(a.obs, 6),
                        v.rebase(a.K ? a.K : a.wsK, m * n * N), v.rebase(a.K ? a.k : a.wsk, m * N),
                        a.lamb[b], a.cost[b], a.iters[b], a.status[b], c.trap);
    return;
  }
  LaneWorker<T, Sys, HASQR, TILED> w(c, v.Bs, v.bl);
  extern __shared__ __align__(16) unsigned char lane_smem[];
  typedef __attribute__((address_space(3))) T lds_t;
  w.lds = (lds_t*)lane_smem;
  w.has_lds = true;
  w.lds_steps = a.lds_steps;
  const T* gxt = v.rebase(a.x_term, n);
  const T* gob = v.rebase(a.obs, 6);
  T xT[n], ob[6];
#pragma unroll
  for (int i = 0; i < n; i++) xT[i] = gxt[(int64_t)i * v.Bs + v.bl];
#pragma unroll
  for (int q = 0; q < 6; q++) ob[q] = gob ? gob[(int64_t)q * v.Bs + v.bl] : T(q == 5 ? -1 : 1);
  T lamb = a.lamb[b];
  T* gK = v.rebase(a.K ? a.K : a.wsK, m * n * N);
  T* gk = v.rebase(a.K ? a.k : a.wsk, m * N);
  // States live in ONE buffer (the caller's X); only the inputs (m N words) are double-buffered
  // per lane.  With X double-buffered per lane too, divergent accept/reject decisions split every
  // row access of a wavefront over two buffers: +38 % HBM traffic per iteration (rocprofv3
  // FETCH_SIZE / WRITE_SIZE, tools/pmc_iters.sh).  Either the forward pass stores no states and
  // accepted steps re-roll them (a.defer), or it writes the candidate states over the nominal
  // ones in place (each x_t is loaded one step ahead of being overwritten) and rejected steps
  // re-roll the nominal ones; both re-rolls are bit-identical to what the rollout stored.
  // THIS kernel runs launches of at most 512 workgroups, bound by the instruction stream of a step
  // and not by HBM: with a second state buffer in the workspace (a.wsX; the host passes one up to
  // 256 workgroups) the forward pass stores the candidate states there and an accepted lane swaps
  // its state and input buffers — no re-roll (15 % of an iteration of the main wavefront at 16384
  // problems against +6 % for the stores and the per-lane addresses, tools/stamp_run_lane.py;
  // +3.5 % it/s at 8192 problems, +4 % at 12288 and 16384, +1 % at 20480, 0 at 24576, -3 % at
  // 32768 where two workgroups share a CU's path to memory).
  T* const X0 = v.rebase(a.X, n * (N + 1));
  T* const U0 = v.rebase(a.U, m * N);
  T *Uc = U0, *Un = v.rebase(a.wsU, m * N);
  const bool two = a.wsX != nullptr && live <= a.two_max;
  T *X = X0, *Xn = two ? v.rebase(a.wsX, n * (N + 1)) : X0;

  // The nominal rollout of iteration i+1 is bit-identical to the forward rollout of an accepted
  // iteration i (same inputs, same code), and unchanged after a rejected one: roll out once.
  // checkpointed states (fp64 only; host: deferred + merged + re-rolling forward pass, Q = R = 0)
  constexpr bool kCanCkpt = false;  // (the host gives this kernel no checkpointed launch)
  const bool ckpt = false;
  // LDS behind the gains: the two record slots and the control words of the pair
  typedef LaneWorker<T, Sys, HASQR, TILED> LW;
  w.rec = (lds_t*)lane_smem + (size_t)a.lds_steps * 64 * m * (n + 1) + 64 * m;
  typedef __attribute__((address_space(3))) int lds_i;
  lds_i* const ctl = (lds_i*)(w.rec + 2 * LW::kRec * 64);  // [64] input buffer of each lane, [64]: go
  if (role == 1) {
    // helper: its half of every backward pass the main wavefront announces (same lanes, same
    // problems; it never touches the gains, the candidate or the accept / reject state)
    const T* const Uh1 = v.rebase(a.wsU, m * N);
    for (;;) {
      __syncthreads();  // B0: the control words of this pass are written
      if (!ctl[64]) return;
      const int sel = ctl[l64];  // bit 0: inputs in the workspace, bit 1: states in the workspace
      w.template backward<true, false, 2>((sel & 2) ? Xn : X0, (sel & 1) ? Uh1 : U0, xT, ob, lamb,
                                          nullptr, nullptr, false);
    }
  }
  if (a.stagger > 0 && ((blockIdx.x >> 9) & 1)) {  // see k_lane_iterate_rows
    for (int q = 0; q < a.stagger; q++) __builtin_amdgcn_s_sleep(127);
  }
  T cost = w.rollout(X, Uc, xT, ckpt);
  const int it0 = a.resume ? a.iters[b] : 0;  // iterations of earlier chunks
  int it = 0, status = a.early_exit ? 2 : 0;
  T cost_ret = cost;
  while (it < a.n_iters && it0 + it < a.max_total) {
    // K_0 goes to HBM from the passes that can be this launch's last one for the problem
    const bool k0_out = a.early_exit || it + 1 >= a.n_iters || it0 + it + 1 >= a.max_total;
    ctl[l64] = (Uc != U0 ? 1 : 0) | (X != X0 ? 2 : 0);  // where this lane's nominal trajectory is
    ctl[64] = 1;
    __syncthreads();  // B0
    w.template backward<true, false, 1>(X, Uc, xT, ob, lamb, gK, gk, k0_out);
#ifdef I2LQR_STAMPS
    {
      auto& st_t0 = w.st_t0; auto& st_t1 = w.st_t1; auto& st_acc = w.st_acc;
      STAMP_BEGIN();
    }
#endif
    T cost_new;
    constexpr int FD = I2LQR_PAIR_FWD_DEPTH;
    if (two) {
      cost_new = a.reroll ? w.template forward<true, true, FD>(X, Uc, gK, gk, Xn, Un, xT)
                          : w.template forward<false, true, FD>(X, Uc, gK, gk, Xn, Un, xT);
    } else if (a.defer) {
      cost_new = a.reroll ? w.template forward<true, false, FD>(X, Uc, gK, gk, X, Un, xT)
                          : w.template forward<false, false, FD>(X, Uc, gK, gk, X, Un, xT);
    } else {
      cost_new = a.reroll ? w.template forward<true, true, FD>(X, Uc, gK, gk, X, Un, xT)
                          : w.template forward<false, true, FD>(X, Uc, gK, gk, X, Un, xT);
    }
#ifdef I2LQR_STAMPS
    {
      auto& st_t0 = w.st_t0; auto& st_t1 = w.st_t1; auto& st_acc = w.st_acc;
      STAMP_END(4);
    }
#endif
    it++;
    const bool accepted = cost_new < cost;
    // X must hold the states of each lane's CURRENT inputs: deferred mode owes them to the lanes
    // that accepted, in-place mode to the lanes that rejected.  If any lane of the wavefront needs
    // it, ALL of them re-roll and store (the others rewrite what is already there, bit for bit):
    // full 64-lane rows instead of masked partial ones, which cost a read-modify-write in HBM.
    if (two) {
      if (accepted) {
        T* tp = Uc; Uc = Un; Un = tp;
        tp = X; X = Xn; Xn = tp;
      }
    } else if (a.defer && a.merge) {
      // the current inputs stay in ONE buffer for the whole wavefront: accepted candidates are
      // merged into it during the re-roll (Uc == U0, Un == workspace throughout)
      if (__all(accepted)) {
        // every lane of the wavefront accepted: the two input buffers change roles for the whole
        // wavefront (still ONE buffer with full rows, no copy) and the states are re-rolled
        T* tp = Uc; Uc = Un; Un = tp;
        if constexpr (kCanCkpt) {
          if (ckpt) w.template restore_states_paired<false, true>(X, Uc);
          else w.restore_states(X, Uc);
        } else {
          w.restore_states(X, Uc);
        }
      } else if (__any(accepted)) {
        if constexpr (kCanCkpt) {
          if (ckpt) w.template merge_and_restore<true>(X, Uc, Un, accepted);
          else w.merge_and_restore(X, Uc, Un, accepted);
        } else {
          w.merge_and_restore(X, Uc, Un, accepted);
        }
      }
    } else {
      if (accepted) {
        T* tp = Uc; Uc = Un; Un = tp;
      }
      if (__any(a.defer ? accepted : !accepted)) w.restore_states(X, Uc);
    }
    if (accepted) {  // control/iterative_ilqr.py:74-80
      lamb /= c.lamb_factor;
      const bool conv = t_abs((cost_new - cost) / cost) < c.eps;
      cost_ret = cost_new;
      // next iteration's nominal cost: stage terms are measured to xtarget, not x_terminal
      cost = HASQR ? w.nominal_cost(X, Uc, xT) : cost_new;
      if (conv) {
        if (a.early_exit) { status = 1; break; }
        if (status == 0) status = 1;
      }
    } else {  // control/iterative_ilqr.py:81-84
      lamb *= c.lamb_factor;
      cost_ret = cost;
      if (lamb > c.max_lamb) {
        if (a.early_exit) { status = 3; break; }
        if (status == 0) status = 3;
      }
    }
#ifdef I2LQR_STAMPS
    {
      auto& st_t0 = w.st_t0; auto& st_t1 = w.st_t1; auto& st_acc = w.st_acc;
      STAMP_END(5);
    }
#endif
  }
  ctl[64] = 0;  // release the helper
  __syncthreads();
  // a chunk that ran out before the problem terminated: RUNNING unless the iteration cap is hit
  if (a.early_exit && status == 2 && it0 + it < a.max_total) status = 0;
  if (!t_isfinite(cost_ret) && (status != 0 || !a.early_exit)) status = 4;
#ifdef I2LQR_STAMPS
  if (a.dbg && threadIdx.x == 0)
    for (int q = 0; q < 8; q++) a.dbg[blockIdx.x * 8 + q] = w.st_acc[q];
#endif
  if constexpr (kCanCkpt) {
    if (ckpt) w.restore_states(X, Uc);  // the caller's X in full: one re-roll per launch
  }
  w.flush_gains(gK, gk);
  // the accepted trajectory sits in the workspace: copy it out, 32 rows in flight at a time (one
  // row per round trip to HBM was 44 us of a 0.46 ms launch)
  auto copy_out = [&](T* dst, const T* src, const int rows) __attribute__((always_inline)) {
    for (int e0 = 0; e0 < rows; e0 += 32) {
      T tmp[32];
#pragma unroll
      for (int j = 0; j < 32; j++)
        if (e0 + j < rows) tmp[j] = src[(int64_t)(e0 + j) * v.Bs + v.bl];
#pragma unroll
      for (int j = 0; j < 32; j++)
        if (e0 + j < rows) dst[(int64_t)(e0 + j) * v.Bs + v.bl] = tmp[j];
    }
  };
  if (Uc != U0) copy_out(U0, Uc, m * N);
  if (X != X0) copy_out(X0, X, n * (N + 1));
  a.lamb[b] = lamb;
  a.cost[b] = cost_ret;
  if (a.iters) a.iters[b] = it0 + it;
  if (a.status) a.status[b] = status;
  if (a.cp.on)
    lane_exit_compact(a.cp, n, m, N, b, v.Bs, v.bl, X0, U0, gxt, gob, gK, gk, lamb, cost_ret, it0 + it,
                      status, c.trap);
}

// The fused kernel of the row-block plants (quad12): k_lane_iterate's deferred, merged form — the
// forward pass stores no states, accepted candidates are merged into the one input buffer while
// the states are re-rolled — with every access through row groups and no LDS (see LaneWorker::
// for_rows, backward_blocked).  Same arguments and results as k_lane_iterate (chunked solves
// included); the defer / reroll / merge / lds / ckpt options do not apply.  HASQR: stage weights
// Q, R != 0 (round 5; an instantiation of its own: the plant's default is Q = R = 0 and the 12 words
// of l_x = 2 Q (x_t - xtarget) that live through the step cost the hot kernel registers it does not have).
template <class T, class Sys, bool HASQR, bool TILED>
__global__ __launch_bounds__(64, 1) void k_lane_iterate_rows(const DevCfg<T, Sys::n, Sys::m> c,
                                                             const LaneArgs<T> a) {
  constexpr int n = Sys::n, m = Sys::m;
  static_assert(Sys::NBLK > 0, "built for the plants with a row-block form");
  const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
  const int64_t live = a.count ? (int64_t)*a.count : a.B;
  if (b >= live) return;
  const int N = c.N;
  const LaneView<TILED> v(a.B);
  if (a.resume && a.status[b] != 0) {
    if (a.cp.on)  // fused compaction (see k_lane_iterate)
      lane_exit_compact(a.cp, n, m, N, b, v.Bs, v.bl, v.rebase(a.X, n * (N + 1)),
                        v.rebase(a.U, m * N), v.rebase(a.x_term, n), v.rebase(a.obs, 6),
                        v.rebase(a.K ? a.K : a.wsK, m * n * N), v.rebase(a.K ? a.k : a.wsk, m * N),
                        a.lamb[b], a.cost[b], a.iters[b], a.status[b], c.trap);
    return;
  }
  LaneWorker<T, Sys, HASQR, TILED> w(c, v.Bs, v.bl);
  const T* gxt = v.rebase(a.x_term, n);
  const T* gob = v.rebase(a.obs, 6);
  T xT[n], ob[6];
#pragma unroll
  for (int i = 0; i < n; i++) xT[i] = gxt[(int64_t)i * v.Bs + v.bl];
#pragma unroll
  for (int q = 0; q < 6; q++) ob[q] = gob ? gob[(int64_t)q * v.Bs + v.bl] : T(q == 5 ? -1 : 1);
  T lamb = a.lamb[b];
  T* gK = v.rebase(a.K ? a.K : a.wsK, m * n * N);
  T* gk = v.rebase(a.K ? a.k : a.wsk, m * N);
  T* const X = v.rebase(a.X, n * (N + 1));
  T* const U0 = v.rebase(a.U, m * N);
  T *Uc = U0, *Un = v.rebase(a.wsU, m * N);
  __shared__ T lds_gains[LaneWorker<T, Sys, HASQR, TILED>::kGainWords];
  __shared__ unsigned lds_sink[64];  // LaneWorker::warm_rows
  w.sink = lds_sink;
  // Stagger: every wavefront of a full-chip launch does the same work in the same order, so all of
  // them stream their gains at once (the forward pass: at the HBM rate, issue slots idle) and all
  // of them compute at once (the backward pass: issue-bound, HBM half idle).  Workgroups 512-1023
  // of every 1024 — two of the four wavefronts a CU holds — start a.stagger x ~8000 cycles late,
  // so that one half streams while the other computes (+5-7 % at 65536 problems; the delay itself
  // is paid once per launch).
  if (a.stagger > 0 && ((blockIdx.x >> 9) & 1)) {
    for (int q = 0; q < a.stagger; q++) __builtin_amdgcn_s_sleep(127);
  }
  T cost = w.rollout(X, Uc, xT);
  const int it0 = a.resume ? a.iters[b] : 0;  // iterations of earlier chunks
  int it = 0, status = a.early_exit ? 2 : 0;
  T cost_ret = cost;
  const bool general = !(sizeof(T) == 8 && c.fast_barrier);  // configurations outside the hot form
  while (it < a.n_iters && it0 + it < a.max_total) {
    // hot, branch-free passes first; the general forms only if a lane asked for them
    if (__builtin_expect(__any(general || w.template backward_blocked<true, false>(X, Uc, xT, ob, lamb, gK, gk, lds_gains)), 0))
      w.template backward_blocked<true, true>(X, Uc, xT, ob, lamb, gK, gk, lds_gains);
    bool big = false;
#ifdef I2LQR_STAMPS
    auto& st_t0 = w.st_t0; auto& st_t1 = w.st_t1; auto& st_acc = w.st_acc;
    STAMP_BEGIN();
#endif
    T cost_new = w.template forward_rows<false>(X, Uc, gK, gk, Un, xT, &big);
    if (__builtin_expect(__any(big), 0))
      cost_new = w.template forward_rows<true>(X, Uc, gK, gk, Un, xT, &big);
    it++;
#ifdef I2LQR_STAMPS
    STAMP_END(4);
#endif
    const bool accepted = cost_new < cost;
    if (__all(accepted)) {  // the two input buffers change roles for the whole wavefront
      T* tp = Uc; Uc = Un; Un = tp;
      if (__builtin_expect(__any(w.template restore_rows<false, false>(X, Uc)), 0))
        w.template restore_rows<false, true>(X, Uc);
    } else if (__any(accepted)) {
      if (__builtin_expect(__any(w.template restore_rows<true, false>(X, Uc, Un, accepted)), 0))
        w.template restore_rows<true, true>(X, Uc, Un, accepted);
    }
#ifdef I2LQR_STAMPS
    STAMP_END(5);
#endif
    if (accepted) {  // control/iterative_ilqr.py:74-80
      lamb /= c.lamb_factor;
      const bool conv = t_abs((cost_new - cost) / cost) < c.eps;
      cost_ret = cost_new;
      // next iteration's nominal cost: stage terms are measured to xtarget, not x_terminal
      cost = HASQR ? w.nominal_cost(X, Uc, xT) : cost_new;
      if (conv) {
        if (a.early_exit) { status = 1; break; }
        if (status == 0) status = 1;
      }
    } else {  // control/iterative_ilqr.py:81-84
      lamb *= c.lamb_factor;
      cost_ret = cost;
      if (lamb > c.max_lamb) {
        if (a.early_exit) { status = 3; break; }
        if (status == 0) status = 3;
      }
    }
  }
  if (a.early_exit && status == 2 && it0 + it < a.max_total) status = 0;
  if (!t_isfinite(cost_ret) && (status != 0 || !a.early_exit)) status = 4;
  if (Uc != U0) {  // the accepted inputs sit in the workspace: copy them out
    for (int e = 0; e < m * N; e++) U0[(int64_t)e * v.Bs + v.bl] = Uc[(int64_t)e * v.Bs + v.bl];
  }
  a.lamb[b] = lamb;
  a.cost[b] = cost_ret;
  if (a.iters) a.iters[b] = it0 + it;
  if (a.status) a.status[b] = status;
#ifdef I2LQR_STAMPS
  if (a.dbg && threadIdx.x == 0)
    for (int q = 0; q < 8; q++) a.dbg[blockIdx.x * 8 + q] = w.st_acc[q];
#endif
  if (a.cp.on)
    lane_exit_compact(a.cp, n, m, N, b, v.Bs, v.bl, X, U0, gxt, gob, gK, gk, lamb, cost_ret, it0 + it,
                      status, c.trap);
}

// ---------------------------------------------------------------------------------------------
// Compaction step of the chunked solve.  Problems of the source set that have terminated are
// scattered to the caller's arrays at their original index (unless the source IS the caller's
// arrays); the survivors are packed densely into the destination work set, so the next chunk's
// wavefronts are full again.  Order inside the destination is arbitrary (atomic slot counter,
// wave-aggregated by the compiler): every problem is independent, results do not depend on it.
// ---------------------------------------------------------------------------------------------
template <class T, bool USER_TILED>
__global__ __launch_bounds__(256) void k_lane_compact(int n, int m, int N, LaneSet<T> src,
                                                      int src_is_user, const int32_t* count_in,
                                                      LaneSet<T> dst, int32_t* count_out,
                                                      LaneSet<T> usr,
                                                      unsigned long long* trap = nullptr) {
  (void)trap;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t live = count_in ? (int64_t)*count_in : src.B;
  if (i >= live) return;
  const int rx = n * (N + 1), ru = m * N, rK = m * n * N;
  // address of (row, problem) in the caller's layout / in a batch-minor work set
  auto uaddr = [&](int rows, int row, int64_t p) -> int64_t {
    I2LQR_DBG_CHECK(trap, TAG_COMPACT, row, rows);
    I2LQR_DBG_CHECK(trap, TAG_COMPACT, p, usr.B);
    if (USER_TILED) return ((p >> 6) * rows + row) * 64 + (p & 63);
    return (int64_t)row * usr.B + p;
  };
  auto saddr = [&](int rows, int row, int64_t p) -> int64_t {
    if (src_is_user) return uaddr(rows, row, p);
    I2LQR_DBG_CHECK(trap, TAG_COMPACT, row, rows);
    I2LQR_DBG_CHECK(trap, TAG_COMPACT, p, src.B);
    return (int64_t)row * src.B + p;
  };
  // rows are moved eight at a time: eight independent loads in flight, then eight stores (the
  // arrays may alias as far as the compiler knows, so a plain loop would serialise every pair)
  auto move_rows = [&](int rows, auto&& dst_at, auto&& src_at) __attribute__((always_inline)) {
    int r = 0;
    for (; r + 8 <= rows; r += 8) {
      T v[8];
#pragma unroll
      for (int q = 0; q < 8; q++) v[q] = src_at(r + q);
#pragma unroll
      for (int q = 0; q < 8; q++) dst_at(r + q, v[q]);
    }
    for (; r < rows; r++) dst_at(r, src_at(r));
  };
  const int st = src.status[i];
  if (st != 0) {
    if (src_is_user) return;  // already in place
    if (st & kStatusDelivered) return;  // the tail kernel wrote it to the caller's arrays itself
    const int64_t o = src.orig[i];
    move_rows(rx, [&](int r, T v) { usr.X[uaddr(rx, r, o)] = v; },
              [&](int r) { return src.X[(int64_t)r * src.B + i]; });
    move_rows(ru, [&](int r, T v) { usr.U[uaddr(ru, r, o)] = v; },
              [&](int r) { return src.U[(int64_t)r * src.B + i]; });
    if (usr.K) {
      move_rows(rK, [&](int r, T v) { usr.K[uaddr(rK, r, o)] = v; },
                [&](int r) { return src.K[(int64_t)r * src.B + i]; });
      move_rows(ru, [&](int r, T v) { usr.k[uaddr(ru, r, o)] = v; },
                [&](int r) { return src.k[(int64_t)r * src.B + i]; });
    }
    usr.lamb[o] = src.lamb[i];
    usr.cost[o] = src.cost[i];
    if (usr.iters) usr.iters[o] = src.iters[i];
    if (usr.status) usr.status[o] = st;
    return;
  }
  if (!dst.X) return;  // final pass: nothing survives (every problem has a terminal status)
  const int64_t j = atomicAdd(count_out, 1);
  I2LQR_DBG_CHECK(trap, TAG_COMPACT, j, dst.B);
  // a survivor carries its inputs and x_0 only (the rows t = 0 of the time-major X): every chunk
  // starts by rolling the states out again
  move_rows(n, [&](int r, T v) { dst.X[(int64_t)r * dst.B + j] = v; },
            [&](int r) { return src.X[saddr(rx, r, i)]; });
  move_rows(ru, [&](int r, T v) { dst.U[(int64_t)r * dst.B + j] = v; },
            [&](int r) { return src.U[saddr(ru, r, i)]; });
  move_rows(n, [&](int r, T v) { dst.x_term[(int64_t)r * dst.B + j] = v; },
            [&](int r) { return src.x_term[saddr(n, r, i)]; });
  if (src.obs)
    move_rows(6, [&](int r, T v) { dst.obs[(int64_t)r * dst.B + j] = v; },
              [&](int r) { return src.obs[saddr(6, r, i)]; });
  dst.lamb[j] = src.lamb[i];
  dst.iters[j] = src.iters[i];
  dst.status[j] = 0;  // RUNNING (a tail launch of the one-problem-per-wavefront kernel may finish it)
  dst.orig[j] = src_is_user ? (int32_t)i : src.orig[i];
}

template <class T, class Sys, bool HASQR, bool TILED>
__global__ __launch_bounds__(64) void k_lane_rollout(const DevCfg<T, Sys::n, Sys::m> c, int64_t B,
                                                     T* X, T* U, const T* x_term, T* cost) {
  constexpr int n = Sys::n, m = Sys::m;
  const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  const LaneView<TILED> v(B);
  LaneWorker<T, Sys, HASQR, TILED> w(c, v.Bs, v.bl);
  const T* gxt = v.rebase(x_term, n);
  T xT[n];
#pragma unroll
  for (int i = 0; i < n; i++) xT[i] = gxt[(int64_t)i * v.Bs + v.bl];
  cost[b] = w.rollout(v.rebase(X, n * (c.N + 1)), v.rebase(U, m * c.N), xT);
}

template <class T, class Sys, bool HASQR, bool TILED>
__global__ __launch_bounds__(64) void k_lane_backward(const DevCfg<T, Sys::n, Sys::m> c, int64_t B,
                                                      const T* X, const T* U, const T* x_term,
                                                      const T* lamb, const T* obs, T* K, T* k) {
  constexpr int n = Sys::n, m = Sys::m;
  const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  const int N = c.N;
  const LaneView<TILED> v(B);
  LaneWorker<T, Sys, HASQR, TILED> w(c, v.Bs, v.bl);
  const T* gxt = v.rebase(x_term, n);
  const T* gob = v.rebase(obs, 6);
  T xT[n], ob[6];
#pragma unroll
  for (int i = 0; i < n; i++) xT[i] = gxt[(int64_t)i * v.Bs + v.bl];
#pragma unroll
  for (int q = 0; q < 6; q++) ob[q] = gob ? gob[(int64_t)q * v.Bs + v.bl] : T(q == 5 ? -1 : 1);
  w.backward(v.rebase(X, n * (N + 1)), v.rebase(U, m * N), xT, ob, lamb[b],
             v.rebase(K, m * n * N), v.rebase(k, m * N), true);  // no LDS: every gain to HBM
}

template <class T, class Sys, bool HASQR, bool TILED>
__global__ __launch_bounds__(64) void k_lane_forward(const DevCfg<T, Sys::n, Sys::m> c, int64_t B,
                                                     const T* X, const T* U, const T* x_term,
                                                     const T* K, const T* k, T* Xn, T* Un,
                                                     T* cost_new) {
  constexpr int n = Sys::n, m = Sys::m;
  const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  const int N = c.N;
  const LaneView<TILED> v(B);
  LaneWorker<T, Sys, HASQR, TILED> w(c, v.Bs, v.bl);
  const T* gxt = v.rebase(x_term, n);
  T xT[n];
#pragma unroll
  for (int i = 0; i < n; i++) xT[i] = gxt[(int64_t)i * v.Bs + v.bl];
  cost_new[b] = w.template forward<false>(v.rebase(X, n * (N + 1)), v.rebase(U, m * N), v.rebase(K, m * n * N),
                          v.rebase(k, m * N), v.rebase(Xn, n * (N + 1)), v.rebase(Un, m * N), xT);
}

}  // namespace i2lqr

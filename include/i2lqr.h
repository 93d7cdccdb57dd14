/*
 * i2lqr.h — C-ABI of the MI355X-native batched iLQR solver (libi2lqr_hip.so).
 *
 * This is the drop-in boundary for the hot path of HybridRobotics/ilqr-iterative-tasks:
 * the reference has no FFI layer, its seam is the plain Python call
 *     uvar, xvar, lamb = ilqr(ilqr_param, num_horizon, xtarget, timestep, obstacle,
 *                             system_param, x_terminal, dX, uvar, xvar, lamb)
 * at iterative_ilqr/utils/base.py:414-426 inside iLqr.calc_input() (:371-479).  Every entry point
 * below names the reference function (file:line, relative to the reference root) it replaces.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no C++/torch types.  Every function returns an int:
 *    0 = ok, <0 = error (message via i2lqr_last_error(), thread-local).  Numerical outcomes
 *    (converged / diverged / non-finite) are per-problem status words, never error codes.
 *  - All data pointers are DEVICE pointers (HBM) owned by the caller; the library never allocates
 *    user-visible memory.  `stream` is a hipStream_t passed as void*; launches are asynchronous.
 *  - `real` below is double (cfg.dtype == I2LQR_F64) or float (I2LQR_F32); all buffers of one
 *    handle use that one type.  The config struct itself is always double.
 *  - Layout I2LQR_LAYOUT_PROBLEM_MAJOR (default): one problem's record is contiguous and inside
 *    it TIME is the fastest axis, exactly as the reference's NumPy arrays
 *    (xvar[n, N+1], uvar[m, N], K[m, n, N], k[m, N]; control/iterative_ilqr.py:109-110,
 *    utils/base.py:405-409):
 *        X[B][n][N+1]  U[B][m][N]  K[B][m][n][N]  k[B][m][N]  x_term[B][n]  obs[B][6]
 *    Layout I2LQR_LAYOUT_BATCH_MINOR: the batch index is the fastest axis and TIME the slowest
 *        X[N+1][n][B]  U[N][m][B]  K[N][m][n][B]  k[N][m][B]  x_term[n][B]  obs[6][B]
 *    (one problem per lane; used by the throughput kernels: the words of one horizon step are
 *    adjacent rows of B).
 *    Layout I2LQR_LAYOUT_BATCH_TILED: the same inside tiles of 64 problems, tiles outermost
 *        X[B/64][N+1][n][64]  U[B/64][N][m][64]  K[B/64][N][m][n][64]  k[B/64][N][m][64]
 *        x_term[B/64][n][64]  obs[B/64][6][64];  lamb, cost, iters, status stay flat [B].
 *    B must be a multiple of 64.  Same kernels; every wavefront's rows are contiguous in HBM.
 *  - obs record = {x, y, width, height, spd, moving_option}; moving_option 0 = static,
 *    1 = moving up (+y), 2 = moving left (-x) (utils/base.py:23-34, control/ilqr_helper.py:34-43);
 *    moving_option < 0 disables the obstacle for that problem (the reference's `obstacle is None`).
 *    A NULL obs pointer disables it for the whole batch.  A record with moving_option 0 is a
 *    STATIC obstacle whatever its spd word holds (the word is not read); the reference raises a
 *    NameError for spd != 0 with moving_option None — the Python host rejects that combination
 *    (ValueError in control/params.py:obstacle_record), the C-ABI defines it.
 */
#ifndef I2LQR_H
#define I2LQR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define I2LQR_ABI_VERSION 3
#define I2LQR_MAX_N 12
#define I2LQR_MAX_M 4
#define I2LQR_MAX_HORIZON 64
#define I2LQR_OBS_WORDS 6
#define I2LQR_QF_NONE 0x7fffffff /* qfun value of an empty candidate slot (i2lqr_select_candidates) */

/* cfg.dtype */
enum { I2LQR_F64 = 0, I2LQR_F32 = 1 };
/* cfg.layout */
enum { I2LQR_LAYOUT_PROBLEM_MAJOR = 0, I2LQR_LAYOUT_BATCH_MINOR = 1, I2LQR_LAYOUT_BATCH_TILED = 2 };
/* cfg.system_id: the plant model (reference: systems/kinetic_bicycle.py:10-52 for BICYCLE4) */
enum {
  I2LQR_SYS_BICYCLE4 = 0, /* reference plant: [x,y,v,theta], [accel,delta]            n=4  m=2 */
  I2LQR_SYS_BICYCLE6 = 1, /* build-defined: bicycle4 + actuator states [a,delta],
                             inputs [jerk, steer rate]                                 n=6  m=2 */
  I2LQR_SYS_QUAD12 = 2    /* build-defined: rigid-body quadrotor, explicit Euler       n=12 m=4 */
};
/* return codes */
enum {
  I2LQR_OK = 0,
  I2LQR_ERR_INVALID = -1,     /* bad argument / config */
  I2LQR_ERR_UNSUPPORTED = -2, /* shape / system / layout combination not built */
  I2LQR_ERR_LAUNCH = -3,      /* HIP launch or runtime failure */
  I2LQR_ERR_NODEVICE = -4     /* no HIP device visible */
};
/* per-problem status word written by i2lqr_iterate / i2lqr_solve */
enum {
  I2LQR_ST_RUNNING = 0,       /* fixed-count iterate finished without a termination event */
  I2LQR_ST_CONVERGED = 1,     /* |dcost/cost| < eps on an accepted step (iterative_ilqr.py:78-80) */
  I2LQR_ST_MAX_ITER = 2,      /* max_ilqr_iter reached (iterative_ilqr.py:29) */
  I2LQR_ST_LAMB_OVERFLOW = 3, /* lamb > max_lamb after a rejected step (iterative_ilqr.py:83-84) */
  I2LQR_ST_NONFINITE = 4      /* returned cost is NaN/Inf */
};

/*
 * Host-side POD configuration.  Mirrors the parameter bundles the reference reads on the path:
 * iLqrParam (utils/base.py:242-302), KineticBicycleParam (utils/base.py:15-20) and the constants
 * in utils/constants_kinetic_bicycle.py:1-6.  Matrices are row-major with leading dimension
 * I2LQR_MAX_N (Q, Qt) / I2LQR_MAX_M (R); only the top-left n x n / m x m block is read.
 */
typedef struct i2lqr_config {
  int32_t struct_size; /* = sizeof(i2lqr_config); guards against ABI drift */
  int32_t n, m, N;     /* X_DIM, U_DIM, num_horizon */
  int32_t dtype;       /* I2LQR_F64 | I2LQR_F32 */
  int32_t layout;      /* I2LQR_LAYOUT_* */
  int32_t system_id;   /* I2LQR_SYS_* */
  int32_t max_iter;    /* iLqrParam.max_ilqr_iter (150) */
  double dt;           /* timestep */
  double eps;          /* iLqrParam.eps (1e-2) */
  double lamb_factor;  /* iLqrParam.lamb_factor (10) */
  double max_lamb;     /* iLqrParam.max_lamb (1000) */
  double ctrl_q1, ctrl_q2; /* iLqrParam.tuning_ctrl_q1/q2 (1, 1) */
  double obs_q1, obs_q2;   /* iLqrParam.tuning_obs_q1/q2 (2.74, 2.74) */
  double safety_margin;    /* iLqrParam.safety_margin (0) */
  double u_max[I2LQR_MAX_M]; /* symmetric input box; bicycle4: {a_max, round(delta_max, 2)}
                                (control/iterative_ilqr.py:33-40, control/ilqr_helper.py:90-100) */
  double xtarget[I2LQR_MAX_N]; /* tracking target of the stage cost (utils/base.py:374: zeros) */
  double Q[I2LQR_MAX_N * I2LQR_MAX_N];  /* iLqrParam.matrix_Q */
  double Qt[I2LQR_MAX_N * I2LQR_MAX_N]; /* iLqrParam.matrix_Qterminal */
  double R[I2LQR_MAX_M * I2LQR_MAX_M];  /* iLqrParam.matrix_R */
  double sys_par[8]; /* plant constants; QUAD12: {mass, g, arm, Ix, Iy, Iz, ctau, 0}; else unused */
} i2lqr_config;

typedef struct i2lqr_handle i2lqr_handle;

/* ABI version of the loaded library (== I2LQR_ABI_VERSION of the header it was built from). */
int i2lqr_version(void);

/* Last error message of the calling thread ("" if none).  Never NULL. */
const char* i2lqr_last_error(void);

/* Fill *cfg with the reference defaults for `system_id` (utils/base.py:243-271, :16) at the
 * system's native n, m and the given horizon; dtype f64, problem-major layout.  Host only. */
int i2lqr_config_default(i2lqr_config* cfg, int system_id, int num_horizon);

/* Validate the config, upload it to the device and size internal state.  No user-visible memory
 * is allocated.  One handle per stream; a handle is not thread-safe. */
int i2lqr_create(const i2lqr_config* cfg, i2lqr_handle** out);
/* i2lqr_destroy(NULL) is a no-op (I2LQR_OK); a pointer that is not a live handle of this library —
 * e.g. a second destroy of the same handle — returns I2LQR_ERR_INVALID and frees nothing. */
int i2lqr_destroy(i2lqr_handle* h);

/*
 * Geometry of the CURRENT device as the library sees it (hipDeviceGetAttribute, once per device;
 * round 6): out[0..count) = {compute units, SIMDs per CU, LDS bytes per CU, most dynamic LDS one
 * workgroup can be given, ... without opt-in, wavefront size, 1 if the debug override
 * I2LQR_FAKE_CUS=<n> replaced the CU count, 1 if the figures come from the runtime (0: no device
 * visible, the MI355X figures: 256, 4, 163840, 163840, 65536, 64)}.  Every LDS budget, every
 * "a SIMD for each wavefront" limit and every batch-size threshold of the kernel choice derives
 * from these (csrc/i2lqr_geometry.hpp); the thresholds were measured on the full chip (256 CUs)
 * and are scaled by CUs / 256 elsewhere (a partitioned device).  No reference counterpart.
 */
int i2lqr_device_geometry(int32_t* out, int32_t count);

/*
 * Sanitizer hook (round 6).  In the AddressSanitizer / UBSan build of the
 * library (make asan -> libi2lqr_hip_asan.so) with I2LQR_DRY_RUN=1 in the environment,
 * i2lqr_create needs no device and every kernel launch behind the handle becomes a RECORD (kernel,
 * grid, workgroup size, dynamic LDS) whose pointer arguments — plain ones and every pointer field of
 * the kernels' argument blocks — must lie in a range declared here: the host code behind a live
 * handle (workspace carving, the chunked solve's scheduler, LDS budgeting, the sharded round) runs
 * under the sanitizers on a box without a GPU (tools/dry_run_fuzz.py).
 *   op 0: 1 if dry-run launches are active;  op 1: declare [p, p + n) (the workspace, a caller's
 *   array);  op 2: forget ranges and records;  op 3: write the records and violations since the
 *   last report into p[0..n) and return the number of violations.
 * The product library answers I2LQR_ERR_UNSUPPORTED to every op.  No reference counterpart.
 */
int64_t i2lqr_dry_run(int32_t op, void* p, int64_t n);

/*
 * Which cfg.layout to create the handle with for batches of B problems (host only, no GPU needed):
 * the layouts are different kernel FAMILIES — problem-major runs one problem per 64, 16 or 8 lanes
 * (latency kernels: up to ~10^4 problems), batch-minor / batch-tiled run one problem per lane (the
 * HBM-bound throughput kernels) — and the crossover is MEASURED PER SHAPE (plant x horizon x
 * precision x entry point: sixteen shapes, tools/threshold_sweep.py; the table is kLaneFrom /
 * kLaneFromQuad in csrc/i2lqr_abi.hip and DESIGN.md section 3.0), not derivable by a caller and not
 * one number: e.g. bicycle6 N = 20 fp64 crosses over at 8193 problems for fixed iteration counts
 * (early_exit 0: i2lqr_iterate) and 10241 for solves to termination (early_exit 1: i2lqr_solve),
 * bicycle6 N = 50 fp64 at 3073 / 4096, quad12 N = 50 fp64 at 4097 / 6145, fp32 at 8193 / 12289.
 * With stage weights (Q or R != 0) every plant crosses over at 2048 problems (the problem-major
 * side is the one-problem-per-wavefront kernel then).  On a device with another CU count than the
 * 256 the table was measured on, the entries are scaled by CUs / 256 (i2lqr_device_geometry).
 * Returns I2LQR_LAYOUT_BATCH_TILED where B is a multiple of 64, I2LQR_LAYOUT_BATCH_MINOR otherwise,
 * I2LQR_LAYOUT_PROBLEM_MAJOR below the crossover and for the configurations the lane kernels do
 * not run: non-symmetric weights, and quad12 in fp32 WITH stage weights (quad12 with stage
 * weights in fp64, and quad12 in fp32 with Q = R = 0, are lane-capable since round 5); < 0 on a
 * bad argument.  cfg->layout itself is not read.  The reference has no counterpart (one NumPy
 * layout).  Inside the problem-major layout the library picks the kernel per call
 * (i2lqr_iterate_kernel).
 */
int i2lqr_recommended_layout(const i2lqr_config* cfg, int64_t B, int32_t early_exit);

/*
 * Scratch in HBM.  Batch-minor / batch-tiled layouts (one problem per lane): candidate inputs, the
 * gains when the caller does not ask for K/k, the work sets of the chunked solve; a call whose
 * batch needs more than the registered size fails with I2LQR_ERR_INVALID.  Problem-major layout,
 * optional (without it the kernels that need none run): for the bicycles above 4096 problems the
 * per-step records and gains of the workspace form of the eight-lane kernel ("group_workspace":
 * 5.6 KB per problem at n = 6, N = 20; 0 up to 4096 problems); for quad12 (n = 12, m = 4) the
 * per-step records, gains and candidate trajectory of the sixteen-lanes-per-problem kernel
 * ("group_lanes" 16: ~60 KB per problem at N = 50).
 * The caller owns the memory (device pointer, 16-byte aligned) and registers it on the handle.
 */
int64_t i2lqr_workspace_bytes(const i2lqr_handle* h, int64_t B);
int i2lqr_set_workspace(i2lqr_handle* h, void* workspace, int64_t bytes);

/*
 * Chunked form of i2lqr_solve for the batch-minor / batch-tiled layouts: from `min_batch`
 * problems the solve runs in chunks of 8, 4, then doubling iterations and packs the still-running
 * problems into dense work sets between chunks (no host synchronisation: the live count stays in
 * device memory and decides there which kernel of a round does the work — the schedule holds no
 * constant derived from a workload); once few problems are left ("wave_tail" option below) they are
 * finished by the
 * speculative sixteen-lane kernel, which writes their results straight to the caller's arrays
 * (the one-problem-per-wavefront kernel where that is not built).
 * ilqr() runs 1..max_iter iterations per problem (control/iterative_ilqr.py:29-84), so the end of a
 * large solve is bound by the slowest problem's iteration latency.
 *   min_batch  > 0  explicit threshold;  0  never (single launch);  < 0  automatic (default:
 *   from 4096 problems when max_iter > 16).
 * Measured on MI355X, n=6, N=20, 65536 problems, fp64: 6.6-7.5 ms single launch, 7.4 ms chunked
 * without the tail kernel, 1.5-1.6 ms with it.  The chunks alone are bit-identical to the single
 * launch; with the tail kernel the outputs agree to the solve tolerance (1e-8 rel).
 */
int i2lqr_set_compaction(i2lqr_handle* h, int64_t min_batch);

/*
 * Scheduling options of the kernels.  They change how the work is laid out on the GPU, not the
 * results (bit-identical, tests/; the one exception is "wave_tail", see there).
 * value -1 restores the automatic choice.  No reference counterpart (the NumPy path has no such
 * degrees of freedom); they exist so that A/B measurements run in one process on one device.
 * One problem per lane (batch-minor / batch-tiled layouts):
 *   "defer_states"    1: the forward pass (control/iterative_ilqr.py:133-160) stores only the
 *                     candidate inputs; the states of an ACCEPTED step are re-rolled from them
 *                     (rejected steps cost no state traffic at all).  0: candidate states are
 *                     written in place and a rejected step re-rolls the nominal ones.
 *                     Automatic: 1 for i2lqr_iterate, 0 for i2lqr_solve.
 *   "merge_inputs"    ("defer_states" 1) 1: the re-roll of a wavefront in which some lane accepted
 *                     also merges the accepted candidate inputs into the one input buffer (every
 *                     lane rewrites its row entry), so all input rows of a wavefront stay full
 *                     64-lane rows of ONE buffer.  0: per-lane buffer swap (no copy, but after
 *                     mixed decisions every input row access touches two buffers).  Automatic: 1 in
 *                     fp64 above 32768 problems (bandwidth-bound), 0 below and in fp32
 *                     (instruction-bound).
 *   "checkpoint_states"  (fp64, "defer_states" 1, "merge_inputs" 1, "reroll_nominal" 1, Q = R = 0)
 *                     1: between the passes of an iteration only every fourth state is kept in
 *                     HBM; the backward pass re-rolls the states of four horizon steps at a time
 *                     from their checkpoint into LDS (bit-identical values).  A quarter of the
 *                     state traffic for one more plant step per horizon step; the segment buffer
 *                     replaces two LDS-resident gain steps.  Automatic: 1 from 65536 problems.
 *   "reroll_nominal"  1: the forward pass re-rolls the nominal states it needs for the feedback
 *                     law instead of reading them back.  Automatic: 1 from 32768 problems.
 *   "lds_gain_steps"  upper bound on the horizon steps 1, 2, ... whose gains stay in LDS between
 *                     the backward and the forward pass (automatic: what fits next to four
 *                     wavefronts per CU: 5 in fp64, 10 in fp32 at n = 6, m = 2; a launch of at most
 *                     256 / 512 workgroups puts one / two of them on a CU: what fits 160 / 80 KiB
 *                     then — in fp64 19 of 20 steps up to 16384 problems, 8-10 up to 32768).  Step 0 always
 *                     stays: the forward pass needs only its k_0 (x_0 is common to the nominal
 *                     and the candidate trajectory, K_0 multiplies zeros).
 *   "wave_tail"       chunked solve (i2lqr_set_compaction) only: once a compaction leaves at most
 *                     this many running problems (<= 65536), they are finished by the speculative
 *                     kernel ("speculate"; the one-problem-per-wavefront kernel where
 *                     that is not built), whose iteration latency is ~2.8x lower.  Same
 *                     algorithm, different summation order: results agree with the single launch
 *                     to the solve tolerance (1e-8), not bit for bit (fp32: 1-2 % of the problems
 *                     settle an accept / reject tie the other way and stop at a different
 *                     iteration).  0: off; automatic: 12288 with the speculative kernel, 2048 with
 *                     the one-problem-per-wavefront kernel.
 *   "helper_wavefront"  (the bicycles: fp64 and fp32, with or without stage weights) 1: workgroups of TWO wavefronts — a helper forms
 *                     the part of every backward step that depends on the nominal trajectory only
 *                     (loads, sin / cos, Jacobian entries, barrier exponentials, obstacle term: 30 %
 *                     of an iteration) a step ahead of the main wavefront, on a SIMD the launch
 *                     leaves idle; bit-identical.  Automatic: up to 512 workgroups (32768 problems):
 *                     16384 problems 302 -> 360 M it/s, 32768: 543 -> 649 M (fp32: 460 -> 595, 865 -> 1035 M).  In the chunked solve
 *                     of a larger batch every lane chunk behind the first is enqueued in both forms
 *                     and the live count picks one on the device (<= 32768 survivors: this one).
 *                     0: one wavefront everywhere.
 *   "state_buffers"   (the helper-wavefront form) 1: the forward pass stores the candidate states
 *                     in a second state buffer of the workspace and an accepted problem swaps its
 *                     buffers, instead of storing no candidate states and re-rolling the accepted
 *                     ones (what the launches that sit on the HBM roof do); bit-identical.
 *                     Automatic: up to 256 workgroups (16384 problems: 360 -> 374 M it/s; 32768:
 *                     -3 %, two workgroups share a CU's path to memory).  0: off.
 *   "first_chunk", "chunk_step"  chunked solve only: length of the first chunk (automatic: 8) and
 *                     of the one behind it (automatic: 4) — hand-tuned schedules, there to measure
 *                     the automatic one against (tools/solve_bench.py).  Same results either way.
 *   "fused_compaction"  chunked solve only (round 6).  1 (automatic): the compaction is folded into
 *                     the EXIT of the chunk kernels — a wavefront that has finished its chunk packs
 *                     its still-running problems into the next work set (one atomic add per
 *                     wavefront claims their slots) and scatters the terminated ones to the
 *                     caller's arrays while the other wavefronts still iterate; 0: a k_lane_compact
 *                     launch between every two chunks and a scatter pass at the end (rounds 3-5).
 *                     Same results (the order of a work set's slots differs; nothing depends on it).
 *   "final_round"     chunked solve with the speculative tail only (round 6): the tail kernel of
 *                     round r = value (1: at 8 iterations, 2: at 12, 3: at 24, ...) takes EVERY
 *                     survivor, however many, and runs it to termination; nothing is enqueued behind
 *                     it (the rounds behind the one whose tail ran were three to four empty launches
 *                     each).  Automatic: 3.  0: never (the schedule runs to max_iter as in round 5).
 *                     Same results either way.
 * Problem-major layout, i2lqr_iterate / i2lqr_solve:
 *   "group_lanes"     lanes of a wavefront that work on one problem: 64 (one problem per
 *                     wavefront, each lane one element of the Riccati step's products), 8 (eight
 *                     problems per wavefront, each lane one COLUMN of them; the bicycles) or 16
 *                     (four problems per wavefront, quad12 with Q = R = 0 and a registered
 *                     workspace; automatic then).  8 is built for the bicycle plants with
 *                     Q = R = 0 and horizons whose eight problem slices fit the LDS,
 *                     I2LQR_ERR_UNSUPPORTED otherwise).  Automatic: 8 from 1024 problems where built.
 *                     The two agree to round-off (1e-10 on one backward pass), not bit for bit:
 *                     K^T Quu K is associated differently.
 *   "group_workspace" eight-lane kernel: 1 keeps the per-step records and the gains in the
 *                     registered HBM workspace instead of LDS (4 KB of LDS per problem instead of
 *                     9.5 KB at n = 6, N = 20: four wavefronts per CU instead of two), fetched a
 *                     step (records) / two steps (gains) ahead.  Same arithmetic, bit-identical.
 *                     Automatic: above 4096 problems whenever the workspace is registered (below,
 *                     every wavefront has its CU's LDS to itself either way).
 *   "speculate"       ("group_lanes" 8 / automatic) 1: the speculative form of the eight-lane
 *                     kernel — V wavefronts per eight problems (three up to 512 problems, two
 *                     above and in the tail of the chunked solves), wavefront v runs the iteration
 *                     that follows v rejects (same nominal trajectory, lamb * 10^v:
 *                     control/iterative_ilqr.py:81-82); after each round the outcomes are resolved
 *                     in order and everything behind the first accept is discarded.  Same
 *                     iterations, same order, same arithmetic: bit-identical results; a run of r
 *                     rejects and one accept costs one round instead of r + 1.  Uses SIMDs a small
 *                     batch leaves idle.  A launch lasts as long as its slowest problem, and the
 *                     slowest problems of a solve to termination alternate accepts and rejects:
 *                     i2lqr_solve of 1024 problems 1.21 -> 0.71 ms.  With a FIXED iteration count
 *                     nothing is gained as soon as one problem of the batch never rejects (0.275
 *                     vs 0.215 ms per 10 iterations at 1024 problems).  Automatic: on for
 *                     early-exit calls (i2lqr_solve, i2lqr_iterate with early_exit) of at most
 *                     12288 problems where built, and for the <= "wave_tail" survivors of the
 *                     chunked solves of the lane layouts; off for fixed iteration counts.
 *                     0: never; 1: always (I2LQR_ERR_UNSUPPORTED where not built).
 *   "per_step_jacobians"  ("group_lanes" 64)  1: the [A | B] matrices of all horizon steps (systems/kinetic_bicycle.py:
 *                     30-52) are written to LDS by the parallel per-step phase, so the serial
 *                     Riccati recursion has no Jacobian refresh; doubles the LDS per problem.
 *                     Automatic: on while every wavefront of the launch fits on the chip at once
 *                     (n <= 6 systems).
 *   "stagger"         lane layouts (k_lane_iterate_rows, k_lane_iterate): every second half-thousand
 *                     of workgroups starts value x ~8000 cycles late, so that half of the
 *                     wavefronts stream their gains (forward pass) while the other half computes
 *                     (backward pass).  Automatic: quad12 45 from 65536 problems; bicycles in fp64
 *                     8 for launches of about one wavefront per SIMD (61441 ... 81920 problems:
 *                     larger launches desynchronise by themselves); else 0.  Result-preserving.
 *   "debug_self_test" index-checked debug build only (make -C ilqr_iterative_tasks_amd/csrc debug ->
 *                     libi2lqr_hip_debug.so): provokes one recorded index violation and returns
 *                     what the next call would, I2LQR_ERR_LAUNCH with the decoded record; the
 *                     product library answers I2LQR_ERR_UNSUPPORTED.  In that build every call
 *                     synchronises its stream and reports violations of its LDS slices, HBM
 *                     workspace slots and row addressing as I2LQR_ERR_LAUNCH.
 */
int i2lqr_set_option(i2lqr_handle* h, const char* name, int64_t value);

/* Name of the kernel i2lqr_iterate (fixed iteration count) launches for a batch of B problems
 * with the handle's current options: "k_iterate", "k_group_iterate", "k_group_iterate (sixteen
 * lanes)", "k_group_iterate (workspace form)", "k_group_spec", "k_group_spec (sixteen lanes)",
 * "k_quad_iterate", "k_lane_iterate", "k_lane_iterate_pair" (the helper-wavefront form),
 * "k_lane_iterate_rows"; "unsupported" if a forced option cannot be honoured and the launch would
 * return I2LQR_ERR_UNSUPPORTED: what to look for in a rocprofv3 kernel trace.  The name comes from
 * the launcher's own decision code run on scratch arguments, not from a copy of its conditions.
 * Host only; "" for a NULL handle. */
const char* i2lqr_iterate_kernel(const i2lqr_handle* h, int64_t B);
/* The same for i2lqr_solve / early-exit calls (the dominant kernel; the chunked solves of the lane
 * layouts also launch a tail kernel, and k_lane_compact with "fused_compaction" 0). */
const char* i2lqr_solve_kernel(const i2lqr_handle* h, int64_t B);

/*
 * Nominal rollout + cost — replaces control/iterative_ilqr.py:32-48.
 * In: X[.,:,0] = x0, U.  Out: U clipped in place, X[.,:,1..N], cost[B] (stage cost to xtarget +
 * terminal cost to x_term; barrier terms are NOT part of the cost, as in the reference).
 */
int i2lqr_rollout(i2lqr_handle* h, int64_t B, void* X, void* U, const void* x_term, void* cost,
                  void* stream);

/*
 * Backward pass: dynamics Jacobians, cost quadratisation, Riccati gain recursion — replaces
 * backward_pass() control/iterative_ilqr.py:88-130 with its callees get_A_matrix/get_B_matrix
 * (systems/kinetic_bicycle.py:30-52), get_cost_derivation (control/ilqr_helper.py:9-56),
 * add_control_constraint (:83-103), repelling_cost_function (:59-64), get_cost_final (:106-150).
 * In: X, U (a rolled-out nominal trajectory), x_term, lamb[B], obs (or NULL).
 * Out: K[B][m][n][N], k[B][m][N].
 */
int i2lqr_backward(i2lqr_handle* h, int64_t B, const void* X, const void* U, const void* x_term,
                   const void* lamb, const void* obs, void* K, void* k, void* stream);

/*
 * Forward pass: closed-loop full-step rollout with input clipping and cost — replaces
 * forward_pass() control/iterative_ilqr.py:133-160.
 * Out: X_new, U_new, cost_new[B] (stage cost measured to x_term, as the reference does).
 */
int i2lqr_forward(i2lqr_handle* h, int64_t B, const void* X, const void* U, const void* x_term,
                  const void* K, const void* k, void* X_new, void* U_new, void* cost_new,
                  void* stream);

/*
 * Fused fixed-count iteration — `n_iters` passes of the loop body control/iterative_ilqr.py:29-84
 * (rollout+cost, backward, forward, accept/reject with the lamb schedule) per problem, WITHOUT the
 * two early exits (:78-80, :83-84): every problem executes exactly n_iters iterations.  This is
 * the unit of the throughput metric.  In/out: X (X[.,:,0] = x0 on entry), U, lamb.  Out: cost[B]
 * (cost of the returned trajectory), K, k of the last iteration (may be NULL to skip the stores),
 * iters[B] (int32, = n_iters), status[B] (int32).
 */
int i2lqr_iterate(i2lqr_handle* h, int64_t B, int32_t n_iters, void* X, void* U,
                  const void* x_term, void* lamb, const void* obs, void* cost, void* K, void* k,
                  int32_t* iters, int32_t* status, void* stream);

/*
 * Solve to termination — replaces ilqr() control/iterative_ilqr.py:7-85 for B problems at once:
 * up to cfg.max_iter iterations with both early exits.  Same buffers as i2lqr_iterate; iters[B]
 * is the number of executed iterations, status[B] the exit reason.
 */
int i2lqr_solve(i2lqr_handle* h, int64_t B, void* X, void* U, const void* x_term, void* lamb,
                const void* obs, void* cost, void* K, void* k, int32_t* iters, int32_t* status,
                void* stream);

/*
 * The chained regularisation of the reference's candidate loops, ONE launch (round 6): inside one
 * lap's candidate list `lamb` is carried from candidate to candidate — it is reset per lap
 * (utils/base.py:393) and ilqr() returns the value the next call starts from (:414-426) —, the laps
 * are independent.  `chains` chains of `chain_len` problems each, problem-major arrays of
 * chains * chain_len problems stored chain after chain (problem c of chain a at index a * chain_len
 * + c); lamb[] is READ for the first problem of every chain only and WRITTEN for all of them (what
 * each ilqr() call returned); everything else as i2lqr_solve.  A workgroup of the sixteen-lane
 * speculative kernel solves its chains' problems one after the other: bit-identical to chain_len
 * calls of i2lqr_solve with lamb copied across, without their launches (~20 us each at N = 6).
 * I2LQR_ERR_UNSUPPORTED where that kernel is not built (stage weights, quad12, lane layouts, more
 * than 512 chains, a horizon whose three-wavefront buffers do not fit the LDS): solve the chain
 * steps one after the other instead (the Python host's HipCandidateSolver.solve_chained does).
 */
int i2lqr_solve_chained(i2lqr_handle* h, int64_t chains, int32_t chain_len, void* X, void* U,
                        const void* x_term, void* lamb, const void* obs, void* cost, void* K, void* k,
                        int32_t* iters, int32_t* status, void* stream);

/*
 * Relaxed terminal cost of each candidate — replaces utils/base.py:427-437:
 * smallest i in [1, max_relax_iter] with ||x_N - x_term||_2 <= 80 i / 10^outer_iter gives
 * cost_it = qfun + N + 100 i; a norm above 80 max_relax_iter / 10^outer_iter (or NaN) gives +inf.
 * qfun[B] is int32 (cost-to-go in steps, utils/base.py:346); cost_it[B] is `real`.
 */
int i2lqr_relax_cost(i2lqr_handle* h, int64_t B, const void* X, const void* x_term,
                     const int32_t* qfun, int32_t outer_iter, int32_t max_relax_iter,
                     void* cost_it, void* stream);

/*
 * Flat arg-min over cost_it[B] with first-index tie-break (the reduction the all-gather feeds;
 * utils/base.py:462-465 applies it per lap, see the Python host for the list-of-lists form).
 * Out (device): best_idx[1] (int64), best_cost[1] (`real`).  `workspace` is device scratch of
 * `workspace_bytes` bytes; i2lqr_argmin_workspace_bytes(B) is what B candidates need (it GROWS
 * with B: one (value, index) pair per workgroup of the kernel that carries the pick) and a smaller
 * workspace is refused with I2LQR_ERR_INVALID — the library never writes past the size it was
 * given.  If no element can win — B == 0, or every cost is NaN — best_idx = -1 and best_cost =
 * +inf (Python's min() of an empty list raises instead).
 */
int64_t i2lqr_argmin_workspace_bytes(int64_t B);
int i2lqr_argmin(i2lqr_handle* h, int64_t B, const void* cost_it, int64_t* best_idx,
                 void* best_cost, void* workspace, int64_t workspace_bytes, void* stream);

/*
 * One control round in one call — i2lqr_iterate followed by i2lqr_relax_cost on the returned X and
 * (best_idx != NULL) i2lqr_argmin over cost_it, with exactly their outputs, bit for bit.  Replaces
 * the body of the candidate loops utils/base.py:414-437 plus the flat form of the pick :462-465.
 * Where the eight-lane kernels run (problem-major layout, the bicycles with Q = R = 0, from 1024
 * problems) the relaxed cost is formed in the kernel's exit block from the x_N it still holds in
 * LDS and the pick is a last-workgroup-done reduction of per-wavefront minima inside the same
 * launch: ONE launch per round instead of four (at 1024 problems the three small launches and
 * their dependent launch boundaries cost 10 % of the round).  Every other kernel family runs the
 * three steps as separate launches behind this call.
 * best_idx NULL: costs only (the sharded path: the all-gather sits between the costs and the
 * pick); then best_cost and workspace are not read.  `workspace` / `workspace_bytes`: as for
 * i2lqr_argmin (at least i2lqr_argmin_workspace_bytes(B), checked).
 * Concurrency: a handle serves ONE stream at a time.  Calls on one handle from
 * two host threads do not share host state (the epilogue of a call is thread-local), and launches
 * of one handle that overlap on two streams draw their last-workgroup-done tickets from different
 * device words as long as fewer than 16 are in flight; each such call must be given its OWN
 * workspace (the partial minima live there).
 */
int i2lqr_iterate_pick(i2lqr_handle* h, int64_t B, int32_t n_iters, void* X, void* U,
                       const void* x_term, void* lamb, const void* obs, void* cost, void* K, void* k,
                       int32_t* iters, int32_t* status, const int32_t* qfun, int32_t outer_iter,
                       int32_t max_relax_iter, void* cost_it, int64_t* best_idx, void* best_cost,
                       void* workspace, int64_t workspace_bytes, void* stream);

/*
 * The one collective of the path (multi-GPU): all-gather of the per-candidate
 * terminal costs, one RCCL ncclAllGather over xGMI.  Rank r owns a contiguous shard of n_local
 * candidates; afterwards every rank holds cost_all[world * n_local] in rank order and evaluates
 * the pick (utils/base.py:462-469; i2lqr_argmin for the flat form) locally.  The reference runs
 * its candidates sequentially in one process and has no counterpart.
 *
 * `comm` is an RCCL communicator (ncclComm_t passed as void*): either the caller's own, or one
 * made by the helpers below, which wrap ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy so
 * that a host language needs no second binding.  i2lqr_comm_unique_id fills I2LQR_COMM_ID_BYTES
 * bytes on ONE rank; the caller distributes them to all ranks (any side channel) and every rank
 * calls i2lqr_comm_create(id, world, rank, &comm) with its GPU current (hipSetDevice).  RCCL is
 * bound at run time: a librccl already loaded in the process is used, else the ROCm one;
 * I2LQR_ERR_UNSUPPORTED if neither is available.
 * cost_local[n_local] and cost_all[world * n_local] are device buffers of `real`; the collective
 * is enqueued on `stream`.  n_local must be the SAME on every rank (ncclAllGather semantics: a
 * mismatch is not detectable from inside one rank); ragged shards are padded by the caller to the
 * largest shard with +inf, which never wins the arg-min (the Python host does that:
 * CostExchange.allgather(total=...)).
 */
#define I2LQR_COMM_ID_BYTES 128
/* I2LQR_OK if RCCL can be bound in this process (no communicator is created, no GPU touched):
 * every rank checks this and the ranks agree on the answer BEFORE any of them enters
 * i2lqr_comm_create, whose bootstrap blocks until all `world` ranks have arrived. */
int i2lqr_comm_available(void);
int i2lqr_comm_unique_id(void* id);
int i2lqr_comm_create(const void* id, int32_t world, int32_t rank, void** comm);
int i2lqr_comm_destroy(void* comm);
/* ncclCommAbort: frees the communicator WITHOUT the collective teardown ncclCommDestroy performs —
 * the call for a communicator whose peers did not all come up (a destroy may wait for them). */
int i2lqr_comm_abort(void* comm);
int i2lqr_comm_info(void* comm, int32_t* world, int32_t* rank);
int i2lqr_allgather_costs(i2lqr_handle* h, void* comm, const void* cost_local, void* cost_all,
                          int64_t n_local, void* stream);

/*
 * The hand-off that follows the pick on a sharded batch: after the all-gather every
 * rank knows WHICH candidate won, only the rank that solved it holds its trajectory — and the
 * reference goes on with exactly that trajectory: it applies u_pred[:, 0] and feeds x_pred[:, -1] to
 * the next round (utils/base.py:466-471).  One ncclBroadcast of `count` reals, in place, from the
 * owning rank `root` (the winner's U[m][N] and X[n][N+1] packed by the caller: <= 6.5 KB at n = 12,
 * N = 50) on the communicator of i2lqr_allgather_costs.  Every rank calls it with the same count and
 * root.  (The alternative — every rank re-solves the winner — costs a whole solve per round.)
 */
int i2lqr_broadcast_winner(i2lqr_handle* h, void* comm, void* buf, int64_t count, int32_t root,
                           void* stream);

/*
 * The same two steps WITHOUT a host round trip, for rounds whose pick is the flat arg-min (the
 * synthetic batches; any caller that does not need the list-of-lists order): the root of a broadcast
 * must be known on the host, i.e. the host would have to read the pick back before it could enqueue
 * the hand-off.  Instead every rank packs the trajectory of its LOCAL winner (the global winner is
 * the local winner of the rank that owns it) and the packs ride along with the costs:
 *
 * i2lqr_allgather_round — ncclAllGather of cost_local[n_local] into cost_all[world * n_local] and
 *   ncclAllGather of pack_local[pack_count] into pack_all[world * pack_count] as ONE grouped RCCL
 *   operation (ncclGroupStart / ncclGroupEnd) on `stream`.  Same n_local / pack_count on every rank
 *   (ragged shards: pad the costs with +inf).  No counterpart in the reference (one process).
 * i2lqr_round_winner — after i2lqr_argmin over cost_all (best_padded[1], device): copies the owner's
 *   pack to winner[pack_count] and writes best_global[2] = {index of the winner in the unpadded
 *   batch of `total` candidates sharded contiguously over `world` ranks (first ranks take the
 *   remainder), owner rank}; `width` = n_local of the all-gather.  Replaces, together with the
 *   gather, utils/base.py:462-471 (pick + "go on with the winner's trajectory") on every rank.
 * i2lqr_pack_problem — pack[m N + n (N+1)] = (U[m][N], X[n][N+1]) of problem idx[0] (device int64,
 *   e.g. a pick; clamped into [0, B)) in the reference's orientation, from X / U in the handle's
 *   layout: what a rank contributes as pack_local, and what the owner broadcasts with
 *   i2lqr_broadcast_winner.  (The deep copy of the winner, utils/base.py:453-455, :466-469.)
 * Everything stays enqueued on the stream: the next round can be launched behind it.
 */
int i2lqr_pack_problem(i2lqr_handle* h, int64_t B, const void* X, const void* U, const int64_t* idx,
                       void* pack, void* stream);
int i2lqr_allgather_round(i2lqr_handle* h, void* comm, const void* cost_local, void* cost_all,
                          int64_t n_local, const void* pack_local, void* pack_all,
                          int64_t pack_count, void* stream);
int i2lqr_round_winner(i2lqr_handle* h, int32_t world, int64_t width, int64_t total,
                       int64_t pack_count, const int64_t* best_padded, const void* pack_all,
                       void* winner, int64_t* best_global, void* stream);

/*
 * ONE sharded control round with the flat pick as ONE call (round 6): what a rank does
 * for utils/base.py:391-471 — the candidate loops over its shard, the pick over all ranks' costs and
 * "go on with the winner's trajectory" — enqueued from C without a host step between the launches:
 *
 *   on `stream`       [guard_previous: wait for the side stream of the previous round on this handle]
 *                     n_iters >= 0: i2lqr_iterate_pick(B, n_iters, ..., cost_it, local_best, ...)
 *                     n_iters <  0: i2lqr_solve + i2lqr_relax_cost + i2lqr_argmin
 *                     event record
 *   on `side_stream`  wait for that event; pack of the LOCAL winner (i2lqr_pack_problem) and, for a
 *                     ragged split, the +inf padding of the costs to `width` = ceil(total / world)
 *                     (one launch); the grouped all-gather of costs and packs
 *                     (i2lqr_allgather_round); the pick over the world x width gathered costs and
 *                     the winner's hand-off (i2lqr_round_pick); event record.
 *
 * With side_stream != stream the launch stream carries the shard's solve only — the same single
 * launch per round as the unsharded i2lqr_iterate_pick — and the exchange of round i runs beside
 * the solve of round i + 1 — provided the two streams do not share a HARDWARE queue (two HIP
 * streams may: the exchange then runs behind the next solve, +10 % per step measured; probe the
 * candidate with a spin kernel as the Python host's dist.exchange_stream() does, and do not make
 * it a high-priority stream: +60 % at 1024 problems).  side_stream NULL (or == stream): everything
 * on `stream`.
 * comm NULL is a world of one without RCCL (the gathers become device copies), or any world with
 * round->loopback set (see there).
 * A rank with B == 0 (more ranks than candidates) takes part in the exchange with `width` costs of
 * +inf and a pack of zeros: no rank raises while the others wait in the collective.
 * Buffers: every pointer of i2lqr_round is device memory owned by the caller and must stay valid
 * and UNTOUCHED by other work until the side stream has passed this round.  A caller that reuses
 * one buffer set round after round sets guard_previous = 1: the round then starts behind the
 * previous round's side stream (handle-owned event) instead of racing it; a caller that gives
 * every round in flight its own buffers (bench.py) sets 0 and keeps the overlap.
 * Outputs, identical on every rank: best_global[2] = {index of the winner in the unpadded batch of
 * `total` candidates (-1: nothing can win), owner rank}, best_cost[1], winner[P] = the winner's
 * (U[m][N], X[n][N+1]), P = m N + n (N + 1); cost_all[world * width] (padded with +inf per rank).
 * Results are bit-identical to the five separate calls (tests/test_gpu_round6.py).
 *
 * i2lqr_round_pick — i2lqr_argmin over cost_all[world * width] followed by i2lqr_round_winner, as
 *   ONE launch up to 16384 gathered costs (two + one above); `workspace` as for i2lqr_argmin over
 *   world * width elements plus 16 bytes.
 */
typedef struct i2lqr_round {
  int32_t struct_size;      /* = sizeof(i2lqr_round) */
  int32_t n_iters;          /* fused iterations per candidate; < 0: solve to termination */
  int64_t B;                /* candidates of THIS rank: shard_range(total, rank, world) */
  int64_t total;            /* candidates of the round over all ranks (>= 1) */
  int32_t world, rank;      /* of `comm` */
  int32_t outer_iter, max_relax_iter; /* as i2lqr_relax_cost */
  int32_t guard_previous;   /* 1: start behind the previous round's side stream (see above) */
  int32_t loopback;         /* 0; 1: NO collective — this rank's costs and pack are copied into
                               ITS slots of cost_all / pack_all and the other slots stay as the
                               caller left them: one process can play the ranks of a world one
                               after the other on shared gather buffers (tests; comm unused) */
  /* the shard: as i2lqr_iterate_pick (all NULL-able ones may be NULL; all unused when B == 0) */
  void* X; void* U; const void* x_term; void* lamb; const void* obs; void* cost;
  void* K; void* k; int32_t* iters; int32_t* status; const int32_t* qfun;
  void* cost_it;            /* [B] out */
  int64_t* local_best;      /* [1] out: the shard's pick */
  void* local_best_cost;    /* [1] out */
  void* pick_ws; int64_t pick_ws_bytes;   /* >= i2lqr_argmin_workspace_bytes(B) */
  /* the exchange */
  void* pack_local;         /* [P] */
  void* cost_padded;        /* [width]; may be NULL when B == width */
  void* cost_all;           /* [world * width] out */
  void* pack_all;           /* [world * P] */
  void* side_ws; int64_t side_ws_bytes;   /* >= i2lqr_argmin_workspace_bytes(world * width) + 16 */
  void* best_cost;          /* [1] out */
  void* winner;             /* [P] out */
  int64_t* best_global;     /* [2] out */
} i2lqr_round;
int i2lqr_sharded_round_flat(i2lqr_handle* h, void* comm, const i2lqr_round* round,
                             void* side_stream, void* stream);
int i2lqr_round_pick(i2lqr_handle* h, int32_t world, int64_t width, int64_t total,
                     int64_t pack_count, const void* cost_all, const void* pack_all,
                     void* best_cost, void* winner, int64_t* best_global, void* workspace,
                     int64_t workspace_bytes, void* stream);

/*
 * Controller round on the device (problem-major layout).
 *
 * i2lqr_select_candidates — replaces iLqr.select_close_ss (utils/base.py:332-341) and the
 * candidate set-up (:405-412) for L laps at once: the k nearest safe-set columns to x_guess in the
 * 1-norm (ascending, ties to the lower column), their states as x_term and their cost-to-go.
 *   ss[L][n][Tmax] (`real`, component-major, time contiguous, padded), T[L], qfun[L][Tmax] (int32),
 *   x_guess: element i at x_guess[i * guess_stride]  (so X_pred[:, N] can be passed in place).
 *   out: idx[L][k] (int32), x_term[L*k][n], qf[L*k] (int32).   Tmax <= 1024, k <= Tmax.
 *   A lap with T[l] < k has only T[l] candidates (the reference's argsort()[0:k] returns fewer):
 *   its surplus slots get idx = -1, qf = I2LQR_QF_NONE and a copy of the lap's last state, and
 *   i2lqr_relax_cost gives a slot with qfun == I2LQR_QF_NONE the cost +inf, so it is never picked.
 * i2lqr_init_candidates — uvar = 0, xvar[:, 0] = x0, lamb = lamb0 for B candidates (:393, :405-408).
 * i2lqr_pick_best — the pick of utils/base.py:462-469 on cost_it[L][k] (lexicographic over the
 *   laps' lists, then first minimum); copies the winner's X, U to x_pred[n][N+1], u_pred[m][N];
 *   best[2] = {lap position, candidate position} (int32, device).  X, U, x_pred, u_pred may ALL be
 *   NULL: the pick alone — what a sharded round runs on the gathered cost vector, the winner's
 *   trajectory being on the rank that solved it (i2lqr_broadcast_winner hands it over).
 */
int i2lqr_select_candidates(i2lqr_handle* h, int32_t L, int32_t Tmax, const void* ss,
                            const int32_t* T, const int32_t* qfun, const void* x_guess,
                            int32_t guess_stride, int32_t k, int32_t* idx, void* x_term,
                            int32_t* qf, void* stream);
int i2lqr_init_candidates(i2lqr_handle* h, int64_t B, const void* x0, double lamb0, void* X,
                          void* U, void* lamb, void* stream);
int i2lqr_pick_best(i2lqr_handle* h, int32_t L, int32_t k, const void* cost_it, const void* X,
                    const void* U, int32_t* best, void* x_pred, void* u_pred, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* I2LQR_H */

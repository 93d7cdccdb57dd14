// Launchers of the eight- / sixteen-lanes-per-problem kernels (i2lqr_group.hpp, i2lqr_quad.hpp),
// compiled in their own translation units (i2lqr_group.hip, i2lqr_quad.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/i2lqr.h"
#include "i2lqr_wave.hpp"

namespace i2lqr {

// true if the configuration can run on the eight-lane kernel (plant, Q = R = 0, LDS budget)
bool group_supported(const i2lqr_config& cfg);
// Enqueue k_group_iterate for B problems (problem-major layout).  Returns hipSuccess or the HIP
// error of the attribute call / launch.
template <class T> hipError_t group_iterate(const i2lqr_config& cfg, const IterArgs<T>& a,
                                            hipStream_t stream);

// The same column scheme with SIXTEEN lanes per problem — one problem per 16-lane DPP row, four per
// wavefront — whose backward step exchanges columns by DPP row broadcasts instead of LDS round trips
// (i2lqr_group.hpp: GroupWorker::backward_row): the latency form, for batches that leave every
// wavefront a SIMD of its own (up to kGroup16Batch problems = one round of 1024 wavefronts), and —
// in rounds of kGroup16Batch — wherever whole rounds beat the other forms (above kGroupWsTop).
constexpr int64_t kGroup16Batch = 4096;
bool group16_supported(const i2lqr_config& cfg);
template <class T> hipError_t group16_iterate(const i2lqr_config& cfg, const IterArgs<T>& a,
                                              hipStream_t stream);

// Workspace form of the same kernel (records and gains in a caller-provided HBM workspace of
// group_workspace_bytes() for B problems: 4 KB of LDS per problem, four wavefronts per CU): the
// choice above kGroupWsBatch problems.  0 bytes if the configuration is not supported.
constexpr int64_t kGroupWsBatch = 4096;
// ... up to kGroupWsTop problems: the LDS form of the sixteen-lane kernel runs in rounds of 4096
// problems (0.168 ms each), the workspace form's time grows with the batch (0.25 ms + 25 us per
// 1024 problems): 5120 problems 0.285 against 0.330 ms, 6144: 0.306 / 0.339, 8192: 0.354 / 0.348,
// 12288: 0.583 / 0.502, 20480: 0.872 / 0.804 (tools/_diag/mid_check.py).
// Round 5 (tools/threshold_sweep.py, three shapes): at 8192 problems in fp64 the workspace form is
// ahead at every shape measured (bicycle6 N=20: 0.313 against 0.337 ms, bicycle4 N=6: 0.106 / 0.136,
// bicycle4 N=20: 0.279 / 0.314), in fp32 it is behind there (0.250 / 0.197): the top of the range
// depends on the precision.
constexpr int64_t kGroupWsTop = 8192;      // fp64
constexpr int64_t kGroupWsTopF32 = 7168;
inline int64_t group_ws_top(const i2lqr_config& cfg) {
  return cfg.dtype == I2LQR_F64 ? kGroupWsTop : kGroupWsTopF32;
}
int64_t group_workspace_bytes(const i2lqr_config& cfg, int64_t B);
template <class T> hipError_t group_iterate_ws(const i2lqr_config& cfg, const IterArgs<T>& a,
                                               void* ws, hipStream_t stream);

// The speculative form (k_group_spec: V wavefronts per workgroup, wavefront v runs the iteration that
// follows v rejects): small batches only.  lanes = 8 (eight problems per workgroup, LDS exchanges)
// or 16 (four problems per workgroup, the DPP passes of the sixteen-lane form: shorter rounds, and
// three workgroups fit a CU's LDS instead of one).
bool group_spec_supported(const i2lqr_config& cfg, int lanes);
template <class T> hipError_t group_spec_iterate(const i2lqr_config& cfg, const IterArgs<T>& a,
                                                 hipStream_t stream, int lanes);
// Chains (k_group_spec<.., CHAIN>): a.B chains of a.chain_len problems each, stored chain after chain,
// solved in ONE launch — the final lamb of a chain's problem c is the initial lamb of its problem
// c + 1 (utils/base.py:393, :414-426: the controller's chained regularisation).  Supported for the
// plants of the speculative kernel with Q = R = 0 in the problem-major layout, up to the batch of
// its three-wavefront form (512 chains on 256 CUs).
bool group_spec_chain_supported(const i2lqr_config& cfg, int64_t chains);
template <class T> hipError_t group_spec_chain(const i2lqr_config& cfg, const IterArgs<T>& a,
                                               hipStream_t stream);
// The same kernel on the first *a.count (<= a.count_max) columns of a batch-minor work set: the tail
// of the chunked solves of the one-problem-per-lane layouts (IterArgs::count / set_stride /
// max_total).  Supported for the plants of the eight-lane kernel with Q = R = 0, whatever the
// handle's layout is; sixteen lanes per problem where that fits the LDS.
bool group_spec_tail_supported(const i2lqr_config& cfg);
template <class T> hipError_t group_spec_tail(const i2lqr_config& cfg, const IterArgs<T>& a,
                                              hipStream_t stream);

// Sixteen lanes per problem (i2lqr_quad.hpp; the n + m = 16 plant quad12, Q = R = 0): needs a
// caller-provided HBM workspace of quad_workspace_bytes() for B problems.
bool quad_supported(const i2lqr_config& cfg);
int64_t quad_workspace_bytes(const i2lqr_config& cfg, int64_t B);
template <class T> hipError_t quad_iterate(const i2lqr_config& cfg, const IterArgs<T>& a, void* ws,
                                           hipStream_t stream);

}  // namespace i2lqr

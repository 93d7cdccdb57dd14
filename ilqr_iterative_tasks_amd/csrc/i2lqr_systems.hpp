// Plant models of the batched iLQR solver, device side (gfx950).
//
// Each system provides, for state dimension n and input dimension m:
//   step()     x_{t+1} = f(x_t, u_t)
//   trig()     the transcendental values the Jacobians need at the evaluation state
//              (computed once per horizon step, in parallel over t, and cached in LDS)
//   jac_var()  the NVAR state/input-dependent entries of F = [A | B]  (n x (n+m), row-major)
//   var_idx()  flat index into F of varying entry v
//   jac_const() value of a non-varying entry of F (identity / dt pattern)
//
// Reference: systems/kinetic_bicycle.py:10-52 (kinetic_bicycle, get_A_matrix, get_B_matrix).
// The reference evaluates the Jacobians of step t with v, theta of x_{t+1} and accel of u_t
// (control/iterative_ilqr.py:92-99); `xe` below is that evaluation state for every system.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

namespace i2lqr {

// sin and cos of one argument.  fp64: Cody-Waite reduction by pi/2 in three FMA steps (exact for
// |x| < 2^20 pi/2) and the classic degree-13 / degree-14 minimax kernels on [-pi/4, pi/4]
// (<= ~1 ulp each); about a third of the instructions of the general-range library routine, which
// stays as the fallback for |x| >= 1e5.  fp32: the same scheme in single precision (degree-7 /
// degree-8 kernels, ~1 ulp, fallback for |x| >= 1e4).
template <class T> __device__ __forceinline__ void t_sincos(T x, T* s, T* c);
// t_sincos_fast: the short kernel alone; *big is set (never cleared) if the argument is outside its
// range (|x| >= 1e5, NaN) and the result must not be used.  Callers that loop over a horizon run
// the loop with it and repeat the whole loop with t_sincos if any argument was out of range: the
// common case has no branch inside the loop at all, so the compiler can interleave the polynomial
// with the neighbouring arithmetic (a branch ends the scheduling region, and a TAKEN branch costs
// a wavefront alone on its SIMD ~100 cycles: tools/ubench_issue.hip).
// SLIT (the one-problem-per-lane kernels of the row-block plants, where registers are the scarce
// resource): every fp64 literal is placed in SCALAR registers at its point of use.  A v_fma_f64
// cannot encode a 64-bit literal, and the compiler otherwise materialises each one in a vector
// register pair, hoists the pair out of the horizon loop as an invariant — ~35 literals of exp /
// sincos = 70 vector registers — and spills it to scratch.  Same values, same arithmetic.
// WHERE and WHEN: the literal is moved into its scalar pair by the two s_mov_b32 of t_lit_at's
// statement, which the compiler may not place before `dep` has been computed (the statement names
// it as an input and does not read it).  With the previous Horner accumulator as `dep` the two
// moves sit between a multiply-add and the one that depends on it — issue slots a wavefront alone
// on its SIMD cannot use anyway — and the pair is live for one step.  (A literal merely hidden
// behind an empty asm was formed at the top of the loop with the ~30 others of the horizon step;
// they outran the scalar registers and were parked in vector-register lanes: 236 v_readlane /
// v_writelane and their s_nop per step.)
template <bool SLIT, unsigned long long BITS>
__device__ __forceinline__ double t_lit_at(double dep) {
  if constexpr (SLIT) {
    unsigned lo, hi;
    asm volatile("s_mov_b32 %0, %2\n\ts_mov_b32 %1, %3"
                 : "=s"(lo), "=s"(hi)
                 : "n"((unsigned)(BITS & 0xffffffffull)), "n"((unsigned)(BITS >> 32)), "v"(dep));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
  } else {
    return __builtin_bit_cast(double, BITS);
  }
}
#define I2LQR_LIT(SLIT, value, dep) \
  t_lit_at<SLIT, __builtin_bit_cast(unsigned long long, (double)(value))>(dep)
template <class T> __device__ __forceinline__ void t_sincos_fast(T x, T* s, T* c, bool* big_out);
// NS arguments at once: the same operations per argument as for one (bit for bit), every literal
// formed once per Horner step and shared by the NS chains
template <bool SLIT, int NS>
__device__ __forceinline__ void t_sincos_fast_dn(const double (&x)[NS], double (&s)[NS],
                                                 double (&c)[NS], bool* big_out) {
  double xs[NS], kf[NS], r[NS], z[NS], ps[NS], pc[NS];
  bool big[NS];
  const double lim = I2LQR_LIT(SLIT, 1.0e5, x[0]);
#pragma unroll
  for (int q = 0; q < NS; q++) {
    big[q] = !(__builtin_fabs(x[q]) < lim);
    *big_out = *big_out || big[q];
    xs[q] = big[q] ? 0.0 : x[q];
  }
  const double tpi = I2LQR_LIT(SLIT, 6.36619772367581382433e-01, xs[0]);  // 2/pi
#pragma unroll
  for (int q = 0; q < NS; q++) kf[q] = __builtin_rint(xs[q] * tpi);
  const double p2h = I2LQR_LIT(SLIT, 1.57079632679489655800e+00, kf[0]);  // pi/2 hi
#pragma unroll
  for (int q = 0; q < NS; q++) r[q] = __builtin_fma(-kf[q], p2h, xs[q]);
  const double p2m = I2LQR_LIT(SLIT, 6.12323399573676603587e-17, r[0]);   // pi/2 mid
#pragma unroll
  for (int q = 0; q < NS; q++) r[q] = __builtin_fma(-kf[q], p2m, r[q]);
  const double p2l = I2LQR_LIT(SLIT, -1.49738490485916983327e-33, r[0]);  // pi/2 lo
#pragma unroll
  for (int q = 0; q < NS; q++) {
    r[q] = __builtin_fma(-kf[q], p2l, r[q]);
    z[q] = r[q] * r[q];
  }
  const double s0 = I2LQR_LIT(SLIT, 1.58969099521155010221e-10, z[0]);
  const double c0 = I2LQR_LIT(SLIT, -1.13596475577881948265e-11, z[0]);
#pragma unroll
  for (int q = 0; q < NS; q++) {
    ps[q] = s0;
    pc[q] = c0;
  }
#define I2LQR_SC_STEP(ls, lc)                                                                \
  {                                                                                          \
    const double ks = I2LQR_LIT(SLIT, ls, pc[0]);                                            \
    _Pragma("unroll") for (int q = 0; q < NS; q++) ps[q] = __builtin_fma(ps[q], z[q], ks);   \
    const double kc = I2LQR_LIT(SLIT, lc, ps[0]);                                            \
    _Pragma("unroll") for (int q = 0; q < NS; q++) pc[q] = __builtin_fma(pc[q], z[q], kc);   \
  }
  I2LQR_SC_STEP(-2.50507602534068634195e-08, 2.08757232129817482790e-09)
  I2LQR_SC_STEP(2.75573137070700676789e-06, -2.75573143513906633035e-07)
  I2LQR_SC_STEP(-1.98412698298579493134e-04, 2.48015872894767294178e-05)
  I2LQR_SC_STEP(8.33333333332248946124e-03, -1.38888888888741095749e-03)
  I2LQR_SC_STEP(-1.66666666666666324348e-01, 4.16666666666666019037e-02)
#undef I2LQR_SC_STEP
#pragma unroll
  for (int q = 0; q < NS; q++) {
    const int qd = (int)kf[q];
    const double sr = __builtin_fma(ps[q] * z[q], r[q], r[q]);
    const double cr = __builtin_fma(pc[q] * z[q], z[q], __builtin_fma(-0.5, z[q], 1.0));
    const bool swap = qd & 1;
    const double sv = swap ? cr : sr, cv = swap ? sr : cr;
    s[q] = (qd & 2) ? -sv : sv;
    c[q] = ((qd + 1) & 2) ? -cv : cv;
  }
}
template <bool SLIT>
__device__ __forceinline__ void t_sincos_fast_d(double x, double* s, double* c, bool* big_out) {
  const double xa[1] = {x};
  double sa[1], ca[1];
  t_sincos_fast_dn<SLIT, 1>(xa, sa, ca, big_out);
  *s = sa[0];
  *c = ca[0];
}
template <> __device__ __forceinline__ void t_sincos_fast<double>(double x, double* s, double* c,
                                                                   bool* big_out) {
  t_sincos_fast_d<false>(x, s, c, big_out);
}
// t_sincos_fast with the Horner steps written as THREE-address multiply-adds (v_fma_f64 d, p, z, k).
// Left to itself the compiler picks the two-address v_fmac_f64 and copies every coefficient — a
// loop-invariant vector register pair — into the accumulator first: 12 v_mov_b64 per evaluation
// in the sixteen-lane forward pass (GroupWorker::forward_row).  Same operations, bit for bit.
__device__ __forceinline__ double t_fma3(double a, double b, double c) {
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ void t_sincos_fast_h3(double x, double* s, double* c, bool* big_out) {
  const bool big = !(__builtin_fabs(x) < 1.0e5);
  *big_out = *big_out || big;
  const double xs = big ? 0.0 : x;
  const double kf = __builtin_rint(xs * 6.36619772367581382433e-01);
  double r = __builtin_fma(-kf, 1.57079632679489655800e+00, xs);
  r = __builtin_fma(-kf, 6.12323399573676603587e-17, r);
  r = __builtin_fma(-kf, -1.49738490485916983327e-33, r);
  const double z = r * r;
  double ps = 1.58969099521155010221e-10, pc = -1.13596475577881948265e-11;
#define I2LQR_SC_STEP3(ls, lc) \
  ps = t_fma3(ps, z, ls);      \
  pc = t_fma3(pc, z, lc);
  I2LQR_SC_STEP3(-2.50507602534068634195e-08, 2.08757232129817482790e-09)
  I2LQR_SC_STEP3(2.75573137070700676789e-06, -2.75573143513906633035e-07)
  I2LQR_SC_STEP3(-1.98412698298579493134e-04, 2.48015872894767294178e-05)
  I2LQR_SC_STEP3(8.33333333332248946124e-03, -1.38888888888741095749e-03)
  I2LQR_SC_STEP3(-1.66666666666666324348e-01, 4.16666666666666019037e-02)
#undef I2LQR_SC_STEP3
  const int qd = (int)kf;
  const double sr = __builtin_fma(ps * z, r, r);
  const double cr = __builtin_fma(pc * z, z, __builtin_fma(-0.5, z, 1.0));
  const bool swap = qd & 1;
  const double sv = swap ? cr : sr, cv = swap ? sr : cr;
  *s = (qd & 2) ? -sv : sv;
  *c = ((qd + 1) & 2) ? -cv : cv;
}
template <> __device__ __forceinline__ void t_sincos<double>(double x, double* s, double* c) {
  // The short kernel runs unconditionally and the library routine, which only arguments of
  // |x| >= 1e5 (or NaN) need, sits behind a WAVE-UNIFORM unlikely branch: the common case falls
  // through instead of jumping over ~170 instructions of Payne-Hanek reduction in every call.
  bool big = false;
  double so, co;
  t_sincos_fast(x, &so, &co, &big);
  if (__builtin_expect(__any(big), 0)) {
    double sl, cl;
    sincos(x, &sl, &cl);
    so = big ? sl : so;
    co = big ? cl : co;
  }
  *s = so;
  *c = co;
}
template <> __device__ __forceinline__ void t_sincos_fast<float>(float x, float* s, float* c,
                                                                  bool* big_out) {
  const bool big = !(__builtin_fabsf(x) < 1.0e4f);
  *big_out = *big_out || big;
  const float xs = big ? 0.0f : x;
  const float kf = __builtin_rintf(xs * 6.36619772367581382433e-01f);
  float r = __builtin_fmaf(-kf, 1.57079637050628662109375f, xs);
  r = __builtin_fmaf(-kf, -4.37113900018624283e-8f, r);
  r = __builtin_fmaf(-kf, -1.71512468793638e-15f, r);
  const int q = (int)kf;
  const float z = r * r;
  float ps = -1.9515295891e-4f;
  ps = __builtin_fmaf(ps, z, 8.3321608736e-3f);
  ps = __builtin_fmaf(ps, z, -1.6666654611e-1f);
  const float sr = __builtin_fmaf(ps * z, r, r);
  float pc = 2.443315711809948e-5f;
  pc = __builtin_fmaf(pc, z, -1.388731625493765e-3f);
  pc = __builtin_fmaf(pc, z, 4.166664568298827e-2f);
  const float cr = __builtin_fmaf(pc * z, z, __builtin_fmaf(-0.5f, z, 1.0f));
  const bool swap = q & 1;
  const float sv = swap ? cr : sr, cv = swap ? sr : cr;
  *s = (q & 2) ? -sv : sv;
  *c = ((q + 1) & 2) ? -cv : cv;
}
__device__ __forceinline__ void t_sincos_fast_h3(float x, float* s, float* c, bool* big_out) {
  t_sincos_fast<float>(x, s, c, big_out);
}
template <> __device__ __forceinline__ void t_sincos<float>(float x, float* s, float* c) {
  bool big = false;
  float so, co;
  t_sincos_fast(x, &so, &co, &big);
  if (__builtin_expect(__any(big), 0)) {
    float sl, cl;
    sincosf(x, &sl, &cl);
    so = big ? sl : so;
    co = big ? cl : co;
  }
  *s = so;
  *c = co;
}
template <class T> __device__ __forceinline__ T t_exp(T x);
// fp64 exp: x = k ln2 + r (Cody-Waite), degree-13 Horner on |r| <= ln2/2, v_ldexp (<= ~2 ulp;
// overflow / underflow / NaN through ldexp and the clamped exponent).
// exp of NE arguments at once with the SLIT literals shared: e[q] = t_exp(x[q]) for q < NE - NB,
// t_exp_bounded(x[q]) for the last NB (bit for bit the single-argument functions below)
template <bool SLIT, int NE, int NB>
__device__ __forceinline__ void t_exp_d(const double (&x0)[NE], double (&e)[NE]) {
  double x[NE], kf[NE], r[NE], p[NE];
#pragma unroll
  for (int q = 0; q < NE; q++)
    x[q] = q < NE - NB ? (x0[q] < I2LQR_LIT(SLIT, -746.0, x0[q]) ? -746.0
                          : (x0[q] > I2LQR_LIT(SLIT, 710.0, x0[q]) ? 710.0 : x0[q]))
                       : x0[q];  // NaN passes through
  const double l2e = I2LQR_LIT(SLIT, 1.44269504088896338700e+00, x[0]);
#pragma unroll
  for (int q = 0; q < NE; q++) kf[q] = __builtin_rint(x[q] * l2e);
  const double ln2h = I2LQR_LIT(SLIT, 6.93147180369123816490e-01, kf[0]);
#pragma unroll
  for (int q = 0; q < NE; q++) r[q] = __builtin_fma(-kf[q], ln2h, x[q]);
  const double ln2l = I2LQR_LIT(SLIT, 1.90821492927058770002e-10, r[0]);
#pragma unroll
  for (int q = 0; q < NE; q++) r[q] = __builtin_fma(-kf[q], ln2l, r[q]);
  const double c13 = I2LQR_LIT(SLIT, 1.6059043836821614599e-10, r[0]);  // 1/13!
#pragma unroll
  for (int q = 0; q < NE; q++) p[q] = c13;
#define I2LQR_EXP_STEP(lit)                                          \
  {                                                                  \
    const double ck = I2LQR_LIT(SLIT, lit, p[0]);                    \
    _Pragma("unroll") for (int q = 0; q < NE; q++) p[q] = __builtin_fma(p[q], r[q], ck); \
  }
  I2LQR_EXP_STEP(2.0876756987868098979e-09)
  I2LQR_EXP_STEP(2.5052108385441718775e-08)
  I2LQR_EXP_STEP(2.7557319223985890653e-07)
  I2LQR_EXP_STEP(2.7557319223985892511e-06)
  I2LQR_EXP_STEP(2.4801587301587301566e-05)
  I2LQR_EXP_STEP(1.9841269841269841253e-04)
  I2LQR_EXP_STEP(1.3888888888888889419e-03)
  I2LQR_EXP_STEP(8.3333333333333332177e-03)
  I2LQR_EXP_STEP(4.1666666666666664354e-02)
  I2LQR_EXP_STEP(1.6666666666666665741e-01)
#undef I2LQR_EXP_STEP
#pragma unroll
  for (int q = 0; q < NE; q++) {
    p[q] = __builtin_fma(p[q], r[q], 0.5);
    p[q] = __builtin_fma(p[q], r[q], 1.0);
    p[q] = __builtin_fma(p[q], r[q], 1.0);
    const double v = __builtin_ldexp(p[q], (int)kf[q]);
    e[q] = q < NE - NB ? (x0[q] < I2LQR_LIT(SLIT, -745.14, v) ? 0.0
                          : (x0[q] > I2LQR_LIT(SLIT, 709.79, v) ? __builtin_inf() : v))
                       : v;
  }
}
template <> __device__ __forceinline__ double t_exp<double>(double x0) {
  const double xa[1] = {x0};
  double ea[1];
  t_exp_d<false, 1, 0>(xa, ea);
  return ea[0];
}
// The same without the range handling, for arguments known to lie in [-700, 700].
template <class T> __device__ __forceinline__ T t_exp_bounded(T x);
template <> __device__ __forceinline__ double t_exp_bounded<double>(double x) {
  const double xa[1] = {x};
  double ea[1];
  t_exp_d<false, 1, 1>(xa, ea);
  return ea[0];
}
template <> __device__ __forceinline__ float t_exp_bounded<float>(float x) { return __expf(x); }
// fp32 exp: the hardware exp2 path (v_exp_f32), ~2 ulp
template <> __device__ __forceinline__ float t_exp<float>(float x) { return __expf(x); }
template <class T> __device__ __forceinline__ T t_sqrt(T x);
template <> __device__ __forceinline__ double t_sqrt<double>(double x) { return sqrt(x); }
template <> __device__ __forceinline__ float t_sqrt<float>(float x) { return sqrtf(x); }
template <class T> __device__ __forceinline__ T t_abs(T x) { return x < T(0) ? -x : x; }
// Reciprocal / square root for the 2x2 regularised inverse: hardware seed (v_rcp_f64 / v_rsq_f64)
// + two Newton steps — <= ~2 ulp on normal-range operands, a third of the instructions of the
// IEEE-exact division / sqrt expansions.  Zero, infinite and NaN operands behave like 1/x, sqrt(x).
// The library is built with -ffp-contract=off so that one source expression rounds the same way
// in every kernel it is inlined into (the bit-exact replay properties of tests/ rely on it);
// the dot products of the Riccati step ask for the fused multiply-add explicitly.
template <class T> __device__ __forceinline__ T t_fma(T a, T b, T c);
template <> __device__ __forceinline__ double t_fma<double>(double a, double b, double c) {
  return __builtin_fma(a, b, c);
}
template <> __device__ __forceinline__ float t_fma<float>(float a, float b, float c) {
  return __builtin_fmaf(a, b, c);
}

template <class T> __device__ __forceinline__ T t_rcp(T x);
template <> __device__ __forceinline__ double t_rcp<double>(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-x, r, 1.0);
  const double r2 = __builtin_fma(r, e, r);
  return (r2 == r2) ? r2 : r;  // x = 0 / inf: keep the seed's inf / 0 instead of the NaN of 0 * inf
}
template <> __device__ __forceinline__ float t_rcp<float>(float x) {
  const float r = __builtin_amdgcn_rcpf(x);  // 1 ulp seed, one Newton step
  const float e = __builtin_fmaf(-x, r, 1.0f);
  const float r2 = __builtin_fmaf(r, e, r);
  return (r2 == r2) ? r2 : r;
}
template <class T> __device__ __forceinline__ T t_sqrt_fast(T x);
template <> __device__ __forceinline__ double t_sqrt_fast<double>(double x) {
  if (!(x > 1.0e-290) || !(x < 1.0e290)) return __builtin_sqrt(x);  // 0, tiny, huge, NaN
  const double y = __builtin_amdgcn_rsq(x);       // ~ x^-1/2
  double g = x * y;                               // ~ sqrt(x)
  double h = 0.5 * y;
  double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  const double d = __builtin_fma(-g, g, x);
  return __builtin_fma(d, h, g);
}
template <> __device__ __forceinline__ float t_sqrt_fast<float>(float x) { return sqrtf(x); }
// Regularised inverse of a 2 x 2 Q_uu: control/iterative_ilqr.py:118-123
//   w, V = eig(Quu); w[w<0] = 0; w += lamb; inv = V diag(1/w) V^T.
// Fast path, the only one taken in practice (l_uu carries the strictly positive input-barrier
// curvature): Quu positive definite -> no eigenvalue is clamped and V diag(1/(w + lamb)) V^T is
// the inverse of Quu + lamb I, written out directly.  On the reference's golden calls this is as
// close to np.linalg.eig's result as the closed-form eigen-decomposition below (G2: 1.1e-9 vs
// 2.1e-9 max relative deviation, same branches).  It is computed unconditionally; the general
// form sits behind a wave-uniform unlikely branch (no jump over it in the common case) and
// follows the reference's NON-symmetric eig in closed form: unit-norm eigenvectors that are not
// orthogonalised, the normalisation folded into the eigenvalue division (v v^T / (|v|^2 w)).
// t_quu_inverse2_pd: the positive-definite form alone; *bad is set (never cleared) if Quu is not
// positive definite and the result must not be used (same protocol as t_sincos_fast).
template <class T>
__device__ __forceinline__ void t_quu_inverse2_pd(const T (&Quu)[4], T lamb, T (&inv)[4],
                                                  bool* bad) {
  const T a = Quu[0], b = Quu[1], cc = Quu[2], d = Quu[3];
  const T det = a * d - b * cc;
  *bad = *bad || !(a > T(0) && det > T(0));
  const T ar = a + lamb, dr = d + lamb;
  const T r = t_rcp(ar * dr - b * cc);
  inv[0] = dr * r;
  inv[1] = -b * r;
  inv[2] = -cc * r;
  inv[3] = ar * r;
}
template <class T>
__device__ __forceinline__ void t_quu_inverse2(const T (&Quu)[4], T lamb, T (&inv)[4]) {
  const T a = Quu[0], b = Quu[1], cc = Quu[2], d = Quu[3];
  const T det = a * d - b * cc;
  const bool pd = a > T(0) && det > T(0);
  {
    const T ar = a + lamb, dr = d + lamb;
    const T r = t_rcp(ar * dr - b * cc);
    inv[0] = dr * r;
    inv[1] = -b * r;
    inv[2] = -cc * r;
    inv[3] = ar * r;
  }
  if (__builtin_expect(__any(!pd), 0)) {
    const T mean = T(0.5) * (a + d), hd = T(0.5) * (a - d);
    T disc = hd * hd + b * cc;
    disc = disc < T(0) ? T(0) : disc;
    const T s = t_sqrt_fast(disc);
    T l1 = (mean >= T(0)) ? mean + s : mean - s;
    T l2 = (l1 != T(0)) ? det * t_rcp(l1) : T(0);
    if (s == T(0)) { l1 = mean; l2 = mean; }
    const T w[2] = {l1, l2};
    T vx[2], vy[2], sc[2];
#pragma unroll
    for (int e = 0; e < 2; e++) {
      // eigenvector of w[e] = a non-zero column of (Quu - w[other] I)  (Cayley-Hamilton)
      const T lo = w[1 - e];
      const T c0x = a - lo, c0y = cc, c1x = b, c1y = d - lo;
      const T n0 = c0x * c0x + c0y * c0y, n1 = c1x * c1x + c1y * c1y;
      const bool first = n0 >= n1;
      T ex = first ? c0x : c1x, ey = first ? c0y : c1y, nn = first ? n0 : n1;
      if (nn == T(0)) { ex = (e == 0) ? T(1) : T(0); ey = (e == 0) ? T(0) : T(1); nn = T(1); }
      vx[e] = ex;
      vy[e] = ey;
      sc[e] = t_rcp(nn * ((w[e] < T(0) ? T(0) : w[e]) + lamb));
    }
    const T g0 = vx[0] * sc[0] * vx[0] + vx[1] * sc[1] * vx[1];
    const T g1 = vx[0] * sc[0] * vy[0] + vx[1] * sc[1] * vy[1];
    const T g2 = vy[0] * sc[0] * vx[0] + vy[1] * sc[1] * vx[1];
    const T g3 = vy[0] * sc[0] * vy[0] + vy[1] * sc[1] * vy[1];
    inv[0] = pd ? inv[0] : g0;
    inv[1] = pd ? inv[1] : g1;
    inv[2] = pd ? inv[2] : g2;
    inv[3] = pd ? inv[3] : g3;
  }
}

// Positive-definite form of the regularised solve, for callers that need (Quu + lamb I)^-1 applied
// to a few vectors rather than the inverse itself: square-root-free LDL^T of the symmetrised Quu
// (pivot test: *bad set, never cleared, if a pivot is not positive — the caller then repeats its
// pass with the general form, t_quu_inverse_m<.., true>) and of Quu + lamb I (unit lower factor Lf,
// reciprocal pivots ir).  Same factorisations as t_quu_inverse_m, without forming Lf^-1 and the
// inverse (about a third of the instructions for m = 4); agrees with it to round-off.
template <class T, int m>
__device__ __forceinline__ void t_quu_factor_pd(const T (&Quu)[m * m], T lamb, T (&Lf)[m * m],
                                                T (&ir)[m], bool* bad) {
  T Sm[m * m];
#pragma unroll
  for (int i = 0; i < m; i++)
#pragma unroll
    for (int j = 0; j <= i; j++) Sm[i * m + j] = T(0.5) * (Quu[i * m + j] + Quu[j * m + i]);
  T Lp[m * m], Wp[m * m], Wr[m * m];  // L and W = L diag(d), strictly lower parts
  bool pd = true;
#pragma unroll
  for (int j = 0; j < m; j++) {
    T dp = Sm[j * m + j], dr = Sm[j * m + j] + lamb;
#pragma unroll
    for (int k = 0; k < j; k++) {
      dp = t_fma(-Wp[j * m + k], Lp[j * m + k], dp);
      dr = t_fma(-Wr[j * m + k], Lf[j * m + k], dr);
    }
    pd = pd && (dp > T(0));
    const T ipj = t_rcp(dp);
    ir[j] = t_rcp(dr);
#pragma unroll
    for (int i = j + 1; i < m; i++) {
      T vp = Sm[i * m + j], vr = Sm[i * m + j];
#pragma unroll
      for (int k = 0; k < j; k++) {
        vp = t_fma(-Wp[i * m + k], Lp[j * m + k], vp);
        vr = t_fma(-Wr[i * m + k], Lf[j * m + k], vr);
      }
      Wp[i * m + j] = vp;
      Wr[i * m + j] = vr;
      Lp[i * m + j] = vp * ipj;
      Lf[i * m + j] = vr * ir[j];
    }
  }
  *bad = *bad || !pd;
}
// x = (Lf diag(1 / ir) Lf^T)^-1 b
template <class T, int m>
__device__ __forceinline__ void t_quu_solve(const T (&Lf)[m * m], const T (&ir)[m], const T (&b)[m],
                                            T (&x)[m]) {
  T y[m];
#pragma unroll
  for (int i = 0; i < m; i++) {
    T acc = b[i];
#pragma unroll
    for (int k = 0; k < i; k++) acc = t_fma(-Lf[i * m + k], y[k], acc);
    y[i] = acc;
  }
#pragma unroll
  for (int i = m - 1; i >= 0; i--) {
    T acc = y[i] * ir[i];
#pragma unroll
    for (int k = i + 1; k < m; k++) acc = t_fma(-Lf[k * m + i], x[k], acc);
    x[i] = acc;
  }
}

// Regularised inverse of an m x m Q_uu, m > 2 (control/iterative_ilqr.py:118-123: eigen-
// decomposition, negative eigenvalues clamped, lamb added; no reference counterpart at m > 2,
// the oracle follows the same construction).  GENERAL = false: the positive-definite form alone,
// *bad set (never cleared) if Quu is not positive definite (protocol of t_sincos_fast).
template <class T, int m, bool GENERAL>
__device__ __forceinline__ void t_quu_inverse_m(const T (&Quu)[m * m], T lamb, T (&inv)[m * m],
                                                bool* bad) {
  // Symmetrised Quu.  Fast path (the only one taken in practice: l_uu carries the strictly
  // positive input-barrier curvature): if Quu is positive definite no eigenvalue is clamped
  // and inv = (Quu + lamb I)^-1, computed by Cholesky.  Otherwise: cyclic Jacobi, clamp, add.
  T Sm[m * m];
#pragma unroll
  for (int i = 0; i < m; i++)
#pragma unroll
    for (int j = 0; j < m; j++) Sm[i * m + j] = T(0.5) * (Quu[i * m + j] + Quu[j * m + i]);
  // Square-root-free LDL^T of Sm (positive-definiteness test: all pivots > 0) and of
  // Sm + lamb I; unit lower factors Lp, Lr, pivots dp, dr (reciprocals ip, ir).
  T Lp[m * m], Lr[m * m], ir[m];
  bool pd = true;
#pragma unroll
  for (int j = 0; j < m; j++) {
    T dp = Sm[j * m + j], dr = Sm[j * m + j] + lamb;
    T wp[m], wr[m];  // L_jk d_k
#pragma unroll
    for (int k = 0; k < j; k++) {
      wp[k] = Lp[j * m + k];
      wr[k] = Lr[j * m + k];
    }
#pragma unroll
    for (int k = 0; k < j; k++) {
      // Lp/Lr hold L_jk d_k below the diagonal until column j is finished (see below)
      dp -= wp[k] * Lp[k * m + j];
      dr -= wr[k] * Lr[k * m + j];
    }
    pd = pd && (dp > T(0));
    const T ipj = t_rcp(dp > T(0) ? dp : T(1));
    ir[j] = t_rcp(dr > T(0) ? dr : T(1));
#pragma unroll
    for (int i = j + 1; i < m; i++) {
      T vp = Sm[i * m + j], vr = Sm[i * m + j];
#pragma unroll
      for (int k = 0; k < j; k++) {
        vp -= Lp[i * m + k] * Lp[k * m + j];
        vr -= Lr[i * m + k] * Lr[k * m + j];
      }
      // lower triangle keeps W_ij = L_ij d_j, upper triangle keeps L_ij (transposed slot)
      Lp[i * m + j] = vp;
      Lr[i * m + j] = vr;
      Lp[j * m + i] = vp * ipj;
      Lr[j * m + i] = vr * ir[j];
    }
  }
  // Li = Lr^-1 (unit lower), then inv = Li^T diag(ir) Li.  L_ij lives at Lr[j * m + i].
  T Li[m * m];
#pragma unroll
  for (int i = 0; i < m; i++)
#pragma unroll
    for (int j = 0; j < m; j++) Li[i * m + j] = (i == j) ? T(1) : T(0);
#pragma unroll
  for (int j = 0; j < m; j++) {
#pragma unroll
    for (int i = j + 1; i < m; i++) {
      T acc = T(0);
#pragma unroll
      for (int k = j; k < i; k++) acc += Lr[k * m + i] * Li[k * m + j];
      Li[i * m + j] = -acc;
    }
  }
#pragma unroll
  for (int i = 0; i < m; i++)
#pragma unroll
    for (int j = i; j < m; j++) {
      T acc = T(0);
#pragma unroll
      for (int k = j; k < m; k++) acc += Li[k * m + i] * ir[k] * Li[k * m + j];
      inv[i * m + j] = acc;
      inv[j * m + i] = acc;
    }
  if constexpr (!GENERAL) {
    *bad = *bad || !pd;
  } else if (__builtin_expect(__any(!pd), 0)) {
    // cyclic Jacobi on the symmetrised matrix, clamp, add lamb (lanes with a positive-definite
    // Quu keep the form above)
    T ginv[m * m];
    {
      T V[m * m];
#pragma unroll
      for (int i = 0; i < m; i++)
#pragma unroll
        for (int j = 0; j < m; j++) V[i * m + j] = (i == j) ? T(1) : T(0);
      for (int sweep = 0; sweep < 12; sweep++) {
#pragma unroll
        for (int p = 0; p < m - 1; p++)
#pragma unroll
          for (int q = p + 1; q < m; q++) {
            const T apq = Sm[p * m + q];
            const T app = Sm[p * m + p], aqq = Sm[q * m + q];
            // rotation angle; apq == 0 gives the identity rotation
            const T tau = (aqq - app) / (T(2) * apq);
            T tt = (tau >= T(0) ? T(1) : T(-1)) / (t_abs(tau) + t_sqrt(T(1) + tau * tau));
            tt = (apq == T(0)) ? T(0) : tt;
            const T cs = T(1) / t_sqrt(T(1) + tt * tt), sn = tt * cs;
#pragma unroll
            for (int k = 0; k < m; k++) {
              const T skp = Sm[k * m + p], skq = Sm[k * m + q];
              Sm[k * m + p] = cs * skp - sn * skq;
              Sm[k * m + q] = sn * skp + cs * skq;
            }
#pragma unroll
            for (int k = 0; k < m; k++) {
              const T spk = Sm[p * m + k], sqk = Sm[q * m + k];
              Sm[p * m + k] = cs * spk - sn * sqk;
              Sm[q * m + k] = sn * spk + cs * sqk;
            }
#pragma unroll
            for (int k = 0; k < m; k++) {
              const T vkp = V[k * m + p], vkq = V[k * m + q];
              V[k * m + p] = cs * vkp - sn * vkq;
              V[k * m + q] = sn * vkp + cs * vkq;
            }
          }
      }
      T wr[m];
#pragma unroll
      for (int e = 0; e < m; e++) {
        const T we = Sm[e * m + e];
        wr[e] = T(1) / ((we < T(0) ? T(0) : we) + lamb);
      }
#pragma unroll
      for (int i = 0; i < m; i++)
#pragma unroll
        for (int j = 0; j < m; j++) {
          T acc = T(0);
#pragma unroll
          for (int e = 0; e < m; e++) acc += V[i * m + e] * wr[e] * V[j * m + e];
          ginv[i * m + j] = acc;
        }
    }
#pragma unroll
    for (int e = 0; e < m * m; e++) inv[e] = pd ? inv[e] : ginv[e];
  }
}

// ---------------------------------------------------------------------------------------------
// bicycle4: the reference plant.  x = [x, y, v, theta], u = [accel, delta].
// ---------------------------------------------------------------------------------------------
template <class T> struct Bicycle4 {
  static constexpr int n = 4, m = 2, NTRIG = 2, NVAR = 6;
  static constexpr int NCONST = 0, NBLK = 0;  // no plant-constant entries in F, no row-block form
  static constexpr int NJX = 0;
  static constexpr int jx(int) { return 0; }
  static constexpr int blk(int) { return 0; }
  template <class Cfg> static __device__ __forceinline__ T plant_const(const Cfg&, int) { return T(0); }
  static constexpr int system_id = 0;
  // theta' = theta + delta dt with delta an INPUT: the heading two steps ahead is not known before
  // the next step's feedback law has run
  static constexpr bool kHeadingAhead = false;

  // {cos(theta), sin(theta)}
  static __device__ __forceinline__ void trig(const T (&xe)[n], T (&tr)[NTRIG]) {
    t_sincos(xe[3], &tr[1], &tr[0]);
  }
  // GENERAL = false: short sincos kernel only, *bad set if its range was left (t_sincos_fast)
  template <bool GENERAL>
  static __device__ __forceinline__ void trig_g(const T (&xe)[n], T (&tr)[NTRIG], bool* bad) {
    if constexpr (GENERAL) t_sincos(xe[3], &tr[1], &tr[0]);
    else t_sincos_fast(xe[3], &tr[1], &tr[0], bad);
  }
  // kinetic_bicycle(): systems/kinetic_bicycle.py:10-27, with trig(x) supplied
  template <class Cfg>
  static __device__ __forceinline__ void step_tr(const Cfg& c, const T (&x)[n], const T (&u)[m],
                                                 const T (&tr)[NTRIG], T (&xn)[n]) {
    const T dt = c.dt;
    const T w = x[2] * dt + (u[0] * dt * dt) / T(2);
    xn[0] = x[0] + tr[0] * w;
    xn[1] = x[1] + tr[1] * w;
    xn[2] = x[2] + u[0] * dt;
    xn[3] = x[3] + u[1] * dt;
  }
  template <class Cfg>
  static __device__ __forceinline__ void step(const Cfg& c, const T (&x)[n], const T (&u)[m],
                                              T (&xn)[n]) {
    T tr[NTRIG];
    trig(x, tr);
    step_tr(c, x, u, tr, xn);
  }
  // compile-time pattern of F = [A | B]: 0 zero, 1 one, 2 dt, 3 + v = varying entry v
  static constexpr int pat(int i, int j) {
    if (i == j) return 1;
    if (i == 0 && j == 2) return 3 + 0;
    if (i == 0 && j == 3) return 3 + 1;
    if (i == 1 && j == 2) return 3 + 2;
    if (i == 1 && j == 3) return 3 + 3;
    if (i == 0 && j == n) return 3 + 4;
    if (i == 1 && j == n) return 3 + 5;
    if ((i == 2 && j == n) || (i == 3 && j == n + 1)) return 2;
    return 0;
  }
  // get_A_matrix / get_B_matrix: systems/kinetic_bicycle.py:30-52
  template <class Cfg>
  static __device__ __forceinline__ void jac_var(const Cfg& c, const T (&xe)[n], const T (&u)[m],
                                                 const T (&tr)[NTRIG], T (&v)[NVAR]) {
    const T dt = c.dt;
    const T w = xe[2] * dt + (u[0] * dt * dt) / T(2);
    v[0] = tr[0] * dt;             // A[0][2]
    v[1] = -w * tr[1];             // A[0][3]
    v[2] = tr[1] * dt;             // A[1][2]
    v[3] = w * tr[0];              // A[1][3]
    v[4] = dt * dt * tr[0] / T(2); // B[0][0]
    v[5] = dt * dt * tr[1] / T(2); // B[1][0]
  }
  // flat index into F (row-major n x (n+m)) of varying entry v
  static constexpr int var_idx_c(int v) {
    constexpr int W = n + m;
    constexpr int idx[NVAR] = {0 * W + 2, 0 * W + 3, 1 * W + 2, 1 * W + 3, 0 * W + n, 1 * W + n};
    return idx[v];
  }
  template <class Cfg> static __device__ __forceinline__ T jac_const(const Cfg& c, int i, int j) {
    if (j < n) return (i == j) ? T(1) : T(0);
    return ((i == 2 && j == n) || (i == 3 && j == n + 1)) ? c.dt : T(0);
  }
};

// ---------------------------------------------------------------------------------------------
// bicycle6 (build-defined): bicycle4 with actuator states.  x = [x, y, v, theta, a, delta],
// u = [jerk, steering rate].
// ---------------------------------------------------------------------------------------------
template <class T> struct Bicycle6 {
  static constexpr int n = 6, m = 2, NTRIG = 2, NVAR = 6;
  static constexpr int NCONST = 0, NBLK = 0;  // no plant-constant entries in F, no row-block form
  static constexpr int NJX = 0;
  static constexpr int jx(int) { return 0; }
  static constexpr int blk(int) { return 0; }
  template <class Cfg> static __device__ __forceinline__ T plant_const(const Cfg&, int) { return T(0); }
  static constexpr int system_id = 1;
  // theta' = theta + delta dt with delta a STATE: the heading of x_{t+2} follows from x_{t+1} alone,
  // so a rollout can evaluate the sin / cos of two consecutive steps side by side (the sixteen-lane
  // forward pass does, on the two halves of a problem's DPP row: GroupWorker::forward_row)
  static constexpr bool kHeadingAhead = true;
  static __device__ __forceinline__ T heading(const T (&xe)[n]) { return xe[3]; }
  // heading of the successor state: the very expression step_tr evaluates for xn[3]
  template <class Cfg>
  static __device__ __forceinline__ T next_heading(const Cfg& c, const T (&xe)[n]) {
    return xe[3] + xe[5] * c.dt;
  }
  // trig_g<false> for a heading given as a number
  static __device__ __forceinline__ void trig_heading_fast(T ang, T (&tr)[NTRIG], bool* bad) {
    t_sincos_fast_h3(ang, &tr[1], &tr[0], bad);
  }

  static __device__ __forceinline__ void trig(const T (&xe)[n], T (&tr)[NTRIG]) {
    t_sincos(xe[3], &tr[1], &tr[0]);
  }
  template <bool GENERAL>
  static __device__ __forceinline__ void trig_g(const T (&xe)[n], T (&tr)[NTRIG], bool* bad) {
    if constexpr (GENERAL) t_sincos(xe[3], &tr[1], &tr[0]);
    else t_sincos_fast(xe[3], &tr[1], &tr[0], bad);
  }
  template <class Cfg>
  static __device__ __forceinline__ void step_tr(const Cfg& c, const T (&x)[n], const T (&u)[m],
                                                 const T (&tr)[NTRIG], T (&xn)[n]) {
    const T dt = c.dt;
    const T w = x[2] * dt + (x[4] * dt * dt) / T(2);
    xn[0] = x[0] + tr[0] * w;
    xn[1] = x[1] + tr[1] * w;
    xn[2] = x[2] + x[4] * dt;
    xn[3] = x[3] + x[5] * dt;
    xn[4] = x[4] + u[0] * dt;
    xn[5] = x[5] + u[1] * dt;
  }
  template <class Cfg>
  static __device__ __forceinline__ void step(const Cfg& c, const T (&x)[n], const T (&u)[m],
                                              T (&xn)[n]) {
    T tr[NTRIG];
    trig(x, tr);
    step_tr(c, x, u, tr, xn);
  }
  static constexpr int pat(int i, int j) {
    if (i == j) return 1;
    if (i == 0 && j == 2) return 3 + 0;
    if (i == 0 && j == 3) return 3 + 1;
    if (i == 0 && j == 4) return 3 + 2;
    if (i == 1 && j == 2) return 3 + 3;
    if (i == 1 && j == 3) return 3 + 4;
    if (i == 1 && j == 4) return 3 + 5;
    if ((i == 2 && j == 4) || (i == 3 && j == 5)) return 2;
    if ((i == 4 && j == n) || (i == 5 && j == n + 1)) return 2;
    return 0;
  }
  template <class Cfg>
  static __device__ __forceinline__ void jac_var(const Cfg& c, const T (&xe)[n], const T (&)[m],
                                                 const T (&tr)[NTRIG], T (&v)[NVAR]) {
    const T dt = c.dt;
    const T w = xe[2] * dt + (xe[4] * dt * dt) / T(2);
    v[0] = tr[0] * dt;             // A[0][2]
    v[1] = -w * tr[1];             // A[0][3]
    v[2] = dt * dt * tr[0] / T(2); // A[0][4]
    v[3] = tr[1] * dt;             // A[1][2]
    v[4] = w * tr[0];              // A[1][3]
    v[5] = dt * dt * tr[1] / T(2); // A[1][4]
  }
  static constexpr int var_idx_c(int v) {
    constexpr int W = n + m;
    constexpr int idx[NVAR] = {0 * W + 2, 0 * W + 3, 0 * W + 4, 1 * W + 2, 1 * W + 3, 1 * W + 4};
    return idx[v];
  }
  template <class Cfg> static __device__ __forceinline__ T jac_const(const Cfg& c, int i, int j) {
    if (j < n) {
      if (i == j) return T(1);
      return ((i == 2 && j == 4) || (i == 3 && j == 5)) ? c.dt : T(0);
    }
    return ((i == 4 && j == n) || (i == 5 && j == n + 1)) ? c.dt : T(0);
  }
};

// ---------------------------------------------------------------------------------------------
// quad12 (build-defined): rigid-body quadrotor, explicit Euler.
// x = [p(3), phi, theta, psi, v(3), p, q, r], u = rotor thrust deviations from hover.
// sys_par = {mass, g, arm, Ix, Iy, Iz, ctau}.
// ---------------------------------------------------------------------------------------------
template <class T> struct Quad12 {
  static constexpr int n = 12, m = 4, NTRIG = 6;
  // 25 state-dependent entries of A and the 3 x 4 thrust-direction entries of B
  static constexpr int NVAR = 37;
  static constexpr int system_id = 2;
  static constexpr bool kHeadingAhead = false;

  // {sin phi, cos phi, sin theta, cos theta, sin psi, cos psi}
  static __device__ __forceinline__ void trig(const T (&xe)[n], T (&tr)[NTRIG]) {
    t_sincos(xe[3], &tr[0], &tr[1]);
    t_sincos(xe[4], &tr[2], &tr[3]);
    t_sincos(xe[5], &tr[4], &tr[5]);
  }
  template <bool GENERAL>
  static __device__ __forceinline__ void trig_g(const T (&xe)[n], T (&tr)[NTRIG], bool* bad) {
    if constexpr (GENERAL) {
      trig(xe, tr);
    } else {
      t_sincos_fast(xe[3], &tr[0], &tr[1], bad);
      t_sincos_fast(xe[4], &tr[2], &tr[3], bad);
      t_sincos_fast(xe[5], &tr[4], &tr[5], bad);
    }
  }
  // hot form with scalar-register literals (LaneWorker::backward_blocked and friends)
  static __device__ __forceinline__ void trig_s(const double (&xe)[n], double (&tr)[NTRIG], bool* bad) {
    const double ang[3] = {xe[3], xe[4], xe[5]};  // the three angles at once: literals shared
    double sn[3], cs[3];
    t_sincos_fast_dn<true, 3>(ang, sn, cs, bad);
#pragma unroll
    for (int q = 0; q < 3; q++) {
      tr[2 * q] = sn[q];
      tr[2 * q + 1] = cs[q];
    }
  }
  static __device__ __forceinline__ void trig_s(const float (&xe)[n], float (&tr)[NTRIG], bool* bad) {
    trig_g<false>(xe, tr, bad);
  }
  // The three angles on three lanes at once: `g` is the lane's index inside a 16-lane DPP row whose
  // lanes all hold the same xe; lane q < 3 evaluates angle q, the results are broadcast along the
  // row (BC<L>(v) = value of lane L of the row).  Same arithmetic as trig_g, a third of the
  // instructions; *bad is per lane (the caller votes over the wavefront).
  template <bool GENERAL, class BC>
  static __device__ __forceinline__ void trig_row(const T (&xe)[n], T (&tr)[NTRIG], bool* bad, int g,
                                                  BC&& bc) {
    const T ang = g == 0 ? xe[3] : (g == 1 ? xe[4] : xe[5]);
    T sv, cv;
    if constexpr (GENERAL) t_sincos(ang, &sv, &cv);
    else t_sincos_fast(ang, &sv, &cv, bad);
    tr[0] = bc(std::integral_constant<int, 0>{}, sv);
    tr[1] = bc(std::integral_constant<int, 0>{}, cv);
    tr[2] = bc(std::integral_constant<int, 1>{}, sv);
    tr[3] = bc(std::integral_constant<int, 1>{}, cv);
    tr[4] = bc(std::integral_constant<int, 2>{}, sv);
    tr[5] = bc(std::integral_constant<int, 2>{}, cv);
  }
  template <class Cfg>
  static __device__ __forceinline__ void step_tr(const Cfg& c, const T (&x)[n], const T (&u)[m],
                                                 const T (&tr)[NTRIG], T (&xn)[n]) {
    // Divisions by plant constants are multiplications by their (loop-invariant) reciprocals and
    // the two divisions by cos(theta) share one reciprocal: an IEEE fp64 division is a dozen
    // instructions on this hardware, six of them per step were a fifth of the rollout.  (A few ulp
    // from the textbook form; every kernel and every rollout uses this one function.)
    // (the quotients are formed once on the host, DevCfg::pd: the same IEEE divisions)
    const T g = c.sys_par[1];
    const T inv_mass = c.pd[0], arm_ix = c.pd[1], arm_iy = c.pd[2], ct_iz = c.pd[3];
    const T sph = tr[0], cph = tr[1], sth = tr[2], cth = tr[3], sps = tr[4], cps = tr[5];
    const T icth = t_rcp(cth);
    const T tth = sth * icth;
    const T Tt = c.pd[7] + (u[0] + u[1] + u[2] + u[3]);
    const T Tm = Tt * inv_mass;
    const T p = x[9], q = x[10], r = x[11], dt = c.dt;
    T f[n];
    f[0] = x[6];
    f[1] = x[7];
    f[2] = x[8];
    f[3] = p + q * sph * tth + r * cph * tth;
    f[4] = q * cph - r * sph;
    f[5] = (q * sph + r * cph) * icth;
    f[6] = Tm * (cph * sth * cps + sph * sps);
    f[7] = Tm * (cph * sth * sps - sph * cps);
    f[8] = Tm * (cph * cth) - g;
    f[9] = c.pd[4] * q * r + arm_ix * (u[1] - u[3]);
    f[10] = c.pd[5] * p * r + arm_iy * (u[2] - u[0]);
    f[11] = c.pd[6] * p * q + ct_iz * (u[0] - u[1] + u[2] - u[3]);
#pragma unroll
    for (int i = 0; i < n; i++) xn[i] = x[i] + dt * f[i];
  }
  template <class Cfg>
  static __device__ __forceinline__ void step(const Cfg& c, const T (&x)[n], const T (&u)[m],
                                              T (&xn)[n]) {
    T tr[NTRIG];
    trig(x, tr);
    step_tr(c, x, u, tr, xn);
  }
  template <class Cfg>
  static __device__ __forceinline__ void jac_var(const Cfg& c, const T (&xe)[n], const T (&u)[m],
                                                 const T (&tr)[NTRIG], T (&v)[NVAR]) {
    const T inv_mass = c.pd[0];
    const T sph = tr[0], cph = tr[1], sth = tr[2], cth = tr[3], sps = tr[4], cps = tr[5];
    const T icth = t_rcp(cth);  // one reciprocal for every division by cos(theta) (see step_tr)
    const T tth = sth * icth, sec2 = icth * icth;
    const T Tt = c.pd[7] + (u[0] + u[1] + u[2] + u[3]);
    const T Tm = Tt * inv_mass, dt = c.dt;
    const T p = xe[9], q = xe[10], r = xe[11];
    // A = I + dt * dF/dx
    v[0] = T(1) + dt * ((q * cph - r * sph) * tth);       // [3][3]
    v[1] = dt * ((q * sph + r * cph) * sec2);             // [3][4]
    v[2] = dt * (sph * tth);                              // [3][10]
    v[3] = dt * (cph * tth);                              // [3][11]
    v[4] = dt * (-q * sph - r * cph);                     // [4][3]
    v[5] = dt * cph;                                      // [4][10]
    v[6] = dt * (-sph);                                   // [4][11]
    v[7] = dt * ((q * cph - r * sph) * icth);             // [5][3]
    v[8] = dt * ((q * sph + r * cph) * sth * sec2);       // [5][4]
    v[9] = dt * (sph * icth);                             // [5][10]
    v[10] = dt * (cph * icth);                            // [5][11]
    v[11] = dt * (Tm * (-sph * sth * cps + cph * sps));   // [6][3]
    v[12] = dt * (Tm * (cph * cth * cps));                // [6][4]
    v[13] = dt * (Tm * (-cph * sth * sps + sph * cps));   // [6][5]
    v[14] = dt * (Tm * (-sph * sth * sps - cph * cps));   // [7][3]
    v[15] = dt * (Tm * (cph * cth * sps));                // [7][4]
    v[16] = dt * (Tm * (cph * sth * cps + sph * sps));    // [7][5]
    v[17] = dt * (Tm * (-sph * cth));                     // [8][3]
    v[18] = dt * (Tm * (-cph * sth));                     // [8][4]
    v[19] = dt * (c.pd[4] * r);                  // [9][10]
    v[20] = dt * (c.pd[4] * q);                  // [9][11]
    v[21] = dt * (c.pd[5] * r);                  // [10][9]
    v[22] = dt * (c.pd[5] * p);                  // [10][11]
    v[23] = dt * (c.pd[6] * q);                  // [11][9]
    v[24] = dt * (c.pd[6] * p);                  // [11][10]
    // B rows 6..8: dt * a{x,y,z} for each of the 4 inputs
    const T ax = dt * ((cph * sth * cps + sph * sps) * inv_mass);
    const T ay = dt * ((cph * sth * sps - sph * cps) * inv_mass);
    const T az = dt * ((cph * cth) * inv_mass);
#pragma unroll
    for (int j = 0; j < m; j++) {
      v[25 + j] = ax;
      v[29 + j] = ay;
      v[33 + j] = az;
    }
  }
  // compile-time pattern of F = [A | B]: 0 zero, 1 one, 2 dt, 3 + v = varying entry v,
  // 100 + q = plant constant q (plant_const below)
  static constexpr int NCONST = 6;
  static constexpr int pat(int i, int j) {
    for (int v = 0; v < NVAR; v++)
      if (var_idx_c(v) == i * (n + m) + j) return 3 + v;
    if (j < n) {
      if (i == j) return 1;
      if (i < 3 && j == i + 6) return 2;  // d pos / d vel
      if (i == 3 && j == 9) return 2;     // d phi / d p
      return 0;
    }
    const int a = j - n;
    if (i == 9) return a == 1 ? 100 : (a == 3 ? 101 : 0);
    if (i == 10) return a == 2 ? 102 : (a == 0 ? 103 : 0);
    if (i == 11) return (a & 1) == 0 ? 104 : 105;
    return 0;
  }
  // Row-block factorisation of A = I + E (one-problem-per-lane kernel, LaneWorker::backward_blocked).
  // With the state rows grouped into the blocks
  //     {p, q, r} = 0   {phi, theta} = 1   {psi} = 2   {vx}, {vy}, {vz} = 3, 4, 5   {x}, {y}, {z} = 6, 7, 8
  // every non-zero E[i][j] has blk(j) <= blk(i) (rates feed angles, angles feed velocities,
  // velocities feed positions; the cycles p-q-r and phi-theta stay inside one block), hence
  //     A = M_0 M_1 ... M_8,   M_k = I + (the rows of block k of E)
  // exactly (all cross products vanish), and A^T V A, K A, A^T v become nine IN-PLACE updates that
  // touch only the columns a block's rows reach — no n x n intermediate next to V.  Checked at
  // compile time where it is used.
  // entries of the evaluation state that jac_var() reads besides the sin / cos values (the body
  // rates): what LaneWorker::backward_blocked keeps to form the A entries a second time
  static constexpr int NJX = 3;
  static constexpr int jx(int k) { return 9 + k; }
  static constexpr int NBLK = 9;
  static constexpr int blk(int i) {
    return i >= 9 ? 0 : (i == 3 || i == 4) ? 1 : i == 5 ? 2 : i >= 6 ? i - 3 : i + 6;
  }
  // {dt arm / Ix, -, dt arm / Iy, -, dt ctau / Iz, -}: the constant entries of B (jac_const)
  template <class Cfg> static __device__ __forceinline__ T plant_const(const Cfg& c, int q) {
    return c.pd[8 + q];
  }
  static constexpr int var_idx_c(int v) {
    constexpr int W = n + m;
    constexpr int idx[NVAR] = {
        3 * W + 3,  3 * W + 4,  3 * W + 10, 3 * W + 11, 4 * W + 3,  4 * W + 10, 4 * W + 11,
        5 * W + 3,  5 * W + 4,  5 * W + 10, 5 * W + 11, 6 * W + 3,  6 * W + 4,  6 * W + 5,
        7 * W + 3,  7 * W + 4,  7 * W + 5,  8 * W + 3,  8 * W + 4,  9 * W + 10, 9 * W + 11,
        10 * W + 9, 10 * W + 11, 11 * W + 9, 11 * W + 10,
        6 * W + n + 0, 6 * W + n + 1, 6 * W + n + 2, 6 * W + n + 3,
        7 * W + n + 0, 7 * W + n + 1, 7 * W + n + 2, 7 * W + n + 3,
        8 * W + n + 0, 8 * W + n + 1, 8 * W + n + 2, 8 * W + n + 3};
    return idx[v];
  }
  template <class Cfg> static __device__ __forceinline__ T jac_const(const Cfg& c, int i, int j) {
    const T dt = c.dt;
    if (j < n) {
      if (i == j) return T(1);
      if (i < 3 && j == i + 6) return dt; // d pos / d vel
      if (i == 3 && j == 9) return dt;    // d phi / d p
      return T(0);
    }
    const int a = j - n;
    const T arm = c.sys_par[2], Ix = c.sys_par[3], Iy = c.sys_par[4], Iz = c.sys_par[5];
    const T ct = c.sys_par[6];
    if (i == 9) return (a == 1) ? dt * arm / Ix : (a == 3) ? -dt * arm / Ix : T(0);
    if (i == 10) return (a == 2) ? dt * arm / Iy : (a == 0) ? -dt * arm / Iy : T(0);
    if (i == 11) return ((a & 1) == 0) ? dt * ct / Iz : -dt * ct / Iz;
    return T(0);
  }
};

}  // namespace i2lqr

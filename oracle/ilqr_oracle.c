/*
 * ilqr_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C, single-thread, fp64 restatement of the reference's iLQR hot path
 * (HybridRobotics/ilqr-iterative-tasks).  Only tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py may load it; nothing under ilqr_iterative_tasks_amd/ does.
 *
 * Parity status: PINNED at n=4, m=2 (system bicycle4) against golden vectors captured by running
 * the reference itself (oracle/gen_golden.py -> tests/golden/ fixtures; checked by
 * tests/test_oracle_golden.py).  The build-defined systems bicycle6 / quad12 have no reference
 * counterpart: for them this file IS the definition ("reference-parity at n=4 only").
 *
 * Every function cites the reference lines it follows (paths relative to the reference root).
 * Per-problem array layout is the reference's NumPy layout: X[n][N+1], U[m][N], K[m][n][N],
 * k[m][N] with time the fastest axis.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "../include/i2lqr.h"

#define MAXN I2LQR_MAX_N
#define MAXM I2LQR_MAX_M
#define MAXH I2LQR_MAX_HORIZON

typedef i2lqr_config cfg_t;

/* ------------------------------------------------------------------------------------------ */
/* Plant models                                                                               */
/* ------------------------------------------------------------------------------------------ */

/* kinetic_bicycle(): systems/kinetic_bicycle.py:10-27 (theta is not wrapped). */
static void bicycle4_step(const cfg_t* c, const double* x, const double* u, double* xn) {
  const double dt = c->dt;
  const double w = x[2] * dt + (u[0] * dt * dt) / 2;
  xn[0] = x[0] + cos(x[3]) * w;
  xn[1] = x[1] + sin(x[3]) * w;
  xn[2] = x[2] + u[0] * dt;
  xn[3] = x[3] + u[1] * dt;
}

/* get_A_matrix / get_B_matrix: systems/kinetic_bicycle.py:30-52.  Called by backward_pass with
 * v, theta of x_{t+1} and accel of u_t (control/iterative_ilqr.py:92-99) — `xe` is that
 * evaluation state. */
static void bicycle4_jac(const cfg_t* c, const double* xe, const double* u, double* A, double* B) {
  const int n = 4, m = 2;
  const double dt = c->dt, v = xe[2], th = xe[3], a = u[0];
  memset(A, 0, sizeof(double) * n * n);
  memset(B, 0, sizeof(double) * n * m);
  for (int i = 0; i < n; i++) A[i * n + i] = 1.0;
  A[0 * n + 2] = cos(th) * dt;
  A[0 * n + 3] = -(v * dt + (a * dt * dt) / 2) * sin(th);
  A[1 * n + 2] = sin(th) * dt;
  A[1 * n + 3] = (v * dt + (a * dt * dt) / 2) * cos(th);
  B[0 * m + 0] = dt * dt * cos(th) / 2;
  B[1 * m + 0] = dt * dt * sin(th) / 2;
  B[2 * m + 0] = dt;
  B[3 * m + 1] = dt;
}

/* bicycle6 (build-defined): state [x, y, v, theta, a, delta], input [jerk, steer_rate];
 * bicycle4 with the two inputs promoted to actuator states. */
static void bicycle6_step(const cfg_t* c, const double* x, const double* u, double* xn) {
  const double dt = c->dt;
  const double w = x[2] * dt + (x[4] * dt * dt) / 2;
  xn[0] = x[0] + cos(x[3]) * w;
  xn[1] = x[1] + sin(x[3]) * w;
  xn[2] = x[2] + x[4] * dt;
  xn[3] = x[3] + x[5] * dt;
  xn[4] = x[4] + u[0] * dt;
  xn[5] = x[5] + u[1] * dt;
}

/* Jacobians of bicycle6 at the evaluation state `xe` (= x_{t+1}, keeping the reference's
 * evaluation-point convention, control/iterative_ilqr.py:92-99). */
static void bicycle6_jac(const cfg_t* c, const double* xe, const double* u, double* A, double* B) {
  (void)u;
  const int n = 6, m = 2;
  const double dt = c->dt, v = xe[2], th = xe[3], a = xe[4];
  const double w = v * dt + (a * dt * dt) / 2;
  memset(A, 0, sizeof(double) * n * n);
  memset(B, 0, sizeof(double) * n * m);
  for (int i = 0; i < n; i++) A[i * n + i] = 1.0;
  A[0 * n + 2] = cos(th) * dt;
  A[0 * n + 3] = -w * sin(th);
  A[0 * n + 4] = dt * dt * cos(th) / 2;
  A[1 * n + 2] = sin(th) * dt;
  A[1 * n + 3] = w * cos(th);
  A[1 * n + 4] = dt * dt * sin(th) / 2;
  A[2 * n + 4] = dt;
  A[3 * n + 5] = dt;
  B[4 * m + 0] = dt;
  B[5 * m + 1] = dt;
}

/* quad12 (build-defined): rigid-body quadrotor, explicit Euler.
 * state [px,py,pz, phi,theta,psi, vx,vy,vz, p,q,r]; input = rotor thrust deviations from hover
 * (thrust_i = mass*g/4 + u_i).  sys_par = {mass, g, arm, Ix, Iy, Iz, ctau}. */
static void quad12_f(const cfg_t* c, const double* x, const double* u, double* f) {
  const double mass = c->sys_par[0], g = c->sys_par[1], arm = c->sys_par[2];
  const double Ix = c->sys_par[3], Iy = c->sys_par[4], Iz = c->sys_par[5], ct = c->sys_par[6];
  const double sph = sin(x[3]), cph = cos(x[3]), sth = sin(x[4]), cth = cos(x[4]);
  const double sps = sin(x[5]), cps = cos(x[5]);
  const double tth = sth / cth;
  const double T = mass * g + (u[0] + u[1] + u[2] + u[3]);
  const double p = x[9], q = x[10], r = x[11];
  f[0] = x[6];
  f[1] = x[7];
  f[2] = x[8];
  f[3] = p + q * sph * tth + r * cph * tth;
  f[4] = q * cph - r * sph;
  f[5] = (q * sph + r * cph) / cth;
  f[6] = (T / mass) * (cph * sth * cps + sph * sps);
  f[7] = (T / mass) * (cph * sth * sps - sph * cps);
  f[8] = (T / mass) * (cph * cth) - g;
  f[9] = ((Iy - Iz) / Ix) * q * r + arm * (u[1] - u[3]) / Ix;
  f[10] = ((Iz - Ix) / Iy) * p * r + arm * (u[2] - u[0]) / Iy;
  f[11] = ((Ix - Iy) / Iz) * p * q + ct * (u[0] - u[1] + u[2] - u[3]) / Iz;
}

static void quad12_step(const cfg_t* c, const double* x, const double* u, double* xn) {
  double f[12];
  quad12_f(c, x, u, f);
  for (int i = 0; i < 12; i++) xn[i] = x[i] + c->dt * f[i];
}

static void quad12_jac(const cfg_t* c, const double* xe, const double* u, double* A, double* B) {
  const int n = 12, m = 4;
  const double dt = c->dt;
  const double mass = c->sys_par[0], g = c->sys_par[1], arm = c->sys_par[2];
  const double Ix = c->sys_par[3], Iy = c->sys_par[4], Iz = c->sys_par[5], ct = c->sys_par[6];
  const double sph = sin(xe[3]), cph = cos(xe[3]), sth = sin(xe[4]), cth = cos(xe[4]);
  const double sps = sin(xe[5]), cps = cos(xe[5]);
  const double tth = sth / cth, sec2 = 1.0 / (cth * cth);
  const double T = mass * g + (u[0] + u[1] + u[2] + u[3]);
  const double Tm = T / mass;
  const double p = xe[9], q = xe[10], r = xe[11];
  double F[12 * 12];
  memset(F, 0, sizeof(F));
  memset(B, 0, sizeof(double) * n * m);
  /* position */
  F[0 * n + 6] = 1.0;
  F[1 * n + 7] = 1.0;
  F[2 * n + 8] = 1.0;
  /* euler rates */
  F[3 * n + 3] = (q * cph - r * sph) * tth;
  F[3 * n + 4] = (q * sph + r * cph) * sec2;
  F[3 * n + 9] = 1.0;
  F[3 * n + 10] = sph * tth;
  F[3 * n + 11] = cph * tth;
  F[4 * n + 3] = -q * sph - r * cph;
  F[4 * n + 10] = cph;
  F[4 * n + 11] = -sph;
  F[5 * n + 3] = (q * cph - r * sph) / cth;
  F[5 * n + 4] = (q * sph + r * cph) * sth * sec2;
  F[5 * n + 10] = sph / cth;
  F[5 * n + 11] = cph / cth;
  /* linear acceleration */
  F[6 * n + 3] = Tm * (-sph * sth * cps + cph * sps);
  F[6 * n + 4] = Tm * (cph * cth * cps);
  F[6 * n + 5] = Tm * (-cph * sth * sps + sph * cps);
  F[7 * n + 3] = Tm * (-sph * sth * sps - cph * cps);
  F[7 * n + 4] = Tm * (cph * cth * sps);
  F[7 * n + 5] = Tm * (cph * sth * cps + sph * sps);
  F[8 * n + 3] = Tm * (-sph * cth);
  F[8 * n + 4] = Tm * (-cph * sth);
  /* body rates */
  F[9 * n + 10] = ((Iy - Iz) / Ix) * r;
  F[9 * n + 11] = ((Iy - Iz) / Ix) * q;
  F[10 * n + 9] = ((Iz - Ix) / Iy) * r;
  F[10 * n + 11] = ((Iz - Ix) / Iy) * p;
  F[11 * n + 9] = ((Ix - Iy) / Iz) * q;
  F[11 * n + 10] = ((Ix - Iy) / Iz) * p;
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) A[i * n + j] = (i == j ? 1.0 : 0.0) + dt * F[i * n + j];
  const double ax = (cph * sth * cps + sph * sps) / mass;
  const double ay = (cph * sth * sps - sph * cps) / mass;
  const double az = (cph * cth) / mass;
  for (int j = 0; j < m; j++) {
    B[6 * m + j] = dt * ax;
    B[7 * m + j] = dt * ay;
    B[8 * m + j] = dt * az;
  }
  B[9 * m + 1] = dt * arm / Ix;
  B[9 * m + 3] = -dt * arm / Ix;
  B[10 * m + 2] = dt * arm / Iy;
  B[10 * m + 0] = -dt * arm / Iy;
  B[11 * m + 0] = dt * ct / Iz;
  B[11 * m + 1] = -dt * ct / Iz;
  B[11 * m + 2] = dt * ct / Iz;
  B[11 * m + 3] = -dt * ct / Iz;
}

static void sys_step(const cfg_t* c, const double* x, const double* u, double* xn) {
  switch (c->system_id) {
    case I2LQR_SYS_BICYCLE4: bicycle4_step(c, x, u, xn); break;
    case I2LQR_SYS_BICYCLE6: bicycle6_step(c, x, u, xn); break;
    default: quad12_step(c, x, u, xn); break;
  }
}

static void sys_jac(const cfg_t* c, const double* xe, const double* u, double* A, double* B) {
  switch (c->system_id) {
    case I2LQR_SYS_BICYCLE4: bicycle4_jac(c, xe, u, A, B); break;
    case I2LQR_SYS_BICYCLE6: bicycle6_jac(c, xe, u, A, B); break;
    default: quad12_jac(c, xe, u, A, B); break;
  }
}

/* ------------------------------------------------------------------------------------------ */
/* Helpers                                                                                    */
/* ------------------------------------------------------------------------------------------ */

static double clipd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* np.clip on every input row: control/iterative_ilqr.py:33-40 and :146-149. */
static void clip_u(const cfg_t* c, double* u) {
  for (int a = 0; a < c->m; a++) u[a] = clipd(u[a], -c->u_max[a], c->u_max[a]);
}

/* d^T M d for the top-left n x n block of a MAX_N-strided matrix. */
static double quad_form(const double* M, int ld, const double* d, int n) {
  double acc = 0.0;
  for (int i = 0; i < n; i++) {
    double row = 0.0;
    for (int j = 0; j < n; j++) row += M[i * ld + j] * d[j];
    acc += d[i] * row;
  }
  return acc;
}

/* Obstacle centre seen at horizon index `i`: control/ilqr_helper.py:34-43 (stage, index i) and
 * :131-139 (terminal, index num_horizon).  No dt factor, as in the reference. */
static void obstacle_diff(const double* obs, double px, double py, int i, double* dz, double* dy) {
  const double ox = obs[0], oy = obs[1], spd = obs[4];
  const int opt = (int)obs[5];
  *dz = px - ox;
  *dy = py - oy;
  if (opt == 1) *dy = py - (oy + i * spd);
  if (opt == 2) *dz = px - (ox - i * spd);
}

/* Exponential barrier on the obstacle ellipse (Gauss-Newton form), added to l_x / l_xx:
 * control/ilqr_helper.py:32-51 with repelling_cost_function :59-64. */
static void add_obstacle_terms(const cfg_t* c, const double* obs, double px, double py, int i,
                               double* lx, double* lxx, int n) {
  if (!obs || obs[5] < 0) return;
  double dz, dy;
  obstacle_diff(obs, px, py, i, &dz, &dy);
  const double pa = 1.0 / (obs[2] * obs[2]), pb = 1.0 / (obs[3] * obs[3]);
  const double h = 1 + c->safety_margin - (dz * pa * dz + dy * pb * dy);
  const double hd0 = -2 * pa * dz, hd1 = -2 * pb * dy;
  const double q1 = c->obs_q1, q2 = c->obs_q2;
  const double e = exp(q2 * h);
  lx[0] += q1 * q2 * e * hd0;
  lx[1] += q1 * q2 * e * hd1;
  lxx[0 * n + 0] += q1 * (q2 * q2) * e * (hd0 * hd0);
  lxx[0 * n + 1] += q1 * (q2 * q2) * e * (hd0 * hd1);
  lxx[1 * n + 0] += q1 * (q2 * q2) * e * (hd1 * hd0);
  lxx[1 * n + 1] += q1 * (q2 * q2) * e * (hd1 * hd1);
}

/* add_control_constraint(): control/ilqr_helper.py:83-103, generalised from m=2 to one symmetric
 * box per input: b = q1 e^{q2 (u-umax)} + q1 e^{q2 (-umax-u)}. */
static void control_barrier(const cfg_t* c, const double* u, double* lu, double* luu) {
  const int m = c->m;
  const double q1 = c->ctrl_q1, q2 = c->ctrl_q2;
  for (int a = 0; a < m; a++) {
    const double e_hi = exp(q2 * (u[a] - c->u_max[a]));
    const double e_lo = exp(q2 * (-c->u_max[a] - u[a]));
    lu[a] += q1 * q2 * e_hi - q1 * q2 * e_lo;
    luu[a * m + a] += q1 * (q2 * q2) * e_hi + q1 * (q2 * q2) * e_lo;
  }
}

/*
 * Regularised inverse of Q_uu: control/iterative_ilqr.py:118-123
 *   w, V = np.linalg.eig(Quu); w[w<0] = 0; w += lamb; inv = V diag(1/w) V^T.
 * np.linalg.eig is the NON-symmetric LAPACK dgeev (unit-2-norm eigenvectors, not orthogonalised).
 * m == 2: closed-form non-symmetric eigen-decomposition with the same normalisation, so the
 * (tiny) asymmetry of Quu propagates as in the reference.  m > 2 (quad12, no reference
 * counterpart): cyclic Jacobi on the symmetrised matrix.
 */
static void eig2_nonsym(const double* M, double* w, double* V /* columns = eigenvectors */) {
  const double a = M[0], b = M[1], c = M[2], d = M[3];
  const double mean = 0.5 * (a + d), hd = 0.5 * (a - d);
  double disc = hd * hd + b * c;
  if (disc < 0) disc = 0;
  const double s = sqrt(disc);
  double l1, l2; /* l1: larger magnitude root first for a stable product form */
  if (mean >= 0) {
    l1 = mean + s;
    l2 = (l1 != 0.0) ? (a * d - b * c) / l1 : 0.0;
  } else {
    l1 = mean - s;
    l2 = (l1 != 0.0) ? (a * d - b * c) / l1 : 0.0;
  }
  if (s == 0.0) l2 = l1 = mean;
  w[0] = l1;
  w[1] = l2;
  for (int e = 0; e < 2; e++) {
    /* eigenvector of w[e] = a non-zero column of (M - w[other] I)  (Cayley-Hamilton) */
    const double lo = w[1 - e];
    const double c0x = a - lo, c0y = c, c1x = b, c1y = d - lo;
    const double n0 = c0x * c0x + c0y * c0y, n1 = c1x * c1x + c1y * c1y;
    double vx, vy, nn;
    if (n0 >= n1) { vx = c0x; vy = c0y; nn = n0; } else { vx = c1x; vy = c1y; nn = n1; }
    if (nn == 0.0) { vx = (e == 0) ? 1.0 : 0.0; vy = (e == 0) ? 0.0 : 1.0; nn = 1.0; }
    const double inv = 1.0 / sqrt(nn);
    V[0 * 2 + e] = vx * inv;
    V[1 * 2 + e] = vy * inv;
  }
}

static void jacobi_sym(int m, double* S /* m x m, destroyed */, double* w, double* V) {
  for (int i = 0; i < m; i++)
    for (int j = 0; j < m; j++) V[i * m + j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 12; sweep++) {
    for (int p = 0; p < m - 1; p++)
      for (int q = p + 1; q < m; q++) {
        const double apq = S[p * m + q];
        if (apq == 0.0) continue;
        const double app = S[p * m + p], aqq = S[q * m + q];
        const double tau = (aqq - app) / (2.0 * apq);
        const double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
        const double cs = 1.0 / sqrt(1.0 + t * t), sn = t * cs;
        for (int k = 0; k < m; k++) { /* columns p, q */
          const double skp = S[k * m + p], skq = S[k * m + q];
          S[k * m + p] = cs * skp - sn * skq;
          S[k * m + q] = sn * skp + cs * skq;
        }
        for (int k = 0; k < m; k++) { /* rows p, q */
          const double spk = S[p * m + k], sqk = S[q * m + k];
          S[p * m + k] = cs * spk - sn * sqk;
          S[q * m + k] = sn * spk + cs * sqk;
        }
        for (int k = 0; k < m; k++) {
          const double vkp = V[k * m + p], vkq = V[k * m + q];
          V[k * m + p] = cs * vkp - sn * vkq;
          V[k * m + q] = sn * vkp + cs * vkq;
        }
      }
  }
  for (int i = 0; i < m; i++) w[i] = S[i * m + i];
}

static void quu_inverse_reg(int m, const double* Quu, double lamb, double* inv) {
  double w[MAXM], V[MAXM * MAXM];
  if (m == 2) {
    eig2_nonsym(Quu, w, V);
  } else {
    double S[MAXM * MAXM];
    for (int i = 0; i < m; i++)
      for (int j = 0; j < m; j++) S[i * m + j] = 0.5 * (Quu[i * m + j] + Quu[j * m + i]);
    jacobi_sym(m, S, w, V);
  }
  for (int e = 0; e < m; e++) {
    if (w[e] < 0) w[e] = 0.0;
    w[e] += lamb;
  }
  for (int i = 0; i < m; i++)
    for (int j = 0; j < m; j++) {
      double acc = 0.0;
      for (int e = 0; e < m; e++) acc += V[i * m + e] * (1.0 / w[e]) * V[j * m + e];
      inv[i * m + j] = acc;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* The path                                                                                   */
/* ------------------------------------------------------------------------------------------ */

/* Nominal rollout + cost: control/iterative_ilqr.py:32-48. */
double orc_rollout(const cfg_t* c, double* X, double* U, const double* x_term) {
  const int n = c->n, m = c->m, N = c->N, S = N + 1;
  double cost = 0.0, x[MAXN], u[MAXM], xn[MAXN], d[MAXN];
  for (int t = 0; t < N; t++) {
    for (int a = 0; a < m; a++) u[a] = U[a * N + t];
    clip_u(c, u);
    for (int a = 0; a < m; a++) U[a * N + t] = u[a];
    for (int i = 0; i < n; i++) x[i] = X[i * S + t];
    sys_step(c, x, u, xn);
    for (int i = 0; i < n; i++) X[i * S + t + 1] = xn[i];
    for (int i = 0; i < n; i++) d[i] = x[i] - c->xtarget[i];
    const double l_state = quad_form(c->Q, MAXN, d, n);
    const double l_ctrl = quad_form(c->R, MAXM, u, m);
    cost = cost + l_state + l_ctrl;
  }
  for (int i = 0; i < n; i++) d[i] = X[i * S + N] - x_term[i];
  cost = cost + quad_form(c->Qt, MAXN, d, n);
  return cost;
}

/* Intermediates of the backward pass, exported for the function-level golden vectors (G1).
 * Any pointer may be NULL. */
typedef struct {
  double* f_x;  /* [n][n][N] */
  double* f_u;  /* [n][m][N] */
  double* l_x;  /* [n][N]    */
  double* l_xx; /* [n][n][N] */
  double* l_u;  /* [m][N]    */
  double* l_uu; /* [m][m][N] */
  double* V_x;  /* [n]   terminal */
  double* V_xx; /* [n][n] terminal */
} orc_bwd_dump;

/* backward_pass(): control/iterative_ilqr.py:88-130 with get_cost_derivation
 * (control/ilqr_helper.py:9-56) and get_cost_final (:106-150). */
void orc_backward_dump(const cfg_t* c, const double* X, const double* U, const double* x_term,
                       double lamb, const double* obs, double* K, double* k, orc_bwd_dump* dump) {
  const int n = c->n, m = c->m, N = c->N, S = N + 1;
  double Vx[MAXN], Vxx[MAXN * MAXN];
  double A[MAXN * MAXN], Bm[MAXN * MAXM];
  double lx[MAXN], lxx[MAXN * MAXN], lu[MAXM], luu[MAXM * MAXM];
  double x[MAXN], xe[MAXN], u[MAXM];

  /* get_cost_final: control/ilqr_helper.py:106-150 (obstacle index = num_horizon, :134-139) */
  for (int i = 0; i < n; i++) x[i] = X[i * S + N] - x_term[i];
  for (int i = 0; i < n; i++) {
    double acc = 0.0;
    for (int j = 0; j < n; j++) acc += 2 * c->Qt[i * MAXN + j] * x[j];
    Vx[i] = acc;
    for (int j = 0; j < n; j++) Vxx[i * n + j] = 2 * c->Qt[i * MAXN + j];
  }
  add_obstacle_terms(c, obs, X[0 * S + N], X[1 * S + N], N, Vx, Vxx, n);
  if (dump && dump->V_x) memcpy(dump->V_x, Vx, sizeof(double) * n);
  if (dump && dump->V_xx) memcpy(dump->V_xx, Vxx, sizeof(double) * n * n);

  for (int t = N - 1; t >= 0; t--) {
    for (int i = 0; i < n; i++) { x[i] = X[i * S + t]; xe[i] = X[i * S + t + 1]; }
    for (int a = 0; a < m; a++) u[a] = U[a * N + t];
    /* f_x, f_u at (x_{t+1}, u_t): control/iterative_ilqr.py:92-99 */
    sys_jac(c, xe, u, A, Bm);
    /* get_cost_derivation: control/ilqr_helper.py:25-55 */
    for (int a = 0; a < m; a++) {
      double acc = 0.0;
      for (int b = 0; b < m; b++) acc += 2 * c->R[a * MAXM + b] * u[b];
      lu[a] = acc;
      for (int b = 0; b < m; b++) luu[a * m + b] = 2 * c->R[a * MAXM + b];
    }
    control_barrier(c, u, lu, luu);
    for (int i = 0; i < n; i++) {
      double acc = 0.0;
      for (int j = 0; j < n; j++) acc += 2 * c->Q[i * MAXN + j] * (x[j] - c->xtarget[j]);
      lx[i] = acc;
      for (int j = 0; j < n; j++) lxx[i * n + j] = 2 * c->Q[i * MAXN + j];
    }
    add_obstacle_terms(c, obs, x[0], x[1], t, lx, lxx, n);
    if (dump) {
      for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
          if (dump->f_x) dump->f_x[(i * n + j) * N + t] = A[i * n + j];
          if (dump->l_xx) dump->l_xx[(i * n + j) * N + t] = lxx[i * n + j];
        }
      for (int i = 0; i < n; i++) {
        for (int a = 0; a < m; a++)
          if (dump->f_u) dump->f_u[(i * m + a) * N + t] = Bm[i * m + a];
        if (dump->l_x) dump->l_x[i * N + t] = lx[i];
      }
      for (int a = 0; a < m; a++) {
        if (dump->l_u) dump->l_u[a * N + t] = lu[a];
        for (int b = 0; b < m; b++)
          if (dump->l_uu) dump->l_uu[(a * m + b) * N + t] = luu[a * m + b];
      }
    }

    /* Q-function blocks: control/iterative_ilqr.py:112-116 (same association as the NumPy
     * expressions: f.T @ V first, then @ f). */
    double Qx[MAXN], Qu[MAXM], Qxx[MAXN * MAXN], Quu[MAXM * MAXM], Qux[MAXM * MAXN];
    double AtV[MAXN * MAXN], BtV[MAXM * MAXN];
    for (int i = 0; i < n; i++) {
      double acc = 0.0;
      for (int r = 0; r < n; r++) acc += A[r * n + i] * Vx[r];
      Qx[i] = lx[i] + acc;
    }
    for (int a = 0; a < m; a++) {
      double acc = 0.0;
      for (int r = 0; r < n; r++) acc += Bm[r * m + a] * Vx[r];
      Qu[a] = lu[a] + acc;
    }
    for (int i = 0; i < n; i++)
      for (int j = 0; j < n; j++) {
        double acc = 0.0;
        for (int r = 0; r < n; r++) acc += A[r * n + i] * Vxx[r * n + j];
        AtV[i * n + j] = acc;
      }
    for (int a = 0; a < m; a++)
      for (int j = 0; j < n; j++) {
        double acc = 0.0;
        for (int r = 0; r < n; r++) acc += Bm[r * m + a] * Vxx[r * n + j];
        BtV[a * n + j] = acc;
      }
    for (int i = 0; i < n; i++)
      for (int j = 0; j < n; j++) {
        double acc = 0.0;
        for (int r = 0; r < n; r++) acc += AtV[i * n + r] * A[r * n + j];
        Qxx[i * n + j] = lxx[i * n + j] + acc;
      }
    for (int a = 0; a < m; a++)
      for (int b = 0; b < m; b++) {
        double acc = 0.0;
        for (int r = 0; r < n; r++) acc += BtV[a * n + r] * Bm[r * m + b];
        Quu[a * m + b] = luu[a * m + b] + acc;
      }
    for (int a = 0; a < m; a++)
      for (int j = 0; j < n; j++) {
        double acc = 0.0;
        for (int r = 0; r < n; r++) acc += BtV[a * n + r] * A[r * n + j];
        Qux[a * n + j] = acc;
      }

    /* regularised inverse and gains: control/iterative_ilqr.py:118-126 */
    double Qinv[MAXM * MAXM], kk[MAXM], KK[MAXM * MAXN];
    quu_inverse_reg(m, Quu, lamb, Qinv);
    for (int a = 0; a < m; a++) {
      double acc = 0.0;
      for (int b = 0; b < m; b++) acc += Qinv[a * m + b] * Qu[b];
      kk[a] = -acc;
      for (int j = 0; j < n; j++) {
        double accK = 0.0;
        for (int b = 0; b < m; b++) accK += Qinv[a * m + b] * Qux[b * n + j];
        KK[a * n + j] = -accK;
      }
    }
    for (int a = 0; a < m; a++) {
      k[a * N + t] = kk[a];
      for (int j = 0; j < n; j++) K[(a * n + j) * N + t] = KK[a * n + j];
    }

    /* value update with the UNregularised Quu: control/iterative_ilqr.py:128-129
     * (K.T @ Quu) @ k and (K.T @ Quu) @ K; Vxx is not re-symmetrised. */
    double KtQ[MAXN * MAXM];
    for (int i = 0; i < n; i++)
      for (int b = 0; b < m; b++) {
        double acc = 0.0;
        for (int a = 0; a < m; a++) acc += KK[a * n + i] * Quu[a * m + b];
        KtQ[i * m + b] = acc;
      }
    for (int i = 0; i < n; i++) {
      double acc = 0.0;
      for (int b = 0; b < m; b++) acc += KtQ[i * m + b] * kk[b];
      Vx[i] = Qx[i] - acc;
    }
    for (int i = 0; i < n; i++)
      for (int j = 0; j < n; j++) {
        double acc = 0.0;
        for (int b = 0; b < m; b++) acc += KtQ[i * m + b] * KK[b * n + j];
        Vxx[i * n + j] = Qxx[i * n + j] - acc;
      }
  }
}

void orc_backward(const cfg_t* c, const double* X, const double* U, const double* x_term,
                  double lamb, const double* obs, double* K, double* k) {
  orc_backward_dump(c, X, U, x_term, lamb, obs, K, k, 0);
}

/* forward_pass(): control/iterative_ilqr.py:133-160.  Stage cost is measured to x_terminal. */
double orc_forward(const cfg_t* c, const double* X, const double* U, const double* x_term,
                   const double* K, const double* k, double* Xn, double* Un) {
  const int n = c->n, m = c->m, N = c->N, S = N + 1;
  double cost = 0.0, x[MAXN], u[MAXM], xn[MAXN], d[MAXN];
  for (int i = 0; i < n; i++) { x[i] = X[i * S + 0]; Xn[i * S + 0] = x[i]; }
  for (int t = 0; t < N; t++) {
    for (int a = 0; a < m; a++) {
      double acc = 0.0;
      for (int j = 0; j < n; j++) acc += K[(a * n + j) * N + t] * (x[j] - X[j * S + t]);
      u[a] = U[a * N + t] + k[a * N + t] + acc;
    }
    clip_u(c, u);
    for (int a = 0; a < m; a++) Un[a * N + t] = u[a];
    sys_step(c, x, u, xn);
    for (int i = 0; i < n; i++) d[i] = x[i] - x_term[i];
    const double l_state = quad_form(c->Q, MAXN, d, n);
    const double l_ctrl = quad_form(c->R, MAXM, u, m);
    cost = cost + l_state + l_ctrl;
    for (int i = 0; i < n; i++) { x[i] = xn[i]; Xn[i * S + t + 1] = xn[i]; }
  }
  for (int i = 0; i < n; i++) d[i] = x[i] - x_term[i];
  cost = cost + quad_form(c->Qt, MAXN, d, n);
  return cost;
}

/*
 * ilqr(): control/iterative_ilqr.py:7-85.  In/out: X (X[:,0] = x0), U, *lamb.
 * early_exit != 0: the reference behaviour (exits at :78-80 and :83-84), at most max_iter passes.
 * early_exit == 0: exactly max_iter passes (the fixed-count benchmark unit).
 * Returns the number of executed iterations; *status is an I2LQR_ST_* word; *cost_out the cost of
 * the returned trajectory; K, k hold the last iteration's gains.
 */
int orc_ilqr(const cfg_t* c, int max_iter, int early_exit, double* X, double* U,
             const double* x_term, double* lamb, const double* obs, double* K, double* k,
             double* cost_out, int* status) {
  const int n = c->n, m = c->m, N = c->N, S = N + 1;
  double Xn[MAXN * (MAXH + 1)], Un[MAXM * MAXH];
  int it = 0, st = early_exit ? I2LQR_ST_MAX_ITER : I2LQR_ST_RUNNING;
  double cost_ret = 0.0;
  for (it = 0; it < max_iter; it++) {
    const double cost = orc_rollout(c, X, U, x_term);
    orc_backward(c, X, U, x_term, *lamb, obs, K, k);
    const double cost_new = orc_forward(c, X, U, x_term, K, k, Xn, Un);
    cost_ret = cost;
    if (cost_new < cost) {
      memcpy(X, Xn, sizeof(double) * n * S);
      memcpy(U, Un, sizeof(double) * m * N);
      *lamb /= c->lamb_factor;
      cost_ret = cost_new;
      if (fabs((cost_new - cost) / cost) < c->eps) {
        if (st == I2LQR_ST_RUNNING || early_exit) st = I2LQR_ST_CONVERGED;
        if (early_exit) { it++; break; }
      }
    } else {
      *lamb *= c->lamb_factor;
      if (*lamb > c->max_lamb) {
        if (st == I2LQR_ST_RUNNING || early_exit) st = I2LQR_ST_LAMB_OVERFLOW;
        if (early_exit) { it++; break; }
      }
    }
  }
  if (!isfinite(cost_ret)) st = I2LQR_ST_NONFINITE;
  if (cost_out) *cost_out = cost_ret;
  if (status) *status = st;
  return it;
}

/* Relaxed terminal cost of one candidate: utils/base.py:427-437. */
double orc_relax_cost(const cfg_t* c, const double* X, const double* x_term, int qfun,
                      int outer_iter, int max_relax_iter) {
  const int n = c->n, N = c->N, S = N + 1;
  double ss = 0.0;
  for (int i = 0; i < n; i++) {
    const double d = X[i * S + N] - x_term[i];
    ss += d * d;
  }
  const double nrm = sqrt(ss), scale = pow(10.0, outer_iter);
  for (int i = 1; i <= max_relax_iter; i++) {
    if (nrm <= 80.0 * i / scale) return (double)qfun + N + 100 * i;
    if (nrm > 80.0 * max_relax_iter / scale) return INFINITY;
  }
  return INFINITY; /* NaN norm: the reference would leave cost_it unset; treated as infeasible */
}

/* ------------------------------------------------------------------------------------------ */
/* Batched drivers (problem-major layout) — used by tests and by bench.py's cpu_baseline leg   */
/* ------------------------------------------------------------------------------------------ */

void orc_ilqr_batch(const cfg_t* c, int64_t B, int max_iter, int early_exit, double* X, double* U,
                    const double* x_term, double* lamb, const double* obs, double* K, double* k,
                    double* cost, int32_t* iters, int32_t* status) {
  const int n = c->n, m = c->m, N = c->N;
#pragma omp parallel for schedule(static)
  for (int64_t b = 0; b < B; b++) {
    double Ktmp[MAXM * MAXN * MAXH], ktmp[MAXM * MAXH];
    int st = 0;
    double cst = 0.0;
    double* Kb = K ? K + b * (int64_t)(m * n * N) : Ktmp;
    double* kb = k ? k + b * (int64_t)(m * N) : ktmp;
    const int it = orc_ilqr(c, max_iter, early_exit, X + b * (int64_t)(n * (N + 1)),
                            U + b * (int64_t)(m * N), x_term + b * n, lamb + b,
                            obs ? obs + b * I2LQR_OBS_WORDS : 0, Kb, kb, &cst, &st);
    if (cost) cost[b] = cst;
    if (iters) iters[b] = it;
    if (status) status[b] = st;
  }
}

void orc_backward_batch(const cfg_t* c, int64_t B, const double* X, const double* U,
                        const double* x_term, const double* lamb, const double* obs, double* K,
                        double* k) {
  const int n = c->n, m = c->m, N = c->N;
  for (int64_t b = 0; b < B; b++)
    orc_backward(c, X + b * (int64_t)(n * (N + 1)), U + b * (int64_t)(m * N), x_term + b * n,
                 lamb[b], obs ? obs + b * I2LQR_OBS_WORDS : 0, K + b * (int64_t)(m * n * N),
                 k + b * (int64_t)(m * N));
}

void orc_forward_batch(const cfg_t* c, int64_t B, const double* X, const double* U,
                       const double* x_term, const double* K, const double* k, double* Xn,
                       double* Un, double* cost) {
  const int n = c->n, m = c->m, N = c->N;
  for (int64_t b = 0; b < B; b++)
    cost[b] = orc_forward(c, X + b * (int64_t)(n * (N + 1)), U + b * (int64_t)(m * N),
                          x_term + b * n, K + b * (int64_t)(m * n * N), k + b * (int64_t)(m * N),
                          Xn + b * (int64_t)(n * (N + 1)), Un + b * (int64_t)(m * N));
}

void orc_rollout_batch(const cfg_t* c, int64_t B, double* X, double* U, const double* x_term,
                       double* cost) {
  const int n = c->n, m = c->m, N = c->N;
  for (int64_t b = 0; b < B; b++)
    cost[b] = orc_rollout(c, X + b * (int64_t)(n * (N + 1)), U + b * (int64_t)(m * N),
                          x_term + b * n);
}

void orc_relax_cost_batch(const cfg_t* c, int64_t B, const double* X, const double* x_term,
                          const int32_t* qfun, int outer_iter, int max_relax_iter,
                          double* cost_it) {
  const int n = c->n, N = c->N;
  for (int64_t b = 0; b < B; b++)
    cost_it[b] = orc_relax_cost(c, X + b * (int64_t)(n * (N + 1)), x_term + b * n, qfun[b],
                                outer_iter, max_relax_iter);
}

/* Plant step / Jacobians exported for the dynamics golden vector (G7) and finite-difference
 * checks of the build-defined systems. */
void orc_sys_step(const cfg_t* c, const double* x, const double* u, double* xn) {
  sys_step(c, x, u, xn);
}
void orc_sys_jac(const cfg_t* c, const double* xe, const double* u, double* A, double* B) {
  sys_jac(c, xe, u, A, B);
}
void orc_quu_inverse_reg(int m, const double* Quu, double lamb, double* inv) {
  quu_inverse_reg(m, Quu, lamb, inv);
}
int orc_config_size(void) { return (int)sizeof(cfg_t); }
int orc_round_size(void) { return (int)sizeof(i2lqr_round); }  /* tests/test_abi.py: ctypes mirror */

/* Threads used by orc_ilqr_batch (OpenMP); 0 restores the runtime default.  Returns the count in
 * effect. */
#ifdef _OPENMP
#include <omp.h>
int orc_set_threads(int n) {
  if (n > 0) omp_set_num_threads(n);
  return omp_get_max_threads();
}
#else
int orc_set_threads(int n) { (void)n; return 1; }
#endif

"""Synthetic problem batches for the BASELINE.json configurations.

Seed numpy.random.default_rng(20230228).  Every terminal target is reachable: it is the end of a
random-input rollout from x0 (mirrors safe-set points being states of a feasible trajectory,
utils/base.py:411).  Half of the batch carries the reference's static obstacle (31, -3, 8, 6),
the rest none.  Host-side NumPy only; returns float64 arrays in the problem-major layout.
"""
from __future__ import annotations

import numpy as np

from . import _abi
from ._abi import I2lqrConfig, default_config

SEED = 20230228

# name -> (system, N, dt, dtype, per-GPU batch): BASELINE.json configs[1..4]
CONFIGS = {
    "config2": dict(system="bicycle6", N=20, dt=0.25, dtype="f64", batch=1024),
    "config3": dict(system="bicycle6", N=20, dt=0.25, dtype="f32", batch=65536),
    "config4": dict(system="bicycle6", N=20, dt=0.25, dtype="f64", batch=131072),
    "config5": dict(system="quad12", N=50, dt=0.02, dtype="f64", batch=65536),
    "reference": dict(system="bicycle4", N=6, dt=1.0, dtype="f64", batch=16),
}


def config_for(name: str, dtype: str | None = None) -> I2lqrConfig:
    w = CONFIGS[name]
    return default_config(w["system"], w["N"], dtype or w["dtype"], dt=w["dt"])


def _step(cfg: I2lqrConfig, x: np.ndarray, u: np.ndarray) -> np.ndarray:
    """Vectorised plant step x[B, n], u[B, m] -> x_next (same formulas as the HIP kernels)."""
    dt = cfg.dt
    xn = x.copy()
    if cfg.system_id == _abi.SYS_BICYCLE4:
        w = x[:, 2] * dt + u[:, 0] * dt * dt / 2
        xn[:, 0] = x[:, 0] + np.cos(x[:, 3]) * w
        xn[:, 1] = x[:, 1] + np.sin(x[:, 3]) * w
        xn[:, 2] = x[:, 2] + u[:, 0] * dt
        xn[:, 3] = x[:, 3] + u[:, 1] * dt
    elif cfg.system_id == _abi.SYS_BICYCLE6:
        w = x[:, 2] * dt + x[:, 4] * dt * dt / 2
        xn[:, 0] = x[:, 0] + np.cos(x[:, 3]) * w
        xn[:, 1] = x[:, 1] + np.sin(x[:, 3]) * w
        xn[:, 2] = x[:, 2] + x[:, 4] * dt
        xn[:, 3] = x[:, 3] + x[:, 5] * dt
        xn[:, 4] = x[:, 4] + u[:, 0] * dt
        xn[:, 5] = x[:, 5] + u[:, 1] * dt
    else:
        mass, g, arm, Ix, Iy, Iz, ct = list(cfg.sys_par)[:7]
        sph, cph = np.sin(x[:, 3]), np.cos(x[:, 3])
        sth, cth = np.sin(x[:, 4]), np.cos(x[:, 4])
        sps, cps = np.sin(x[:, 5]), np.cos(x[:, 5])
        tth = sth / cth
        T = mass * g + u.sum(axis=1)
        p, q, r = x[:, 9], x[:, 10], x[:, 11]
        f = np.zeros_like(x)
        f[:, 0:3] = x[:, 6:9]
        f[:, 3] = p + q * sph * tth + r * cph * tth
        f[:, 4] = q * cph - r * sph
        f[:, 5] = (q * sph + r * cph) / cth
        f[:, 6] = (T / mass) * (cph * sth * cps + sph * sps)
        f[:, 7] = (T / mass) * (cph * sth * sps - sph * cps)
        f[:, 8] = (T / mass) * (cph * cth) - g
        f[:, 9] = ((Iy - Iz) / Ix) * q * r + arm * (u[:, 1] - u[:, 3]) / Ix
        f[:, 10] = ((Iz - Ix) / Iy) * p * r + arm * (u[:, 2] - u[:, 0]) / Iy
        f[:, 11] = ((Ix - Iy) / Iz) * p * q + ct * (u[:, 0] - u[:, 1] + u[:, 2] - u[:, 3]) / Iz
        xn = x + dt * f
    return xn


def make_batch(cfg: I2lqrConfig, B: int, seed: int = SEED, offset: int = 0,
               variant: str | None = None) -> dict:
    """B problems: X[B,n,N+1] (x0 in [:, :, 0], rest 0), U = 0, x_term, lamb = 1, obs[B,6].

    `offset` selects a disjoint slice of the (conceptually infinite) problem stream so that rank r
    of a sharded run draws problems [offset, offset + B).
    `variant`: other distributions of the same stream, for schedules and thresholds that must not be
    fitted to the default one (tools/solve_bench.py): "all_obstacle" — every problem carries the
    obstacle; "far_targets" — the target is the end of a rollout twice as long as the horizon (not
    reachable within it: longer accept / reject histories, more survivors per chunk)."""
    rng = np.random.default_rng([seed, offset])
    n, m, N = cfg.n, cfg.m, cfg.N
    x0 = np.zeros((B, n))
    if cfg.system_id in (_abi.SYS_BICYCLE4, _abi.SYS_BICYCLE6):
        x0[:, 0] = rng.uniform(0, 200, B)
        x0[:, 1] = rng.uniform(-5, 5, B)
        x0[:, 2] = rng.uniform(0, 10, B)
        x0[:, 3] = rng.uniform(-0.5, 0.5, B)
        if cfg.system_id == _abi.SYS_BICYCLE4:
            urand = np.stack([rng.uniform(-2, 2, (N, B)), rng.uniform(-0.3, 0.3, (N, B))], -1)
        else:
            urand = np.stack([rng.uniform(-1, 1, (N, B)), rng.uniform(-0.3, 0.3, (N, B))], -1)
    else:
        x0[:] = rng.normal(0.0, 0.1, (B, n))
        # common-mode thrust +-0.5 N, differential +-0.02 N: stays near hover over the horizon
        urand = rng.uniform(-0.5, 0.5, (N, B, 1)) + rng.uniform(-0.02, 0.02, (N, B, m))
    x = x0.copy()
    for t in range(N):
        x = _step(cfg, x, urand[t])
    if variant == "far_targets":
        for t in range(N):
            x = _step(cfg, x, urand[N - 1 - t])
    X = np.zeros((B, n, N + 1))
    X[:, :, 0] = x0
    obs = np.tile(np.array([31.0, -3.0, 8.0, 6.0, 0.0, 0.0]), (B, 1))
    if variant != "all_obstacle":
        obs[1::2, 5] = -1.0  # every second problem: no obstacle
    if variant not in (None, "all_obstacle", "far_targets"):
        raise ValueError(f"unknown workload variant {variant!r}")
    if cfg.system_id == _abi.SYS_QUAD12:
        obs[:, :4] = [2.0, 2.0, 0.5, 0.5]
    return dict(X=X, U=np.zeros((B, m, N)), x_term=x, lamb=np.ones(B), obs=obs)


def algorithmic_bytes_per_iteration(cfg: I2lqrConfig) -> int:
    """compulsory HBM bytes per iLQR iteration per problem if the state
    round-trips HBM once per iteration: read X, U, x_term, lamb; write X', U', K, k, cost, lamb.
    words = 2 n (N+1) + 3 m N + m n N + n + 3."""
    n, m, N = cfg.n, cfg.m, cfg.N
    words = 2 * n * (N + 1) + 3 * m * N + m * n * N + n + 3
    return words * (8 if cfg.dtype == _abi.F64 else 4)


def algorithmic_flops_per_iteration(cfg: I2lqrConfig) -> int:
    """flops of one iLQR iteration per problem in the reference's dense form,
    N (4 n^3 + 6 m n^2 + 6 m^2 n + 2 n^2 + 4 n m) for the backward pass (the rollouts and barrier
    terms add a few per cent): 600 k at n=12, m=4, N=50; 35 k at n=6, m=2, N=20.  The kernels
    execute fewer (they fold the sparsity of [A | B] into the instruction stream)."""
    n, m, N = cfg.n, cfg.m, cfg.N
    return N * (4 * n ** 3 + 6 * m * n ** 2 + 6 * m ** 2 * n + 2 * n ** 2 + 4 * n * m)


#!/usr/bin/env python3
"""Randomised parity campaign: every kernel family against the CPU oracle on configurations the
test-suite does not pin — random horizons, ragged batch sizes, time steps, regularisation, static
and moving obstacles, inputs outside the box — fused iterations (with gains) and solves to
termination.  Round-off is amplified by the exponential barriers on some of these problems (inputs
1.5 x outside the box), so every problem's deviation is priced against the oracle's OWN sensitivity:
the same problem solved by the oracle with U0 moved by one ulp.  Prints one line per (plant, family):
the worst deviation on the problems whose sensitivity is below 1e-12, and the worst ratio deviation /
sensitivity over the others; exit code 1 if a well-conditioned problem exceeds the
test-suite's bounds or a ratio exceeds 100.  Run on a GPU box: python tools/parity_campaign.py [trials]"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: E402
from helpers import dev_batch, to_host  # noqa: E402
from ilqr_iterative_tasks_amd import BatchedILQR, default_config, workloads  # noqa: E402
from oracle import oracle as orc  # noqa: E402



def per_problem(a, b):
    """rel_err of helpers.py per leading-axis entry."""
    a = np.asarray(a, dtype=np.float64).reshape(len(a), -1)
    b = np.asarray(b, dtype=np.float64).reshape(len(b), -1)
    return np.abs(a - b).max(axis=1) / np.maximum(np.abs(b).max(axis=1), 1e-300)


def worst_of(err, sens, mask):
    """(worst err among well-conditioned problems, worst err / sensitivity) over `mask`."""
    if not mask.any():
        return 0.0, 0.0
    e, s_ = err[mask], sens[mask]
    well = s_ < 1e-12
    return float(e[well].max(initial=0.0)), float((e[~well] / s_[~well]).max(initial=0.0))


TRIALS = int(sys.argv[1]) if len(sys.argv) > 1 else 6
# family -> (layout id, options)
FAMILIES = {
    "wave": (0, {"group_lanes": 64}),
    "group": (0, {"group_lanes": 8, "speculate": 0, "group_workspace": 0}),
    "group-ws": (0, {"group_lanes": 8, "speculate": 0, "group_workspace": 1}),
    "spec": (0, {"group_lanes": 8, "speculate": 1}),
    "quad16": (0, {"group_lanes": 16}),
    "row16": (0, {"group_lanes": 16, "speculate": 0}),  # bicycles: sixteen lanes, DPP row broadcasts
    "lane": (1, {}),
    "tiled": (2, {}),
}
PLANTS = {
    "bicycle4": dict(N=[1, 2, 5, 6, 13, 20], dt=[0.25, 1.0], fam=["wave", "group", "row16", "group-ws", "spec", "lane", "tiled"]),
    "bicycle6": dict(N=[2, 7, 20, 31], dt=[0.1, 0.25], fam=["wave", "group", "row16", "group-ws", "spec", "lane", "tiled"]),
    "quad12": dict(N=[3, 10, 50], dt=[0.02], fam=["wave", "quad16", "lane", "tiled"]),
    # stage weights Q, R != 0 and a non-zero xtarget (utils/base.py:243-246): the column kernels are
    # built for Q = R = 0; round 5: quad12's one-problem-per-lane kernels take them
    "bicycle6+QR": dict(system="bicycle6", weights=True, N=[2, 7, 20], dt=[0.1, 0.25],
                        fam=["wave", "lane", "tiled"]),
    "quad12+QR": dict(system="quad12", weights=True, N=[3, 10, 50], dt=[0.02],
                      fam=["wave", "lane", "tiled"]),
}


def with_weights(cfg, wrng):
    """Random symmetric positive semi-definite Q, positive definite R, a target off the origin."""
    n, m = cfg.n, cfg.m
    A = wrng.normal(0, 0.1, (n, n))
    cfg.set_matrix("Q", A @ A.T + np.diag(wrng.uniform(0.0, 0.1, n)))
    Bm = wrng.normal(0, 0.05, (m, m))
    cfg.set_matrix("R", Bm @ Bm.T + np.diag(wrng.uniform(0.02, 0.1, m)))
    cfg.xtarget[:n] = wrng.normal(0, 0.2, n)
    # With Q != 0 the reference compares a forward cost measured to x_terminal with a nominal cost
    # measured to xtarget (control/iterative_ilqr.py:43 vs :151): on these problems every step is
    # "accepted", no solve converges, and after 150 iterations lamb has decayed to 1e-150 — an
    # unregularised, chaotic recursion in which three implementations give three answers (oracle
    # 722, one-problem-per-wavefront kernel 85.2, lane kernel 84.6 on one bicycle6 problem, with
    # identical accept histories).  Solves are compared over the first 12 iterations.
    cfg.max_iter = 12
    return cfg


rng = np.random.default_rng(20261003)
worst = {}
bad = 0
t0 = time.time()
for plant, spec in PLANTS.items():
    system = spec.get("system", plant)
    for trial in range(TRIALS):
        N = int(rng.choice(spec["N"]))
        dt = float(rng.choice(spec["dt"]))
        B = int(rng.choice([64, 128, 192])) if system == "quad12" else int(rng.choice([64, 192, 448, 1024]))
        cfg0 = default_config(system, N, "f64", dt=dt)
        if spec.get("weights"):  # (drawn for these plants only: the other plants' sample is round 4's)
            wseed = int(rng.integers(1 << 30))
            with_weights(cfg0, np.random.default_rng(wseed))
        host = workloads.make_batch(cfg0, B, seed=1000 + trial)
        host["lamb"] = 10.0 ** rng.integers(-6, 3, B).astype(float)
        u_max = np.array(cfg0.u_max[:cfg0.m])[None, :, None]
        scale = 0.05 if system == "quad12" else 1.5  # beyond the box for the bicycles
        host["U"] = rng.uniform(-1, 1, host["U"].shape) * scale * u_max
        if system != "quad12":
            ob = host["obs"]
            ob[:, 0] = host["X"][:, 0, 0] + rng.uniform(5, 40, B)
            ob[:, 1] = rng.uniform(-6, 6, B)
            ob[:, 2:4] = rng.uniform(4, 30, (B, 2))
            ob[:, 4] = rng.uniform(0, 1.0, B)
            ob[:, 5] = rng.choice([-1, 0, 1, 2], B)
        iters = int(rng.choice([1, 3, 6]))
        ref_it = orc.ilqr_batch(cfg0, host["X"], host["U"], host["x_term"], host["lamb"], host["obs"],
                                max_iter=iters, early_exit=False)
        ref_so = orc.ilqr_batch(cfg0, host["X"], host["U"], host["x_term"], host["lamb"], host["obs"])
        U1 = host["U"] * (1.0 + rng.choice([-1.0, 1.0], host["U"].shape) * 2.0 ** -52)
        p_it = orc.ilqr_batch(cfg0, host["X"], U1, host["x_term"], host["lamb"], host["obs"],
                              max_iter=iters, early_exit=False)
        p_so = orc.ilqr_batch(cfg0, host["X"], U1, host["x_term"], host["lamb"], host["obs"])
        sens_it = np.maximum(per_problem(p_it["X"], ref_it["X"]), per_problem(p_it["K"], ref_it["K"]))
        sens_so = per_problem(p_so["X"], ref_so["X"])
        # a perturbed solve that took another accept / reject history: maximally sensitive
        sens_it[p_it["lamb"] != ref_it["lamb"]] = 1.0
        sens_so[(p_so["iters"] != ref_so["iters"]) | (p_so["lamb"] != ref_so["lamb"])] = 1.0
        for fam in spec["fam"]:
            lid, opts = FAMILIES[fam]
            if fam == "tiled" and B % 64:
                continue
            cfg = default_config(system, N, "f64", dt=dt, layout=lid)
            if spec.get("weights"):
                with_weights(cfg, np.random.default_rng(wseed))
            s = BatchedILQR(cfg)
            try:
                for k, v in opts.items():
                    s.set_option(k, v)
                it = s.iterate(dev_batch(s, host), iters)
                so = s.solve(dev_batch(s, host))
            except Exception as e:  # noqa: BLE001
                msg = str(e)
                if "UNSUPPORTED" in msg or "needs" in msg or "-3" in msg or "built for" in msg:
                    s.close()
                    continue  # e.g. the speculative kernel's buffers do not fit this horizon
                raise
            w = worst.setdefault((plant, fam), dict(n=0, flips_it=0, flips_so=0, probs=0, X=0.0, K=0.0,
                                                    cost=0.0, Xs=0.0, ratio=0.0, ill=0))
            same = it["lamb"].cpu().numpy() == ref_it["lamb"]
            w["n"] += 1
            w["probs"] += B
            w["flips_it"] += int((~same).sum())
            w["ill"] += int((sens_it >= 1e-12).sum())
            c = it["cost"].cpu().numpy()
            for key, err in (("X", per_problem(to_host(s, it["X"]), ref_it["X"])),
                             ("K", per_problem(to_host(s, it["K"]), ref_it["K"])),
                             ("cost", np.abs(c - ref_it["cost"]) / np.maximum(np.abs(ref_it["cost"]), 1e-300))):
                e, r = worst_of(err, sens_it, same)
                w[key] = max(w[key], e)
                w["ratio"] = max(w["ratio"], r / (10.0 if key == "cost" else 1.0))  # cost: 10 x looser, as the suite
            sames = (so["iters"].cpu().numpy() == ref_so["iters"]) & (so["lamb"].cpu().numpy() == ref_so["lamb"])
            w["flips_so"] += int((~sames).sum())
            e, r = worst_of(per_problem(to_host(s, so["X"]), ref_so["X"]), sens_so, sames)
            w["Xs"], w["ratio"] = max(w["Xs"], e), max(w["ratio"], r)
            st = so["status"].cpu().numpy()
            if not set(np.unique(st)) <= {1, 2, 3} or not (st[sames] == ref_so["status"][sames]).all():
                print("STATUS mismatch", plant, fam, N, B)
                bad += 1
            s.close()
print("deviations: worst over the problems whose oracle sensitivity (one ulp on U0) is < 1e-12; ratio: worst "
      "deviation / sensitivity over the others ('sensitive')")
print(f"{'plant':11s} {'family':9s} configs problems sensitive flipped(iterate) flipped(solve)   X(iterate)  K(iterate)  cost(iterate)  X(solve)   ratio")
for (plant, fam), w in worst.items():
    fi, fs = w["flips_it"] / w["probs"], w["flips_so"] / w["probs"]
    print(f"{plant:11s} {fam:9s} {w['n']:7d} {w['probs']:8d} {w['ill']:9d} {fi:16.4f} {fs:14.4f}   {w['X']:.2e}   {w['K']:.2e}   {w['cost']:.2e}   {w['Xs']:.2e}   {w['ratio']:.1f}")
    if (w["X"] > 1e-8 or w["Xs"] > 1e-8 or w["cost"] > 1e-7 or w["K"] > 1e-6 or fi > 0.03 or fs > 0.03
            or w["ratio"] > 100):
        bad += 1
print(f"bounds of the test-suite (X 1e-8, K 1e-6, cost 1e-7, <= 3 % flipped accept / reject histories; ratio <= 100): "
      f"{'EXCEEDED on %d lines' % bad if bad else 'held everywhere'}; {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)

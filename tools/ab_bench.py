#!/usr/bin/env python3
"""Interleaved A/B timing of the fused iterate kernel across layouts / dtypes / batches in ONE
process (cdna guide §5.4 rule 24): variants x rounds, median and min per variant."""
import argparse
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

from ilqr_iterative_tasks_amd import BatchedILQR, workloads

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="config2")
ap.add_argument("--variants", default="lane:f64:65536,tiled:f64:65536")
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--cold", action="store_true", help="every timed launch runs on its own copy of "
                "the batch, filled before the first launch (no restore copy right before a launch "
                "leaves its inputs warm in the 256 MB Infinity Cache): what bench.py measures")
ap.add_argument("--weights", action="store_true", help="stage weights Q, R != 0 (seeded random "
                "positive semi-definite Q, positive definite R, a target off the origin)")
ap.add_argument("--json", default=None, help="write the run as data (VERDICT r5 #10): per-round times "
                "of every variant, medians, library hashes, device — the file a delta is quoted from")
args = ap.parse_args()

LAY = {"wave": 0, "lane": 1, "tiled": 2}
runs = []
for v in args.variants.split(","):
    parts = v.split(":")
    layout, dtype, B = parts[:3]
    lib_path = (parts[3] if len(parts) > 3 else None) or None  # e.g. tools/_diag/libold.so
    # optional 5th field: i2lqr_set_option settings of this variant, e.g. defer_states=1;reroll_nominal=0
    opts = dict(kv.split("=") for kv in parts[4].split(";")) if len(parts) > 4 else {}
    B = int(B)
    cfg = workloads.config_for(args.workload, dtype)
    cfg.layout = LAY[layout]
    if args.weights:
        wr = np.random.default_rng(11)
        A = wr.normal(0, 0.1, (cfg.n, cfg.n))
        cfg.set_matrix("Q", A @ A.T + np.diag(wr.uniform(0.0, 0.1, cfg.n)))
        Bm = wr.normal(0, 0.05, (cfg.m, cfg.m))
        cfg.set_matrix("R", Bm @ Bm.T + np.diag(wr.uniform(0.02, 0.1, cfg.m)))
        cfg.xtarget[:cfg.n] = wr.normal(0, 0.2, cfg.n)
    solver = BatchedILQR(cfg, lib_path=lib_path)
    for key, val in opts.items():
        solver.set_option(key, int(val))
    host = workloads.make_batch(cfg, B)
    dev = lambda a: solver.to_native(torch.as_tensor(a).to(solver.device, solver.dtype))
    buf = solver.alloc(B)
    for key in ("X", "U", "x_term", "lamb"):
        buf[key].copy_(dev(host[key]))
    buf["obs"] = dev(host["obs"])
    init = {k: buf[k].clone() for k in ("X", "U", "lamb")}
    if args.cold:  # one set of in/out arrays per round, the read-only inputs and outputs shared
        sets = []
        for _ in range(args.rounds + 1):
            b2 = dict(buf)
            b2.update({k: init[k].clone() for k in init})
            sets.append(b2)
        buf = sets
    runs.append((v, solver, buf, init, B, []))

e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
rng = np.random.default_rng(0)
for r in range(args.rounds + 1):
    # a fresh order every round: what ran just before (cache / MALL state, clocks) is not
    # systematically the same variant
    for v, solver, buf, init, B, times in [runs[i] for i in rng.permutation(len(runs))]:
        if args.cold:
            buf = buf[r]
        else:
            for k in init:
                buf[k].copy_(init[k])
        torch.cuda.synchronize()
        e0.record()
        solver.iterate(buf, args.iters)
        e1.record()
        torch.cuda.synchronize()
        if r > 0:
            times.append(e0.elapsed_time(e1))
doc = {"tool": "tools/ab_bench.py", "argv": sys.argv[1:], "workload": args.workload,
       "iterations_per_launch": args.iters, "rounds": args.rounds, "cold": bool(args.cold),
       "device": torch.cuda.get_device_name(0), "variants": []}
for v, solver, buf, init, B, times in runs:
    t = np.array(times)
    import hashlib
    lib_file = getattr(solver.lib, "_name", None)
    doc["variants"].append({
        "variant": v, "batch": B, "ms_per_round": [float(x) for x in t], "ms_median": float(np.median(t)),
        "ms_min": float(t.min()), "Mits_median": float(B * args.iters / np.median(t) / 1e3),
        "kernel": solver.iterate_kernel(B), "library": lib_file,
        "library_sha256": hashlib.sha256(Path(lib_file).read_bytes()).hexdigest()[:16] if lib_file else None})
    print(f"{v:56s} median {np.median(t):9.3f} ms  min {t.min():9.3f} ms  -> "
          f"{B * args.iters / np.median(t) / 1e3:8.1f} M it/s (median)  "
          f"{B * args.iters / t.min() / 1e3:8.1f} (best)")
if len(doc["variants"]) >= 2:
    base = doc["variants"][0]["ms_median"]
    for rec in doc["variants"][1:]:
        rec["speedup_over_first"] = base / rec["ms_median"]
if args.json:
    Path(args.json).parent.mkdir(parents=True, exist_ok=True)
    Path(args.json).write_text(json.dumps(doc, indent=1))

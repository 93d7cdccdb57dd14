#!/usr/bin/env python3
"""Solve-to-termination timing (reference exits, 1..150 iterations per problem).

    solve_bench.py spec...            spec = layout:dtype:batch[:compaction_min_batch[:wave_tail]]
    solve_bench.py --schedule [B...]  the data-driven schedule of the chunked solve against
                                      hand-tuned ones ("first_chunk" 6 .. 20, no extension chunks)
                                      on three distributions of the workload: the bench's, every
                                      problem with the obstacle, targets twice as far
                                      (VERDICT r4 #3); JSON to gpurun_out/solve_schedule.json
    solve_bench.py --ab opt=val [B...]  the automatic solve against the same solve with one option
                                      set (e.g. helper_wavefront=0: one-wavefront chunks), same
                                      process, alternating, on the three distributions
"""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

from ilqr_iterative_tasks_amd import BatchedILQR, workloads


def time_solve(solver, host, reps=5):
    B = host["X"].shape[0]
    dev = lambda a: solver.to_native(torch.as_tensor(a).to(solver.device, solver.dtype))
    base = solver.alloc(B, want_gains=False)
    for key in ("X", "U", "x_term", "lamb"):
        base[key].copy_(dev(host[key]))
    base["obs"] = dev(host["obs"])
    sets = []
    for _ in range(reps + 1):  # every solve on its own copy of the batch
        b2 = dict(base)
        b2.update({k: base[k].clone() for k in ("X", "U", "lamb", "cost", "iters", "status")})
        sets.append(b2)
    ts = []
    for r in range(reps + 1):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        solver.solve(sets[r])
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts[1:])), sets[-1]["iters"].double()


def schedule_sweep(batches):
    doc = {}
    for B in batches:
        for variant in (None, "all_obstacle", "far_targets"):
            cfg = workloads.config_for("config2", "f64")
            cfg.layout = 2
            host = workloads.make_batch(cfg, B, variant=variant)
            row = {}
            for first in (-1, 6, 8, 10, 12, 14, 16, 20):
                solver = BatchedILQR(cfg)
                solver.set_option("first_chunk", first)
                ms, it = time_solve(solver, host)
                row["auto" if first < 0 else str(first)] = ms
                solver.close()
            hand = min(v for k, v in row.items() if k != "auto")
            best = min((v, k) for k, v in row.items() if k != "auto")[1]
            surv = {str(c): float((it > c).double().mean()) for c in (8, 10, 12, 14)}
            rec = dict(ms=row, best_hand_tuned=best, auto_over_best=row["auto"] / hand,
                       iterations_mean=float(it.mean()), iterations_max=int(it.max()),
                       survivors_after=surv)
            doc[f"B{B}:{variant or 'bench'}"] = rec
            print(f"B={B:7d} {variant or 'bench':13s} auto {row['auto']:7.3f} ms  best hand-tuned "
                  f"first_chunk={best} {hand:7.3f} ms  auto/best {row['auto'] / hand:5.3f}  "
                  f"survivors after 8/10/12/14: " + "/".join(f"{surv[str(c)]:.3f}" for c in (8, 10, 12, 14)),
                  flush=True)
    Path("gpurun_out").mkdir(exist_ok=True)
    Path("gpurun_out/solve_schedule.json").write_text(json.dumps(doc, indent=1))


def plan_sweep(batches):
    """The automatic schedule and structural variants of it (length of the chunk behind the first,
    the tail's cap) against hand-tuned first chunks, as ratios to the best hand-tuned one."""
    plans = {}
    plans["auto"] = {}
    for step in (2, 4):
        for tail in (8192, 12288, 16384):
            plans[f"s{step}t{tail // 1024}"] = dict(chunk_step=step, wave_tail=tail)
    for fc in (8, 10, 12, 14):
        plans[f"hand{fc}"] = dict(first_chunk=fc)
    for B in batches:
        for variant in (None, "all_obstacle", "far_targets"):
            cfg = workloads.config_for("config2", "f64")
            cfg.layout = 2
            host = workloads.make_batch(cfg, B, variant=variant)
            row = {}
            for name, opts in plans.items():
                solver = BatchedILQR(cfg)
                for k, v in opts.items():
                    solver.set_option(k, v)
                row[name], _ = time_solve(solver, host)
                solver.close()
            hand = min(v for k, v in row.items() if k.startswith("hand"))
            print(f"B={B:7d} {variant or 'bench':13s} hand {hand:6.3f} | " +
                  "  ".join(f"{k} {v / hand:5.3f}" for k, v in row.items() if not k.startswith("hand")),
                  flush=True)


def ab_sweep(option, batches):
    key, val = option.split("=")
    for B in batches:
        for variant in (None, "all_obstacle", "far_targets"):
            cfg = workloads.config_for("config2", "f64")
            cfg.layout = 2
            host = workloads.make_batch(cfg, B, variant=variant)
            res, its = {"auto": [], option: []}, {}
            for _ in range(2):
                for name in res:
                    solver = BatchedILQR(cfg)
                    if name != "auto":
                        solver.set_option(key, int(val))
                    ms, it = time_solve(solver, host, reps=7)
                    res[name].append(ms)
                    its[name] = it
                    solver.close()
            print(f"B={B:7d} {variant or 'bench':13s} auto {min(res['auto']):6.3f} ms   {option} "
                  f"{min(res[option]):6.3f} ms   iteration counts equal: "
                  f"{bool(torch.equal(its['auto'], its[option]))}", flush=True)


if len(sys.argv) > 2 and sys.argv[1] == "--ab":
    ab_sweep(sys.argv[2], [int(b) for b in sys.argv[3:]] or [49152, 65536, 131072])
    sys.exit(0)

if len(sys.argv) > 1 and sys.argv[1] == "--plans":
    plan_sweep([int(b) for b in sys.argv[2:]] or [16384, 65536, 262144])
    sys.exit(0)

if len(sys.argv) > 1 and sys.argv[1] == "--schedule":
    schedule_sweep([int(b) for b in sys.argv[2:]] or [16384, 65536, 262144])
    sys.exit(0)

for spec in sys.argv[1:]:
    parts = spec.split(":")
    layout, dtype, B = parts[:3]
    B = int(B)
    cfg = workloads.config_for("config2", dtype)
    cfg.layout = {"wave": 0, "lane": 1, "tiled": 2}[layout]
    solver = BatchedILQR(cfg)
    if len(parts) > 3:
        solver.set_compaction(int(parts[3]))
    if len(parts) > 4:
        solver.set_option("wave_tail", int(parts[4]))
    ms, it = time_solve(solver, workloads.make_batch(cfg, B), reps=3)
    print(f"{spec:32s} {ms:9.3f} ms  iterations mean {float(it.mean()):5.2f} max {int(it.max()):3d}  "
          f"-> {float(it.sum()) / ms / 1e3:8.1f} M executed it/s, {B / ms / 1e3:7.2f} M problems/s")
    solver.close()

#!/usr/bin/env python3
"""Run ONE kernel configuration a few times — the target of `rocprofv3 --pmc ...` runs
(counters are collected in their own passes, never together with the traces)."""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from ilqr_iterative_tasks_amd import BatchedILQR, workloads

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="config2")
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--dtype", default="f64")
ap.add_argument("--layout", default="wave")
ap.add_argument("--iters", type=int, default=10)
ap.add_argument("--launches", type=int, default=5)
ap.add_argument("--options", default="", help="i2lqr_set_option settings: 'name=value name=value'")
ap.add_argument("--solve", action="store_true", help="i2lqr_solve (to termination) instead of "
                "a fixed iteration count")
ap.add_argument("--sync-each", action="store_true", help="synchronise and pause 2 ms after every "
                "launch (tools/solve_timeline.py separates the solves of a kernel trace by the gaps)")
args = ap.parse_args()
cfg = workloads.config_for(args.workload, args.dtype)
cfg.layout = {"wave": 0, "lane": 1, "tiled": 2}[args.layout]
solver = BatchedILQR(cfg)
for kv in args.options.split():
    key, val = kv.split("=")
    solver.set_option(key, int(val))
host = workloads.make_batch(cfg, args.batch)
dev = lambda a: solver.to_native(torch.as_tensor(a).to(solver.device, solver.dtype))
bufs = []
for _ in range(args.launches):
    # a solve returns what the reference's ilqr() returns (U, X, lamb) + cost / iters / status;
    # the fixed-count launches also write the gains of their last backward pass
    buf = solver.alloc(args.batch, want_gains=not args.solve)
    for key in ("X", "U", "x_term", "lamb"):
        buf[key].copy_(dev(host[key]))
    buf["obs"] = dev(host["obs"])
    bufs.append(buf)
torch.cuda.synchronize()
for buf in bufs:
    if args.solve:
        solver.solve(buf)
    else:
        solver.iterate(buf, args.iters)
    if args.sync_each:
        import time
        torch.cuda.synchronize()
        time.sleep(0.002)
torch.cuda.synchronize()
print("done", args)

#!/usr/bin/env python3
"""Latency of small solves to termination (bicycle4, N = 6: the reference's shape) against the
iteration count: what a solve LAUNCH costs and what an iteration costs (the chained controller
runs 24 such solves per control step).  python tools/small_solve_probe.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tests"))
import numpy as np, torch
from helpers import dev_batch
from ilqr_iterative_tasks_amd import BatchedILQR, default_config, workloads
cfg = default_config("bicycle4", 6)
for B in (1, 2, 4, 16):
    for max_iter in (1, 2, 4, 8, 150):
        c = cfg.copy(); c.max_iter = max_iter
        solver = BatchedILQR(c)
        host = workloads.make_batch(c, B)
        bufs = [dev_batch(solver, host, want_gains=False) for _ in range(12)]
        ts = []
        for b in bufs:
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); solver.solve(b); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        it = bufs[-1]["iters"].cpu().numpy()
        print(f"B={B:3d} max_iter={max_iter:3d}  solve {np.median(ts[2:])*1e3:7.1f} us  iterations {it.tolist() if B <= 4 else (int(it.min()), int(it.max()))}  kernel {solver.solve_kernel(B)}")
        solver.close()

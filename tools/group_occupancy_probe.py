#!/usr/bin/env python3
"""How does k_group_iterate scale with wavefronts per CU?  A short horizon (N = 8) shrinks the LDS
slice to ~4 KB per problem, so four wavefronts fit a CU; the time per launch from 2048 to 16384
problems then shows what a lighter LDS footprint at N = 20 could buy."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

from ilqr_iterative_tasks_amd import BatchedILQR, default_config, workloads

# argv: list of "N:ws" pairs (ws = 0 LDS form, 1 workspace form, -1 automatic); default both forms
CASES = [tuple(int(v) for v in a.split(":")) for a in sys.argv[1:]] or [(8, 0), (8, 1), (20, 0), (20, 1)]
for N, ws in CASES:
    for B in (1024, 2048, 4096, 8192, 16384):
        cfg = default_config("bicycle6", N, "f64", dt=0.25)
        solver = BatchedILQR(cfg)
        solver.set_option("group_lanes", 8)
        solver.set_option("group_workspace", ws)
        host = workloads.make_batch(cfg, B)
        dev = lambda a: solver.to_native(torch.as_tensor(a).to(solver.device, solver.dtype))
        bufs = []
        for _ in range(8):
            buf = solver.alloc(B)
            for key in ("X", "U", "x_term", "lamb"):
                buf[key].copy_(dev(host[key]))
            buf["obs"] = dev(host["obs"])
            bufs.append(buf)
        ts = []
        for i, buf in enumerate(bufs):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            solver.iterate(buf, 10)
            e1.record()
            torch.cuda.synchronize()
            if i >= 2:
                ts.append(e0.elapsed_time(e1))
        t = float(np.median(ts))
        print(f"N={N:2d} ws={ws:2d} B={B:6d} {solver.iterate_kernel(B):16s} {t:7.3f} ms  {B * 10 / t / 1e3:7.1f} M it/s")

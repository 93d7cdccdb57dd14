#!/usr/bin/env python3
"""A/B of the two problem-major kernel families ("group_lanes" 64 vs 8) over batch sizes:
kernel time of 10 fused iterations (HIP events, interleaved launches, median)."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from ilqr_iterative_tasks_amd import BatchedILQR, workloads

dtype = sys.argv[1] if len(sys.argv) > 1 else "f64"
cfg = workloads.config_for("config2", dtype)
solver = BatchedILQR(cfg)
for B in (64, 256, 1024, 2048, 4096, 8192, 16384, 32768):
    host = workloads.make_batch(cfg, B)
    dev = lambda a: solver.to_native(torch.as_tensor(a).to(solver.device, solver.dtype))
    X0, U0, l0 = dev(host["X"]), dev(host["U"]), dev(host["lamb"])
    base = solver.alloc(B)
    base["x_term"].copy_(dev(host["x_term"]))
    base["obs"] = dev(host["obs"])
    times = {64: [], 8: [], "spec": []}
    for rep in range(12):
        for lanes in (64, 8, "spec"):
            solver.set_option("group_lanes", 8 if lanes == "spec" else lanes)
            solver.set_option("speculate", 1 if lanes == "spec" else 0)
            if lanes == "spec" and B > 4096:
                continue
            base["X"].copy_(X0); base["U"].copy_(U0); base["lamb"].copy_(l0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            solver.iterate(base, 10)
            e1.record()
            torch.cuda.synchronize()
            if rep >= 2:
                times[lanes].append(e0.elapsed_time(e1))
    t64, t8 = np.median(times[64]), np.median(times[8])
    ts = np.median(times["spec"]) if times["spec"] else float("nan")
    print(f"B={B:6d} {dtype}: wave {t64:8.4f} ms ({B * 10 / t64 / 1e3:8.1f} M it/s)   "
          f"group {t8:8.4f} ms ({B * 10 / t8 / 1e3:8.1f} M it/s)   spec {ts:8.4f} ms "
          f"({B * 10 / ts / 1e3:8.1f} M it/s)")

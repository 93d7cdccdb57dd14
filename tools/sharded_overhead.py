#!/usr/bin/env python3
"""bench.py's measure_sharded_overhead at other per-rank batch sizes (the sharded step beside the
unsharded one, one process, one GPU, world of one over the library's RCCL communicator):
    python tools/sharded_overhead.py 4096 16384 65536 [--json out.json] [--opt name=value ...]"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch

import bench
from ilqr_iterative_tasks_amd import dist as dist_mod, workloads

argv = [a for a in sys.argv[1:] if not a.startswith("--")]
opts = dict(a.split("=") for a in argv if "=" in a)
args = bench.parse_args([])
out = {}
for B in [int(a) for a in argv if a.isdigit()] or [4096, 16384, 65536]:
    cfg = workloads.config_for("config2", "f64")
    steps = 50 if B <= 16384 else 12
    r = bench.measure_sharded_overhead(args, cfg, B, torch, dist_mod, steps=steps, reps=4, options=opts)
    out[f"B{B}"] = r
    f = r["forms"]
    print(f"B={B:7d} unsharded {f['unsharded']['step_ms']:.4f} ms  one-call {f['sharded_one_call']['step_ms']:.4f} "
          f"(x{r['sharded_over_unsharded']:.3f}, host {f['sharded_one_call']['host_enqueue_ms'] * 1e3:.0f} us)  "
          f"host-driven {f['sharded_host_driven']['step_ms']:.4f} (x{r['host_driven_over_unsharded']:.3f})", flush=True)
if "--json" in sys.argv:
    Path(sys.argv[sys.argv.index("--json") + 1]).write_text(json.dumps(out, indent=1))

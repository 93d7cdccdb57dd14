#!/usr/bin/env python3
"""Phase cycle shares of the lane kernel from the diagnostic (stamped) build:
python tools/stamp_run_lane.py [f64|f32] [batch] [option=value ...]"""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch

from ilqr_iterative_tasks_amd import _abi, workloads

_abi.LIB_PATH = ROOT / "tools" / "_diag" / "libi2lqr_stamps.so"
from ilqr_iterative_tasks_amd import BatchedILQR

dtype = sys.argv[1] if len(sys.argv) > 1 else "f64"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
WL = os.environ.get("WORKLOAD", "config2")  # config5: quad12 (k_lane_iterate_rows)
iters = 4 if WL == "config5" else 10
cfg = workloads.config_for(WL, dtype)
cfg.layout = 2
solver = BatchedILQR(cfg)
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    solver.set_option(k, int(v))
host = workloads.make_batch(cfg, B)
dev = lambda a: solver.to_native(torch.as_tensor(a).to(solver.device, solver.dtype))
buf = solver.alloc(B)
for key in ("X", "U", "x_term", "lamb"):
    buf[key].copy_(dev(host[key]))
buf["obs"] = dev(host["obs"])
dbg = torch.zeros(B // 64, 8, dtype=torch.int64, device=solver.device)
os.environ["I2LQR_DBG_PTR"] = hex(dbg.data_ptr())
solver.iterate(buf, iters)
torch.cuda.synchronize()
d = dbg.double().mean(0).cpu().numpy() / iters
names = ["bwd trig+jac+barriers+loads", "bwd Riccati products", "bwd inverse+gains+store",
         "bwd value update", "forward", "accept/reject (+re-roll)", "-", "-"]
if WL == "config5":
    names = ["bwd trig+jac+barriers", "bwd G, Quu, inverse, Kc, W", "bwd K = Kc A, loads, gain stores",
             "bwd state blocks", "forward", "re-roll", "bwd loop top (wait for loads)", "-"]
tot = d.sum()
for nm, v in zip(names, d):
    print(f"{nm:30s} {v:9.0f} ticks/iteration  {100 * v / tot:5.1f} %  ({v / cfg.N:7.0f} per step)")
print(f"{'sum':30s} {tot:9.0f} ticks/iteration")

#!/usr/bin/env python3
"""Phase cycle shares of the wave kernel from the diagnostic (stamped) build."""
import ctypes as C
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch

from ilqr_iterative_tasks_amd import _abi, workloads

_abi.LIB_PATH = ROOT / "tools" / "_diag" / "libi2lqr_stamps.so"
from ilqr_iterative_tasks_amd import BatchedILQR

B = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
wl = sys.argv[2] if len(sys.argv) > 2 else "config2"
iters = 10 if wl == "config2" else 4
cfg = workloads.config_for(wl, sys.argv[1] if len(sys.argv) > 1 else "f64")
solver = BatchedILQR(cfg)
lanes = int(sys.argv[3]) if len(sys.argv) > 3 else 64   # "group_lanes": 64 or 8
solver.set_option("group_lanes", lanes)
host = workloads.make_batch(cfg, B)
buf = solver.alloc(B)
for key in ("X", "U", "x_term", "lamb"):
    buf[key].copy_(torch.as_tensor(host[key]).to(solver.device, solver.dtype))
buf["obs"] = torch.as_tensor(host["obs"]).to(solver.device, solver.dtype)
dbg = torch.zeros(B, 8, dtype=torch.int64, device=solver.device)
if lanes == 16 and cfg.system_id == 2:  # quad12's sixteen-lane kernel keeps its real workspace: debug buffer through the env
    import os
    os.environ["I2LQR_DBG_PTR"] = str(dbg.data_ptr())
    solver.ensure_workspace(B)
else:
    solver.lib.i2lqr_set_workspace(solver._handle, C.c_void_p(dbg.data_ptr()), dbg.numel() * 8)
    solver.ensure_workspace = lambda B: None
solver.iterate(buf, iters)
torch.cuda.synchronize()
d = dbg.double().mean(0).cpu().numpy() / iters
names = (["prep", "bwd P1", "bwd P2", "bwd quu_inv", "bwd gains+value", "bwd refreshF+sync", "forward", "-"]
         if lanes == 64 else ["record phase", "bwd P1 + T1 exchange", "forward", "accept + adopt", "bwd P2 (H column)",
          "bwd record fetch, Quu, inverse, gain column", "bwd gain exchange + value update", "rollout at entry + stores at exit (per launch / iters)"]
         if lanes == 16 and cfg.system_id == 2 else
         ["records (sum / iters)", "backward (sum / iters)", "ENTRY: loads + rollout (per launch / iters)",
          "LOOP: all iterations (per launch / iters)", "EXIT: stores (per launch / iters)", "-",
          "forward (sum / iters)", "-"] if lanes == 16 else
         ["prep", "bwd P1 + T1 exchange", "bwd P2 (H column)", "bwd Quu + inverse",
          "bwd gains + exchange", "bwd value update", "forward", "-"])
tot = d.sum()
for nm, v in zip(names, d):
    print(f"{nm:20s} {v:9.0f} cycles/iteration  {100 * v / tot:5.1f} %")
print(f"{'sum':20s} {tot:9.0f} cycles/iteration")

#!/usr/bin/env python3
"""Time the individual kernels of the path (rollout / backward / forward / fused iterate) with
HIP events — a development aid for finding which phase dominates."""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch

from ilqr_iterative_tasks_amd import BatchedILQR, workloads


def timeit(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="config2")
    ap.add_argument("--batches", default="1024,65536")
    ap.add_argument("--dtypes", default="f64,f32")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--layouts", default="wave,lane")
    args = ap.parse_args()
    for layout, dtype in [(l, d) for l in args.layouts.split(",") for d in args.dtypes.split(",")]:
        cfg = workloads.config_for(args.workload, dtype)
        cfg.layout = {"wave": 0, "lane": 1, "tiled": 2}[layout]
        solver = BatchedILQR(cfg)
        for B in [int(b) for b in args.batches.split(",")]:
            host = workloads.make_batch(cfg, B)
            buf = solver.alloc(B)
            for key in ("X", "U", "x_term", "lamb"):
                buf[key].copy_(solver.to_native(torch.as_tensor(host[key]).to(solver.device,
                                                                              solver.dtype)))
            buf["obs"] = solver.to_native(torch.as_tensor(host["obs"]).to(solver.device,
                                                                          solver.dtype))
            X0, U0, l0 = buf["X"].clone(), buf["U"].clone(), buf["lamb"].clone()
            solver.iterate(buf, 3)  # a non-trivial nominal
            Xn, Un, cn = torch.empty_like(buf["X"]), torch.empty_like(buf["U"]), torch.empty_like(buf["cost"])
            t_roll = timeit(lambda: solver.rollout(buf["X"], buf["U"], buf["x_term"], buf["cost"]))
            t_bwd = timeit(lambda: solver.backward(buf["X"], buf["U"], buf["x_term"], buf["lamb"],
                                                   buf["obs"], buf["K"], buf["k"]))
            t_fwd = timeit(lambda: solver.forward(buf["X"], buf["U"], buf["x_term"], buf["K"],
                                                  buf["k"], Xn, Un, cn))

            def it():
                buf["X"].copy_(X0); buf["U"].copy_(U0); buf["lamb"].copy_(l0)
                solver.iterate(buf, args.iters)
            def cp():
                buf["X"].copy_(X0); buf["U"].copy_(U0); buf["lamb"].copy_(l0)
            t_it = timeit(it) - timeit(cp)
            print(f"{args.workload} {layout} {dtype} B={B}: rollout {t_roll:9.1f} us  backward {t_bwd:9.1f} us  "
                  f"forward {t_fwd:9.1f} us  iterate/{args.iters} {t_it / args.iters:9.1f} us  "
                  f"-> {B * args.iters / t_it:8.2f} M it/s")
        solver.close()


if __name__ == "__main__":
    main()

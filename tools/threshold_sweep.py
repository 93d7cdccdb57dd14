#!/usr/bin/env python3
"""Where the problem-major (latency) kernels hand over to the one-problem-per-lane (throughput)
kernels, measured at MORE THAN ONE SHAPE: the thresholds behind
i2lqr_recommended_layout / select_fused were measured at bicycle6 N=20 fp64 only.

For every shape x batch: the fused iterate (10 iterations, no early exit) and the solve to
termination on (a) the problem-major layout with the library's automatic kernel choice, (b) the
problem-major layout pinned to the eight-lane workspace form, (c) the batch-tiled lane layout.
Interleaved rounds, every timed launch on its own copy of the batch, median per variant.
Writes one JSON document (default gpurun_out/threshold_sweep.json) and a table on stdout.
"""
import argparse
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

from ilqr_iterative_tasks_amd import BatchedILQR, default_config, workloads
from ilqr_iterative_tasks_amd import _abi

SHAPES = {
    # name: (system, N, dtype, dt)
    "bicycle6_N20_f64": ("bicycle6", 20, "f64", 0.25),   # the shape the thresholds were set on
    "bicycle4_N6_f64": ("bicycle4", 6, "f64", 1.0),      # the reference's own shape
    "bicycle6_N50_f64": ("bicycle6", 50, "f64", 0.25),
    "bicycle6_N20_f32": ("bicycle6", 20, "f32", 0.25),
    "bicycle4_N6_f32": ("bicycle4", 6, "f32", 1.0),
    "bicycle4_N20_f64": ("bicycle4", 20, "f64", 0.5),
    "bicycle6_N6_f64": ("bicycle6", 6, "f64", 0.25),
    "bicycle6_N6_f32": ("bicycle6", 6, "f32", 0.25),
    "bicycle4_N20_f32": ("bicycle4", 20, "f32", 0.5),
    "bicycle4_N50_f64": ("bicycle4", 50, "f64", 0.25),
    "bicycle4_N50_f32": ("bicycle4", 50, "f32", 0.25),
    "bicycle6_N50_f32": ("bicycle6", 50, "f32", 0.25),
    "quad12_N50_f64": ("quad12", 50, "f64", 0.02),
    "quad12_N20_f64": ("quad12", 20, "f64", 0.02),
    "quad12_N50_f32": ("quad12", 50, "f32", 0.02),
    "quad12_N20_f32": ("quad12", 20, "f32", 0.02),
}

ap = argparse.ArgumentParser()
ap.add_argument("--shapes", default=",".join(SHAPES))
ap.add_argument("--batches", default="2048,4096,6144,8192,10240,12288,14336,16384,20480,24576,32768")
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--iters", type=int, default=10)  # (quad12: 4, as in bench.py)
ap.add_argument("--solve", type=int, default=1)
ap.add_argument("--out", default="gpurun_out/threshold_sweep.json")
args = ap.parse_args()

LAY = {"pm": _abi.LAYOUT_PROBLEM_MAJOR, "pm_g8ws": _abi.LAYOUT_PROBLEM_MAJOR,
       "tiled": _abi.LAYOUT_BATCH_TILED}
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
doc = {"iters": args.iters, "rounds": args.rounds, "device": torch.cuda.get_device_name(0),
       "lib_sha256": None, "shapes": {}}
try:
    import hashlib
    doc["lib_sha256"] = hashlib.sha256(Path(_abi.LIB_PATH).read_bytes()).hexdigest()
except Exception:
    pass


def variants_for(B):
    out = ["pm", "tiled"]
    if 4096 < B <= 8192:
        out.insert(1, "pm_g8ws")
    return out


for shape in args.shapes.split(","):
    system, N, dtype, dt = SHAPES[shape]
    if system == "quad12":
        args.iters = 4
    rows = {}
    for B in [int(b) for b in args.batches.split(",")]:
        runs = []
        for var in variants_for(B):
            cfg = default_config(system, N, dtype, dt=dt)
            cfg.layout = LAY[var]
            solver = BatchedILQR(cfg)
            if var == "pm_g8ws":
                try:
                    solver.ensure_workspace(B)
                    solver.set_option("group_lanes", 8)
                    solver.set_option("group_workspace", 1)
                except Exception as exc:  # shape the eight-lane form is not built for
                    print(f"{shape} B={B} {var}: skipped ({exc})")
                    continue
            host = workloads.make_batch(cfg, B)
            dev = lambda a: solver.to_native(torch.as_tensor(a).to(solver.device, solver.dtype))
            base = solver.alloc(B, want_gains=False)
            for key in ("X", "U", "x_term", "lamb"):
                base[key].copy_(dev(host[key]))
            base["obs"] = dev(host["obs"])
            sets = []
            for _ in range(2 * (args.rounds + 1)):
                b2 = dict(base)
                b2.update({k: base[k].clone() for k in ("X", "U", "lamb", "cost", "iters", "status")})
                sets.append(b2)
            try:
                kern = solver.iterate_kernel(B)
                skern = solver.solve_kernel(B)
            except Exception:
                kern = skern = "?"
            runs.append(dict(var=var, solver=solver, sets=sets, it=[], so=[], kernel=kern,
                             solve_kernel=skern))
        rng = np.random.default_rng(0)
        for r in range(args.rounds + 1):
            for i in rng.permutation(len(runs)):
                run = runs[i]
                for mode in ("it", "so") if args.solve else ("it",):
                    buf = run["sets"][2 * r + (mode == "so")]
                    torch.cuda.synchronize()
                    e0.record()
                    try:
                        if mode == "it":
                            run["solver"].iterate(buf, args.iters)
                        else:
                            run["solver"].solve(buf)
                    except Exception as exc:
                        run[mode] = None
                        run.setdefault("error", str(exc))
                        continue
                    e1.record()
                    torch.cuda.synchronize()
                    if r > 0 and run[mode] is not None:
                        run[mode].append(e0.elapsed_time(e1))
        row = {}
        for run in runs:
            it = float(np.median(run["it"])) if run["it"] else None
            so = float(np.median(run["so"])) if run["so"] else None
            row[run["var"]] = dict(iterate_ms=it, solve_ms=so, kernel=run["kernel"],
                                   solve_kernel=run["solve_kernel"], error=run.get("error"))
            print(f"{shape:18s} B={B:6d} {run['var']:8s} iterate {it if it else float('nan'):8.4f} ms "
                  f"({B * args.iters / it / 1e3 if it else 0:7.1f} M it/s)  solve "
                  f"{so if so else float('nan'):8.4f} ms   {run['kernel'][:50]}", flush=True)
            run["solver"].close()
        rows[str(B)] = row
        del runs
        torch.cuda.empty_cache()
    doc["shapes"][shape] = rows
    Path(args.out).parent.mkdir(parents=True, exist_ok=True)
    Path(args.out).write_text(json.dumps(doc, indent=1))
print("wrote", args.out)

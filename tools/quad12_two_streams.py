#!/usr/bin/env python3
"""quad12 (BASELINE configs[4]: n=12, m=4, N=50, 65536 problems, fp64) overlap experiment, VERDICT r5
#5 (ii): the batch as TWO half-launches on two streams, the second offset by a fraction of an
iteration, against the ONE launch with the s_sleep stagger (the product).  The question: does a real
dependency structure (two kernels, one streaming its gains while the other computes) overlap better
than a sleep constant?  Same process, interleaved, every launch on its own copy of the batch.

    python tools/quad12_two_streams.py --json profiles/r06_ab_quad12_two_streams.json"""
import argparse
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

from ilqr_iterative_tasks_amd import BatchedILQR, workloads

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=65536)
ap.add_argument("--iters", type=int, default=4)
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--json", default=None)
args = ap.parse_args()
B, H = args.batch, args.batch // 2
cfg = workloads.config_for("config5", "f64")
cfg.layout = 2
host = workloads.make_batch(cfg, B)


def fill(solver, sl):
    dev = lambda a: solver.to_native(torch.as_tensor(a[sl]).to(solver.device, solver.dtype))
    buf = solver.alloc(sl.stop - sl.start)
    for key in ("X", "U", "x_term", "lamb"):
        buf[key].copy_(dev(host[key]))
    buf["obs"] = dev(host["obs"])
    return buf


one = BatchedILQR(cfg)
halves = [BatchedILQR(cfg), BatchedILQR(cfg)]
for h in halves:
    h.set_option("stagger", 0)  # the offset between the two launches replaces the sleep
R = args.rounds + 1
bufs_one = [fill(one, slice(0, B)) for _ in range(R)]
bufs_half = [[fill(halves[q], slice(q * H, (q + 1) * H)) for q in range(2)] for _ in range(R)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
spin = torch.zeros(1, device="cuda")
# ~2.1 cycles of the spin loop per clock tick: calibrate the delay kernel in milliseconds
torch.cuda.synchronize()
t0 = time.perf_counter(); torch.cuda._sleep(20_000_000); torch.cuda.synchronize()
ticks_per_ms = 20_000_000 / ((time.perf_counter() - t0) * 1e3)

variants = {"one_launch_stagger_auto": None}
for off in (0.0, 0.15, 0.3, 0.45, 0.6):
    variants[f"two_streams_offset_{off:g}ms"] = off
times = {k: [] for k in variants}
rng = np.random.default_rng(0)
for r in range(R):
    for name in [list(variants)[i] for i in rng.permutation(len(variants))]:
        off = variants[name]
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if off is None:
            e0.record()
            one.iterate(bufs_one[r], args.iters)
            e1.record()
        else:
            main = torch.cuda.current_stream()
            e0.record()
            for q, st in enumerate(streams):
                st.wait_stream(main)
                with torch.cuda.stream(st):
                    if q == 1 and off > 0:
                        torch.cuda._sleep(int(off * ticks_per_ms))
                    halves[q].iterate(bufs_half[r][q], args.iters)
            for st in streams:
                main.wait_stream(st)
            e1.record()
        torch.cuda.synchronize()
        if r > 0:
            times[name].append(e0.elapsed_time(e1))
        # (the half launches of one round restart from fresh copies in the next: R copies)
doc = {"tool": "tools/quad12_two_streams.py", "batch": B, "iterations_per_launch": args.iters,
       "rounds": args.rounds, "device": torch.cuda.get_device_name(0), "variants": []}
base = float(np.median(times["one_launch_stagger_auto"]))
for name, t in times.items():
    med = float(np.median(t))
    doc["variants"].append({"variant": name, "ms_per_round": [float(x) for x in t], "ms_median": med,
                            "Mits_median": B * args.iters / med / 1e3,
                            "speedup_over_one_launch": base / med})
    print(f"{name:36s} median {med:8.3f} ms  {B * args.iters / med / 1e3:7.1f} M it/s  x{base / med:5.3f}")
if args.json:
    Path(args.json).parent.mkdir(parents=True, exist_ok=True)
    Path(args.json).write_text(json.dumps(doc, indent=1))

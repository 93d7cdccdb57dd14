#!/usr/bin/env python3
"""Experiment (candidate b): a batch of 8 k - 32 k problems split over BOTH kernel
families at once — the one-problem-per-lane kernel (one wavefront per 64 problems: at 16384
problems one SIMD per CU, launch floor ~0.53 ms whatever the count) on one stream and the
sixteen-lane kernel (four problems per wavefront, rounds of 4096) on another — against each family
alone.  bicycle6, N = 20, fp64, 10 fused iterations; every timed launch on its own copy of the
batch; median of the rounds.  python tools/hetero_bench.py [B ...]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

from ilqr_iterative_tasks_amd import BatchedILQR, workloads

ITERS, ROUNDS = 10, 8


def prepare(layout, B, offset):
    cfg = workloads.config_for("config2", "f64")
    cfg.layout = layout
    s = BatchedILQR(cfg)
    host = workloads.make_batch(cfg, B, offset=offset)
    dev = lambda a: s.to_native(torch.as_tensor(a).to(s.device, s.dtype))
    base = s.alloc(B, want_gains=False)
    for k in ("X", "U", "x_term", "lamb"):
        base[k].copy_(dev(host[k]))
    base["obs"] = dev(host["obs"])
    sets = []
    for _ in range(ROUNDS + 1):
        b = dict(base)
        b.update({k: base[k].clone() for k in ("X", "U", "lamb", "cost", "iters", "status")})
        sets.append(b)
    s.ensure_workspace(B)
    return s, sets


def timed(parts):
    """parts: list of (solver, sets, stream); all launched back to back on their streams."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for r in range(ROUNDS + 1):
        torch.cuda.synchronize()
        e0.record()
        evs = []
        for s, sets, st in parts:
            st.wait_event(e0)
            with torch.cuda.stream(st):
                s.iterate(sets[r], ITERS)
                ev = torch.cuda.Event()
                ev.record(st)
                evs.append(ev)
        for ev in evs:
            torch.cuda.current_stream().wait_event(ev)
        e1.record()
        torch.cuda.synchronize()
        if r:
            ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for B in [int(b) for b in sys.argv[1:]] or [12288, 16384, 20480, 24576, 28672, 32768]:
    lane, lsets = prepare(2, B, 0)
    t_lane = timed([(lane, lsets, s1)])
    pm, psets = prepare(0, B, 0)
    t_pm = timed([(pm, psets, s1)])
    best = (None, 1e9)
    rows = []
    for x in range(4096, B, 4096):  # x problems on the sixteen-lane kernel, the rest per lane
        a, asets = prepare(0, x, 0)
        b, bsets = prepare(2, B - x, x)
        t = timed([(b, bsets, s1), (a, asets, s2)])
        rows.append(f"{x}+{B - x}: {t:.3f}")
        if t < best[1]:
            best = (x, t)
        a.close(), b.close()
    print(f"B={B:6d}  lane alone {t_lane:.3f} ms ({B * ITERS / t_lane / 1e3:6.1f} M it/s)  sixteen-lane "
          f"alone {t_pm:.3f} ms ({B * ITERS / t_pm / 1e3:6.1f})  split (sixteen-lane + lane): "
          + "  ".join(rows) + f"  -> best {best[1]:.3f} ms ({B * ITERS / best[1] / 1e3:6.1f} M it/s)", flush=True)
    lane.close(), pm.close()

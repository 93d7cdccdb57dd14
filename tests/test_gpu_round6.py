"""GPU suite, round 6: the sharded control round as ONE C-ABI call (i2lqr_sharded_round_flat,
i2lqr_round_pick) against the five host-driven calls it replaces — VERDICT r5 #1 / ADVICE r5."""
import numpy as np
import pytest

from helpers import dev_batch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available(), "gpu-marked test without a HIP device"
    return torch


def _round_inputs(torch, cfg, B, seed=0):
    from ilqr_iterative_tasks_amd import workloads
    host = workloads.make_batch(cfg, B)
    x0 = torch.as_tensor(host["X"][0, :, 0]).cuda()
    x_terms = torch.as_tensor(host["x_term"]).cuda()
    qfun = torch.as_tensor(np.random.default_rng(B + seed).integers(0, 100, B).astype(np.int32)).cuda()
    return x0, x_terms, qfun


@pytest.mark.parametrize("B,n_iters", [(1024, 10), (2048, 3), (16448, 10), (200, None), (4160, None)])
@pytest.mark.parametrize("exchange", ["native", "copies"])
def test_one_call_round_is_bit_identical_to_the_host_driven_round(torch_mod, B, n_iters, exchange):
    """HipCandidateSolver.sharded_round in a world of one: the ONE-call form
    (i2lqr_sharded_round_flat: solve on the launch stream, pack + gather + pick + hand-off enqueued
    from C on the exchange stream) returns exactly what the host-driven form returns (the same
    steps as five Python-driven calls) — pick, owner, winner's trajectory, gathered costs — with
    and without an exchange stream, over the library's RCCL communicator (self-gathers) and over
    device copies (no communicator), for fused rounds (sixteen-lane kernel), lane-layout rounds
    and solves to termination."""
    torch = torch_mod
    from ilqr_iterative_tasks_amd import BatchedILQR, default_config, dist as idist
    from ilqr_iterative_tasks_amd.control.iterative_ilqr import HipCandidateSolver
    cfg = default_config("bicycle6", 20, "f64", dt=0.25)
    x0, x_terms, qfun = _round_inputs(torch, cfg, B)
    obs = (31, -3, 8, 6, 0, 0)
    xch = idist.CostExchange(BatchedILQR(default_config("bicycle4", 6), "cuda:0")) \
        if exchange == "native" else idist.TorchExchange()
    side = torch.cuda.Stream()
    ref = HipCandidateSolver().sharded_round(cfg, x0, x_terms, qfun, 1.0, xch, B, obs_rec=obs,
                                             n_iters=n_iters, host_driven=True)
    torch.cuda.synchronize()
    want = {k: ref[k].clone() for k in ("best_idx", "U", "X", "cost_all", "best_cost")}
    for stream in (None, side):
        hs = HipCandidateSolver()
        for rep in range(3):  # cached buffers, reused round after round (guard_previous)
            got = hs.sharded_round(cfg, x0, x_terms, qfun, 1.0, xch, B, obs_rec=obs,
                                   n_iters=n_iters, exchange_stream=stream)
        torch.cuda.synchronize()
        for k, w in want.items():
            assert torch.equal(got[k].reshape(w.shape), w), (k, stream is not None)
        assert int(got["best_idx"][1]) == 0
    if exchange == "native":
        xch.close()


def _loopback_world(torch, solver, cfg, host, qfun_all, total, world, n_iters):
    """One process plays the ranks of `world` one after the other on SHARED gather buffers
    (i2lqr_round.loopback): after the last rank's call the buffers hold every rank's contribution
    and that rank's pick is the global one."""
    from ilqr_iterative_tasks_amd import dist as idist
    width = idist.padded_width(total, world)
    P = cfg.m * cfg.N + cfg.n * (cfg.N + 1)
    shared = dict(cost_all=torch.full((world * width,), float("nan"), dtype=solver.dtype, device="cuda"),
                  pack_all=torch.zeros(world, P, dtype=solver.dtype, device="cuda"))
    plans, bufs = [], []
    for rank in range(world):
        lo, hi = idist.shard_range(total, rank, world)
        buf = None
        if hi > lo:
            sl = {k: v[lo:hi] for k, v in host.items()}
            buf = dev_batch(solver, sl, want_gains=False)
        cost_it = solver.empty(hi - lo)
        plan = solver.plan_round(buf, qfun_all[lo:hi].contiguous() if hi > lo else None, cost_it,
                                 total, world, rank, n_iters, bufs=dict(shared), loopback=True,
                                 guard_previous=False)
        solver.round_flat(plan, None, None)
        plans.append(plan)
        bufs.append(buf)
    torch.cuda.synchronize()
    return plans, bufs, width


@pytest.mark.parametrize("world,total", [(2, 13), (3, 4), (8, 8 * 256), (4, 2), (8, 8 * 2304 - 5)])
def test_ranks_played_one_after_the_other_pad_gather_and_translate(torch_mod, world, total):
    """The multi-rank paths of the one-call round on ONE GPU (loopback: every call fills its own
    slots of shared gather buffers): ragged shards are padded with +inf, a rank WITHOUT candidates
    (4 ranks, 2 candidates) contributes +inf and zeros instead of raising, the last rank's pick is
    the first-index arg-min over ALL candidates and its hand-off is the owner's trajectory; more
    than 16384 gathered costs take the two-level pick."""
    torch = torch_mod
    from ilqr_iterative_tasks_amd import BatchedILQR, default_config, workloads, dist as idist
    cfg = default_config("bicycle6", 20, "f64", dt=0.25)
    solver = BatchedILQR(cfg)
    host = workloads.make_batch(cfg, total)
    qfun_all = torch.as_tensor(np.random.default_rng(total).integers(0, 100, total).astype(np.int32)).cuda()
    plans, bufs, width = _loopback_world(torch, solver, cfg, host, qfun_all, total, world, 4)
    # one process over all candidates
    full = dev_batch(solver, host, want_gains=False)
    cost_ref, (idx_ref, val_ref) = solver.iterate_pick(full, 4, qfun_all, 0)
    torch.cuda.synchronize()
    last = plans[-1]
    g, owner = (int(v) for v in last["best_global"].cpu())
    assert g == int(idx_ref) and owner == idist.owner_of(g, total, world)[0]
    assert float(last["best_cost"]) == float(val_ref)
    U, X = solver.unpack(last["winner"])
    assert torch.equal(U, full["U"][g]) and torch.equal(X, full["X"][g])
    ca = last["cost_all"].cpu().numpy().reshape(world, width)
    for r in range(world):
        lo, hi = idist.shard_range(total, r, world)
        np.testing.assert_array_equal(ca[r, :hi - lo], cost_ref[lo:hi].cpu().numpy())
        assert np.all(np.isposinf(ca[r, hi - lo:]))
    solver.close()


def test_round_pick_equals_argmin_plus_round_winner(torch_mod):
    """i2lqr_round_pick (one launch up to 16384 gathered costs, two-level above) against
    i2lqr_argmin + i2lqr_round_winner: ties resolve to the first index, NaN never wins, nothing
    can win -> (-1, owner 0, +inf)."""
    torch = torch_mod
    from ilqr_iterative_tasks_amd import BatchedILQR, default_config, dist as idist
    for dtype in ("f64", "f32"):
        solver = BatchedILQR(default_config("bicycle4", 6, dtype))
        P = solver.m * solver.N + solver.n * (solver.N + 1)
        rng = np.random.default_rng(7)
        for world, total in ((1, 9), (2, 13), (8, 8 * 1024), (8, 8 * 4096 - 3), (3, 4)):
            width = idist.padded_width(total, world)
            c = rng.integers(0, 50, world * width).astype(np.float64)  # many ties
            c[rng.integers(0, world * width, 3)] = np.nan
            cost = torch.as_tensor(c).to(solver.dtype).cuda()
            packs = torch.arange(world * P, device="cuda").to(solver.dtype).view(world, P)
            i, v = solver.argmin(cost)
            win, best = solver.round_winner(world, width, total, i, packs)
            bc, w2, b2 = solver.round_pick(world, width, total, cost, packs)
            torch.cuda.synchronize()
            assert torch.equal(b2, best) and torch.equal(w2, win) and torch.equal(bc, v)
        nothing = torch.full((16,), float("nan"), dtype=solver.dtype, device="cuda")
        bc, _, b2 = solver.round_pick(2, 8, 16, nothing, torch.zeros(2, P, dtype=solver.dtype, device="cuda"))
        assert [int(x) for x in b2.cpu()] == [-1, 0] and float(bc) == float("inf")
        solver.close()


def test_one_call_round_refuses_what_it_cannot_run(torch_mod):
    """Argument errors are codes with a message, raised before any launch: a shard size that is not
    this rank's share, a world above one without a communicator, a ragged split without its padding
    buffer, a struct of another size, no candidates at all."""
    torch = torch_mod
    from ilqr_iterative_tasks_amd import BatchedILQR, default_config, workloads
    from ilqr_iterative_tasks_amd.solver import I2lqrError
    cfg = default_config("bicycle6", 20, "f64", dt=0.25)
    solver = BatchedILQR(cfg)
    B = 256
    buf = dev_batch(solver, workloads.make_batch(cfg, B), want_gains=False)
    qfun = torch.zeros(B, dtype=torch.int32, device="cuda")
    cost_it = solver.empty(B)

    def refused(match, **kw):
        plan = solver.plan_round(buf, qfun, cost_it, kw.pop("total", B), kw.pop("world", 1),
                                 kw.pop("rank", 0), 2, **kw)
        for key, val in kw.pop("patch", {}).items():
            setattr(plan["round"], key, val)
        with pytest.raises(I2lqrError, match=match):
            solver.round_flat(plan)

    refused("communicator", total=2 * B, world=2)
    refused("owns", total=2 * B + 1, world=2, rank=0, loopback=True)
    plan = solver.plan_round(buf, qfun, cost_it, 2 * B - 1, 2, 0, 2, loopback=True)
    plan["round"].cost_padded = None  # (rank 0 of 511 owns 256 = width: not ragged, no padding needed)
    solver.round_flat(plan)
    plan = solver.plan_round(None, None, solver.empty(0), 1, 2, 1, 2, loopback=True)
    plan["round"].cost_padded = None
    with pytest.raises(I2lqrError, match="cost_padded"):
        solver.round_flat(plan)
    plan = solver.plan_round(buf, qfun, cost_it, B, 1, 0, 2)
    plan["round"].struct_size = 8
    with pytest.raises(I2lqrError, match="struct_size"):
        solver.round_flat(plan)
    plan = solver.plan_round(buf, qfun, cost_it, B, 1, 0, 2)
    plan["round"].total = 0
    with pytest.raises(I2lqrError, match="at least one candidate"):
        solver.round_flat(plan)
    torch.cuda.synchronize()
    from ilqr_iterative_tasks_amd.control.iterative_ilqr import HipCandidateSolver
    from ilqr_iterative_tasks_amd import dist as idist
    with pytest.raises(ValueError, match="at least one candidate"):
        HipCandidateSolver().sharded_round(cfg, None, torch.zeros(0, cfg.n), None, 1.0,
                                           idist.TorchExchange(), 0)
    solver.close()


def test_device_geometry_comes_from_the_runtime(torch_mod):
    """CU count, LDS per CU and the dynamic-LDS limits are hipDeviceGetAttribute's
    answers.  On the full MI355X they must be exactly what rounds 1-5 had compiled in (256 CUs,
    160 KiB per CU, 64 KiB without opt-in, 64-lane wavefronts): every derived budget and threshold
    is then unchanged, which the bit-identity and kernel-name tests of the suite check."""
    import ctypes as C
    from ilqr_iterative_tasks_amd import _abi
    torch_mod.cuda.current_device()
    geo = (C.c_int32 * 8)()
    assert _abi.load_library().i2lqr_device_geometry(geo, 8) == 0
    cus, simds, lds, max_dyn, dflt, wave, faked, queried = list(geo)
    assert queried == 1 and faked == 0 and wave == 64 and simds == 4
    assert cus == torch_mod.cuda.get_device_properties(0).multi_processor_count
    if cus == 256:  # (an unpartitioned MI355X)
        assert (lds, max_dyn, dflt) == (160 * 1024, 160 * 1024, 64 * 1024)


def test_a_device_with_a_quarter_of_the_cus_still_passes_parity():
    """I2LQR_FAKE_CUS=64 (the debug override of the queried CU count: a CPX-like partition): the
    thresholds scale by 64 / 256 — 4096 problems already take the one-problem-per-lane layout, the
    helper-wavefront kernel stops at 128 workgroups, LDS-resident gain steps are budgeted for 64
    CUs — and the results still match the CPU oracle at 4096 and 16384 problems."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, I2LQR_FAKE_CUS="64")
    out = subprocess.run([sys.executable, str(root / "tools" / "fake_cus_parity.py")], env=env,
                         capture_output=True, text=True, timeout=900, cwd=str(root))
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["geometry"][0] == 64 and d["geometry"][6] == 1
    by = {(c["B"], c["solve"]): c for c in d["cases"]}
    assert by[(4096, False)]["layout"] == 2 and by[(4096, False)]["kernel"] == "k_lane_iterate_pair"
    assert by[(16384, False)]["kernel"] == "k_lane_iterate"  # 256 workgroups > 128: one wavefront
    for c in d["cases"]:
        assert c["same_branch"] > 0.97, c
        assert c["X_err"] < 1e-8 and c["U_err"] < 1e-7 and c["cost_err"] < 1e-7, c


def test_exchange_stream_runs_beside_the_launch_stream(torch_mod):
    """dist.exchange_stream(): two HIP streams may share a hardware queue (every fourth stream a
    process creates lands on the launch stream's), and an exchange on a shared queue runs BEHIND
    the next solve instead of beside it.  The probed stream's kernels finish while a spin kernel on
    the current stream is still running — for several streams in a row, so at least one unlucky
    candidate has been skipped on the way."""
    torch = torch_mod
    from ilqr_iterative_tasks_amd import dist as idist
    main = torch.cuda.current_stream()
    x = torch.zeros(64, device="cuda")
    ev = lambda: torch.cuda.Event(enable_timing=True)
    a, b = ev(), ev()
    torch.cuda._sleep(1000)
    torch.cuda.synchronize()
    a.record()
    torch.cuda._sleep(200_000)
    b.record()
    torch.cuda.synchronize()
    ticks = int(200_000 * 0.5 / a.elapsed_time(b))  # a spin of ~0.5 ms
    seen = set()
    for _ in range(6):
        s = idist.exchange_stream()
        assert s != main
        seen.add(s.cuda_stream)
        e0, e1, es = ev(), ev(), ev()
        torch.cuda.synchronize()
        e0.record()
        torch.cuda._sleep(ticks)
        e1.record()
        with torch.cuda.stream(s):
            x.add_(1.0)
            es.record(s)
        torch.cuda.synchronize()
        assert e0.elapsed_time(es) < 0.5 * e0.elapsed_time(e1), (e0.elapsed_time(es), e0.elapsed_time(e1))
    assert len(seen) >= 2


def test_chain_on_the_device_equals_the_step_by_step_chain(torch_mod):
    """HipCandidateSolver.solve_chained (round 6): the reference's chained lamb — candidate c + 1 of
    a lap starts from candidate c's final lamb, utils/base.py:393, :414-426 — with the whole round
    on the device: one upload, one launch per chain step whose lamb is gathered on the device from
    the step before, one read-back.  Bit-identical to the step-by-step form (a host round trip per
    step), for laps of equal and of different lengths, with and without the obstacle."""
    from ilqr_iterative_tasks_amd import default_config
    from ilqr_iterative_tasks_amd.control.iterative_ilqr import HipCandidateSolver
    cfg = default_config("bicycle4", 6)
    rng = np.random.default_rng(5)
    x0 = np.array([0.0, 0.0, 1.0, 0.0])
    hs = HipCandidateSolver()
    hs_steps = HipCandidateSolver()
    hs_steps.use_chain_kernel = False  # a launch per chain step (hipGraph replay) for equal laps too
    for widths, obs in (((8, 8), (31.0, -3.0, 8.0, 6.0, 0.0, 0.0)), ((8, 5, 3), None), ((1,), None),
                        ((4, 8), (12.0, 1.0, 6.0, 4.0, 0.0, 0.0)), ((6, 6, 6, 6, 6), None)):
        chains = [np.column_stack([rng.uniform(5, 40, w), rng.uniform(-4, 4, w),
                                   rng.uniform(0.5, 3, w), rng.uniform(-0.3, 0.3, w)]) for w in widths]
        got = hs.solve_chained(cfg, x0, chains, 1.0, obs)
        # laps of equal length: ONE launch (i2lqr_solve_chained: k_group_spec<.., CHAIN>)
        assert hs.chain_info["one_launch"] == (len(set(widths)) == 1)
        for rep in range(2):  # (second call: the captured graph is replayed)
            again = hs_steps.solve_chained(cfg, x0, chains, 1.0, obs)
            assert hs_steps.chain_info["one_launch"] is False
            for a in range(len(widths)):
                for key in ("U", "X", "lamb", "cost", "iters", "status"):
                    assert np.array_equal(again[a][key], got[a][key]), (widths, a, key, rep)
        lamb = np.full(len(widths), 1.0)
        for c in range(max(widths)):
            rows = [a for a, w in enumerate(widths) if c < w]
            ref = hs.solve(cfg, x0, np.stack([chains[a][c] for a in rows]), lamb[rows], obs)
            for r, a in enumerate(rows):
                for key in ("U", "X", "lamb", "cost", "iters", "status"):
                    assert np.array_equal(got[a][key][c], ref[key][r]), (widths, a, c, key)
                lamb[a] = ref["lamb"][r]
        assert all(len(g["lamb"]) == w for g, w in zip(got, widths))

"""GPU suite: bench.py honours the driver's contract — one JSON line with the BASELINE metric, the
roofline object of the dominant kernel and the CPU baseline — and __graft_entry__.smoke() passes."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_bench_json_line_contract():
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "3", "--warmup", "1",
                          "--no-extra", "--cpu-seconds", "1"], capture_output=True, text=True,
                         timeout=600, cwd=str(ROOT))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["metric"].startswith("batched iLQR iterations/s") and d["unit"] == "iLQR iterations/s"
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic"
    assert "configs[1]" in d["config"]["workload"] or "config2" in d["config"]["workload"]
    assert d["config"]["batch_per_gpu"] == 1024 and d["config"]["iterations_per_step"] == 10
    assert d["value"] > 1e6  # north_star floor: 1e6 batched iLQR iterations/s on one MI355X
    assert abs(d["value"] - 1024 * 10 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["algorithmic_bytes_per_iteration"] == 4968
    assert abs(r["achieved"] - 4968 * 1024 * 10 / (r["kernel_ms_avg"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
    assert r["traffic"] is None or r["traffic"] > 0
    assert r["traffic_stale"] in (True, False) and (r["traffic"] is None or not r["traffic_stale"])
    ri = d["roofline_issue"]
    # 1024 problems on the sixteen-lane form (four problems per wavefront): 256 main wavefronts,
    # one SIMD each (+ 256 helper wavefronts when the counter file of this library knows about
    # them: their instructions are in the numerator, so their SIMDs are in the denominator)
    assert ri["bound"] == "issue" and ri["simds_occupied"] in (256, 512)
    assert abs(ri["peak"] - ri["simds_occupied"] * 2.4 / 4) < 1e-9
    assert r["kernel_ms_samples"] >= 3 and r["kernel_ms_avg"] <= d["ms_per_step"]
    assert r["kernel_ms_min"] <= r["kernel_ms_avg"] <= r["kernel_ms_max"]
    assert r["kernel"] == "k_group_iterate (sixteen lanes)"
    # the control round is ONE launch: iterations, relaxed cost and pick
    assert d["config"]["launches_per_step"] == 1 and "ONE launch" in d["config"]["step"]
    assert 0.0 < r["accepted_fraction"] <= 1.0 and r["waves_per_simd"] == 0.25
    assert r["step_ms_outside_kernel"] >= 0.0
    assert ri["frac"] is None or 0 < ri["frac"] <= 1.0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
    assert c["reference_python"]["value"] == 756.0 and "BASELINE.md" in c["reference_python"]["provenance"]


CLAIMS = ("large_batch_B65536_f64_frac", "large_batch_B131072_f64_frac", "f32_B65536_frac",
          "quad12_frac", "quad12_Mits", "quad12_f32_frac", "sharded_step_ms", "unsharded_step_ms",
          "sharded_host_enqueue_ms", "sharded_over_unsharded", "sharded_over_unsharded_B131072",
          "control_step_ms", "control_step_ms_device_rounds", "solve_B65536_ms", "solve_B1024_ms",
          "solve_frac_of_fixed_count_rate", "mid_4096_Mits", "mid_8192_Mits", "mid_12288_Mits",
          "mid_16384_Mits", "mid_24576_Mits", "mid_32768_Mits")


def test_default_bench_line_carries_the_claim_scalars_at_the_head_of_roofline():
    """the driver keeps the head of `parsed.roofline`; round 5's claim scalars sat
    behind a dict and most were cut.  They are flat numbers now, immediately behind `frac`: a
    consumer that re-parses the line and keeps only the first 32 SCALAR keys of `roofline` (the six
    mandated keys + the claims) still finds every one."""
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "20", "--warmup", "5",
                          "--cpu-seconds", "1"], capture_output=True, text=True, timeout=1200,
                         cwd=str(ROOT))
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    r = d["roofline"]
    kept = {}
    for k, v in r.items():  # (json.loads keeps the order of the line)
        if isinstance(v, (dict, list)):
            break  # a consumer that stops at the first non-scalar
        kept[k] = v
        if len(kept) == 32:
            break
    for key in CLAIMS:
        assert key in kept, (key, list(kept))
        assert isinstance(kept[key], float) and kept[key] > 0, key
    assert list(r)[:6] == ["bound", "kernel", "achieved", "peak", "unit", "frac"]
    assert not any(isinstance(v, (dict, list)) for v in r.values())  # nothing nested in roofline
    assert 0.3 < r["large_batch_B65536_f64_frac"] < 1 and 0.2 < r["quad12_frac"] < 1
    assert r["large_batch_B65536_f64_frac"] == d["roofline_large_batch"]["B65536"]["hbm_frac"]
    assert r["solve_B65536_ms"] == d["roofline_solve"]["ms_per_solve"] < 3.0
    # the range VERDICT r4 #4 asked about: the lane side (helper-wavefront kernel) carries it upward
    assert r["mid_12288_Mits"] < r["mid_16384_Mits"] < r["mid_24576_Mits"] < r["mid_32768_Mits"]
    assert d["extra"]["B16384_f64"]["kernel"] == "k_lane_iterate_pair"
    assert "no gains" not in d["roofline_solve"]["outputs"] and "K" not in d["roofline_solve"]["outputs"]
    assert d["roofline_solve"]["ms_per_solve_with_gains_out"] > 0
    # the sharded step (one C-ABI call, exchange on a side stream) costs a rank at
    # most 15 % more than the unsharded one-launch step at 1024 problems, 5 % at 131072, and the
    # host needs less time to enqueue a step than the GPU to run it
    so = d["extra"]["sharded_overhead"]
    assert so["forms"]["sharded_one_call"]["step_ms"] == r["sharded_step_ms"]
    assert r["sharded_over_unsharded"] <= 1.15, so
    assert r["sharded_over_unsharded_B131072"] <= 1.05, d["extra"]["sharded_overhead_B131072"]
    assert r["sharded_host_enqueue_ms"] < r["sharded_step_ms"]
    # the reference's metric (control-step latency on configs[0]) through the product
    cl = d["extra"]["config1_closed_loop"]
    assert cl["chained"]["laps"] == [121, 54, 29, 23]
    assert r["control_step_ms"] == cl["chained"]["control_step_ms_mean"] < 700.0
    assert cl["reference"]["control_step_ms_mean"] == 700.0


def test_bench_under_torchrun_world_of_one_uses_the_native_rccl_exchange():
    """The driver's N > 1 command form with one rank (all this box has): RCCL process group,
    communicator created through the C-ABI, i2lqr_allgather_costs on the side stream, pick checked
    against the gathered vector inside bench.py; the strong-scaled configs[3] line beside it."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                          "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port",
                          str(port), str(ROOT / "bench.py"), "--gpus", "1", "--steps", "5",
                          "--warmup", "2", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["exchange"]["nccl_world"] == 1
    assert d["exchange"]["path"] == "native" and "i2lqr_allgather_round" in d["exchange"]["what"]
    assert d["exchange"]["ms_per_step"] > 0
    # the timed step is the product's sharded round — ONE C-ABI call (round 6) —, the winner's
    # hand-off included; the host-driven form of the same round and the two-collective form
    # (read-back + i2lqr_broadcast_winner) are timed beside it
    assert "sharded_round" in d["config"]["step"] and "hand-off" in d["config"]["step"]
    assert "i2lqr_sharded_round_flat" in d["config"]["step"] and d["exchange"]["one_call_round"] is True
    assert set(d["exchange"]["phases_ms"]) == {"gather_costs_and_packs", "pick_and_winner"}
    hv = d["exchange"]["host_driven_variant"]
    assert hv["ms_per_step"] > 0 and hv["one_call_ms_per_step"] == pytest.approx(d["ms_per_step"])
    bv = d["exchange"]["broadcast_variant"]
    assert bv["ms_per_step"] > 0 and bv["phases_ms"]["broadcast_winner"] > 0
    assert d["config"]["launches_per_step"] == 4  # solve | pack, grouped gather, pick + hand-off
    # the driver keeps the head of the line: the multi-rank facts come before the long objects
    head = lines[0][:2000]
    assert '"exchange"' in head and '"per_rank_iterations_per_s"' in head and '"nccl_world"' in head
    assert lines[0].index('"exchange"') < lines[0].index('"roofline"') < lines[0].index('"extra"')
    assert d["exchange"]["bytes_per_rank"] == (1024 + 2 * 20 + 6 * 21) * 8  # costs + the local winner's pack
    assert len(d["per_rank_iterations_per_s"]) == 1
    s4 = d["extra"]["config4_strong"]
    assert s4["global_batch"] == 1 << 20 and s4["batch_per_gpu"] == 1 << 20
    assert s4["nccl_world"] == 1 and s4["iterations_per_s"] > 1e8


def test_bench_mismatched_world_exits_nonzero():
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1",
                          "--warmup", "0", "--no-extra", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600, cwd=str(ROOT))
    # one GPU here: the two ranks cannot both get a device -> the launcher reports the failure
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith('{"metric"')]


def test_two_ranks_sharing_the_gpu_solve_their_shards_and_fall_back_together():
    """The N = 2 path with the real solve on this one-GPU box (I2LQR_BENCH_SHARE_GPU: both ranks on
    device 0, process group on gloo).  RCCL itself refuses two ranks on one device, so the bring-up
    of the native exchange fails inside the real library — as an agreed error or as a timed-out
    attempt the launcher ends — and the line that comes out is the torch exchange's, for twice the
    batch, with the pick checked against the gathered vector inside bench.py."""
    import os
    env = dict(os.environ, I2LQR_BENCH_SHARE_GPU="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "6",
                          "--warmup", "2", "--no-extra", "--no-cpu-baseline", "--launch-timeout", "240"],
                         capture_output=True, text=True, timeout=900, cwd=str(ROOT), env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 2048 and d["value"] > 1e6
    assert len(d["per_rank_iterations_per_s"]) == 2
    ex, att = d["exchange"], d["launcher"]["attempts"]
    print("exchange:", ex, "launcher:", d["launcher"])
    assert ex["path"] == "torch" and "torch.distributed" in ex["what"]
    assert "sharded_round" in d["config"]["step"]
    assert ex["broadcast_variant"]["phases_ms"]["broadcast_winner"] > 0
    (ROOT / "gpurun_out").mkdir(exist_ok=True)  # kept: copied to profiles/ as the round's record
    (ROOT / "gpurun_out" / "two_ranks_shared_gpu.json").write_text(json.dumps(d, indent=1))
    # two ranks: the contract's 6 steps and the 200-step loop beside them
    assert d["long_run"]["steps"] == 200 and d["long_run"]["value"] > 1e6
    assert len(d["long_run"]["per_rank_iterations_per_s"]) == 2
    # either the ranks agreed on the failure inside one attempt, or the launcher restarted them
    assert ("native_exchange_error" in ex and len(att) == 1) or \
        (len(att) == 2 and att[1]["argv"] == ["--exchange", "torch"])


def test_native_allgather_matches_torch_world_of_one():
    """i2lqr_comm_* / i2lqr_allgather_costs without any process group: a world of one."""
    import torch
    from ilqr_iterative_tasks_amd import BatchedILQR, default_config
    from ilqr_iterative_tasks_amd.dist import CostExchange
    for dtype in ("f64", "f32"):
        solver = BatchedILQR(default_config("bicycle4", 6, dtype))
        ex = CostExchange(solver)
        assert (ex.world, ex.rank, ex.comm_world, ex.comm_rank) == (1, 0, 1, 0)
        cost = torch.rand(4099, dtype=solver.dtype, device=solver.device)
        out = ex.allgather(cost)
        torch.cuda.synchronize()
        assert torch.equal(out, cost) and out.data_ptr() != cost.data_ptr()
        with pytest.raises(ValueError):
            ex.allgather(cost, torch.zeros(5, dtype=solver.dtype, device=solver.device))
        ex.close()
        solver.close()


def test_graft_entry_smoke():
    out = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"],
                         capture_output=True, text=True, timeout=600, cwd=str(ROOT))
    assert out.returncode == 0, out.stderr[-2000:]
    assert "smoke ok" in out.stdout


@pytest.mark.parametrize("handoff", ["gather", "broadcast"])
def test_two_gpus_native_rccl_round_hands_over_the_owners_trajectory(handoff):
    """the grouped all-gather of costs and packs (and the two-collective form beside it)
    over the library's own RCCL communicator with a world ABOVE one — gated on a second GPU (the
    test boxes of this repository have one: skipped there; the driver's multi-GPU node runs it).
    bench.py itself asserts, on every rank, that the pick is the first-index arg-min of the
    gathered costs and, on the owner, that the handed-over pack is the trajectory it solved."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two HIP devices")
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "5",
                          "--warmup", "2", "--no-cpu-baseline", "--no-extra", "--handoff", handoff],
                         capture_output=True, text=True, timeout=1200, cwd=str(ROOT))
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 2048
    assert d["exchange"]["path"] == "native" and d["exchange"]["nccl_world"] == 2
    assert d["launcher"]["fallback"] is None
    if handoff == "gather":
        assert d["exchange"]["one_call_round"] is True
        assert d["exchange"]["broadcast_variant"]["ms_per_step"] > 0
    else:
        assert "ncclBroadcast" in d["exchange"]["handoff"]
    assert len(d["per_rank_iterations_per_s"]) == 2 and d["value"] > 1e6

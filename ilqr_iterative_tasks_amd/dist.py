"""Multi-GPU sharding of the candidate batch: one process per GPU, no data-path collective except
ONE all-gather of the per-candidate terminal costs (RCCL over xGMI via torch.distributed, backend
"nccl"; "gloo" in the CPU tests), followed by the arg-min every rank evaluates locally.

The batch axis is the safe-set terminal candidates of iLqr.calc_input (utils/base.py:391-455, the
`for id` / `for j` loops); candidates are independent given their own lamb0, and the only
cross-candidate step is the pick at utils/base.py:462-469.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend: str | None = None) -> tuple[int, int, int]:
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun).
    Returns (rank, world_size, local_rank); a no-op single-process world if WORLD_SIZE is unset."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # under a launcher (torchrun exports RANK / WORLD_SIZE) the process group is created even for
    # a world of one, so the single-GPU box exercises the same collective code path
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if (world > 1 or launched) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(total: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous shard [lo, hi) of `total` candidates owned by `rank` (first ranks take the
    remainder, so shards differ by at most one candidate)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def allgather_costs(cost_local: torch.Tensor, total: int | None = None, group=None) -> torch.Tensor:
    """The one collective of the path: every rank contributes its shard of cost_it and receives the
    full vector in shard order.  Equal shards use a single all_gather_into_tensor (one RCCL
    ncclAllGather, B/world elements per rank); ragged shards are padded with +inf to the largest
    shard and compacted afterwards."""
    if not (dist.is_available() and dist.is_initialized()):
        return cost_local
    world = dist.get_world_size(group)
    n_local = cost_local.numel()
    if cost_local.device.type != "cpu" and dist.get_backend(group) == "gloo":
        # gloo gathers host tensors only: stage through the host (a debugging set-up, e.g. several
        # ranks sharing one GPU; the production group is "nccl" and stays on the device)
        return allgather_costs(cost_local.cpu(), total, group).to(cost_local.device)
    if total is None or total == n_local * world:
        out = torch.empty(n_local * world, dtype=cost_local.dtype, device=cost_local.device)
        try:
            dist.all_gather_into_tensor(out, cost_local.contiguous(), group=group)
        except (RuntimeError, NotImplementedError):  # backend without the fused form
            parts = list(out.chunk(world))
            dist.all_gather(parts, cost_local.contiguous(), group=group)
        return out
    sizes = [shard_range(total, r, world) for r in range(world)]
    width = max(hi - lo for lo, hi in sizes)
    padded = torch.full((width,), float("inf"), dtype=cost_local.dtype, device=cost_local.device)
    padded[:n_local] = cost_local
    gathered = allgather_costs(padded, None, group).view(world, width)
    return torch.cat([gathered[r, : hi - lo] for r, (lo, hi) in enumerate(sizes)])


# Host-side side channel of a process group: the group itself if it runs on gloo, otherwise ONE gloo
# group made next to it.  The bring-up below agrees on its outcomes over this channel only, so that
# no device collective is issued by a rank whose helper thread may still sit inside ncclCommInitRank.
#
# dist.new_group is a collective of the DEFAULT group: every rank of the world must enter it, in
# the same order.  So this function makes the gloo group itself only for the world group (where
# "every rank of `group` calls host_group at the same point" IS "every rank of the world does");
# for a sub-group on a device backend the caller creates the host-side group where all world ranks
# pass (dist.new_group(ranks=..., backend="gloo")) and hands it in as `host`.
# The cache holds the group object next to its channel: the entry keeps the object alive, so its
# id() cannot be reused by another group.
_host_groups: dict = {}


class _DeviceChannel:
    """host_group(): no host-side transport could be made — the agreement runs on the group
    itself, with device tensors (the pre-round-4 behaviour)."""

    def __init__(self, group):
        self.group = group


def host_group(group=None, host=None):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return None
    if host is not None:
        return host
    if dist.get_backend(group) == "gloo":
        return group if group is not None else dist.group.WORLD
    if group is not None and group is not dist.group.WORLD:
        raise ValueError(
            "host_group(): a sub-group on a device backend needs its host-side (gloo) group passed "
            "in (host=...): dist.new_group must be entered by every rank of the default group, "
            "which the members of a sub-group cannot arrange on their own")
    key = id(group) if group is not None else None
    if key not in _host_groups:
        hg = None
        try:
            hg = dist.new_group(backend="gloo")
        except Exception:  # noqa: BLE001  (no gloo transport on this node)
            hg = None
        # Agree on the outcome over the EXISTING group before anyone uses the new one: a rank whose
        # new_group failed while the others' succeeded would otherwise sit on another channel than
        # its peers and the next collective would hang.
        flag = torch.tensor([1 if hg is not None else 0], dtype=torch.int32,
                            device="cuda" if torch.cuda.is_available() else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        _host_groups[key] = (group, hg if int(flag.item()) == 1 else _DeviceChannel(group))
    return _host_groups[key][1]


def _agree(flags, hgroup) -> list[bool]:
    """For every flag: true on every rank iff it is true on every rank (ONE all-reduce of host
    integers over the host-side group; a world of one agrees with itself)."""
    flags = [bool(f) for f in flags]
    if hgroup is None:
        return flags
    t = torch.tensor([1 if f else 0 for f in flags], dtype=torch.int32)
    if isinstance(hgroup, _DeviceChannel):
        t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=hgroup.group)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=hgroup)
    return [bool(int(v)) for v in t.cpu()]


class CostExchangeUnavailable(RuntimeError):
    """The native communicator could not be brought up — raised on EVERY rank of the group (the
    ranks agree on the outcome before and after the bootstrap), so the callers' fallbacks stay in
    step."""


class CostExchangePoisoned(CostExchangeUnavailable):
    """... because some rank's ncclCommInitRank did not return within its timeout (raised on every
    rank; `self.here` tells whether THIS process holds the abandoned helper thread).  Such a
    process group should not go on sharing a device with the abandoned bootstrap: a launcher that
    can (bench.py's own) ends the ranks and starts fresh ones on the torch exchange; a rank started
    by a foreign launcher carries on with the torch exchange and leaves through os._exit."""

    def __init__(self, msg, here: bool):
        super().__init__(msg)
        self.here = here


# Bring-ups whose ncclCommInitRank never returned: the daemon thread that made the call is still
# inside the library.  A host that sees a non-zero count should end with os._exit after flushing
# its output (bench.py does): the library's teardown may wait for that thread.
_abandoned_bring_ups = 0


def abandoned_bring_ups() -> int:
    return _abandoned_bring_ups


def _bring_up_timeout() -> float:
    """Seconds a rank waits for ncclCommInitRank (I2LQR_COMM_TIMEOUT, default 180; 0: forever)."""
    try:
        return float(os.environ.get("I2LQR_COMM_TIMEOUT", "180"))
    except ValueError:
        return 180.0


class CostExchange:
    """The all-gather of the candidates' terminal costs through the C-ABI (i2lqr_allgather_costs:
    one RCCL ncclAllGather on a communicator the library creates itself).

    The ranks agree on the communicator's unique id and on every step of the bring-up over a
    HOST-side channel (host_group(): gloo; the id is 128 bytes of host data); without a process
    group this is a world of one.  A host language other than Python does the same with
    i2lqr_comm_available / i2lqr_comm_unique_id / i2lqr_comm_create and its own side channel.

    Bring-up never leaves the ranks in different collectives: (1) every rank checks that RCCL can
    be bound and the ranks agree on that; (2) rank 0 makes the id and broadcasts it — an EMPTY id
    if it failed, so the others do not sit in the broadcast; (3) every rank joins
    ncclCommInitRank; (4) the ranks agree on the outcome — a rank whose communicator came up while
    another's did not frees it with i2lqr_comm_abort (ncclCommAbort: a destroy would wait for
    peers that never arrived).  Any failure raises CostExchangeUnavailable on all ranks.  Step (3)
    blocks inside RCCL until every rank has arrived; it runs in a helper thread and a rank that
    has waited `timeout` seconds (default I2LQR_COMM_TIMEOUT = 180) gives up and reports that in
    step (4) — where the other ranks, which are waiting for the same bootstrap, arrive the same
    way —; every rank then raises CostExchangePoisoned.  No device collective is issued between
    (1) and (4).  After a successful bring-up the device is synchronised and the ranks pass a
    host-side barrier, so the first collective of the library's communicator and the next one of
    torch's are ordered the same way on every rank."""

    def __init__(self, solver, group=None, timeout: float | None = None, host=None):
        """`host`: the host-side (gloo) group of a sub-group on a device backend, created where
        every rank of the world passes (see host_group); not needed for the world group."""
        import ctypes as C
        from . import _abi
        self.solver, self.lib = solver, solver.lib
        self._comm = None
        grouped = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if grouped else 1
        self.rank = dist.get_rank(group) if grouped else 0
        self._group = group
        hg = self._hgroup = host_group(group, host)
        # (1) can every rank bind RCCL?
        err = None
        try:
            solver._check(self.lib.i2lqr_comm_available())
        except Exception as e:  # noqa: BLE001
            err = e
        if not _agree([err is None], hg)[0]:
            raise CostExchangeUnavailable(f"RCCL cannot be bound on every rank ({err or 'another rank'})")
        # (2) the unique id, or an empty marker
        uid = C.create_string_buffer(_abi.COMM_ID_BYTES)
        payload = b""
        if self.rank == 0:
            try:
                solver._check(self.lib.i2lqr_comm_unique_id(uid))
                payload = uid.raw
            except Exception as e:  # noqa: BLE001
                err = e
        box = [payload]
        if isinstance(hg, _DeviceChannel):
            src0 = dist.get_process_group_ranks(hg.group)[0] if hg.group is not None else 0
            dist.broadcast_object_list(box, src=src0, group=hg.group)
        elif hg is not None:
            dist.broadcast_object_list(box, src=dist.get_process_group_ranks(hg)[0], group=hg)
        if not box[0]:
            raise CostExchangeUnavailable(f"rank 0 could not create the RCCL unique id ({err or 'see rank 0'})")
        # (3) the bootstrap, (4) agreement on its outcome (host side)
        comm = C.c_void_p()
        err = self._create(box[0], comm, _bring_up_timeout() if timeout is None else timeout)
        timed_out = isinstance(err, TimeoutError)
        all_ok, none_timed_out = _agree([err is None, not timed_out], hg)
        if not all_ok:
            if err is None and comm.value:
                self.lib.i2lqr_comm_abort(comm)
            if not none_timed_out:
                raise CostExchangePoisoned(
                    f"ncclCommInitRank timed out ({err if timed_out else 'on another rank'})",
                    here=timed_out)
            raise CostExchangeUnavailable(f"ncclCommInitRank failed ({err or 'on another rank'})")
        self._comm = comm
        w, r = C.c_int32(), C.c_int32()
        solver._check(self.lib.i2lqr_comm_info(self._comm, C.byref(w), C.byref(r)))
        self.comm_world, self.comm_rank = int(w.value), int(r.value)  # what RCCL itself reports
        # two communicators in this process from here on (torch's and the library's): start them
        # from a common point on every rank
        if torch.device(solver.device).type == "cuda":
            torch.cuda.synchronize(solver.device)
        if isinstance(hg, _DeviceChannel):
            dist.barrier(group=hg.group)
        elif hg is not None:
            dist.barrier(group=hg)

    def _create(self, uid: bytes, comm, timeout: float):
        """i2lqr_comm_create in a helper thread (the device current there too); returns the error
        or None.  After `timeout` seconds without an answer: a TimeoutError, the thread stays
        behind."""
        import ctypes as C
        import threading
        global _abandoned_bring_ups
        result = []

        def call():
            try:
                with _device_ctx(self.solver.device):
                    self.solver._check(self.lib.i2lqr_comm_create(C.c_char_p(uid), self.world,
                                                                  self.rank, C.byref(comm)))
                result.append(None)
            except Exception as e:  # noqa: BLE001
                result.append(e)

        th = threading.Thread(target=call, name="i2lqr-comm-bring-up", daemon=True)
        th.start()
        th.join(timeout if timeout and timeout > 0 else None)
        if th.is_alive():
            _abandoned_bring_ups += 1
            return TimeoutError(f"ncclCommInitRank did not return within {timeout:g} s on rank {self.rank}")
        return result[0]

    def allgather(self, cost_local: torch.Tensor, out: torch.Tensor | None = None,
                  total: int | None = None) -> torch.Tensor:
        """cost_local[n] on this rank -> the costs of all ranks in rank order, on the current
        stream.  Equal shards (total None or world * n): cost_all[world * n], written into `out`
        if given.  Ragged shards (`total` candidates split by shard_range): every shard is padded
        to the largest with +inf — which never wins the arg-min — gathered, and compacted to
        cost_all[total] (as allgather_costs does for the torch path)."""
        import ctypes as C
        n = cost_local.numel()
        if total is not None and total != n * self.world:
            sizes = [shard_range(total, r, self.world) for r in range(self.world)]
            if sizes[self.rank][1] - sizes[self.rank][0] != n:
                raise ValueError(f"rank {self.rank} owns {sizes[self.rank][1] - sizes[self.rank][0]} "
                                 f"of {total} candidates, got {n}")
            width = max(hi - lo for lo, hi in sizes)
            padded = torch.full((width,), float("inf"), dtype=cost_local.dtype,
                                device=cost_local.device)
            padded[:n] = cost_local
            gathered = self.allgather(padded).view(self.world, width)
            res = torch.cat([gathered[r, : hi - lo] for r, (lo, hi) in enumerate(sizes)])
            if out is not None:
                out.copy_(res)
                return out
            return res
        if out is None:
            out = torch.empty(n * self.world, dtype=cost_local.dtype, device=cost_local.device)
        if out.numel() != n * self.world or not out.is_contiguous() or not cost_local.is_contiguous():
            raise ValueError("cost_all must be a contiguous tensor of world * n_local elements")
        if cost_local.dtype != self.solver.dtype or out.dtype != self.solver.dtype:
            raise ValueError(f"cost tensors must be {self.solver.dtype}")
        with _device_ctx(self.solver.device):
            self.solver._check(self.lib.i2lqr_allgather_costs(
                self.solver._handle, self._comm, C.c_void_p(cost_local.data_ptr()),
                C.c_void_p(out.data_ptr()), n, self.solver._stream()))
        return out

    def allgather_round(self, cost_local: torch.Tensor, pack_local: torch.Tensor,
                        cost_all: torch.Tensor | None = None,
                        pack_all: torch.Tensor | None = None):
        """Costs AND every rank's local-winner pack in ONE grouped RCCL operation
        (i2lqr_allgather_round), on the current stream: (cost_all[world * n], pack_all[world, P]).
        Same n and P on every rank (ragged shards are padded by the caller)."""
        import ctypes as C
        n, P = cost_local.numel(), pack_local.numel()
        dt, dev = self.solver.dtype, cost_local.device
        if cost_all is None:
            cost_all = torch.empty(n * self.world, dtype=dt, device=dev)
        if pack_all is None:
            pack_all = torch.empty(self.world, P, dtype=dt, device=dev)
        for t in (cost_local, pack_local, cost_all, pack_all):
            if t.dtype != dt or not t.is_contiguous():
                raise ValueError(f"round tensors must be contiguous {dt}")
        if cost_all.numel() != n * self.world or pack_all.numel() != P * self.world:
            raise ValueError("cost_all / pack_all must hold world x the local sizes")
        with _device_ctx(self.solver.device):
            self.solver._check(self.lib.i2lqr_allgather_round(
                self.solver._handle, self._comm, C.c_void_p(cost_local.data_ptr()),
                C.c_void_p(cost_all.data_ptr()), n, C.c_void_p(pack_local.data_ptr()),
                C.c_void_p(pack_all.data_ptr()), P, self.solver._stream()))
        return cost_all, pack_all

    def broadcast(self, buf: torch.Tensor, root: int) -> torch.Tensor:
        """The winner's hand-off (i2lqr_broadcast_winner: one ncclBroadcast, in place): `buf` on rank
        `root` reaches every rank's `buf` (same shape and dtype everywhere), on the current stream."""
        import ctypes as C
        if not buf.is_contiguous() or buf.dtype != self.solver.dtype:
            raise ValueError(f"broadcast buffer must be a contiguous {self.solver.dtype} tensor")
        with _device_ctx(self.solver.device):
            self.solver._check(self.lib.i2lqr_broadcast_winner(
                self.solver._handle, self._comm, C.c_void_p(buf.data_ptr()), buf.numel(), int(root),
                self.solver._stream()))
        return buf

    def close(self):
        if getattr(self, "_comm", None) is not None and self._comm.value:
            if torch.device(self.solver.device).type == "cuda":
                torch.cuda.synchronize(self.solver.device)
            self.lib.i2lqr_comm_destroy(self._comm)
            self._comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _device_ctx(device):
    """torch.cuda.device(device) for a HIP device, nothing for the CPU doubles of the tests."""
    import contextlib
    return torch.cuda.device(device) if torch.device(device).type == "cuda" else contextlib.nullcontext()


class TorchExchange:
    """The exchanges of a sharded round over torch.distributed — the same three calls as
    CostExchange (allgather / allgather_round / broadcast) for process groups without the library's
    own communicator: gloo in the CPU tests and on the shared-GPU box (device tensors are staged
    through the host there), torch's NCCL group as the fallback of a failed native bring-up.  A
    world of one returns its inputs."""

    def __init__(self, group=None):
        self.group = group
        grouped = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if grouped else 1
        self.rank = dist.get_rank(group) if grouped else 0
        self._host = grouped and dist.get_backend(group) == "gloo"

    def _stage(self, t):
        return t.cpu() if (self._host and t.device.type != "cpu") else t

    def allgather(self, cost_local, out=None, total=None):
        res = allgather_costs(cost_local, total, self.group) if self.world > 1 else cost_local
        if out is not None:
            out.copy_(res)
            return out
        return res

    def allgather_round(self, cost_local, pack_local, cost_all=None, pack_all=None):
        if self.world == 1:
            return cost_local, pack_local.reshape(1, -1)
        ca = allgather_costs(cost_local.contiguous(), None, self.group)
        pa = allgather_costs(pack_local.contiguous(), None, self.group).view(self.world, -1)
        if cost_all is not None:
            cost_all.copy_(ca)
            ca = cost_all
        if pack_all is not None:
            pack_all.copy_(pa.view_as(pack_all))
            pa = pack_all
        return ca, pa

    def broadcast(self, buf, root):
        if self.world == 1:
            return buf
        src = dist.get_process_group_ranks(self.group)[root] if self.group is not None else root
        t = self._stage(buf)
        dist.broadcast(t, src=src, group=self.group)
        if t is not buf:
            buf.copy_(t)
        return buf

    def close(self):
        pass


def native_comm(exchange):
    """(communicator, usable) for the one-call form of the flat round (i2lqr_sharded_round_flat):
    the library's own RCCL communicator of a CostExchange (also behind a ShardedRound), or
    (None, True) for a world of one over torch.distributed (the gathers become device copies);
    (None, False) for every other exchange — those rounds are driven from Python (flat_round)."""
    ex = getattr(exchange, "exchange", exchange)  # ShardedRound -> what it wraps
    if isinstance(ex, CostExchange):
        return ex._comm, ex._comm is not None
    if isinstance(ex, TorchExchange) and ex.world == 1:
        return None, True
    return None, False


def exchange_stream(device=None, tries: int = 8):
    """A stream for the exchange of a sharded round whose work really runs BESIDE the current
    stream's (HipCandidateSolver.sharded_round(exchange_stream=...), i2lqr_sharded_round_flat's
    side_stream).  Two HIP streams may share a hardware queue — on MI355X every fourth stream a
    process creates lands on the launch stream's — and work on a shared queue runs BEHIND the launch
    stream's kernels, not beside them: the exchange of round i then waits for the solve of round
    i + 1 (measured: +10 % per step at 8192 problems per rank, +60 % with a high-priority stream
    at 1024; profiles/r06_sharded_overhead_sweep.json).  So the candidates are PROBED: a spin kernel
    of ~0.3 ms on the current stream, a tiny kernel on the candidate enqueued behind it — the first
    candidate whose kernel finishes while the spin is still running is returned (the first
    candidate if none does)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    with torch.cuda.device(dev):
        main = torch.cuda.current_stream(dev)
        probe = torch.zeros(64, device=dev)
        ev = lambda: torch.cuda.Event(enable_timing=True)
        # what a spin of 200000 ticks lasts here (the tick of torch.cuda._sleep is the device's)
        a, b = ev(), ev()
        torch.cuda._sleep(1000)
        torch.cuda.synchronize(dev)
        a.record(main)
        torch.cuda._sleep(200_000)
        b.record(main)
        torch.cuda.synchronize(dev)
        ticks = int(200_000 * 0.3 / max(a.elapsed_time(b), 1e-3))  # ~0.3 ms
        first = None
        for _ in range(max(1, tries)):
            s = torch.cuda.Stream(dev)
            first = first or s
            e0, e1, es = ev(), ev(), ev()
            torch.cuda.synchronize(dev)
            e0.record(main)
            torch.cuda._sleep(ticks)
            e1.record(main)
            with torch.cuda.stream(s):
                probe.add_(1.0)
                es.record(s)
            torch.cuda.synchronize(dev)
            if e0.elapsed_time(es) < 0.5 * e0.elapsed_time(e1):
                return s
        return first


def padded_width(total: int, world: int) -> int:
    """Largest shard of shard_range(total, ., world): what every rank pads its costs to."""
    return (total + world - 1) // world


def _noop(name):
    pass


def flat_round(exchange, cost_local, pack_local, total, argmin, round_winner, bufs=None,
               on_phase=_noop):
    """The exchange of ONE sharded round whose pick is the flat arg-min — no host round trip.

    cost_local[n_local]: this rank's relaxed costs (its shard_range share of `total` candidates);
    pack_local[P]: the packed trajectory (U, X) of this rank's LOCAL winner.  One grouped all-gather
    carries both (the global winner is the local winner of its owner), then every rank evaluates
      argmin(cost_all) -> (best_padded[1], best_cost[1])              (i2lqr_argmin)
      round_winner(width, total, best_padded, pack_all) -> (winner[P], best_global[2])
    (i2lqr_round_winner: owner's pack + index in the unpadded batch) on the same gathered data.
    Returns dict(cost_all (padded to world x width with +inf), best_idx int64[2] = (index, owner),
    best_cost, pack).  Everything is enqueued on the current stream."""
    world = exchange.world
    width = padded_width(total, world)
    n_local = cost_local.numel()
    if n_local != width:  # ragged: pad with +inf, which never wins
        padded = bufs["padded"] if bufs else torch.empty(width, dtype=cost_local.dtype,
                                                         device=cost_local.device)
        padded.fill_(float("inf"))
        padded[:n_local] = cost_local
        cost_local = padded
    on_phase("start")
    cost_all, pack_all = exchange.allgather_round(
        cost_local, pack_local, bufs["cost_all"] if bufs else None,
        bufs["pack_all"] if bufs else None)
    on_phase("gathered")
    best_padded, best_cost = argmin(cost_all)
    pack, best_global = round_winner(width, total, best_padded, pack_all)
    on_phase("picked")
    return dict(cost_all=cost_all, best_idx=best_global, best_cost=best_cost, pack=pack,
                width=width)


def lexi_round(exchange, cost_local, total, pick, pack_of, pack_numel, on_phase=_noop):
    """The exchange of ONE sharded round with the reference's list-of-lists pick (the controller:
    utils/base.py:462-471): all-gather of the costs, pick(cost_all) -> flat index of the winner on
    the HOST (the controller needs it there anyway: its safe-set bookkeeping is host state, and the
    root of a broadcast must be known on the host), then ONE broadcast of the winner's pack from the
    rank that solved it (pack_of(local index) -> tensor[pack_numel], called on the owner only).
    on_phase(name) is called at "start", "gathered", "picked" and "handed_over".
    Returns dict(cost_all[total], index, owner, pack)."""
    on_phase("start")
    cost_all = exchange.allgather(cost_local, total=total)
    on_phase("gathered")
    index = int(pick(cost_all))
    on_phase("picked")
    owner, loc = owner_of(max(index, 0), total, exchange.world)
    if exchange.rank == owner:
        pack = pack_of(loc).contiguous()
    else:
        pack = torch.empty(pack_numel, dtype=cost_all.dtype, device=cost_all.device)
    exchange.broadcast(pack, owner)
    on_phase("handed_over")
    return dict(cost_all=cost_all, index=index, owner=owner, pack=pack)


def select_best_flat(cost_all: torch.Tensor) -> tuple[int, float]:
    """Flat arg-min with first-index tie-break (used for the synthetic 10^4..10^6 batches)."""
    val, idx = torch.min(cost_all, dim=0)
    # torch.min does not promise the first index on ties: resolve explicitly
    first = int(torch.nonzero(cost_all == val, as_tuple=False)[0, 0]) if not torch.isnan(val) \
        else int(idx)
    return first, float(val)


def select_best_lexicographic(cost_rows: list[list[float]]) -> tuple[int, int]:
    """The reference's pick (utils/base.py:462-465): `cost_list.index(min(cost_list))` on a LIST OF
    LISTS — Python compares the per-lap lists lexicographically — then the first minimum inside
    that lap's list.  Returns (lap_position, candidate_position)."""
    best_lap = cost_rows.index(min(cost_rows))
    row = cost_rows[best_lap]
    return best_lap, row.index(min(row))


def owner_of(index: int, total: int, world: int) -> tuple[int, int]:
    """(rank, local index) of candidate `index` under shard_range's contiguous split."""
    if not 0 <= index < total:
        raise ValueError(f"candidate {index} outside [0, {total})")
    for r in range(world):
        lo, hi = shard_range(total, r, world)
        if lo <= index < hi:
            return r, index - lo
    raise AssertionError("shard_range does not cover the batch")


class ShardedRound:
    """What a sharded controller is given (control.iLqr(sharded=...)): the exchange of its process
    group — `native`, a CostExchange (RCCL through the C-ABI), else a TorchExchange on `group` — and
    a count of the collectives issued (tests: two per solved round, the all-gather of cost_it and
    the winner's hand-off; SURVEY.md §8e, utils/base.py:391-471).  The round itself is
    HipCandidateSolver.sharded_round (device tensors end to end); without a process group this
    is a world of one."""

    def __init__(self, group=None, native: "CostExchange | None" = None):
        self.group, self.native = group, native
        self.exchange = native if native is not None else TorchExchange(group)
        self.world, self.rank = self.exchange.world, self.exchange.rank
        self.collectives = 0

    def shard(self, total: int) -> tuple[int, int]:
        return shard_range(total, self.rank, self.world)

    # counted pass-throughs (the round functions above take `self` as their exchange)
    def allgather(self, cost_local, out=None, total=None):
        self.collectives += 1
        return self.exchange.allgather(cost_local, out, total=total)

    def allgather_round(self, cost_local, pack_local, cost_all=None, pack_all=None):
        self.collectives += 1
        return self.exchange.allgather_round(cost_local, pack_local, cost_all, pack_all)

    def broadcast(self, buf, root):
        self.collectives += 1
        return self.exchange.broadcast(buf, root)

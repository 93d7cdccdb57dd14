"""Drop-in for the reference seam `ilqr()` (iterative_ilqr/control/iterative_ilqr.py:7-85) and the
batched candidate solver the controller uses in its place.

`ilqr(...)` keeps the reference's positional signature and return value `(uvar, xvar, lamb)` and
solves the single problem on the GPU through the C-ABI; `HipCandidateSolver.solve()` is the batched
form: all safe-set terminal candidates of one controller round in one launch.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from .params import config_from_params, obstacle_record


class HipCandidateSolver:
    """Batched ilqr() on the GPU: host NumPy in, host NumPy out, device work through BatchedILQR.

    One BatchedILQR (one C-ABI handle) is cached per distinct configuration."""

    def __init__(self, device="cuda:0", dtype="f64"):
        self.device = device
        self.dtype = dtype
        self._solvers = {}

    def _solver(self, cfg, B=None, early_exit=True):
        """The handle for this configuration; with B given, in the layout the LIBRARY recommends
        for batches of that size (i2lqr_recommended_layout): a caller who hands 65536 candidates
        to solve() gets the one-problem-per-lane kernels without knowing that they exist."""
        from ..solver import BatchedILQR
        if B is not None:
            lay = BatchedILQR.recommended_layout(cfg, B, early_exit)
            if lay != cfg.layout:
                cfg = cfg.copy()
                cfg.layout = lay
        key = bytes(C.string_at(C.byref(cfg), C.sizeof(cfg)))
        if key not in self._solvers:
            self._solvers[key] = BatchedILQR(cfg, self.device)
        return self._solvers[key]

    def _round_buffers(self, cfg, x0, x_terms, qfun, lamb0, obs_rec, early_exit):
        """Device buffers of one round's candidates (cached per solver and batch size) with the
        round's inputs written: X[:, :, 0] = x0, U = 0, lamb = lamb0 (utils/base.py:393, :405-408),
        x_term, the round's obstacle record.  Returns (solver, buf, qfun int32, cost_it)."""
        import torch
        B = int(x_terms.shape[0])
        solver = self._solver(cfg, B, early_exit=early_exit)
        key = (id(solver), B)
        if not hasattr(self, "_round_bufs"):
            self._round_bufs = {}
        if key not in self._round_bufs:
            if len(self._round_bufs) > 8:
                self._round_bufs.clear()
            self._round_bufs[key] = (solver.alloc(B, want_gains=False),
                                     torch.zeros(B, dtype=solver.dtype, device=solver.device))
        buf, cost_it = self._round_bufs[key]
        solver.set_initial_state(buf, x0, lamb0, zero_states=False)  # (the buffer starts zeroed)
        buf["x_term"].copy_(solver.to_native(x_terms.to(solver.device, solver.dtype)))
        obs_key = None if obs_rec is None else tuple(float(v) for v in obs_rec)
        if buf.get("_obs_key", ()) != obs_key:  # the round's obstacle record, shared by its candidates
            if obs_rec is not None:
                rec = torch.as_tensor(np.asarray(obs_rec, float)).to(solver.device, solver.dtype)
                buf["obs"] = solver.to_native(rec[None, :].expand(B, -1).contiguous())
            else:
                buf["obs"] = None
            buf["_obs_key"] = obs_key
        return solver, buf, qfun.to(solver.device, torch.int32), cost_it

    @staticmethod
    def _solve_and_cost(solver, buf, qfun, cost_it, n_iters, outer_iter, max_relax_iter, pick):
        """The candidate loops' body (utils/base.py:414-437) on a prepared buffer: solve (or
        n_iters fused iterations), relaxed cost, and (pick) the flat arg-min over THIS buffer's
        candidates — one launch on the eight- / sixteen-lane kernels.  Returns (idx, val) or None."""
        if n_iters is None:
            solver.solve(buf)
            solver.relax_cost(buf["X"], buf["x_term"], qfun, outer_iter, max_relax_iter, cost_it)
            return solver.argmin(cost_it) if pick else None
        return solver.iterate_pick(buf, int(n_iters), qfun, outer_iter, max_relax_iter, cost_it,
                                   pick=pick)[1]

    def candidate_round(self, cfg, x0, x_terms, qfun, lamb0, obs_rec=None, n_iters=None,
                        outer_iter=0, max_relax_iter=55):
        """One control round on DEVICE tensors, for any number of candidates: the body of the
        candidate loops utils/base.py:403-437 and the flat pick :462-465 without a host round trip.
        x0[n] (shared by the candidates), x_terms[B, n], qfun[B] (int32) are device tensors;
        n_iters None: solve to termination (i2lqr_solve), else that many fused iterations.
        Returns device tensors: cost_it[B], best_idx[1] (int64), best_cost[1] and the winner's
        U[m, N], X[n, N+1]; `solver` / `buf` (the layout the library chose and the full batch in
        it) ride along for callers that want more than the winner."""
        solver, buf, qfun, cost_it = self._round_buffers(cfg, x0, x_terms, qfun, lamb0, obs_rec,
                                                         n_iters is None)
        idx, val = self._solve_and_cost(solver, buf, qfun, cost_it, n_iters, outer_iter,
                                        max_relax_iter, True)
        U, X = solver.unpack(solver.pack_problem(buf, idx))
        return dict(cost_it=cost_it, best_idx=idx, best_cost=val, U=U, X=X, solver=solver, buf=buf)

    def sharded_round(self, cfg, x0, x_terms_local, qfun_local, lamb0, exchange, total,
                      obs_rec=None, n_iters=None, outer_iter=0, max_relax_iter=55, lexi=None,
                      prepared=None, bufs=None, exchange_stream=None, on_phase=None, plan=None,
                      host_driven=False):
        """ONE sharded control round, device-resident end to end (the loops
        utils/base.py:391-455 sharded, the pick and the hand-off :462-471): this rank holds the
        candidates [lo, hi) = dist.shard_range(total, rank, world) of the round — x_terms_local
        [n_local, n], qfun_local[n_local] (device tensors; x0[n] is shared) —, solves them and
        forms their relaxed costs (as candidate_round does), and the ranks exchange through
        `exchange` (dist.CostExchange: RCCL through the C-ABI; dist.TorchExchange; or a
        dist.ShardedRound wrapping either).  Every rank returns the same
            best_idx   index of the winner in the round's flat candidate list
            U, X       the winner's trajectory (device tensors [m, N], [n, N+1])
            cost_all   the gathered costs
        Two forms of the pick:
          lexi=None    the flat arg-min (first index wins).  No host round trip: every rank packs
                       its LOCAL winner and the packs ride with the costs in one grouped
                       all-gather; the pick and the owner's pack are then selected on the gathered
                       data.  best_idx is a device int64[2] = (index, owner rank), cost_all is
                       padded per rank to the largest shard with +inf.  Over the library's own
                       communicator (and in a world of one) the whole round is ONE C-ABI call,
                       i2lqr_sharded_round_flat (round 6): the launch stream carries the shard's
                       solve only and everything else is enqueued on the exchange stream from C.
                       Over torch.distributed (gloo in the CPU tests, torch's NCCL group as the
                       fallback of a failed native bring-up), or with host_driven=True, the same
                       steps are driven from Python (dist.flat_round: i2lqr_pack_problem,
                       exchange.allgather_round, i2lqr_argmin, i2lqr_round_winner) — results are
                       bit-identical.  A rank WITHOUT candidates (total < world) contributes +inf
                       costs and a pack of zeros: no rank raises while the others sit in the
                       collective; total == 0 raises on every rank.
          lexi=(L, k)  the reference's list-of-lists order over L laps of k candidates
                       (i2lqr_pick_best on the gathered vector); the index is read back — the
                       controller's bookkeeping is host state and a broadcast needs its root on the
                       host — and the owner hands the winner over with ONE broadcast
                       (i2lqr_broadcast_winner).  best_idx is a host int.  (dist.lexi_round)
          lexi=callable  cost_all -> flat index on the host (ragged laps: the controller's
                       list-of-lists pick on the gathered costs).
        prepared = (solver, buf, qfun, cost_it): buffers already resident and initialised (bench.py:
        the timed step starts with its inputs in HBM); bufs: preallocated exchange buffers
        (BatchedILQR.round_buffers); plan: a BatchedILQR.plan_round() made from both (the
        argument block of the one-call form, built once per buffer set).  exchange_stream: the
        exchange and everything behind it is enqueued THERE, behind an event recorded after the
        shard's solve — the next round's solve can then run on the current stream beside it (the
        results are valid on exchange_stream).  Without `bufs` / `plan` the round runs on buffers
        cached per batch size and starts behind the previous round's exchange (guard_previous):
        reusing them is safe whatever the streams.  on_phase(name): "solved" on the current stream
        behind the shard's solve, then (host-driven form) see dist.flat_round / lexi_round."""
        from .. import dist as idist
        import torch
        if int(total) < 1:  # (every rank knows `total`: all raise together, before any collective)
            raise ValueError("a sharded round needs at least one candidate over all ranks")
        on_phase = on_phase or (lambda name: None)
        empty_shard = plan is None and prepared is None and int(x_terms_local.shape[0]) == 0
        if plan is None and prepared is None and exchange_stream is not None:
            # this round's inputs are written into buffers cached per batch size: not before the
            # previous round's exchange (which packs its winner out of them) has passed them
            torch.cuda.current_stream(self.device).wait_stream(exchange_stream)
        if plan is not None:
            solver, buf, cost_it = plan["solver"], plan["keep"][0], plan["cost_local"]
            qfun = plan["keep"][1]
        elif empty_shard:  # more ranks than candidates: this rank only takes part in the exchange
            solver = self._solver(cfg)
            buf, qfun = None, None
            cost_it = torch.zeros(0, dtype=solver.dtype, device=solver.device)
        elif prepared is None:
            solver, buf, qfun, cost_it = self._round_buffers(cfg, x0, x_terms_local, qfun_local,
                                                             lamb0, obs_rec, n_iters is None)
        else:
            solver, buf, qfun, cost_it = prepared
        P = solver.m * solver.N + solver.n * (solver.N + 1)
        import contextlib

        def side():  # the exchange's stream: behind everything enqueued so far on the current one
            if exchange_stream is None:
                return contextlib.nullcontext()
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream(solver.device))
            exchange_stream.wait_event(ready)
            return torch.cuda.stream(exchange_stream)

        if plan is not None and (lexi is not None or host_driven or
                                 not idist.native_comm(exchange)[1]):
            raise ValueError("plan= belongs to the one-call form of the flat round (lexi None, the "
                             "library's own communicator or a world of one)")
        if lexi is None:
            comm, one_call = idist.native_comm(exchange)
            if one_call and not host_driven:
                if plan is None:
                    own = bufs is None
                    if own:  # cached per (solver, shard size, split): reused round after round
                        key = (id(solver), cost_it.numel(), int(total), exchange.world)
                        cache = self.__dict__.setdefault("_xbufs", {})
                        if len(cache) > 8 and key not in cache:
                            cache.clear()
                        bufs = cache.setdefault(key, {})
                    plan = solver.plan_round(buf, qfun, cost_it, total, exchange.world,
                                             exchange.rank, n_iters, outer_iter, max_relax_iter,
                                             bufs=bufs, guard_previous=own)
                solver.round_flat(plan, comm, exchange_stream)
                if hasattr(exchange, "collectives"):
                    exchange.collectives += 1
                on_phase("solved")
                res = dict(cost_all=plan["cost_all"], best_idx=plan["best_global"],
                           best_cost=plan["best_cost"], pack=plan["winner"], width=plan["width"])
            else:
                if empty_shard:
                    pack_local = (bufs["pack_local"].zero_() if bufs else
                                  torch.zeros(P, dtype=solver.dtype, device=solver.device))
                else:
                    lidx, _ = self._solve_and_cost(solver, buf, qfun, cost_it, n_iters, outer_iter,
                                                   max_relax_iter, True)
                on_phase("solved")
                if not empty_shard:
                    pack_local = solver.pack_problem(buf, lidx, bufs["pack_local"] if bufs else None)
                with side():
                    if exchange_stream is not None:
                        # handed across streams: the caching allocator must not recycle them for
                        # the launch stream while the exchange is still queued
                        for t in (pack_local, cost_it):
                            t.record_stream(exchange_stream)
                    res = idist.flat_round(
                        exchange, cost_it, pack_local, total,
                        lambda cost_all: solver.argmin(cost_all, side=True),  # (its own workspace)
                        lambda width, tot, best, pack_all: solver.round_winner(
                            exchange.world, width, tot, best, pack_all,
                            bufs["winner"] if bufs else None, bufs["best_global"] if bufs else None),
                        bufs, on_phase)
        else:
            if not empty_shard:
                self._solve_and_cost(solver, buf, qfun, cost_it, n_iters, outer_iter,
                                     max_relax_iter, False)
            on_phase("solved")
            if callable(lexi):
                pick = lexi
            else:
                L, k = lexi

                def pick(cost_all):
                    a, c = (int(v) for v in solver.pick_index(L, k, cost_all).cpu())
                    return a * k + c
            with side():
                if exchange_stream is not None:
                    cost_it.record_stream(exchange_stream)
                res = idist.lexi_round(
                    exchange, cost_it, total, pick,
                    lambda loc: solver.pack_problem(
                        buf, torch.tensor([loc], dtype=torch.int64, device=solver.device)), P,
                    on_phase)
            res["best_idx"] = res.pop("index")
            res["best_cost"] = None
        U, X = solver.unpack(res["pack"])
        res.update(U=U, X=X, solver=solver, buf=buf, cost_local=cost_it)
        return res

    def solve(self, cfg, x0, x_terms, lamb0, obs_rec, U0=None):
        """x0[n] (shared) or [B,n]; x_terms[B,n]; lamb0[B]; obs_rec[6] (shared) or None.
        Returns dict(U[B,m,N], X[B,n,N+1], lamb[B], iters[B], status[B], cost[B]) on the host."""
        x_terms = np.atleast_2d(np.asarray(x_terms, float))
        B = x_terms.shape[0]
        solver = self._solver(cfg, B, early_exit=True)
        if not (solver.batch_minor or solver.batch_tiled) and B <= 4096:
            return self._solve_packed(solver, cfg, B, x0, x_terms, lamb0, obs_rec, U0)
        return self._solve_generic(solver, cfg, B, x0, x_terms, lamb0, obs_rec, U0)

    def _solve_packed(self, solver, cfg, B, x0, x_terms, lamb0, obs_rec, U0):
        """The controller's call (a handful of candidates, problem-major): the chained-lamb mode
        makes 48 of these per control step, so the host side is what counts.  ONE pinned host block
        holds every input and output of the call in the layout of the device block behind it: one
        host-to-device copy, one launch, one copy back (the separate tensors of the generic path
        cost ten fills and eleven transfers per call: 150 us against 45)."""
        import torch
        n, m, N = cfg.n, cfg.m, cfg.N
        key = (id(solver), B, obs_rec is not None)
        if not hasattr(self, "_packs"):
            self._packs = {}
        pk = self._packs.get(key)
        if pk is None:
            if len(self._packs) > 16:
                self._packs.clear()
            item = 8 if solver.dtype == torch.float64 else 4
            npdt = np.float64 if item == 8 else np.float32
            sizes = [("X", B * n * (N + 1)), ("U", B * m * N), ("x_term", B * n), ("lamb", B),
                     ("obs", B * 6), ("cost", B)]
            off, o = {}, 0
            for name, cnt in sizes:
                off[name] = (o, cnt)
                o += (cnt * item + 15) // 16 * 16
            ioff = o
            o += 2 * ((B * 4 + 15) // 16 * 16)
            host = torch.empty(o, dtype=torch.uint8).pin_memory()
            dev = torch.empty(o, dtype=torch.uint8, device=solver.device)
            hview = {k: host[a:a + c * item].view(solver.dtype).numpy() for k, (a, c) in off.items()}
            dview = {k: dev[a:a + c * item].view(solver.dtype) for k, (a, c) in off.items()}
            step = (B * 4 + 15) // 16 * 16
            hint = {"iters": host[ioff:ioff + B * 4].view(torch.int32).numpy(),
                    "status": host[ioff + step:ioff + step + B * 4].view(torch.int32).numpy()}
            dint = {"iters": dev[ioff:ioff + B * 4].view(torch.int32),
                    "status": dev[ioff + step:ioff + step + B * 4].view(torch.int32)}
            buf = dict(X=dview["X"].view(B, n, N + 1), U=dview["U"].view(B, m, N),
                       x_term=dview["x_term"].view(B, n), lamb=dview["lamb"], cost=dview["cost"],
                       obs=dview["obs"].view(B, 6) if obs_rec is not None else None,
                       iters=dint["iters"], status=dint["status"], K=None, k=None)
            pk = self._packs[key] = dict(host=host, dev=dev, h=hview, hi=hint, buf=buf,
                                         in_bytes=off["cost"][0], npdt=npdt)
        h = pk["h"]
        X = h["X"].reshape(B, n, N + 1)
        X[:] = 0
        X[:, :, 0] = np.asarray(x0, float)
        U = h["U"].reshape(B, m, N)
        U[:] = 0 if U0 is None else np.asarray(U0, float).reshape(B, m, N)
        h["x_term"].reshape(B, n)[:] = x_terms
        h["lamb"][:] = np.asarray(lamb0, float).reshape(B)
        if obs_rec is not None:
            h["obs"].reshape(B, 6)[:] = np.asarray(obs_rec, float)
        nb = pk["in_bytes"]
        pk["dev"][:nb].copy_(pk["host"][:nb], non_blocking=True)
        solver.solve(pk["buf"])
        pk["host"].copy_(pk["dev"], non_blocking=True)
        torch.cuda.current_stream(solver.device).synchronize()
        f = lambda a: np.array(a, dtype=np.float64)
        return dict(U=f(U), X=f(X), lamb=f(h["lamb"]), cost=f(h["cost"]),
                    iters=np.array(pk["hi"]["iters"]), status=np.array(pk["hi"]["status"]))

    def solve_chained(self, cfg, x0, x_terms_by_chain, lamb0, obs_rec):
        """The reference's chained regularisation on the DEVICE (round 6): inside one lap's
        candidate list the final lamb of candidate c seeds candidate c + 1 (utils/base.py:393,
        :414-426 — `lamb` is reset per lap and carried through the `for j` loop), the laps are
        independent.  x_terms_by_chain: one array [k_a, n] per lap (their lengths may differ).
        All candidates go up in ONE host-to-device copy and come back in ONE.  Laps of equal
        length (the rule): ONE launch, i2lqr_solve_chained — a workgroup of the speculative kernel
        solves its chains' candidates one after the other, lamb carried in a register.  Otherwise
        (ragged laps; configurations the chain kernel is not built for): step c of every chain is
        one launch (i2lqr_solve over the chains that still have a candidate c), its lamb input
        gathered on the device from step c - 1's output, the launches replayed as a hipGraph.
        Either way ONE synchronisation per round — against a host round trip per step (the
        controller's loop before: 24 per control step on BASELINE configs[0]).  Same kernel code on
        the same inputs: bit-identical to the step-by-step form.
        Returns one dict(U, X, lamb, cost, iters, status) per chain (host arrays, leading axis k_a)."""
        import torch
        chains = [np.atleast_2d(np.asarray(xt, float)) for xt in x_terms_by_chain]
        widths = [len(c) for c in chains]
        width = max(widths)
        rows = [[a for a, w in enumerate(widths) if c < w] for c in range(width)]
        counts = [len(r) for r in rows]
        B = sum(counts)
        solver = self._solver(cfg, max(counts), early_exit=True)
        if solver.batch_minor or solver.batch_tiled or B > 4096:
            raise ValueError("solve_chained is the controller's path: problem-major batches")
        n, m, N = cfg.n, cfg.m, cfg.N
        has_obs = obs_rec is not None
        key = ("chain", id(solver), tuple(widths), has_obs)
        if not hasattr(self, "_packs"):
            self._packs = {}
        pk = self._packs.get(key)
        if pk is None:
            if len(self._packs) > 16:
                self._packs.clear()
            item = 8 if solver.dtype == torch.float64 else 4
            sizes = [("X", B * n * (N + 1)), ("U", B * m * N), ("x_term", B * n), ("lamb", B),
                     ("obs", B * 6), ("cost", B)]
            off, o = {}, 0
            for name, cnt in sizes:
                off[name] = (o, cnt)
                o += (cnt * item + 15) // 16 * 16
            ioff = o
            istep = (B * 4 + 15) // 16 * 16
            o += 2 * istep
            host = torch.empty(o, dtype=torch.uint8).pin_memory()
            dev = torch.empty(o, dtype=torch.uint8, device=solver.device)
            hv = {k: host[a:a + c * item].view(solver.dtype).numpy() for k, (a, c) in off.items()}
            dv = {k: dev[a:a + c * item].view(solver.dtype) for k, (a, c) in off.items()}
            hi = {"iters": host[ioff:ioff + B * 4].view(torch.int32).numpy(),
                  "status": host[ioff + istep:ioff + istep + B * 4].view(torch.int32).numpy()}
            di = {"iters": dev[ioff:ioff + B * 4].view(torch.int32),
                  "status": dev[ioff + istep:ioff + istep + B * 4].view(torch.int32)}
            # one buffer dict per step: contiguous views of the step's problems (step-major order)
            starts = np.concatenate([[0], np.cumsum(counts)]).astype(int)
            steps = []
            for c in range(width):
                lo, hi_ = int(starts[c]), int(starts[c + 1])
                steps.append(dict(
                    X=dv["X"].view(B, n, N + 1)[lo:hi_], U=dv["U"].view(B, m, N)[lo:hi_],
                    x_term=dv["x_term"].view(B, n)[lo:hi_], lamb=dv["lamb"][lo:hi_],
                    cost=dv["cost"][lo:hi_], obs=dv["obs"].view(B, 6)[lo:hi_] if has_obs else None,
                    iters=di["iters"][lo:hi_], status=di["status"][lo:hi_], K=None, k=None))
            # where step c's chains sat in step c - 1 (chains only ever drop out: ordered subsets)
            gather = [None]
            for c in range(1, width):
                pos = [rows[c - 1].index(a) for a in rows[c]]
                gather.append(None if pos == list(range(len(rows[c - 1]))) else
                              torch.tensor(pos, dtype=torch.int64, device=solver.device))
            pk = self._packs[key] = dict(host=host, dev=dev, h=hv, hi=hi, steps=steps,
                                         gather=gather, in_bytes=off["cost"][0], starts=starts)
        h = pk["h"]
        X = h["X"].reshape(B, n, N + 1)
        X[:] = 0
        X[:, :, 0] = np.asarray(x0, float)
        h["U"][:] = 0
        xt = h["x_term"].reshape(B, n)
        starts = pk["starts"]
        for c in range(width):
            for r, a in enumerate(rows[c]):
                xt[starts[c] + r] = chains[a][c]
        h["lamb"][:] = float(lamb0)  # (steps behind the first get theirs on the device)
        if has_obs:
            h["obs"].reshape(B, 6)[:] = np.asarray(obs_rec, float)
        nb = pk["in_bytes"]
        if pk.get("one_launch") is None:
            pk["one_launch"] = len(set(widths)) == 1 and getattr(self, "use_chain_kernel", True)
            if pk["one_launch"]:  # the steps' views are consecutive slices of ONE block: the whole of it
                first = pk["steps"][0]
                whole = {name: torch.as_strided(first[name], (B,) + tuple(first[name].shape[1:]),
                                                first[name].stride(), first[name].storage_offset())
                         for name in ("X", "U", "x_term", "lamb", "cost", "iters", "status")}
                whole["obs"] = (torch.as_strided(first["obs"], (B, 6), first["obs"].stride(),
                                                 first["obs"].storage_offset()) if has_obs else None)
                whole["K"] = whole["k"] = None
                pk["whole"] = whole
        if pk["one_launch"]:
            from ..solver import I2lqrError
            L, k = len(widths), widths[0]
            for a in range(L):  # chain-major order: candidate c of lap a at a * k + c
                xt[a * k:(a + 1) * k] = chains[a]
            pk["dev"][:nb].copy_(pk["host"][:nb], non_blocking=True)
            try:
                solver.solve_chained(pk["whole"], L, k)
            except I2lqrError:  # not built for this configuration: a launch per chain step below
                pk["one_launch"] = False
                for c in range(width):
                    for r, a in enumerate(rows[c]):
                        xt[starts[c] + r] = chains[a][c]
            else:
                pk["host"].copy_(pk["dev"], non_blocking=True)
                torch.cuda.current_stream(solver.device).synchronize()
                U = h["U"].reshape(B, m, N)
                out = []
                for a in range(L):
                    sel = slice(a * k, (a + 1) * k)
                    f = lambda arr: np.array(arr[sel], dtype=np.float64)
                    out.append(dict(U=f(U), X=f(X), lamb=f(h["lamb"]), cost=f(h["cost"]),
                                    iters=np.array(pk["hi"]["iters"][sel]),
                                    status=np.array(pk["hi"]["status"][sel])))
                self.chain_info = {"one_launch": True, "graph": False, "graph_error": None}
                return out

        def launches():  # the device side of the round: a launch per chain step, lamb carried over
            for c, buf in enumerate(pk["steps"]):
                if c > 0:
                    prev = pk["steps"][c - 1]["lamb"]
                    if pk["gather"][c] is None:
                        buf["lamb"].copy_(prev)
                    else:
                        torch.index_select(prev, 0, pk["gather"][c], out=buf["lamb"])
                solver.solve(buf)

        pk["dev"][:nb].copy_(pk["host"][:nb], non_blocking=True)
        if pk.get("graph") is not None:
            pk["graph"].replay()  # (the same launches, captured once: ~2 instead of ~10 us per node)
        else:
            launches()
        pk["host"].copy_(pk["dev"], non_blocking=True)
        torch.cuda.current_stream(solver.device).synchronize()
        if "graph" not in pk:  # first round on this buffer set: capture its launches for the next
            pk["graph"], pk["graph_error"] = None, None
            if getattr(self, "use_graph", True):
                try:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g):
                        launches()
                    pk["graph"] = g
                except RuntimeError as e:  # refused: the rounds stay eager, and say so
                    pk["graph_error"] = f"{type(e).__name__}: {e}"
                    torch.cuda.synchronize(solver.device)
            self.chain_info = {"one_launch": False, "graph": pk["graph"] is not None,
                               "graph_error": pk["graph_error"]}
        U = h["U"].reshape(B, m, N)
        out = []
        for a, w in enumerate(widths):
            sel = [int(starts[c]) + rows[c].index(a) for c in range(w)]
            f = lambda arr: np.array(arr[sel], dtype=np.float64)
            out.append(dict(U=f(U), X=f(X), lamb=f(h["lamb"]), cost=f(h["cost"]),
                            iters=np.array(pk["hi"]["iters"][sel]),
                            status=np.array(pk["hi"]["status"][sel])))
        return out

    def _solve_generic(self, solver, cfg, B, x0, x_terms, lamb0, obs_rec, U0):
        import torch
        buf = solver.alloc(B, want_gains=False)
        X = np.zeros((B, cfg.n, cfg.N + 1))
        X[:, :, 0] = np.asarray(x0, float)
        dev = lambda a: solver.to_native(torch.as_tensor(np.ascontiguousarray(a)).to(
            solver.device, solver.dtype))
        buf["X"].copy_(dev(X))
        if U0 is not None:
            buf["U"].copy_(dev(np.asarray(U0, float).reshape(B, cfg.m, cfg.N)))
        buf["x_term"].copy_(dev(x_terms))
        buf["lamb"].copy_(dev(np.asarray(lamb0, float).reshape(B)))
        if obs_rec is not None:
            buf["obs"] = dev(np.tile(np.asarray(obs_rec, float), (B, 1)))
        solver.solve(buf)
        host = lambda t: solver.to_problem_major(t).double().cpu().numpy()
        return dict(U=host(buf["U"]), X=host(buf["X"]), lamb=host(buf["lamb"]),
                    cost=host(buf["cost"]), iters=buf["iters"].cpu().numpy(),
                    status=buf["status"].cpu().numpy())

    def rollout(self, cfg, x0, U0):
        """Clipped inputs and their rollout (control/iterative_ilqr.py:32-48) for U0[B,m,N] from
        x0[n] (shared) or [B,n]; host arrays dict(U, X, cost)."""
        import torch
        solver = self._solver(cfg)
        U0 = np.asarray(U0, float)
        B = U0.shape[0]
        X = np.zeros((B, cfg.n, cfg.N + 1))
        X[:, :, 0] = np.asarray(x0, float)
        dev = lambda a: solver.to_native(torch.as_tensor(np.ascontiguousarray(a)).to(
            solver.device, solver.dtype))
        Xd, Ud = dev(X), dev(U0)
        # the terminal target only enters the returned cost
        cost = solver.rollout(Xd, Ud, dev(np.zeros((B, cfg.n))))
        host = lambda t: solver.to_problem_major(t).double().cpu().numpy()
        return dict(U=host(Ud), X=host(Xd), cost=host(cost))


_default_solver = None


def default_solver() -> HipCandidateSolver:
    global _default_solver
    if _default_solver is None:
        _default_solver = HipCandidateSolver()
    return _default_solver


def ilqr(ilqr_param, num_horizon, xtarget, timestep, obstacle, system_param, x_terminal, dX, uvar,
         xvar, lamb, solver=None):
    """Same contract as the reference's ilqr() (control/iterative_ilqr.py:7-85): x0 = xvar[:, 0],
    initial inputs uvar, regularisation lamb in; (uvar, xvar, lamb) out.

    Side effects on the caller's arrays, as in the reference: its first iteration clips `uvar` in
    place and writes the nominal rollout into `xvar[:, 1:]` in place (:33-42) before any accepted
    step rebinds the names to fresh arrays (:75-76).  If `uvar` / `xvar` are writable ndarrays they
    are left in exactly that state here too (the clipped initial inputs and their rollout, from
    i2lqr_rollout); the returned arrays are always new ones.  `dX[:, 1:]` is left as the returned
    trajectory's deviation from xtarget (the reference leaves the deviation of the last nominal
    it rolled out, which is the previous one after a final accepted step; no caller reads it:
    utils/base.py:409 re-creates dX per candidate)."""
    solver = default_solver() if solver is None else solver
    cfg = config_from_params(ilqr_param, system_param, num_horizon, timestep, xtarget)
    x0 = np.array(np.asarray(xvar, float)[:, 0])
    U0 = np.array(uvar, float)
    obs = None if obstacle is None else obstacle_record(obstacle)
    out = solver.solve(cfg, x0, np.asarray(x_terminal, float)[None], [float(lamb)], obs,
                       U0=U0[None])
    if int(ilqr_param.max_ilqr_iter) > 0 and isinstance(uvar, np.ndarray) and \
            isinstance(xvar, np.ndarray) and uvar.flags.writeable and xvar.flags.writeable:
        first = solver.rollout(cfg, x0, U0[None])
        uvar[...] = first["U"][0]
        xvar[:, 1:] = first["X"][0][:, 1:]
    uvar_new, xvar_new = out["U"][0], out["X"][0]
    if dX is not None:
        dX[:, 1:] = xvar_new[:, 1:] - np.asarray(xtarget, float).reshape(-1, 1)
    return uvar_new, xvar_new, float(out["lamb"][0])

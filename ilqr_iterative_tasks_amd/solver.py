"""BatchedILQR — Python host of the C-ABI (include/i2lqr.h) over PyTorch-ROCm device tensors.

PyTorch is plumbing only: it owns the HBM allocations and the HIP stream; every arithmetic step
runs in the hand-written HIP kernels of libi2lqr_hip.so.  There is no CPU path: constructing a
solver without a visible HIP device, or without the built extension, raises.

Tensor layouts (cfg.layout):
  problem-major (default; the reference's NumPy layout with a leading batch axis, time contiguous;
  control/iterative_ilqr.py:109-110, utils/base.py:405-409) — one problem per wavefront kernels:
    X[B, n, N+1]   U[B, m, N]   K[B, m, n, N]   k[B, m, N]   x_term[B, n]   lamb[B]   obs[B, 6]
  batch-minor (batch index fastest, time slowest) — one problem per lane kernels, large batches:
    X[N+1, n, B]   U[N, m, B]   K[N, m, n, B]   k[N, m, B]   x_term[n, B]   lamb[B]   obs[6, B]
  batch-tiled (the same inside tiles of 64 problems; B % 64 == 0) — same kernels:
    X[B/64, N+1, n, 64]   U[B/64, N, m, 64]   K[B/64, N, m, n, 64]   ...   lamb[B] (flat)
`to_native()` / `to_problem_major()` convert between the two.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _abi
from ._abi import I2lqrConfig, OBS_WORDS


class I2lqrError(RuntimeError):
    """Non-zero return code of the C-ABI.  A RuntimeError, so the reference's
    `try: calc_input() except RuntimeError` (utils/base.py:146-155) semantics still hold."""


class BatchedILQR:
    def __init__(self, cfg: I2lqrConfig, device: str | torch.device = "cuda:0", lib_path=None):
        self._handle = None  # set first so __del__ is safe if loading fails
        self.lib = _abi.load_library(lib_path)  # lib_path: A/B builds in tools/ only
        if not torch.cuda.is_available():
            raise I2lqrError("no HIP device visible: BatchedILQR has no CPU fallback")
        self.device = torch.device(device)
        self.cfg = cfg.copy()
        self.n, self.m, self.N = cfg.n, cfg.m, cfg.N
        self.dtype = torch.float64 if cfg.dtype == _abi.F64 else torch.float32
        handle = C.c_void_p()
        with torch.cuda.device(self.device):
            self._check(self.lib.i2lqr_create(C.byref(self.cfg), C.byref(handle)))
        self._handle = handle
        self._argmin_ws = None
        self.batch_minor = cfg.layout == _abi.LAYOUT_BATCH_MINOR
        self.batch_tiled = cfg.layout == _abi.LAYOUT_BATCH_TILED
        self._ws = None  # scratch of the batch-minor kernels (registered on the handle)

    # -- plumbing ---------------------------------------------------------------------------
    def _check(self, rc: int) -> None:
        if rc != 0:
            raise I2lqrError(f"i2lqr error {rc}: {self.lib.i2lqr_last_error().decode()}")

    def close(self) -> None:
        if getattr(self, "_handle", None) is not None:
            self.lib.i2lqr_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self) -> C.c_void_p:
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _ptr(self, t: torch.Tensor | None, shape, dtype=None, name="tensor") -> C.c_void_p:
        if t is None:
            return C.c_void_p(None)
        dtype = self.dtype if dtype is None else dtype
        if t.device != self.device and not (t.device.type == self.device.type and
                                            (t.device.index or 0) == (self.device.index or 0)):
            raise ValueError(f"{name}: expected device {self.device}, got {t.device}")
        if t.dtype != dtype:
            raise ValueError(f"{name}: expected dtype {dtype}, got {t.dtype}")
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
        if not t.is_contiguous():
            raise ValueError(f"{name}: must be contiguous")
        return C.c_void_p(t.data_ptr())

    # -- layout ---------------------------------------------------------------------------------
    def shape(self, name: str, B: int) -> tuple:
        n, m, N = self.n, self.m, self.N
        core = {"X": (n, N + 1), "U": (m, N), "K": (m, n, N), "k": (m, N), "x_term": (n,),
                "obs": (OBS_WORDS,), "lamb": (), "cost": (), "iters": (), "status": (),
                "qfun": (), "cost_it": ()}[name]
        if (self.batch_minor or self.batch_tiled) and len(core) >= 2:
            core = core[-1:] + core[:-1]  # the lane layouts are time-major
        if self.batch_tiled and core:
            if B % 64:
                raise ValueError(f"batch-tiled layout needs B % 64 == 0, got {B}")
            return (B // 64,) + core + (64,)
        return core + (B,) if self.batch_minor else (B,) + core

    def to_native(self, t: torch.Tensor) -> torch.Tensor:
        """problem-major [B, ...] tensor -> this solver's layout (contiguous)."""
        if t.dim() == 1 or not (self.batch_minor or self.batch_tiled):
            return t.contiguous()
        if t.dim() >= 3:  # trajectories and gains: time (the last axis) becomes the slowest
            t = t.movedim(-1, 1)
        if self.batch_tiled:
            if t.shape[0] % 64:
                raise ValueError(f"batch-tiled layout needs B % 64 == 0, got {t.shape[0]}")
            return t.reshape(t.shape[0] // 64, 64, *t.shape[1:]).movedim(1, -1).contiguous()
        return t.movedim(0, -1).contiguous()

    def to_problem_major(self, t: torch.Tensor) -> torch.Tensor:
        if t.dim() == 1 or not (self.batch_minor or self.batch_tiled):
            return t
        if self.batch_tiled:
            u = t.movedim(-1, 1)
            u = u.reshape(u.shape[0] * 64, *u.shape[2:])
        else:
            u = t.movedim(-1, 0)
        if u.dim() >= 3:
            u = u.movedim(1, -1)
        return u.contiguous()

    def batch_of(self, X: torch.Tensor) -> int:
        if self.batch_tiled:
            return X.shape[0] * 64
        return X.shape[-1] if self.batch_minor else X.shape[0]

    def ensure_workspace(self, B: int) -> None:
        need = int(self.lib.i2lqr_workspace_bytes(self._handle, B))
        if need == 0:
            return
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            self._check(self.lib.i2lqr_set_workspace(self._handle, C.c_void_p(self._ws.data_ptr()),
                                                     self._ws.numel()))

    def set_compaction(self, min_batch: int) -> None:
        """Threshold of the chunked, compacting form of solve() on the lane layouts
        (i2lqr_set_compaction in include/i2lqr.h): min_batch > 0 explicit, 0 never (single launch),
        < 0 automatic — the handle's default: chunked from 4096 problems when max_iter > 16, with
        the last <= 8192 survivors finished by the speculative eight-lane kernel ("wave_tail"; the
        one-problem-per-wavefront kernel from 2048 survivors where that is not built).
        The chunks alone are bit-identical to the single launch; with the tail the outputs
        agree to 1e-8 (fp64), not bit for bit: for bit-reproducibility against the single launch
        call set_compaction(0) or set_option("wave_tail", 0).  The chunked form always runs with
        in-place candidate states and stored nominal states ("defer_states" / "reroll_nominal"
        overrides apply to the single launch only)."""
        self._check(self.lib.i2lqr_set_compaction(self._handle, int(min_batch)))

    def set_option(self, name: str, value: int) -> None:
        """Scheduling options (i2lqr_set_option in include/i2lqr.h); -1 restores the automatic
        choice.  Lane layouts: "defer_states", "reroll_nominal", "lds_gain_steps", "wave_tail";
        problem-major layout: "group_lanes" (8 / 16 / 64), "speculate", "per_step_jacobians".  All
        but "wave_tail" and "group_lanes" leave the results bit-identical."""
        self._check(self.lib.i2lqr_set_option(self._handle, name.encode(), int(value)))

    def iterate_kernel(self, B: int) -> str:
        """Name of the kernel iterate() launches for B problems (rocprofv3 traces)."""
        return self.lib.i2lqr_iterate_kernel(self._handle, int(B)).decode()

    def solve_kernel(self, B: int) -> str:
        """Name of the (dominant) kernel solve() launches for B problems."""
        return self.lib.i2lqr_solve_kernel(self._handle, int(B)).decode()

    def empty(self, *shape, dtype=None) -> torch.Tensor:
        return torch.empty(*shape, dtype=self.dtype if dtype is None else dtype,
                           device=self.device)

    def alloc(self, B: int, want_gains: bool = True) -> dict:
        """Zero-initialised buffer set for B problems (U = 0, lamb = 1: utils/base.py:393, :405)."""
        z = lambda name, dtype=None: torch.zeros(
            self.shape(name, B), dtype=self.dtype if dtype is None else dtype, device=self.device)
        buf = dict(X=z("X"), U=z("U"), x_term=z("x_term"), lamb=z("lamb") + 1, cost=z("cost"),
                   iters=z("iters", torch.int32), status=z("status", torch.int32), obs=None)
        buf["K"] = z("K") if want_gains else None
        buf["k"] = z("k") if want_gains else None
        return buf

    @staticmethod
    def recommended_layout(cfg: I2lqrConfig, B: int, early_exit: bool = False, lib=None) -> int:
        """cfg.layout to create a solver with for batches of B problems (i2lqr_recommended_layout:
        the crossover between the problem-major latency kernels and the one-problem-per-lane
        throughput kernels is measured and lives behind the C-ABI)."""
        lib = _abi.load_library() if lib is None else lib
        rc = int(lib.i2lqr_recommended_layout(C.byref(cfg), int(B), 1 if early_exit else 0))
        if rc < 0:
            raise I2lqrError(f"i2lqr error {rc}: {lib.i2lqr_last_error().decode()}")
        return rc

    def set_initial_state(self, buf: dict, x0: torch.Tensor, lamb0: float | None = None,
                          zero_states: bool = True) -> None:
        """Candidates of one control round share the current state: X[:, :, 0] = x0, X elsewhere
        and U zero (utils/base.py:405-408), lamb = lamb0 (:393) — written in this solver's layout
        (fills and one strided copy: no arithmetic).  zero_states=False leaves X[:, :, 1:] as it
        is: every solver entry point rolls the states out from X[:, :, 0] and U before it reads
        them (control/iterative_ilqr.py:32-42), so a caller that reuses a buffer round after round
        saves the fill of the largest array."""
        X = buf["X"]
        if zero_states:
            X.zero_()
        buf["U"].zero_()
        x0 = x0.to(self.device, self.dtype)
        if self.batch_tiled:
            X[:, 0] = x0[None, :, None]      # [B/64, N+1, n, 64]
        elif self.batch_minor:
            X[0] = x0[:, None]               # [N+1, n, B]
        else:
            X[:, :, 0] = x0[None, :]         # [B, n, N+1]
        if lamb0 is not None:
            buf["lamb"].fill_(float(lamb0))

    def problem(self, buf: dict, idx) -> dict:
        """U[m, N] and X[n, N+1] of problem `idx` in the reference's orientation, whatever the
        layout.  `idx`: a device int64 tensor of one element (e.g. the pick) — gathered with
        index_select, so NO host synchronisation (indexing with a 0-dim device tensor would read
        it back) and the call can sit inside a captured graph — or a host int."""
        if isinstance(idx, torch.Tensor):
            i = idx.reshape(1).clamp(min=0).to(torch.int64)
        else:
            i = torch.tensor([max(int(idx), 0)], dtype=torch.int64, device=self.device)
        out = {}
        for key in ("U", "X"):
            t = buf[key]
            if self.batch_tiled:    # [B/64, T, c, 64]: tile i // 64, lane i % 64
                tile = t.index_select(0, torch.div(i, 64, rounding_mode="floor"))[0]
                out[key] = tile.index_select(2, i % 64)[:, :, 0].transpose(0, 1).contiguous()
            elif self.batch_minor:  # [T, c, B]
                out[key] = t.index_select(2, i)[:, :, 0].transpose(0, 1).contiguous()
            else:                   # [B, c, T]
                out[key] = t.index_select(0, i)[0].clone()
        return out

    # -- the path ---------------------------------------------------------------------------
    def rollout(self, X, U, x_term, cost=None):
        """control/iterative_ilqr.py:32-48.  X[:, :, 0] = x0; U is clipped in place."""
        B, sh = self.batch_of(X), self.shape
        cost = self.empty(B) if cost is None else cost
        self.ensure_workspace(B)
        with torch.cuda.device(self.device):
            self._check(self.lib.i2lqr_rollout(
                self._handle, B, self._ptr(X, sh("X", B), name="X"),
                self._ptr(U, sh("U", B), name="U"), self._ptr(x_term, sh("x_term", B), name="x_term"),
                self._ptr(cost, (B,), name="cost"), self._stream()))
        return cost

    def backward(self, X, U, x_term, lamb, obs=None, K=None, k=None):
        """control/iterative_ilqr.py:88-130.  Returns (k[B,m,N], K[B,m,n,N])."""
        B, sh = self.batch_of(X), self.shape
        K = self.empty(*sh("K", B)) if K is None else K
        k = self.empty(*sh("k", B)) if k is None else k
        self.ensure_workspace(B)
        with torch.cuda.device(self.device):
            self._check(self.lib.i2lqr_backward(
                self._handle, B, self._ptr(X, sh("X", B), name="X"),
                self._ptr(U, sh("U", B), name="U"), self._ptr(x_term, sh("x_term", B), name="x_term"),
                self._ptr(lamb, (B,), name="lamb"), self._ptr(obs, sh("obs", B), name="obs"),
                self._ptr(K, sh("K", B), name="K"), self._ptr(k, sh("k", B), name="k"),
                self._stream()))
        return k, K

    def forward(self, X, U, x_term, K, k, X_new=None, U_new=None, cost_new=None):
        """control/iterative_ilqr.py:133-160.  Returns (X_new, U_new, cost_new)."""
        B, sh = self.batch_of(X), self.shape
        X_new = self.empty(*sh("X", B)) if X_new is None else X_new
        U_new = self.empty(*sh("U", B)) if U_new is None else U_new
        cost_new = self.empty(B) if cost_new is None else cost_new
        self.ensure_workspace(B)
        with torch.cuda.device(self.device):
            self._check(self.lib.i2lqr_forward(
                self._handle, B, self._ptr(X, sh("X", B), name="X"),
                self._ptr(U, sh("U", B), name="U"), self._ptr(x_term, sh("x_term", B), name="x_term"),
                self._ptr(K, sh("K", B), name="K"), self._ptr(k, sh("k", B), name="k"),
                self._ptr(X_new, sh("X", B), name="X_new"),
                self._ptr(U_new, sh("U", B), name="U_new"),
                self._ptr(cost_new, (B,), name="cost_new"), self._stream()))
        return X_new, U_new, cost_new

    def _iter_args(self, buf, B):
        sh = self.shape
        self.ensure_workspace(B)
        return (self._ptr(buf["X"], sh("X", B), name="X"),
                self._ptr(buf["U"], sh("U", B), name="U"),
                self._ptr(buf["x_term"], sh("x_term", B), name="x_term"),
                self._ptr(buf["lamb"], (B,), name="lamb"),
                self._ptr(buf.get("obs"), sh("obs", B), name="obs"),
                self._ptr(buf["cost"], (B,), name="cost"),
                self._ptr(buf.get("K"), sh("K", B), name="K"),
                self._ptr(buf.get("k"), sh("k", B), name="k"),
                self._ptr(buf.get("iters"), (B,), torch.int32, name="iters"),
                self._ptr(buf.get("status"), (B,), torch.int32, name="status"))

    def iterate(self, buf: dict, n_iters: int) -> dict:
        """`n_iters` fused iLQR iterations per problem without early exits (the throughput unit);
        in place on buf['X'], buf['U'], buf['lamb'].  control/iterative_ilqr.py:29-84."""
        B = self.batch_of(buf["X"])
        with torch.cuda.device(self.device):
            self._check(self.lib.i2lqr_iterate(self._handle, B, int(n_iters),
                                               *self._iter_args(buf, B), self._stream()))
        return buf

    def solve(self, buf: dict) -> dict:
        """ilqr() to termination for every problem (control/iterative_ilqr.py:7-85), in place."""
        B = self.batch_of(buf["X"])
        with torch.cuda.device(self.device):
            self._check(self.lib.i2lqr_solve(self._handle, B, *self._iter_args(buf, B),
                                             self._stream()))
        return buf

    def solve_chained(self, buf: dict, chains: int, chain_len: int) -> dict:
        """i2lqr_solve_chained: `chains` chains of `chain_len` problems each (problem-major buffers of
        chains * chain_len problems, chain after chain) solved in ONE launch, the final lamb of a
        chain's problem c seeding its problem c + 1 (utils/base.py:393, :414-426).  Raises I2lqrError
        (I2LQR_ERR_UNSUPPORTED) where the chain kernel is not built."""
        B = int(chains) * int(chain_len)
        if self.batch_of(buf["X"]) != B:
            raise ValueError(f"buffers hold {self.batch_of(buf['X'])} problems, {chains} chains of "
                             f"{chain_len} need {B}")
        with torch.cuda.device(self.device):
            self._check(self.lib.i2lqr_solve_chained(self._handle, int(chains), int(chain_len),
                                                     *self._iter_args(buf, B), self._stream()))
        return buf

    def relax_cost(self, X, x_term, qfun, outer_iter: int, max_relax_iter: int = 55,
                   cost_it=None):
        """utils/base.py:427-437 for every candidate."""
        B = self.batch_of(X)
        cost_it = self.empty(B) if cost_it is None else cost_it
        with torch.cuda.device(self.device):
            self._check(self.lib.i2lqr_relax_cost(
                self._handle, B, self._ptr(X, self.shape("X", B), name="X"),
                self._ptr(x_term, self.shape("x_term", B), name="x_term"),
                self._ptr(qfun, (B,), torch.int32, name="qfun"), int(outer_iter),
                int(max_relax_iter), self._ptr(cost_it, (B,), name="cost_it"), self._stream()))
        return cost_it

    # -- controller round on the device (problem-major layout) -----------------------------------
    def select_candidates(self, ss, T, qfun, x_guess, guess_stride: int, k: int, idx, x_term, qf):
        """utils/base.py:332-341 + :411-412 for L laps: ss[L,n,Tmax], T[L] int32, qfun[L,Tmax]
        int32; x_guess element i at x_guess.data_ptr() + i*guess_stride.  Fills idx[L,k] int32,
        x_term[L*k,n], qf[L*k] int32."""
        L, n, Tmax = ss.shape
        with torch.cuda.device(self.device):
            self._check(self.lib.i2lqr_select_candidates(
                self._handle, L, Tmax, self._ptr(ss, (L, self.n, Tmax), name="ss"),
                self._ptr(T, (L,), torch.int32, name="T"),
                self._ptr(qfun, (L, Tmax), torch.int32, name="qfun"),
                C.c_void_p(x_guess.data_ptr()), int(guess_stride), int(k),
                self._ptr(idx, (L, k), torch.int32, name="idx"),
                self._ptr(x_term, (L * k, self.n), name="x_term"),
                self._ptr(qf, (L * k,), torch.int32, name="qf"), self._stream()))

    def init_candidates(self, x0, lamb0: float, buf: dict):
        """uvar = 0, xvar[:, 0] = x0, lamb = lamb0 (utils/base.py:393, :405-408) for every problem
        of `buf`."""
        B = buf["X"].shape[0]
        with torch.cuda.device(self.device):
            self._check(self.lib.i2lqr_init_candidates(
                self._handle, B, self._ptr(x0, (self.n,), name="x0"), float(lamb0),
                self._ptr(buf["X"], self.shape("X", B), name="X"),
                self._ptr(buf["U"], self.shape("U", B), name="U"),
                self._ptr(buf["lamb"], (B,), name="lamb"), self._stream()))

    def pick_best(self, L: int, k: int, cost_it, X, U, best, x_pred, u_pred):
        """utils/base.py:462-469 on the device; best[2] int32, x_pred[n,N+1], u_pred[m,N]."""
        B = L * k
        with torch.cuda.device(self.device):
            self._check(self.lib.i2lqr_pick_best(
                self._handle, L, k, self._ptr(cost_it, (B,), name="cost_it"),
                self._ptr(X, (B, self.n, self.N + 1), name="X"),
                self._ptr(U, (B, self.m, self.N), name="U"),
                self._ptr(best, (2,), torch.int32, name="best"),
                self._ptr(x_pred, (self.n, self.N + 1), name="x_pred"),
                self._ptr(u_pred, (self.m, self.N), name="u_pred"), self._stream()))

    def pick_index(self, L: int, k: int, cost_it, best=None):
        """The pick of utils/base.py:462-465 alone (list-of-lists order over L laps of k candidates,
        then the first minimum of that lap) on a device vector cost_it[L * k], e.g. the gathered
        costs of a sharded round; returns best[2] int32 (device) = (lap position, candidate
        position).  i2lqr_pick_best without the trajectory gather: any layout."""
        best = self.empty(2, dtype=torch.int32) if best is None else best
        with torch.cuda.device(self.device):
            self._check(self.lib.i2lqr_pick_best(
                self._handle, int(L), int(k), self._ptr(cost_it, (L * k,), name="cost_it"),
                C.c_void_p(None), C.c_void_p(None), self._ptr(best, (2,), torch.int32, name="best"),
                C.c_void_p(None), C.c_void_p(None), self._stream()))
        return best

    def pack_problem(self, buf: dict, idx: torch.Tensor, pack: torch.Tensor | None = None):
        """pack[m N + n (N+1)] = (U, X) of problem idx (device int64[1]) in the reference's
        orientation (i2lqr_pack_problem: one launch, any layout, no host synchronisation)."""
        B = self.batch_of(buf["X"])
        P = self.m * self.N + self.n * (self.N + 1)
        pack = self.empty(P) if pack is None else pack
        with torch.cuda.device(self.device):
            self._check(self.lib.i2lqr_pack_problem(
                self._handle, B, self._ptr(buf["X"], self.shape("X", B), name="X"),
                self._ptr(buf["U"], self.shape("U", B), name="U"),
                self._ptr(idx.reshape(1), (1,), torch.int64, name="idx"),
                self._ptr(pack, (P,), name="pack"), self._stream()))
        return pack

    def unpack(self, pack: torch.Tensor) -> tuple:
        """(U[m, N], X[n, N+1]) views of a pack."""
        nu = self.m * self.N
        return pack[:nu].view(self.m, self.N), pack[nu:].view(self.n, self.N + 1)

    def round_winner(self, world: int, width: int, total: int, best_padded, pack_all,
                     winner=None, best_global=None):
        """i2lqr_round_winner: the owner's pack and (index in the unpadded batch, owner rank) from
        the flat arg-min over the world x width gathered costs; device tensors, one launch."""
        P = pack_all.numel() // world
        winner = self.empty(P) if winner is None else winner
        best_global = self.empty(2, dtype=torch.int64) if best_global is None else best_global
        with torch.cuda.device(self.device):
            self._check(self.lib.i2lqr_round_winner(
                self._handle, int(world), int(width), int(total), P,
                self._ptr(best_padded.reshape(1), (1,), torch.int64, name="best_padded"),
                self._ptr(pack_all.reshape(world, P), (world, P), name="pack_all"),
                self._ptr(winner, (P,), name="winner"),
                self._ptr(best_global, (2,), torch.int64, name="best_global"), self._stream()))
        return winner, best_global

    # -- one sharded round as ONE C-ABI call (i2lqr_sharded_round_flat) -------------------------
    def round_buffers(self, B: int, total: int, world: int, bufs: dict | None = None) -> dict:
        """Exchange buffers of one sharded round in flight (see plan_round): the local winner's
        pack, the padded cost vector of a ragged split, the gathered costs and packs, the two pick
        workspaces and the outputs.  `bufs`: a partly filled dict, completed in place."""
        P = self.m * self.N + self.n * (self.N + 1)
        width = (total + world - 1) // world
        z = lambda *shape, dtype=None: torch.zeros(*shape, dtype=dtype or self.dtype,
                                                   device=self.device)
        ws = lambda nbytes: torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        make = dict(
            pack_local=lambda: z(P), cost_all=lambda: z(world * width), pack_all=lambda: z(world, P),
            winner=lambda: z(P), best_global=lambda: z(2, dtype=torch.int64),
            best_cost=lambda: z(1), local_best=lambda: z(1, dtype=torch.int64),
            local_best_cost=lambda: z(1),
            pick_ws=lambda: ws(self.lib.i2lqr_argmin_workspace_bytes(max(B, 1))),
            side_ws=lambda: ws(self.lib.i2lqr_argmin_workspace_bytes(world * width) + 16))
        if B != width:
            make["padded"] = lambda: z(width)
        bufs = {} if bufs is None else bufs
        for key, fn in make.items():
            if bufs.get(key) is None:
                bufs[key] = fn()
        return bufs

    def plan_round(self, buf: dict | None, qfun, cost_it, total: int, world: int, rank: int,
                   n_iters: int | None, outer_iter: int = 0, max_relax_iter: int = 55,
                   bufs: dict | None = None, guard_previous: bool = True,
                   loopback: bool = False) -> dict:
        """The argument block of i2lqr_sharded_round_flat for this rank's shard `buf` (None: a rank
        without candidates) — built once for a buffer set, enqueued with round_flat() as often as
        the round is run on it.  Returns the plan: the ctypes struct, the tensors it points to (kept
        alive) and the round's outputs `best_global` int64[2] = (index, owner), `best_cost`,
        `winner` [P], `cost_all` [world x width]."""
        B = self.batch_of(buf["X"]) if buf is not None else 0
        if buf is not None:
            self.ensure_workspace(B)
        bufs = self.round_buffers(B, total, world, bufs)
        width = (total + world - 1) // world
        P = self.m * self.N + self.n * (self.N + 1)
        r = _abi.I2lqrRound()
        r.struct_size = C.sizeof(_abi.I2lqrRound)
        r.n_iters = -1 if n_iters is None else int(n_iters)
        r.B, r.total, r.world, r.rank = B, int(total), int(world), int(rank)
        r.outer_iter, r.max_relax_iter = int(outer_iter), int(max_relax_iter)
        r.guard_previous = 1 if guard_previous else 0
        r.loopback = 1 if loopback else 0  # (tests: one process plays the ranks one after the other)
        val = lambda p: p.value  # c_void_p -> int or None
        if buf is not None:
            X, U, xt, lamb, obs, cost, K, k, iters, status = (val(p) for p in self._iter_args(buf, B))
            r.X, r.U, r.x_term, r.lamb, r.obs, r.cost = X, U, xt, lamb, obs, cost
            r.K, r.k, r.iters, r.status = K, k, iters, status
            r.qfun = val(self._ptr(qfun, (B,), torch.int32, name="qfun"))
            r.cost_it = val(self._ptr(cost_it, (B,), name="cost_it"))
        r.local_best = bufs["local_best"].data_ptr()
        r.local_best_cost = bufs["local_best_cost"].data_ptr()
        r.pick_ws, r.pick_ws_bytes = bufs["pick_ws"].data_ptr(), bufs["pick_ws"].numel()
        r.pack_local = val(self._ptr(bufs["pack_local"], (P,), name="pack_local"))
        if bufs.get("padded") is not None:
            r.cost_padded = val(self._ptr(bufs["padded"], (width,), name="padded"))
        r.cost_all = val(self._ptr(bufs["cost_all"], (world * width,), name="cost_all"))
        r.pack_all = val(self._ptr(bufs["pack_all"].reshape(world, P), (world, P), name="pack_all"))
        r.side_ws, r.side_ws_bytes = bufs["side_ws"].data_ptr(), bufs["side_ws"].numel()
        r.best_cost = val(self._ptr(bufs["best_cost"], (1,), name="best_cost"))
        r.winner = val(self._ptr(bufs["winner"], (P,), name="winner"))
        r.best_global = val(self._ptr(bufs["best_global"], (2,), torch.int64, name="best_global"))
        return dict(round=r, ref=C.byref(r), solver=self, keep=(buf, qfun, cost_it, bufs), bufs=bufs,
                    width=width,
                    best_global=bufs["best_global"], best_cost=bufs["best_cost"],
                    winner=bufs["winner"], cost_all=bufs["cost_all"], cost_local=cost_it)

    def round_flat(self, plan: dict, comm=None, side_stream=None, device_is_current: bool = False):
        """Enqueue ONE sharded round (i2lqr_sharded_round_flat): the shard's solve + relaxed costs +
        local pick on the current stream; the local winner's pack, the grouped all-gather of costs
        and packs over `comm` (an RCCL communicator as c_void_p, None: a world of one) and the pick
        + hand-off on `side_stream` (None: the current stream too).  The outputs are the tensors of
        the plan.  device_is_current: skip the device guard (hot loops that set the device once)."""
        main = torch.cuda.current_stream(self.device).cuda_stream
        side = C.c_void_p(side_stream.cuda_stream) if side_stream is not None else C.c_void_p(None)
        if device_is_current:
            rc = self.lib.i2lqr_sharded_round_flat(self._handle, comm, plan["ref"], side,
                                                   C.c_void_p(main))
        else:
            with torch.cuda.device(self.device):
                rc = self.lib.i2lqr_sharded_round_flat(self._handle, comm, plan["ref"], side,
                                                       C.c_void_p(main))
        self._check(rc)
        return plan

    def round_pick(self, world: int, width: int, total: int, cost_all, pack_all, best_cost=None,
                   winner=None, best_global=None, workspace=None):
        """i2lqr_round_pick: the pick over the world x width gathered costs and the owner's pack in
        one call (one launch up to 16384 gathered costs)."""
        P = pack_all.numel() // world
        best_cost = self.empty(1) if best_cost is None else best_cost
        winner = self.empty(P) if winner is None else winner
        best_global = self.empty(2, dtype=torch.int64) if best_global is None else best_global
        if workspace is None:
            workspace = torch.empty(int(self.lib.i2lqr_argmin_workspace_bytes(world * width)) + 16,
                                    dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            self._check(self.lib.i2lqr_round_pick(
                self._handle, int(world), int(width), int(total), P,
                self._ptr(cost_all, (world * width,), name="cost_all"),
                self._ptr(pack_all.reshape(world, P), (world, P), name="pack_all"),
                self._ptr(best_cost, (1,), name="best_cost"), self._ptr(winner, (P,), name="winner"),
                self._ptr(best_global, (2,), torch.int64, name="best_global"),
                C.c_void_p(workspace.data_ptr()), C.c_int64(workspace.numel()), self._stream()))
        return best_cost, winner, best_global

    def _argmin_workspace(self, B: int, side: bool = False) -> tuple:
        """(pointer, bytes) of the pick's device scratch, grown to i2lqr_argmin_workspace_bytes(B);
        the size travels with the pointer and the library refuses a workspace that is too small.
        side: a SECOND workspace, for picks enqueued on another stream than the solves' (the pick
        on the gathered costs of a sharded round runs beside the next round's local pick: two
        picks in flight must not share their partial minima)."""
        need = int(self.lib.i2lqr_argmin_workspace_bytes(B))
        name = "_argmin_ws_side" if side else "_argmin_ws"
        ws = getattr(self, name, None)
        if ws is None or ws.numel() < need:
            ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            setattr(self, name, ws)
        return C.c_void_p(ws.data_ptr()), C.c_int64(ws.numel())

    def iterate_pick(self, buf: dict, n_iters: int, qfun, outer_iter: int, max_relax_iter: int = 55,
                     cost_it=None, pick: bool = True, best=None):
        """One control round in one call (i2lqr_iterate_pick): iterate(buf, n_iters), then
        relax_cost on the returned X, then (pick) the flat arg-min — a single launch on the
        eight-lane kernels, the same three steps as launches elsewhere; outputs bit-identical to
        the separate calls.  Returns (cost_it, (best_idx int64[1], best_cost[1]) or None);
        `best`: preallocated (idx, val) pair to write into."""
        B = self.batch_of(buf["X"])
        cost_it = self.empty(B) if cost_it is None else cost_it
        idx = val = None
        if pick:
            idx, val = best if best is not None else (self.empty(1, dtype=torch.int64), self.empty(1))
        with torch.cuda.device(self.device):
            self._check(self.lib.i2lqr_iterate_pick(
                self._handle, B, int(n_iters), *self._iter_args(buf, B),
                self._ptr(qfun, (B,), torch.int32, name="qfun"), int(outer_iter),
                int(max_relax_iter), self._ptr(cost_it, (B,), name="cost_it"),
                C.c_void_p(idx.data_ptr()) if pick else C.c_void_p(None),
                C.c_void_p(val.data_ptr()) if pick else C.c_void_p(None),
                *(self._argmin_workspace(B) if pick else (C.c_void_p(None), C.c_int64(0))),
                self._stream()))
        return cost_it, ((idx, val) if pick else None)

    def argmin(self, cost_it, side: bool = False):
        """Flat arg-min with first-index tie-break.  Returns (best_idx int64[1], best_cost[1]).
        side=True: on the second workspace (see _argmin_workspace)."""
        B = cost_it.shape[0]
        idx = self.empty(1, dtype=torch.int64)
        val = self.empty(1)
        with torch.cuda.device(self.device):
            self._check(self.lib.i2lqr_argmin(
                self._handle, B, self._ptr(cost_it, (B,), name="cost_it"),
                C.c_void_p(idx.data_ptr()), C.c_void_p(val.data_ptr()),
                *self._argmin_workspace(B, side), self._stream()))
        return idx, val

"""Row-sharded Gramian MVM across the GPUs of one node.

Output rows are independent (src/gramian.jl:81: `@threads for i in 1:n`), so rank g of P owns the rows
[lo_g, hi_g) of G, keeps ALL column points y and the weight vector a replicated, computes its shard of b
with the single-GPU kernel, and ONE all-gather (RCCL over xGMI; torch.distributed backend "nccl") makes b
complete on every rank — which is exactly the replicated `a` the next Krylov iteration needs.  There is
no other exchange.  Shards are padded to ceil(n/P) rows so the collective is a single
all_gather_into_tensor of equal pieces.

Symmetric form (gramian(k, x), vectors, where a symmetric kernel of the library applies — fp32 on the matrix cores, fp64 on the
direct-difference kernel): the n(n+1)/2
unordered pairs are independent too, so rank g evaluates the upper-triangle tiles of the 256-row panels p ≡ g (mod P) — a
cyclic assignment that gives every rank the same share of the triangle — and produces the partial product of those entries
and their mirror images (covgram_mvm_sym_partial); ONE all-reduce (sum) completes b on every rank.  Half the kernel
evaluations of the row-sharded form for one n-vector all-reduce instead of the all-gather.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


# The symmetric form pays from ~1e9 evaluated pairs per rank (tools/sym_shard_probe.py, n = 131072 on one GPU emulating rank r of
# P: 636 vs 820 us per rank at P = 2, 333 vs 427 at P = 4, 173-199 vs 226 at P = 8; below that its 8-wave workgroups leave the
# chip under-filled) — its all-reduce moves n scalars per rank where the all-gather moves n / P, a few microseconds apart at these sizes.
SYM_MIN_PAIRS_PER_RANK = 1.0e9
# fp64 (direct-difference symmetric kernel, 64-row blocks): a pair costs ~10x the fp32 matrix-core pair, so the form pays from fewer of them
# (one GPU: from n = 8192, tools/fp64_sym_sweep.py)
SYM_MIN_PAIRS_PER_RANK_F64 = 2.5e8


def shard_bounds(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Rows [lo, hi) of rank `rank`: ceil(n/world)-sized contiguous shards (the last ones may be short/empty)."""
    per = (n + world - 1) // world
    lo = min(n, rank * per)
    hi = min(n, lo + per)
    return lo, hi


def _host_staged(group) -> bool:
    """gloo has no device collectives for every op: with that backend the ONE collective of the MVM is staged through host
    memory (two processes sharing a GPU in the tests, or a box without RCCL); under "nccl" (= RCCL) tensors go as they are."""
    try:
        return dist.get_backend(group) == "gloo"
    except Exception:
        return False


def _all_reduce_sum(t: torch.Tensor, group) -> None:
    if t.is_cuda and _host_staged(group):
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


def _all_gather_into(target: torch.Tensor, shard: torch.Tensor, group) -> None:
    if shard.is_cuda and _host_staged(group):
        hs = shard.cpu()
        ht = torch.empty(target.shape, dtype=target.dtype)
        dist.all_gather_into_tensor(ht, hs, group=group)
        target.copy_(ht)
    else:
        dist.all_gather_into_tensor(target, shard, group=group)


# ---- the collective behind the C ABI (include/covgram.h: covgram_comm_*; round 5) -----------------------------------------------------
# Under the "nccl" (= RCCL) backend on GPU tensors the ONE collective of the MVM runs inside libcovgram.so, on the ctx stream, right behind
# the kernels (covgram_mvm_sharded / covgram_mvm_sym_allreduce): torch.distributed only carries the 128-byte unique id once.  torch's own
# collective stays as the route for gloo (CPU tests, host-staged), for process groups other than the default one, and when
# COVGRAM_ABI_COLLECTIVE=0 asks for it (bench.py times both).
_ABI_COMM = {}          # device index -> (rank, world) of the communicator the ctx owns; None: creation failed on some rank, torch's route for good
_ABI_FAILED = []        # why (reprs), for the bench line


def abi_comm(device: torch.device, group=None) -> bool:
    """Make sure the library ctx of `device` owns an RCCL communicator over the default process group; False when this route does not apply."""
    if os.environ.get("COVGRAM_ABI_COLLECTIVE", "1") == "0" or device.type != "cuda" or not dist.is_initialized():
        return False
    if group is not None and group is not dist.group.WORLD:
        return False
    try:
        if dist.get_backend(group) != "nccl":
            return False
    except Exception:
        return False
    from . import _ffi
    from .gramian import get_ctx
    rank, world = dist.get_rank(), dist.get_world_size()
    ctx = get_ctx(device)
    if device.index in _ABI_COMM:
        if _ABI_COMM[device.index] is None:
            return False
        if _ABI_COMM[device.index] == (rank, world):
            return True
    lib = _ffi.lib()
    ident = [None]
    if rank == 0:
        buf = (C.c_ubyte * _ffi.COMM_ID_BYTES)()
        _ffi.check(lib.covgram_comm_unique_id(C.cast(buf, C.c_void_p), _ffi.COMM_ID_BYTES))
        ident[0] = bytes(buf)
    if world > 1:
        dist.broadcast_object_list(ident, src=0)             # the only thing torch.distributed moves for this route
    raw = (C.c_ubyte * _ffi.COMM_ID_BYTES).from_buffer_copy(ident[0])
    ok = 1
    try:
        _ffi.check(lib.covgram_comm_create(ctx.handle, C.cast(raw, C.c_void_p), rank, world))
    except Exception as e:                                   # e.g. an RCCL build the library cannot load: every rank must take the same route
        ok = 0
        _ABI_FAILED.append(repr(e))
    if world > 1:
        flag = torch.tensor([ok], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        agreed = int(flag.item())
    else:
        agreed = ok
    if not agreed:
        if ok:
            lib.covgram_comm_destroy(ctx.handle)
        _ABI_COMM[device.index] = None
        return False
    _ABI_COMM[device.index] = (rank, world)
    return True


class ShardedGramian:
    """Row shard of gramian(k, x, y) owned by this rank + the all-gather that completes b.

    `local_factory(k, x_rows, y)` builds the local operator; by default the device `gramian`.  (The CPU
    multi-process tests inject a factory so that the sharding + collective logic runs under gloo.)"""

    def __init__(self, k, x, y=None, group=None, local_factory: Optional[Callable] = None, block: Optional[int] = None,
                 sym_partial_factory: Optional[Callable] = None, symmetric: Optional[bool] = None):
        """block: entries per point of the flat vectors — 1 for scalar kernels, d for GradientKernel, d + 1 for
        ValueGradientKernel Gramians (inferred from the kernel when not given): rank g then owns the rows
        [lo_g * block, hi_g * block) of the flat output and the all-gather moves ceil(n/P) * block entries per rank."""
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.n = x.shape[0]
        self.per = (self.n + self.world - 1) // self.world
        self.lo, self.hi = shard_bounds(self.n, self.world, self.rank)
        y_full = x if y is None else y
        self.m = y_full.shape[0]
        local_factory_is_default = local_factory is None
        if local_factory is None:
            from .gramian import gramian as local_factory
        x_rows = x[self.lo:self.hi]
        if self.hi > self.lo:
            self.local = local_factory(k, x_rows, y_full)
        else:
            self.local = None
        if block is None:
            from . import kernels as _K
            d = x.shape[1] if x.dim() > 1 else 1
            block = d if isinstance(k, _K.GradientKernel) else (d + 1 if isinstance(k, _K.ValueGradientKernel) else 1)
        self.block = int(block)
        self.shape = (self.n * self.block, self.m * self.block)
        # exercise the collective even on one rank (used to validate the RCCL path on single-GPU boxes)
        self.force_collective = bool(int(__import__("os").environ.get("COVGRAM_FORCE_COLLECTIVE", "0"))) and dist.is_initialized()
        # symmetric form: `sym_partial(out, a, rank, world)` fills out with this rank's partial product.  By default the
        # device Gramian's own method when it says the symmetric kernel applies (the answer depends on k and x only, so all
        # ranks agree); the CPU multi-process tests inject a factory.  symmetric=False forces row shards.
        # optional split of a step into its two parts (bench.py's per-rank report): when `timing` is set, events on the current
        # stream bracket the local kernel(s) and the one collective; read with timing_ms()
        self.timing = False
        self._ev = []
        # the C-ABI route: the library's own kernels on GPU points, the default process group on RCCL
        self._abi = bool(local_factory_is_default and sym_partial_factory is None and x.is_cuda and (self.world > 1 or self.force_collective)
                         and abi_comm(x.device, group))
        self._abi_full = None
        self.sym_partial = None
        if symmetric is not False and y is None and self.block == 1 and (self.world > 1 or self.force_collective):
            if sym_partial_factory is not None:
                self.sym_partial = sym_partial_factory(k, x)
            elif local_factory_is_default and (symmetric is True or self.n * (self.n / 2.0) / self.world >=
                                               (SYM_MIN_PAIRS_PER_RANK_F64 if x.dtype == torch.float64 else SYM_MIN_PAIRS_PER_RANK)):
                full = local_factory(k, x)
                if hasattr(full, "sym_partial_supported") and full.sym_partial_supported(self.world):
                    self._full_op = full
                    self.sym_partial = full.sym_partial_
        self._abi_op = None
        if self._abi and self.block == 1:
            from .gramian import Gramian
            op = getattr(self, "_full_op", None) or local_factory(k, x, y)
            self._abi_op = op if isinstance(op, Gramian) else None

    # -- the C-ABI route --------------------------------------------------------------------------------------------------
    def _abi_gramian(self, k, x, y):
        if self._abi_full is None:
            from .gramian import Gramian
            self._abi_full = getattr(self, "_full_op", None) or Gramian(k, x, y)
        return self._abi_full

    def _abi_collective(self, kind: str, send: torch.Tensor, recv: Optional[torch.Tensor] = None):
        from . import _ffi
        from .gramian import get_ctx, _dtype_code
        ctx = get_ctx(send.device).bind_stream()
        lib = _ffi.lib()
        if kind == "gather":
            _ffi.check(lib.covgram_comm_all_gather(ctx, _ffi._P(send.data_ptr()), _ffi._P(recv.data_ptr()), send.numel(), _dtype_code(send.dtype)))
        else:
            _ffi.check(lib.covgram_comm_all_reduce_sum(ctx, _ffi._P(send.data_ptr()), send.numel(), _dtype_code(send.dtype)))

    def _abi_mul(self, y: torch.Tensor, a: torch.Tensor, alpha: float, beta: float) -> bool:
        """mul! in ONE library call — shard kernel + its collective enqueued on the ctx stream — for a vector on a scalar-kernel Gramian."""
        G = self._abi_op
        if G is None or a.dim() != 1 or self.block != 1 or not (y.is_contiguous() and y.is_cuda and y.dtype == G.dtype and y.shape == (self.n,)):
            return False
        from . import _ffi
        a_c = a.to(device=G.device, dtype=G.dtype).contiguous()
        lib = _ffi.lib()
        ctx = G._px.ctx.bind_stream()
        if self.sym_partial is not None:
            _ffi.check(lib.covgram_mvm_sym_allreduce(ctx, _ffi.kref(G._spec()), G._px.handle, _ffi._P(a_c.data_ptr()), _ffi._P(y.data_ptr()),
                                                     float(alpha), float(beta)))
        else:
            _ffi.check(lib.covgram_mvm_sharded(ctx, _ffi.kref(G._spec()), G._px.handle, G._py.handle, _ffi._P(a_c.data_ptr()), _ffi._P(y.data_ptr()),
                                               float(alpha), float(beta)))
        return True

    def _buffers(self, a: torch.Tensor):
        key = (tuple(a.shape[1:]), a.dtype, a.device)
        if getattr(self, "_buf_key", None) != key:
            tail = tuple(a.shape[1:])
            self._shard = torch.zeros((self.per * self.block,) + tail, dtype=a.dtype, device=a.device)
            self._full = torch.empty((self.per * self.world * self.block,) + tail, dtype=a.dtype, device=a.device)
            self._buf_key = key
        return self._shard, self._full

    def _mark(self, dev):
        """An event on the current stream (timing only).  The collective runs on the backend's own stream between two event
        waits of the current stream, so the interval between the marks around it is its full cost to this stream."""
        if not (self.timing and dev.type == "cuda"):
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def timing_ms(self, reset: bool = True):
        """(local_ms, collective_ms, steps) summed over the steps recorded since the last reset; synchronises."""
        if self._ev:
            torch.cuda.synchronize()
        loc = sum(a.elapsed_time(b) for a, b, _ in self._ev)
        col = sum(b.elapsed_time(c) for _, b, c in self._ev)
        n = len(self._ev)
        if reset:
            self._ev = []
        return loc, col, n

    def _local_into(self, view: torch.Tensor, a: torch.Tensor):
        from .gramian import LazyOperator
        if isinstance(self.local, LazyOperator):
            self.local.mul_(view, a, 1.0, 0.0)
        else:
            view.copy_(self.local(a))

    def matmul(self, a: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """b = G a, complete on every rank.  a: (m,) or (m, p), replicated.  Buffers are allocated once and reused
        (Krylov callers multiply with the same shapes every iteration)."""
        rows = (self.hi - self.lo) * self.block
        if self.world == 1 and not self.force_collective:
            if out is None:
                out = torch.empty((self.n * self.block,) + tuple(a.shape[1:]), dtype=a.dtype, device=a.device)
            if self.local is not None:
                self._local_into(out, a)
            return out
        if self._abi and not self.timing and a.dim() == 1 and self.block == 1:
            # the C-ABI route: kernel(s) + the one collective in ONE library call, all on the ctx stream
            res = out if (out is not None and out.is_contiguous()) else torch.empty(self.n, dtype=a.dtype, device=a.device)
            if self._abi_mul(res, a, 1.0, 0.0):
                if out is not None and res is not out:
                    out.copy_(res)
                    return out
                return res
        if self.sym_partial is not None and a.dim() == 1:
            # symmetric form: this rank's partial product, then ONE all-reduce (the only collective of the MVM)
            if out is None or not out.is_contiguous():
                res = torch.empty(self.n, dtype=a.dtype, device=a.device)
            else:
                res = out
            e0 = self._mark(a.device)
            self.sym_partial(res, a, self.rank, self.world)
            e1 = self._mark(a.device)
            if self._abi:
                self._abi_collective("reduce", res)
            else:
                _all_reduce_sum(res, self.group)
            e2 = self._mark(a.device)
            if e0 is not None:
                self._ev.append((e0, e1, e2))
            if out is not None and res is not out:
                out.copy_(res)
                return out
            return res
        shard, full = self._buffers(a)
        e0 = self._mark(a.device)
        if self.local is not None:
            self._local_into(shard[:rows], a)
        e1 = self._mark(a.device)
        exact = (self.per * self.world == self.n)
        target = out if (exact and out is not None and out.is_contiguous()) else full
        if self._abi and shard.is_contiguous() and target.is_contiguous():
            self._abi_collective("gather", shard, target)                  # the ONLY collective of the MVM, on the library's stream
        else:
            _all_gather_into(target, shard, self.group)
        e2 = self._mark(a.device)
        if e0 is not None:
            self._ev.append((e0, e1, e2))
        if target is out:
            return out
        b = full[: self.n * self.block]
        if out is not None:
            out.copy_(b)
            return out
        return b.clone()

    __matmul__ = matmul

    def mul_(self, y: torch.Tensor, a: torch.Tensor, alpha: float = 1.0, beta: float = 0.0) -> torch.Tensor:
        """mul!(y, G, a, α, β) with b complete on every rank: lets the Krylov callers (covgram.cg) run unchanged on a
        row-sharded Gramian — vectors are replicated, so their dot products need no further collective."""
        if self._abi and not self.timing and (self.world > 1 or self.force_collective) and self._abi_mul(y, a, alpha, beta):
            return y
        if alpha == 1.0 and beta == 0.0:
            return self.matmul(a, out=y)
        t = self.matmul(a)
        return y.copy_(alpha * t) if beta == 0 else y.mul_(beta).add_(t, alpha=alpha)

    @property
    def dtype(self):
        return self.local.dtype if hasattr(self.local, "dtype") else torch.float64

    @property
    def device(self):
        return self.local.device if hasattr(self.local, "device") else torch.device("cpu")

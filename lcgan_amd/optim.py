"""Multi-tensor Adam, gradient bucket all-reduce (the DDP replacement) and the descriptor tables they share.

reference: torch.optim.Adam worker.py:98-110 ; DistributedDataParallel(find_unused_parameters=True, broadcast_buffers=False)
worker.py:88-96.  One kernel launch (lcgan_multi_tensor) walks every parameter tensor; tensors whose .grad is None are
skipped exactly as torch.optim.Adam / DDP-with-unused-parameters skip them (their step counters do not advance).
"""
from __future__ import annotations

import math
from typing import Iterable, List, Optional, Sequence

import numpy as np
import torch
import torch.distributed as dist

from . import kernels as KM

MT_ADAM, MT_EMA, MT_PACK = 0, 1, 2
MT_CHUNK = 65536
_DESC = np.dtype([("p0", "<u8"), ("p1", "<u8"), ("p2", "<u8"), ("p3", "<u8"), ("n", "<i8"), ("f0", "<f4"), ("f1", "<f4")])
assert _DESC.itemsize == 48


class TensorTable:
    """Device-side table for lcgan_multi_tensor: 48-byte descriptors + (tensor, chunk) index arrays.
    rows: sequence of (t0, t1, t2, t3, f0, f1) with t* fp32 contiguous tensors of equal numel (or None).
    The chunk arrays depend only on the tensors' sizes and are uploaded once; `refresh()` re-uploads the descriptors (pointers of
    the second operand and the two per-row scalars change from step to step: Adam's gradients and bias corrections) through pinned
    memory, one small asynchronous copy."""

    def __init__(self, rows: Sequence[tuple], device):
        self.device = device
        ct, ci = [], []
        total = 0
        for i, row in enumerate(rows):
            n = row[0].numel()
            nch = (n + MT_CHUNK - 1) // MT_CHUNK
            ct += [i] * nch
            ci += list(range(nch))
            total += n
        self.total = total
        self.n_chunks = len(ct)
        self.chunk_tensor = self._upload(np.asarray(ct, dtype=np.int32))
        self.chunk_index = self._upload(np.asarray(ci, dtype=np.int32))
        self._arr = np.zeros(len(rows), dtype=_DESC)
        self.refresh(rows)

    def _upload(self, a: np.ndarray) -> torch.Tensor:
        t = torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).copy())
        if torch.device(self.device).type == "cuda":
            t = t.pin_memory()                    # (caching host allocator: the block is not reused before the copy below has run)
        t = t.to(self.device, non_blocking=True)
        return t.view(torch.int32) if a.dtype == np.int32 else t

    def refresh(self, rows: Sequence[tuple]) -> "TensorTable":
        self.entries = list(rows)            # keeps the tensors alive (and feeds the CPU emulation in tests)
        arr = self._arr
        for i, (t0, t1, t2, t3, f0, f1) in enumerate(rows):
            n = t0.numel()
            for t in (t0, t1, t2, t3):
                if t is not None:
                    assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == n
            arr[i] = (t0.data_ptr(), t1.data_ptr() if t1 is not None else 0, t2.data_ptr() if t2 is not None else 0,
                      t3.data_ptr() if t3 is not None else 0, n, f0, f1)
        self.descs = self._upload(arr)
        return self


class Adam:
    """torch.optim.Adam(params, lr, betas, eps) restricted to what worker.py:98-110 uses (no weight decay / amsgrad)."""

    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float, betas=(0.0, 0.99), eps: float = 1e-8, on_zero_grad=None):
        """on_zero_grad: called by zero_grad() -- DataParallel.reset_reduction of the network these parameters belong to, so that the
        start of a step also clears whatever an aborted step left behind in the gradient buckets"""
        self.on_zero_grad = on_zero_grad
        self.params: List[torch.nn.Parameter] = list(params)
        self.lr, self.beta1, self.beta2, self.eps = float(lr), float(betas[0]), float(betas[1]), float(eps)
        self.exp_avg = [torch.zeros_like(p, memory_format=torch.contiguous_format) for p in self.params]
        self.exp_avg_sq = [torch.zeros_like(p, memory_format=torch.contiguous_format) for p in self.params]
        self.steps = [0] * len(self.params)
        self._tables = {}                     # which parameters have a gradient (structural: a few sets per run) -> TensorTable

    def zero_grad(self, set_to_none: bool = True):
        if self.on_zero_grad is not None:
            self.on_zero_grad()
        for p in self.params:
            p.grad = None

    @torch.no_grad()
    def step(self):
        rows, key = [], []
        for i, p in enumerate(self.params):
            if p.grad is None:
                continue                                     # unused / frozen parameter: skipped like torch.optim.Adam does
            self.steps[i] += 1
            t = self.steps[i]
            bc1 = 1.0 - self.beta1 ** t
            bc2 = 1.0 - self.beta2 ** t
            g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
            rows.append((p.data, g, self.exp_avg[i], self.exp_avg_sq[i], self.lr / bc1, 1.0 / math.sqrt(bc2)))
            key.append(i)
        if rows:
            key = tuple(key)
            tab = self._tables.get(key)
            if tab is None:
                if len(self._tables) > 16:
                    self._tables.clear()
                tab = self._tables[key] = TensorTable(rows, self.params[0].device)
            else:
                tab.refresh(rows)                            # same chunk arrays; new gradient pointers + bias corrections
            KM.K.multi_tensor(tab, MT_ADAM, self.beta1, self.beta2, self.eps)
            from . import ops
            ops.invalidate_weights(p for p in self.params if p.grad is not None)   # changed behind torch's version counters

    def state_dict(self):
        return {"steps": list(self.steps), "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq}

    def load_state_dict(self, sd):
        self.steps = list(sd["steps"])
        for a, b in zip(self.exp_avg, sd["exp_avg"]):
            a.copy_(b)
        for a, b in zip(self.exp_avg_sq, sd["exp_avg_sq"]):
            a.copy_(b)


class _Bucket:
    __slots__ = ("params", "views", "flat", "expected", "fired", "launched", "tables")

    def __init__(self, params, views, flat):
        self.params, self.views, self.flat = params, views, flat
        self.expected, self.fired, self.launched, self.tables = 0, 0, False, {}


class DataParallel(torch.nn.Module):
    """Single-node data parallelism: one process per GPU, full replicas, gradient mean over ranks (what the reference gets
    from DistributedDataParallel, worker.py:88-96).  Exposes `.module` and the `module.`-prefixed state_dict the reference's
    checkpoints carry.

    Gradients are reduced in BUCKETS (DDP's default 25 MB, walked in reverse registration order = roughly the order in which
    autograd finishes them) from inside the backward pass: a post-accumulate hook on every parameter counts its bucket down, and the
    bucket whose last gradient has arrived is packed (one multi-tensor kernel, pre-scaled by 1/world, into its fixed slice of one flat
    fp32 buffer) and all-reduced asynchronously over RCCL while the backward pass goes on; `.grad` is re-pointed at the slice so Adam
    reads the reduced values in place.  `sync_gradients()` after the backward launches what is left (buckets that hold parameters the
    iteration did not use never count down to zero) and returns a handle on all reductions.  Parameters whose grad is None on this rank
    are treated as unused -- the sets are structural in LC-GAN (projection heads on odd iterations, frozen layers), hence identical on
    every rank; a bucket without any gradient is not reduced at all.

    Contract: ONE backward pass, then sync_gradients().  A gradient that arrives for a bucket already handed to the all-reduce (a
    second backward before the sync: gradient accumulation, or the other network's parameters left with requires_grad=True) would be
    added in place into a buffer a collective may still be reading and would never be reduced: the hook raises instead.
    `reset_reduction()` (wired to the optimiser's zero_grad()) drops whatever a step that died between backward and sync left behind.

    The ORDER of the collectives is the order in which autograd finishes the buckets; it is the same on every rank because every rank
    runs the same graph (same code, same shapes, same grad-None sets), not because anything enforces it.  `check=True` (or
    LCGAN_DDP_CHECK=1 in the environment) is the debug mode that verifies exactly that before any gradient collective is issued:
    nothing is launched from the hooks, and sync_gradients() first all-reduces (min and max in one call) the bitmap of parameters that
    received a gradient together with the order in which the buckets completed, and raises on EVERY rank -- naming the parameters --
    when the ranks disagree, where the unchecked path would hang in RCCL or silently mix buckets."""

    BUCKET_BYTES = 25 << 20

    def __init__(self, module: torch.nn.Module, process_group=None, broadcast: bool = True, check: Optional[bool] = None):
        super().__init__()
        import os
        self.module = module
        self._pg = process_group
        self._buckets: Optional[List[_Bucket]] = None
        self._works: list = []
        self._armed = False                    # counts of the running backward pass are initialised
        self._order: list = []                 # bucket indices in the order their last gradient arrived (this backward pass)
        self.check = bool(int(os.environ.get("LCGAN_DDP_CHECK", "0"))) if check is None else bool(check)
        if broadcast and self.world_size > 1:
            with torch.no_grad():
                for t in list(module.parameters()) + list(module.buffers()):     # DDP ctor broadcast of rank-0 state
                    dist.broadcast(t.data, src=0, group=self._pg)
        if self.world_size > 1:
            self._build_buckets()

    @property
    def world_size(self) -> int:
        return dist.get_world_size(self._pg) if dist.is_available() and dist.is_initialized() else 1

    def forward(self, *a, **kw):
        return self.module(*a, **kw)

    def __deepcopy__(self, memo):
        import copy
        new = DataParallel.__new__(DataParallel)
        torch.nn.Module.__init__(new)
        new.module = copy.deepcopy(self.module, memo)
        new._pg, new._buckets, new._works, new._armed = self._pg, None, [], False     # (the EMA copy never runs a backward pass)
        new._order, new.check = [], self.check
        return new

    # ---- buckets ----------------------------------------------------------------------------------------------------------
    def _build_buckets(self) -> None:
        params = list(self.module.parameters())
        total = sum(p.numel() for p in params)
        flat = torch.zeros(total, dtype=torch.float32, device=params[0].device)      # holes (unused parameters) stay finite
        cap = self.BUCKET_BYTES // 4
        self._buckets, self._bucket_of = [], {}
        cur, cur_n, off = [], 0, 0
        groups = []
        for p in reversed(params):                                # the last layers' gradients arrive first
            if cur and cur_n + p.numel() > cap:
                groups.append(cur)
                cur, cur_n = [], 0
            cur.append(p)
            cur_n += p.numel()
        if cur:
            groups.append(cur)
        for g in groups:
            n = sum(p.numel() for p in g)
            sl, views, o = flat[off:off + n], [], 0
            for p in g:
                views.append(sl[o:o + p.numel()].view_as(p))
                o += p.numel()
            b = _Bucket(g, views, sl)
            for p in g:
                self._bucket_of[id(p)] = b
                p.register_post_accumulate_grad_hook(self._on_grad)
            self._buckets.append(b)
            off += n

    def _arm(self) -> None:
        for b in self._buckets:
            b.expected = sum(1 for p in b.params if p.requires_grad)
            b.fired, b.launched = 0, False
        self._order = []
        self._armed = True

    def reset_reduction(self) -> None:
        """Start of a step (optimiser.zero_grad()): forget the bucket state of a step that never reached sync_gradients() -- an
        exception between backward and sync -- after waiting for the reductions it had already issued (they read the flat buffer)."""
        if self._buckets is None:
            return
        works, self._works = self._works, []
        for w in works:
            w.wait()
        for b in self._buckets:
            b.fired, b.launched = 0, False
        self._order = []
        self._armed = False

    def _on_grad(self, p: torch.nn.Parameter) -> None:
        if not self._armed:
            self._arm()
        b = self._bucket_of[id(p)]
        if b.launched or (b.expected and b.fired >= b.expected):
            raise RuntimeError(
                "DataParallel: a gradient arrived for a bucket that was already handed to the all-reduce -- a second backward pass "
                "before sync_gradients() (gradient accumulation, or parameters of the other network left with requires_grad=True). "
                "It would be added in place into a buffer the collective may still be reading and never be reduced. Call "
                "sync_gradients() after every backward pass (zero_grad() / reset_reduction() clears an aborted step).")
        b.fired += 1
        if b.fired == b.expected:
            self._order.append(self._buckets.index(b))
            if not self.check:
                self._launch(b)

    @torch.no_grad()
    def _launch(self, b: _Bucket) -> None:
        b.launched = True
        rows, key = [], []
        for i, (p, v) in enumerate(zip(b.params, b.views)):
            if p.grad is None:
                continue
            g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
            rows.append((v.view(-1), g.view(-1), None, None, 0.0, 0.0))
            key.append(i)
        if not rows:
            return                                              # nothing of this bucket was used: no reduction (heads on odd iterations)
        key = tuple(key)
        tab = b.tables.get(key)
        if tab is None:
            tab = b.tables[key] = TensorTable(rows, b.flat.device)
        else:
            tab.refresh(rows)
        KM.K.multi_tensor(tab, MT_PACK, 1.0 / self.world_size)
        self._works.append(dist.all_reduce(b.flat, op=dist.ReduceOp.SUM, group=self._pg, async_op=True))
        for i in key:
            b.params[i].grad = b.views[i]

    @torch.no_grad()
    def sync_gradients(self, async_op: bool = False):
        """Mean-all-reduce of the gradients; returns a handle with .wait() (no-op handle for one rank).  Buckets that completed during
        the backward pass are already in flight; the rest are launched here."""
        if self.world_size == 1:
            return _Done()
        if not self._armed:
            self._arm()                                         # (a backward pass that touched no parameter of this network)
        if self.check:
            self._verify_ranks_agree()
        for b in self._buckets:
            if not b.launched:
                self._launch(b)
        works, self._works, self._armed = self._works, [], False
        handle = _Works(works)
        if async_op:
            return handle
        handle.wait()
        return _Done()


    def _verify_ranks_agree(self) -> None:
        """debug mode (self.check): one all-reduce of [v, -v] under MAX gives max and min over the ranks of
        v = (gradient-present bit per parameter, bucket completion order); any difference raises on every rank"""
        params = [p for b in self._buckets for p in b.params]
        nb = len(self._buckets)
        v = [1 if p.grad is not None else 0 for p in params] + (self._order + [-1] * nb)[:nb]
        t = torch.tensor(v + [-x for x in v], dtype=torch.int32, device=self._buckets[0].flat.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self._pg)
        t = t.cpu().tolist()
        hi, lo = t[:len(v)], [-x for x in t[len(v):]]
        if hi != lo:
            names = {id(p): k for k, p in self.module.named_parameters()}
            bad = [names.get(id(p), "?") for i, p in enumerate(params) if hi[i] != lo[i]]
            order_differs = hi[len(params):] != lo[len(params):]
            raise RuntimeError(
                f"DataParallel(check): the ranks disagree on this step's gradients (rank {dist.get_rank(self._pg)}): "
                f"{len(bad)} parameter(s) have a gradient on some ranks and none on others {bad[:8]}"
                + ("; the buckets completed in a different order" if order_differs else "")
                + ". The unchecked path would issue mismatched collectives here.")


class _Works:
    def __init__(self, works):
        self.works = works

    def wait(self):
        for w in self.works:
            w.wait()
        return True


class _Done:
    def wait(self):
        return True

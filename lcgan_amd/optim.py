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
    rows: sequence of (t0, t1, t2, t3, f0, f1) with t* fp32 contiguous tensors of equal numel (or None)."""

    def __init__(self, rows: Sequence[tuple], device):
        self.entries = list(rows)            # keeps the tensors alive (and feeds the CPU emulation in tests)
        arr = np.zeros(len(rows), dtype=_DESC)
        ct, ci = [], []
        total = 0
        for i, (t0, t1, t2, t3, f0, f1) in enumerate(rows):
            n = t0.numel()
            for t in (t0, t1, t2, t3):
                if t is not None:
                    assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == n
            arr[i] = (t0.data_ptr(), t1.data_ptr() if t1 is not None else 0, t2.data_ptr() if t2 is not None else 0,
                      t3.data_ptr() if t3 is not None else 0, n, f0, f1)
            nch = (n + MT_CHUNK - 1) // MT_CHUNK
            ct += [i] * nch
            ci += list(range(nch))
            total += n
        self.total = total
        self.n_chunks = len(ct)
        self.descs = torch.from_numpy(arr.view(np.uint8).copy()).to(device, non_blocking=True)
        self.chunk_tensor = torch.tensor(ct, dtype=torch.int32).to(device, non_blocking=True)
        self.chunk_index = torch.tensor(ci, dtype=torch.int32).to(device, non_blocking=True)


class Adam:
    """torch.optim.Adam(params, lr, betas, eps) restricted to what worker.py:98-110 uses (no weight decay / amsgrad)."""

    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float, betas=(0.0, 0.99), eps: float = 1e-8):
        self.params: List[torch.nn.Parameter] = list(params)
        self.lr, self.beta1, self.beta2, self.eps = float(lr), float(betas[0]), float(betas[1]), float(eps)
        self.exp_avg = [torch.zeros_like(p, memory_format=torch.contiguous_format) for p in self.params]
        self.exp_avg_sq = [torch.zeros_like(p, memory_format=torch.contiguous_format) for p in self.params]
        self.steps = [0] * len(self.params)

    def zero_grad(self, set_to_none: bool = True):
        for p in self.params:
            p.grad = None

    @torch.no_grad()
    def step(self):
        rows = []
        for i, p in enumerate(self.params):
            if p.grad is None:
                continue                                     # unused / frozen parameter: skipped like torch.optim.Adam does
            self.steps[i] += 1
            t = self.steps[i]
            bc1 = 1.0 - self.beta1 ** t
            bc2 = 1.0 - self.beta2 ** t
            g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
            rows.append((p.data, g, self.exp_avg[i], self.exp_avg_sq[i], self.lr / bc1, 1.0 / math.sqrt(bc2)))
        if rows:
            KM.K.multi_tensor(TensorTable(rows, self.params[0].device), MT_ADAM, self.beta1, self.beta2, self.eps)
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


class DataParallel(torch.nn.Module):
    """Single-node data parallelism: one process per GPU, full replicas, gradient mean over ranks (what the reference gets
    from DistributedDataParallel, worker.py:88-96).  Exposes `.module` and the `module.`-prefixed state_dict the reference's
    checkpoints carry.  `sync_gradients()` packs every present gradient into one flat fp32 bucket (one multi-tensor kernel,
    pre-scaled by 1/world), all-reduces the bucket over RCCL and re-points `.grad` at the bucket so Adam reads it in place.
    Parameters whose grad is None on this rank are treated as unused (the flag sets are structural in LC-GAN -- projection
    heads on odd iterations, frozen layers -- hence identical on every rank)."""

    def __init__(self, module: torch.nn.Module, process_group=None, broadcast: bool = True):
        super().__init__()
        self.module = module
        self._pg = process_group
        self._bucket: Optional[torch.Tensor] = None
        self._comm_stream = None
        if broadcast and self.world_size > 1:
            with torch.no_grad():
                for t in list(module.parameters()) + list(module.buffers()):     # DDP ctor broadcast of rank-0 state
                    dist.broadcast(t.data, src=0, group=self._pg)

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
        new._pg, new._bucket, new._comm_stream = self._pg, None, None
        return new

    @torch.no_grad()
    def sync_gradients(self, async_op: bool = False):
        """Mean-all-reduce of the gradients; returns a handle with .wait() (no-op handle for one rank)."""
        ws = self.world_size
        if ws == 1:
            return _Done()
        used = [p for p in self.module.parameters() if p.grad is not None]
        if not used:
            return _Done()
        total = sum(p.numel() for p in used)
        if self._bucket is None or self._bucket.numel() < total:
            self._bucket = torch.empty(sum(p.numel() for p in self.module.parameters()), dtype=torch.float32, device=used[0].device)
        flat = self._bucket[:total]
        rows, off, views = [], 0, []
        for p in used:
            v = flat[off:off + p.numel()]
            g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
            rows.append((v, g.view(-1), None, None, 0.0, 0.0))
            views.append(v.view_as(p))
            off += p.numel()
        KM.K.multi_tensor(TensorTable(rows, flat.device), MT_PACK, 1.0 / ws)
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self._pg, async_op=True)
        for p, v in zip(used, views):
            p.grad = v
        if async_op:
            return work
        work.wait()
        return _Done()


class _Done:
    def wait(self):
        return True

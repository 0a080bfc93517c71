"""EMA of the generator as ONE multi-tensor kernel launch (reference ema.py:4-32: a Python loop of 141 lerps)."""
from __future__ import annotations

import torch

from . import kernels as KM
from .optim import MT_EMA, TensorTable


class Ema(object):
    """Same constructor / update() contract as the reference (ema.py:5, :19)."""

    def __init__(self, source, target, decay=0.9999, start_iter=0):
        self.source = source
        self.target = target
        self.decay = decay
        self.start_iter = start_iter
        with torch.no_grad():                                   # ema.py:13-17: target starts as a copy of the source
            for p_ema, p in zip(self.target.parameters(), self.source.parameters()):
                p_ema.copy_(p)
            for b_ema, b in zip(self.target.buffers(), self.source.buffers()):
                b_ema.copy_(b)
        self._table = None
        self._copy_buffers = []

    def _build(self):
        rows = []
        for p_ema, p in zip(self.target.parameters(), self.source.parameters()):
            rows.append((p_ema.data, p.data, None, None, 0.0, 0.0))
        for (name, b_ema), (_, b) in zip(self.target.named_buffers(), self.source.named_buffers()):
            if "num_batches_tracked" in name or not torch.is_floating_point(b):   # ema.py:29-30
                self._copy_buffers.append((b_ema, b))
            else:
                rows.append((b_ema.data, b.data, None, None, 0.0, 0.0))
        for t0, t1, *_ in rows:
            assert t0.is_contiguous() and t1.is_contiguous()
        self._table = TensorTable(rows, rows[0][0].device)

    @torch.no_grad()
    def update(self, iter=None):
        decay = 0.0 if (iter is not None and 0 <= iter < self.start_iter) else self.decay     # ema.py:20-23
        if self._table is None:
            self._build()
        KM.K.multi_tensor(self._table, MT_EMA, decay)            # p_ema <- p + decay * (p_ema - p)   (ema.py:26-32)
        from . import ops
        ops.invalidate_weights(self.target.parameters())
        for b_ema, b in self._copy_buffers:
            b_ema.copy_(b)

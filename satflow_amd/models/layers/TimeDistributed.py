"""Apply a module to every timestep (surface of reference ``satflow/models/layers/TimeDistributed.py``).

Folding time into the batch is a view, not arithmetic; the wrapped module does the work (for the
MetNet encoder that is the HIP DownSampler pipeline, which MetNet additionally batches over lead times).
"""
from __future__ import annotations

import torch
from torch import nn


class TimeDistributed(nn.Module):
    def __init__(self, module: nn.Module, low_mem: bool = False, tdim: int = 1):
        super().__init__()
        self.module, self.low_mem, self.tdim = module, low_mem, tdim

    def forward(self, *tensors: torch.Tensor, **kwargs):
        """Inputs ``[bs, seq_len, ...]``; one batched call (reference ``:21-29``) or one call per step (``:31-40``)."""
        if self.low_mem or self.tdim != 1:
            return self.low_mem_forward(*tensors, **kwargs)
        bs, seq_len = tensors[0].shape[:2]
        out = self.module(*(t.reshape(bs * seq_len, *t.shape[2:]) for t in tensors), **kwargs)
        return self.format_output(out, bs, seq_len)

    def low_mem_forward(self, *tensors: torch.Tensor, **kwargs):
        steps = tensors[0].shape[self.tdim]
        outs = [self.module(*(t.select(self.tdim, i) for t in tensors), **kwargs) for i in range(steps)]
        if isinstance(outs[0], tuple):
            return tuple(torch.stack([o[k] for o in outs], dim=self.tdim) for k in range(len(outs[0])))
        return torch.stack(outs, dim=self.tdim)

    def format_output(self, out, bs: int, seq_len: int):
        if isinstance(out, tuple):
            return tuple(o.reshape(bs, seq_len, *o.shape[1:]) for o in out)
        return out.reshape(bs, seq_len, *out.shape[1:])

    def __repr__(self) -> str:
        return f"TimeDistributed({self.module})"

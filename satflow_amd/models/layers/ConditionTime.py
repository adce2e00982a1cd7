"""Lead-time conditioning planes (surface of reference ``satflow/models/layers/ConditionTime.py``).

Inside MetNet these planes are never materialised: they enter conv1 as a constant second source
(``satflow_amd/models/metnet.py``).  This module keeps the stand-alone layer surface - pure data
movement (a one-hot constant concatenated on the channel axis), expressed with tensor views.
"""
from __future__ import annotations

import torch
from torch import nn


def condition_time(x: torch.Tensor, i: int = 0, size=(12, 16), seq_len: int = 15) -> torch.Tensor:
    """One-hot time image-layers ``[seq_len, *size]``; plane ``i`` is all ones (reference ``:5-10``)."""
    assert i < seq_len
    planes = torch.zeros(seq_len, *size, dtype=x.dtype, device=x.device)
    planes[i] = 1
    return planes


class ConditionTime(nn.Module):
    """Appends ``horizon`` one-hot planes on ``ch_dim`` (reference ``:13-33``): 5-D ``[B,T,C,H,W]`` or 4-D channels-last."""

    def __init__(self, horizon: int, ch_dim: int = 2, num_dims: int = 5):
        super().__init__()
        self.horizon, self.ch_dim, self.num_dims = horizon, ch_dim, num_dims

    def forward(self, x: torch.Tensor, fstep: int = 0) -> torch.Tensor:
        if self.num_dims == 5:
            bs, seq_len, ch, h, w = x.shape
            ct = condition_time(x, fstep, (h, w), seq_len=self.horizon).expand(bs, seq_len, self.horizon, h, w)
        else:
            bs, h, w, ch = x.shape
            ct = condition_time(x, fstep, (h, w), seq_len=self.horizon).permute(1, 2, 0).expand(bs, h, w, self.horizon)
        out = torch.cat((x, ct), dim=self.ch_dim)
        assert out.shape[self.ch_dim] == ch + self.horizon
        return out

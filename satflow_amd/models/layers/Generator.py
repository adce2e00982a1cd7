"""DVD-GAN style generator on the HIP kernels - SURVEY 8f-3.  Mirror of reference ``satflow/models/layers/Generator.py:12-131``: same
constructor, same sub-module names (``embedding``, ``affine_transfrom`` [sic], ``conv.0`` ... ``conv.11``, ``colorize``),
``forward(x, class_id) -> [B, n_frames, 3, W, H]``.

PARITY UNPINNED: the reference class cannot be instantiated (its ``ConvGRU`` import points at a module that is not in the tree);
``.ConvGRU`` supplies that module, the rest follows the reference line by line and is checked against ``oracle/dgmr.py``.
``hierar_flag=True`` is not implemented (the reference indexes ``noise_emb[0]`` for the affine map but concatenates the whole tuple
for the conditions, which fails in ``torch.cat``).  ``out_channels`` (default 3) extends the constructor surface.

Frames are kept time-major (image ``t*B + b``) through the whole network: the ConvGRU blocks run layer by layer over the sequence,
the GResBlocks see all frames as one batch; the reference's ``condition.repeat(n_frames, 1)`` pairs image ``j = b*T + t`` of ITS
batch-major order with condition row ``j % B`` - reproduced through an index table.
"""
from __future__ import annotations

import torch
import torch.nn as nn
from torch import Tensor

from ... import functional as F
from ... import functional_gan as FG
from ..._hip import cpad, require_device
from .ConvGRU import ConvGRU
from .GResBlock import GResBlock
from .Normalization import SpectralNorm


class Generator(nn.Module):
    def __init__(self, in_dim=120, latent_dim=4, n_class=4, ch=32, n_frames=48, hierar_flag=False, out_channels=3):
        super().__init__()
        if hierar_flag:
            raise NotImplementedError("hierar_flag=True is not implemented (it fails in the reference's own torch.cat, Generator.py:110)")
        self.in_dim, self.latent_dim, self.n_class, self.ch, self.hierar_flag, self.n_frames = in_dim, latent_dim, n_class, ch, hierar_flag, n_frames
        self.out_channels = out_channels
        self.embedding = nn.Embedding(n_class, in_dim)
        self.affine_transfrom = nn.Linear(in_dim * 2, latent_dim * latent_dim * 8 * ch)
        gru = lambda c, hs, ks: ConvGRU(c, hidden_sizes=hs, kernel_sizes=ks, n_layers=3)
        res = lambda ci, co, **kw: GResBlock(ci, co, n_class=in_dim * 2, **kw)
        self.conv = nn.ModuleList([
            gru(8 * ch, [8 * ch, 16 * ch, 8 * ch], [3, 5, 3]), res(8 * ch, 8 * ch, upsample_factor=1), res(8 * ch, 8 * ch),
            gru(8 * ch, [8 * ch, 16 * ch, 8 * ch], [3, 5, 3]), res(8 * ch, 8 * ch, upsample_factor=1), res(8 * ch, 8 * ch),
            gru(8 * ch, [8 * ch, 16 * ch, 8 * ch], [3, 5, 3]), res(8 * ch, 8 * ch, upsample_factor=1), res(8 * ch, 4 * ch),
            gru(4 * ch, [4 * ch, 8 * ch, 4 * ch], [3, 5, 5]), res(4 * ch, 4 * ch, upsample_factor=1), res(4 * ch, 2 * ch),
        ])
        self.colorize = SpectralNorm(nn.Conv2d(2 * ch, out_channels, kernel_size=(3, 3), padding=1))

    def run(self, x: Tensor, class_id: Tensor) -> Tensor:
        """``x [B, in_dim]`` noise -> time-major NHWC frames ``[T*B, W, H, cpad(out_channels)]`` after the tanh."""
        B, Tn, ld, c0 = x.shape[0], self.n_frames, self.latent_dim, 8 * self.ch
        class_emb = self.embedding(class_id)                                   # row gather
        cond = torch.cat((x.float(), class_emb), 1)                            # [B, 2*in_dim]
        k = cond.shape[1]
        y = F.linear(torch.nn.functional.pad(cond, (0, cpad(k) - k)), self.affine_transfrom.weight, self.affine_transfrom.bias,
                     out_lanes=cpad(self.affine_transfrom.out_features))[:, : c0 * ld * ld]
        y = F.nchw_to_nhwc(y.reshape(B, c0, ld, ld))                           # the reference's .view(-1, 8*ch, ld, ld)
        # image (t, b) of the time-major batch is image j = b*T + t of the reference's batch and takes condition row j % B
        t_idx, b_idx = torch.arange(Tn, device=x.device).view(Tn, 1), torch.arange(B, device=x.device).view(1, B)
        # selection matrix (a constant), built by comparison: an indexed assignment cannot be captured into a hipGraph
        rows = (((b_idx * Tn + t_idx) % B).reshape(-1, 1) == torch.arange(B, device=x.device).view(1, B)).float()
        for k_, layer in enumerate(self.conv):
            if isinstance(layer, ConvGRU):
                y = layer.run_sequence(y, Tn, constant_input=(k_ == 0))
            else:
                y = layer.run(y, cond, embed_rows=rows)
        y = FG.relu(y)
        return FG.tanh(self.colorize.run(y))

    def forward(self, x, class_id):
        require_device(x, "x")
        y = self.run(x, class_id)
        B, Tn, C = x.shape[0], self.n_frames, self.out_channels
        h, w = y.shape[1], y.shape[2]
        return F._FromNHWC.apply(y, (B, Tn, C, h, w), B, Tn, C, h, w, (Tn * C * h * w, C * h * w, h * w))

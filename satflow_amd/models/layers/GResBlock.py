"""GResBlock on the HIP kernels - SURVEY 8f-3.  Mirror of reference ``satflow/models/layers/GResBlock.py:8-99`` (same constructor,
same sub-module names ``conv0 / conv1 / conv_sc / CBNorm1 / CBNorm2``, ``forward(x, condition)`` on NCHW frames).

Execution: conditional BatchNorm -> ReLU -> nearest up-sampling is ONE pass (``sf_film_act_fwd``); the 3x3 convolutions run on the
MFMA kernels with the spectrally normalised weights; the 1x1 projection of the skip path is applied BEFORE the up-sampling and
AFTER the pooling (it commutes with both - bit-identical for the up-sampling, equal up to fp32 summation order for the pooling -
and is 4x cheaper there); the residual sum rides on the pooling kernel.
"""
from __future__ import annotations

import torch.nn as nn
from torch import Tensor
from torch.nn import functional as TF

from ... import functional as F
from ... import functional_gan as FG
from ..._hip import require_device
from .Normalization import ConditionalNorm, SpectralNorm


def residual_block_run(blk, x: Tensor, condition, *, bn: bool, up: bool, down: bool, norms=None, embed_rows=None, pool3_nb: int = 0) -> Tensor:
    """Shared body of GResBlock / GBlock / Res3dBlock on NHWC ``x`` (``pool3_nb`` > 0: time-major frames, ``avg_pool3d`` over
    frame pairs with ``pool3_nb`` images per frame, convolutions given by the block's ``_conv`` hook)."""
    conv = blk._conv
    if bn:
        out = norms[0].run(x, condition, relu=True, up=up, embed_rows=embed_rows)
    else:
        out = FG.relu(x)
        if up:
            out = FG.upsample2(out)
    out = conv(blk.conv0, out)
    out = norms[1].run(out, condition, relu=True, embed_rows=embed_rows) if bn else FG.relu(out)
    out = conv(blk.conv1, out)
    if not getattr(blk, "skip_proj", True):  # GBlock without projection: same width, no resampling (Discriminator.py:182-185)
        return FG.add(out, x)
    if down:
        pooled_x = FG.avg_pool3(x, pool3_nb) if pool3_nb else FG.avg_pool2(x)
        skip = conv(blk.conv_sc, pooled_x)
        return FG.avg_pool3(out, pool3_nb, skip) if pool3_nb else FG.avg_pool2(out, skip)
    skip = conv(blk.conv_sc, x)
    if up:
        skip = FG.upsample2(skip)
    return FG.add(out, skip)


class GResBlock(nn.Module):
    def __init__(self, in_channel, out_channel, kernel_size=None, padding=1, stride=1, n_class=96, bn=True, activation=TF.relu, upsample_factor=2,
                 downsample_factor=1):
        super().__init__()
        self.upsample_factor = upsample_factor if downsample_factor == 1 else 1
        self.downsample_factor = downsample_factor
        self.activation = activation
        self.bn = bn if downsample_factor == 1 else False
        if kernel_size is None:
            kernel_size = [3, 3]
        if list(kernel_size) != [3, 3] or padding != 1 or stride != 1 or activation is not TF.relu:
            raise NotImplementedError("the HIP GResBlock implements kernel_size [3, 3], padding 1, stride 1, ReLU (every use in the reference)")
        if self.upsample_factor not in (1, 2) or downsample_factor not in (1, 2):
            raise NotImplementedError("the HIP GResBlock implements up- / down-sampling factors 1 and 2 (every use in the reference)")
        self.conv0 = SpectralNorm(nn.Conv2d(in_channel, out_channel, kernel_size, stride, padding, bias=True))
        self.conv1 = SpectralNorm(nn.Conv2d(out_channel, out_channel, kernel_size, stride, padding, bias=True))
        self.skip_proj = True
        self.conv_sc = SpectralNorm(nn.Conv2d(in_channel, out_channel, 1, 1, 0))
        if bn:
            self.CBNorm1 = ConditionalNorm(in_channel, n_class)
            self.CBNorm2 = ConditionalNorm(out_channel, n_class)
        self.out_channel = out_channel

    @staticmethod
    def _conv(sn: SpectralNorm, x: Tensor) -> Tensor:
        return sn.run(x)

    def run(self, x: Tensor, condition=None, embed_rows=None) -> Tensor:
        return residual_block_run(self, x, condition, bn=self.bn, up=self.upsample_factor == 2, down=self.downsample_factor == 2,
                                  norms=(self.CBNorm1, self.CBNorm2) if self.bn else None, embed_rows=embed_rows)

    def forward(self, x, condition=None):
        require_device(x, "x")
        return F.nhwc_to_nchw(self.run(F.nchw_to_nhwc(x.float()), condition), self.out_channel)

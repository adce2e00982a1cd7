"""Spatial and temporal discriminators on the HIP kernels - SURVEY 8f-3.

Mirror of reference ``satflow/models/layers/Discriminator.py``: ``SelfAttention`` (``:83-126``), ``ConditionalNorm`` (``:129-155``),
``GBlock`` (``:158-228``), ``SpatialDiscriminator`` (``:231-314``), ``Res3dBlock`` (``:322-397``), ``TemporalDiscriminator``
(``:400-478``) - same constructors, sub-module names and ``state_dict`` keys (pinned: tests/golden/dgmr_*_discriminator_keys.txt),
``forward(x, class_id)`` with the reference's tensor layouts.

Execution on NHWC frames: 3x3 convolutions on the MFMA kernels (a ``Conv3d(3, 3, 3)`` as ONE 3x3 convolution over the channel stack
of its three temporal taps, frames time-major so that a temporal shift is an address offset), 1x1 / 1x1x1 convolutions on
``sf_linear_*``, ``avg_pool2d / avg_pool3d`` (+ the residual sum) in ``sf_pool2``, the attention products in ``sf_bmm_f32`` /
``sf_softmax_rows_*``, ReLU + sum over pixels in ``sf_relu_sum_pixels_*``.  ``in_channels`` (default 3, the reference's hard-coded
value) is the only extension of the constructor surface.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as TF
from torch import Tensor
from torch.nn import init

from ... import functional as F
from ... import functional_gan as FG
from ..._hip import cpad, require_device
from .GResBlock import residual_block_run
from .Normalization import SpectralNorm


def init_conv(conv, glu=True):
    init.xavier_uniform_(conv.weight)
    if conv.bias is not None:
        conv.bias.data.zero_()


class SelfAttention(nn.Module):
    """Self attention layer over the ``W*H`` positions of a frame (reference ``:83-126``)."""

    def __init__(self, in_dim, activation=TF.relu):
        super().__init__()
        self.chanel_in = in_dim
        self.activation = activation
        self.query_conv = nn.Conv2d(in_channels=in_dim, out_channels=in_dim // 8, kernel_size=1)
        self.key_conv = nn.Conv2d(in_channels=in_dim, out_channels=in_dim // 8, kernel_size=1)
        self.value_conv = nn.Conv2d(in_channels=in_dim, out_channels=in_dim, kernel_size=1)
        self.gamma = nn.Parameter(torch.zeros(1))
        self.softmax = nn.Softmax(dim=-1)
        init_conv(self.query_conv)
        init_conv(self.key_conv)
        init_conv(self.value_conv)

    def run(self, x: Tensor) -> Tensor:
        """NHWC ``[N,H,W,Cp]``: ``energy = q k^T`` ``[N, HW, HW]``, softmax over the keys, ``out = attention v``, ``gamma * out + x``."""
        n, h, w, cp = x.shape
        d, c = self.chanel_in // 8, self.chanel_in
        # (lowp: 1x1 convolutions take the 16-bit compute mode's operands, as under the reference's autocast; exact fp32 products in f32 mode)
        q = F.linear(x, self.query_conv.weight.view(d, c), self.query_conv.bias, lowp=True).view(n, h * w, -1)   # pad lanes are zero
        k = F.linear(x, self.key_conv.weight.view(d, c), self.key_conv.bias, lowp=True).view(n, h * w, -1)
        v = F.linear(x, self.value_conv.weight.view(c, c), self.value_conv.bias, out_lanes=cp, lowp=True).view(n, h * w, cp)
        out = FG.attention(q, k, v).view(n, h, w, cp)
        return FG.gamma_residual(out, x, self.gamma)

    def forward(self, x):
        require_device(x, "x")
        return F.nhwc_to_nchw(self.run(F.nchw_to_nhwc(x.float())), self.chanel_in)


class ConditionalNorm(nn.Module):
    """The discriminator file's own conditional BatchNorm (reference ``:129-155``; used by ``GBlock(bn=True)``): as
    ``Normalization.ConditionalNorm`` with a constant initialisation of the embedding."""

    def __init__(self, in_channel, n_condition=148):
        super().__init__()
        from .Normalization import ConditionalNorm as _CN

        self.in_channel = in_channel
        self.bn = nn.BatchNorm2d(in_channel, affine=False)
        self.embed = nn.Linear(n_condition, in_channel * 2)
        self.embed.weight.data[:, :in_channel] = 1
        self.embed.weight.data[:, in_channel:] = 0
        self.embedding = _CN.embedding.__get__(self)
        self.run = _CN.run.__get__(self)

    def forward(self, input, class_id):
        require_device(input, "input")
        return F.nhwc_to_nchw(self.run(F.nchw_to_nhwc(input.float()), class_id), self.in_channel)


class GBlock(nn.Module):
    def __init__(self, in_channel, out_channel, kernel_size=[3, 3], padding=1, stride=1, n_class=None, bn=True, activation=TF.relu, upsample=True,
                 downsample=False):
        super().__init__()
        if list(kernel_size) != [3, 3] or padding != 1 or stride != 1 or activation is not TF.relu:
            raise NotImplementedError("the HIP GBlock implements kernel_size [3, 3], padding 1, stride 1, ReLU (every use in the reference)")
        self.conv0 = SpectralNorm(nn.Conv2d(in_channel, out_channel, kernel_size, stride, padding, bias=True))
        self.conv1 = SpectralNorm(nn.Conv2d(out_channel, out_channel, kernel_size, stride, padding, bias=True))
        self.skip_proj = False
        if in_channel != out_channel or upsample or downsample:
            self.conv_sc = SpectralNorm(nn.Conv2d(in_channel, out_channel, 1, 1, 0))
            self.skip_proj = True
        self.upsample = upsample
        self.downsample = downsample
        self.activation = activation
        self.bn = bn
        if bn:
            self.HyperBN = ConditionalNorm(in_channel, 148)
            self.HyperBN_1 = ConditionalNorm(out_channel, 148)
        self.out_channel = out_channel

    @staticmethod
    def _conv(sn: SpectralNorm, x: Tensor) -> Tensor:
        return sn.run(x)

    def run(self, x: Tensor, condition=None) -> Tensor:
        return residual_block_run(self, x, condition, bn=self.bn, up=bool(self.upsample), down=bool(self.downsample),
                                  norms=(self.HyperBN, self.HyperBN_1) if self.bn else None)

    def forward(self, input, condition=None):
        require_device(input, "input")
        return F.nhwc_to_nchw(self.run(F.nchw_to_nhwc(input.float()), condition), self.out_channel)


def _head(owner, feat: Tensor, class_id: Tensor, frames_per_sample: int, order: Tensor = None, time_major: bool = False) -> Tensor:
    """Shared tail (reference ``:286-314`` / ``:452-478``): ReLU, sum over the pixels, spectral-normed linear score plus the
    projection onto the spectral-normed class embedding.  ``feat`` NHWC ``[N,h,w,Cp]`` -> ``[N]`` scores (rows permuted by ``order``)."""
    c = owner.linear.module.in_features
    pooled = FG.relu_sum_pixels(feat)                                        # [N, Cp]
    if order is not None:
        pooled = pooled[order]                                               # time-major rows -> the reference's (b, t) order
    n = pooled.shape[0]
    w_lin = owner.linear.compute_weight()                                    # [1, C]
    out_linear = F.linear(pooled, w_lin, owner.linear.module.bias)[:, 0]
    ids = class_id.repeat(frames_per_sample) if time_major else class_id.view(-1, 1).repeat(1, frames_per_sample).view(-1)
    emb = owner.embed.compute_weight()[ids]                                  # [N, C] row gather
    prod = FG.bmm(pooled[:, :c].unsqueeze(1), emb.unsqueeze(2)).view(n)      # per-row dot products on the matrix cores
    return out_linear + prod


class SpatialDiscriminator(nn.Module):
    def __init__(self, chn=128, n_class=4, in_channels=3):
        super().__init__()
        self.in_channels = in_channels
        self.pre_conv = nn.Sequential(
            SpectralNorm(nn.Conv2d(in_channels, 2 * chn, 3, padding=1)),
            nn.ReLU(),
            SpectralNorm(nn.Conv2d(2 * chn, 2 * chn, 3, padding=1)),
            nn.AvgPool2d(2),
        )
        self.pre_skip = SpectralNorm(nn.Conv2d(in_channels, 2 * chn, 1))
        self.conv1 = GBlock(2 * chn, 4 * chn, bn=False, upsample=False, downsample=True)
        self.attn = SelfAttention(4 * chn)
        self.conv2 = nn.Sequential(
            GBlock(4 * chn, 8 * chn, bn=False, upsample=False, downsample=True),
            GBlock(8 * chn, 16 * chn, bn=False, upsample=False, downsample=True),
            GBlock(16 * chn, 16 * chn, bn=False, upsample=False, downsample=True),
        )
        self.linear = SpectralNorm(nn.Linear(16 * chn, 1))
        self.embed = nn.Embedding(n_class, 16 * chn)
        self.embed.weight.data.uniform_(-0.1, 0.1)
        self.embed = SpectralNorm(self.embed)

    def run(self, frames: Tensor, class_id: Tensor, T_frames: int, time_major: bool = False) -> Tensor:
        """NHWC frames ``[B*T,H,W,Cp]`` in the reference's (b, t) order (``time_major``: image t*B + b) -> scores in the same order."""
        out = self.pre_conv[0].run(frames)
        out = FG.relu(out)
        out = self.pre_conv[2].run(out)
        out = FG.avg_pool2(out, self.pre_skip.run(FG.avg_pool2(frames)))
        out = self.conv1.run(out)
        out = self.attn.run(out)
        for blk in self.conv2:
            out = blk.run(out)
        return _head(self, out, class_id, T_frames, time_major=time_major)

    def forward(self, x, class_id):
        """``x [B,T,C,W,H]`` -> one score per frame ``[B*T]`` (reference ``:263-314``)."""
        require_device(x, "x")
        B, Tn, C, W, H = x.shape
        frames = F.nchw_to_nhwc(x.float().reshape(B * Tn, C, H, W))
        return self.run(frames, class_id, Tn)


def conv3x3x3(in_planes, out_planes, stride=1):
    return nn.Conv3d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False)


def conv3d_run(sn: SpectralNorm, x: Tensor, T_frames: int) -> Tensor:
    """A spectral-normed ``nn.Conv3d`` (k = 3 padding 1, or k = 1) on time-major NHWC frames ``[T*nb,H,W,Cp]``."""
    m = sn.module
    w = sn.compute_weight()                                  # [O, I, kt, kh, kw]
    O, I = w.shape[0], w.shape[1]
    if w.shape[2:] == (1, 1, 1):
        return F.linear(x, w.view(O, I), m.bias, lowp=True)
    if w.shape[2:] != (3, 3, 3):
        raise NotImplementedError("Conv3d kernels 1 and 3 are implemented")
    cp = x.shape[-1]
    # weight of the equivalent 3x3 convolution over the stacked taps: lanes dt * cp + c  <-  w[:, c, dt]
    w3 = TF.pad(w.permute(0, 2, 1, 3, 4), (0, 0, 0, 0, 0, cp - I)).reshape(O, 3 * cp, 3, 3)
    return F.conv3x3(sn.conv_engine([3 * cp], O), FG.time_stack3(x, T_frames), w3, m.bias)


class Res3dBlock(nn.Module):
    def __init__(self, in_channel, out_channel, kernel_size=[3, 3, 3], padding=1, stride=1, n_class=None, bn=True, activation=TF.relu, upsample=True,
                 downsample=False):
        super().__init__()
        if list(kernel_size) != [3, 3, 3] or padding != 1 or stride != 1 or activation is not TF.relu or bn or upsample:
            raise NotImplementedError("the HIP Res3dBlock implements the discriminator's configuration: 3x3x3, bn=False, upsample=False")
        self.conv0 = SpectralNorm(nn.Conv3d(in_channel, out_channel, kernel_size, stride, padding, bias=True))
        self.conv1 = SpectralNorm(nn.Conv3d(out_channel, out_channel, kernel_size, stride, padding, bias=True))
        self.skip_proj = False
        if in_channel != out_channel or upsample or downsample:
            self.conv_sc = SpectralNorm(nn.Conv3d(in_channel, out_channel, 1, 1, 0))
            self.skip_proj = True
        self.upsample, self.downsample, self.activation, self.bn = upsample, downsample, activation, bn
        self.out_channel = out_channel
        self._frames = 0

    def _conv(self, sn: SpectralNorm, x: Tensor) -> Tensor:
        # the second convolution and the projection see the block's current frame count (the skip path is pooled first)
        return conv3d_run(sn, x, self._cur_frames if sn is not self.conv_sc else self._sc_frames)

    def run(self, x: Tensor, T_frames: int, nb: int) -> Tensor:
        self._cur_frames = T_frames
        self._sc_frames = T_frames // 2 if self.downsample else T_frames
        return residual_block_run(self, x, None, bn=False, up=False, down=bool(self.downsample), pool3_nb=nb)


class TemporalDiscriminator(nn.Module):
    def __init__(self, chn=128, n_class=4, in_channels=3):
        super().__init__()
        self.in_channels = in_channels
        self.pre_conv = nn.Sequential(
            SpectralNorm(nn.Conv3d(in_channels, 2 * chn, 3, padding=1)),
            nn.ReLU(),
            SpectralNorm(nn.Conv3d(2 * chn, 2 * chn, 3, padding=1)),
            nn.AvgPool3d(2),
        )
        self.pre_skip = SpectralNorm(nn.Conv3d(in_channels, 2 * chn, 1))
        self.res3d = Res3dBlock(2 * chn, 4 * chn, bn=False, upsample=False, downsample=True)
        self.self_attn = SelfAttention(4 * chn)
        self.conv = nn.Sequential(
            GBlock(4 * chn, 8 * chn, bn=False, upsample=False, downsample=True),
            GBlock(8 * chn, 16 * chn, bn=False, upsample=False, downsample=True),
            GBlock(16 * chn, 16 * chn, bn=False, upsample=False, downsample=True),
        )
        self.linear = SpectralNorm(nn.Linear(16 * chn, 1))
        self.embed = nn.Embedding(n_class, 16 * chn)
        self.embed.weight.data.uniform_(-0.1, 0.1)
        self.embed = SpectralNorm(self.embed)

    def run(self, frames: Tensor, class_id: Tensor, T_frames: int, nb: int) -> Tensor:
        """TIME-MAJOR NHWC frames ``[T*B,H,W,Cp]`` (image t*B + b) -> scores ``[B * T/4]`` in the reference's (b, t) order."""
        out = conv3d_run(self.pre_conv[0], frames, T_frames)
        out = FG.relu(out)
        out = conv3d_run(self.pre_conv[2], out, T_frames)
        out = FG.avg_pool3(out, nb, conv3d_run(self.pre_skip, FG.avg_pool3(frames, nb), T_frames // 2))
        out = self.res3d.run(out, T_frames // 2, nb)
        t4 = T_frames // 4
        out = self.self_attn.run(out)      # frames are scored independently from here on
        for blk in self.conv:
            out = blk.run(out)
        order = (torch.arange(nb, device=out.device).view(nb, 1) + nb * torch.arange(t4, device=out.device).view(1, t4)).reshape(-1)  # row b*t4 + t <- t*nb + b
        return _head(self, out, class_id, t4, order)

    def forward(self, x, class_id):
        """``x [B,C,T,W,H]`` -> ``[B * T/4]`` scores (reference ``:431-478``)."""
        require_device(x, "x")
        B, C, Tn, W, H = x.shape
        if Tn % 4:
            raise RuntimeError("TemporalDiscriminator: the clip length must be a multiple of 4 (two temporal poolings)")
        frames = F._ToNHWC.apply(x.float().contiguous(), B, Tn, C, W, H, (C * Tn * W * H, W * H, Tn * W * H))
        return self.run(frames, class_id, Tn, B)

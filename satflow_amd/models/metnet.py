"""MetNet (preprocessor -> ConditionTime -> DownSampler -> ConvGRU -> axial attention -> 1x1 head) on the HIP kernels.

The reference does not contain this network: ``LitMetNet`` wraps ``from metnet import MetNet``
(``satflow/models/pl_metnet.py:6,46-59``), an un-vendored package.  This module supplies the same
constructor surface and the upstream ``state_dict`` key names (``image_encoder.module.module.0.weight``
... ``temporal_enc.rnn.cell_list.0.conv_zr.weight`` ... ``temporal_agg.0.axial_attentions.0.fn.to_q.weight``
... ``head.bias``) following SURVEY.md Appendix A; numerical parity for it is "unpinned" (oracle/metnet.py).

Execution (differs from upstream by design, same results):
  * all ``forecast_steps`` lead times are ONE batch (upstream loops and recomputes everything per lead
    time); BatchNorm keeps upstream's per-call statistics by reducing per lead-time group;
  * conv1 is linear and the ConditionTime planes are constant one-hot images, so its image part is
    computed once per frame (not once per frame and lead time) and the per-lead-time contribution - a
    border-aware constant per output channel - is added inside the fused first max-pooling
    (``sf_leadtime_pool_fwd/bwd``); the sequence is never replicated ``forecast_steps`` times;
  * the ConvGRU's input convolutions run for all timesteps in one launch; only the hidden-state
    convolution is sequential, fused with the gate arithmetic.
"""
from __future__ import annotations

import math
import os
from typing import List, Optional

import torch
from torch import nn

from .. import functional as F
from .._hip import cpad, encoder_storage_dtype, require_device
from ..functional import ConvEngine, GRUEngine
from .layers.ConditionTime import ConditionTime
from .layers.TimeDistributed import TimeDistributed
from .utils import get_conv_layer

Tensor = torch.Tensor


class MetNetPreprocessor(nn.Module):
    """Upstream ``MetNetPreprocessor(sat_channels, crop_size, use_space2depth=True, split_input=True)`` (no parameters)."""

    def __init__(self, sat_channels: int = 12, crop_size: int = 256, use_space2depth: bool = True, split_input: bool = True):
        super().__init__()
        if not (use_space2depth and split_input):
            raise NotImplementedError("only the configuration MetNet itself uses (space2depth + split input) is on the path")
        self.sat_channels, self.crop_size = sat_channels, crop_size

    def forward(self, x: Tensor) -> Tensor:
        """``x[B,T,C,H,W] -> [B,T,8*sat+(C-sat),crop,crop]`` (module-surface form; MetNet itself stays in NHWC)."""
        from .. import kernels as K

        B, T, C, H, W = x.shape
        frames = F.metnet_preprocess(x.float(), self.sat_channels, self.crop_size)  # [T*B, S, S, Cp]
        c = 8 * self.sat_channels + (C - self.sat_channels)
        S = self.crop_size
        return F._FromNHWC.apply(frames, (B, T, c, S, S), B, T, c, S, S, (T * c * S * S, c * S * S, S * S))


class DownSampler(nn.Module):
    """Parameter container with upstream's ``nn.Sequential`` indices (0 conv, 3 BN, 4 conv, 5 BN, 6 conv, 7 BN, 8 conv)."""

    def __init__(self, in_channels: int, output_channels: int = 256, conv_type: str = "standard"):
        super().__init__()
        conv2d = get_conv_layer(conv_type)
        self.output_channels = output_channels
        self.in_channels = in_channels
        self.module = nn.Sequential(
            conv2d(in_channels, 160, 3, padding=1),
            nn.MaxPool2d((2, 2), stride=2),
            nn.Identity(),
            nn.BatchNorm2d(160),
            conv2d(160, output_channels, 3, padding=1),
            nn.BatchNorm2d(output_channels),
            conv2d(output_channels, output_channels, 3, padding=1),
            nn.BatchNorm2d(output_channels),
            conv2d(output_channels, output_channels, 3, padding=1),
            nn.MaxPool2d((2, 2), stride=2),
            nn.Identity(),
        )
        oc = output_channels
        self.capture = None  # tests set a dict: receives the tensor in front of the last pooling ("y4")
        self._eng = [ConvEngine([in_channels], 160), ConvEngine([160], oc), ConvEngine([oc], oc), ConvEngine([oc], oc)]

    def run(self, x: Tensor, groups: int, pooled: Optional[Tensor] = None, perm=None, dropout=None, out_dtype=torch.float32,
            pooled_stats=None) -> Tensor:
        """NHWC pipeline from the first pooling's output (``pooled``, optionally with its per-group statistics ``pooled_stats`` from
        ``F.leadtime_pool(..., want_stats=True)``) or from the input ``x``; ``groups`` BatchNorm batches.

        The activations keep the storage type they arrive in (fp32, or bf16 in "bf16a" mode); the last pooling returns
        ``out_dtype`` (fp32 at the module surface; MetNet keeps the encoder's storage type - its only consumers are the
        ConvGRU's input convolution and weight gradient, i.e. bf16 MFMA operands)."""
        m = self.module
        y = pooled if pooled is not None else F.maxpool2(F.conv3x3(self._eng[0], x, m[0].weight, m[0].bias, out_dtype=x.dtype))
        st = y.dtype
        if self.training:
            # BatchNorm -> Conv2d pairs: with bf16-stored activations the normalisation is folded into the convolution
            # (F.batchnorm_conv3x3; otherwise it runs the two ops below, statistics from the producing convolution's epilogue)
            if pooled_stats is not None and pooled_stats.n != groups:
                raise RuntimeError("pooled_stats were taken for a different number of BatchNorm groups")
            y, cs = F.batchnorm_conv3x3(y, m[3], groups, pooled_stats, self._eng[1], m[4].weight, m[4].bias, out_dtype=st, want_stats=True)
            y, cs = F.batchnorm_conv3x3(y, m[5], groups, cs, self._eng[2], m[6].weight, m[6].bias, out_dtype=st, want_stats=True)
            if self.capture is None:
                # conv4 + the last pooling (+ the two dropouts): the pooling rides in the convolution's epilogue where the kernel allows
                n, h, w, c = y.shape
                drop = (dropout[0], dropout[1], (n // dropout[2]) * (h // 2) * (w // 2) * self._eng[3].coutp) if dropout is not None else None
                return F.batchnorm_conv3x3_maxpool(y, m[7], groups, cs, self._eng[3], m[8].weight, m[8].bias, perm, out_dtype, drop)
            y, _ = F.batchnorm_conv3x3(y, m[7], groups, cs, self._eng[3], m[8].weight, m[8].bias, out_dtype=st)
            return self._exit(y, perm, dropout, out_dtype)
        y = F.batchnorm(y, m[3], groups, self.training)
        # the two convolutions that feed a BatchNorm also emit its statistics from their epilogue (bf16 kernels, training)
        y, cs = F.conv3x3(self._eng[1], y, m[4].weight, m[4].bias, out_dtype=st, want_stats=self.training)
        y = F.batchnorm(y, m[5], groups, self.training, cs)
        y, cs = F.conv3x3(self._eng[2], y, m[6].weight, m[6].bias, out_dtype=st, want_stats=self.training)
        y = F.batchnorm(y, m[7], groups, self.training, cs)
        y = F.conv3x3(self._eng[3], y, m[8].weight, m[8].bias, out_dtype=st)
        return self._exit(y, perm, dropout, out_dtype)

    def _exit(self, y: Tensor, perm, dropout, out_dtype) -> Tensor:
        if self.capture is not None:
            self.capture["y4"] = y.detach()
        if dropout is not None:  # (p1, p2, timesteps): period = elements of one timestep of the pooled tensor
            n, h, w, c = y.shape
            dropout = (dropout[0], dropout[1], (n // dropout[2]) * (h // 2) * (w // 2) * c)
        return F.maxpool2(y, perm, out_dtype=out_dtype, dropout=dropout)

    def forward(self, x: Tensor) -> Tensor:
        """``[N,C,H,W] -> [N,out,H/4,W/4]`` (module-surface form, one BatchNorm batch)."""
        require_device(x, "x")
        return F.nhwc_to_nchw(self.run(F.nchw_to_nhwc(x.float()), 1), self.output_channels)


class ConvGRUCell(nn.Module):
    """Parameter container of upstream ``ConvGRUCell`` (conv_zr over [x;h] -> z,r; conv_h1(x); conv_h2(h))."""

    def __init__(self, input_dim: int, hidden_dim: int, kernel_size=(3, 3), bias: bool = True):
        super().__init__()
        if tuple(kernel_size) != (3, 3) or not bias:
            raise NotImplementedError("the HIP ConvGRU implements the 3x3 / bias=True cell MetNet uses (kernel_size: 3 in metnet.yaml)")
        self.input_dim, self.hidden_dim = input_dim, hidden_dim
        self.conv_zr = nn.Conv2d(input_dim + hidden_dim, 2 * hidden_dim, 3, padding=1)
        self.conv_h1 = nn.Conv2d(input_dim, hidden_dim, 3, padding=1)
        self.conv_h2 = nn.Conv2d(hidden_dim, hidden_dim, 3, padding=1)
        self.reset_parameters()
        self._eng = GRUEngine(input_dim, hidden_dim)

    def reset_parameters(self) -> None:
        gain = nn.init.calculate_gain("tanh")
        for conv in (self.conv_zr, self.conv_h1, self.conv_h2):
            nn.init.xavier_uniform_(conv.weight, gain=gain)
            conv.bias.data.zero_()

    def regrouped(self):
        """(Wx, bx, Wh, bh): x-part [z|r|n] and h-part [z|r|h2] of the three convolutions (autograd-tracked cats)."""
        ci, hid = self.input_dim, self.hidden_dim
        wzr = self.conv_zr.weight
        if wzr.is_cuda and not os.environ.get("SF_NO_PARAM_BLOCKS"):
            # one launch (and one for all six gradients) instead of four cats, two copies and a fill: parameters as [rows, in * 9] blocks
            params = (wzr, self.conv_h1.weight, self.conv_h2.weight, self.conv_zr.bias, self.conv_h1.bias, self.conv_h2.bias)
            blocks = ((0, 0, 0, 0, 0, 0, 2 * hid, ci * 9), (1, 0, 0, 0, 2 * hid, 0, hid, ci * 9),           # Wx = [zr's x columns ; h1]
                      (3, 0, 0, 1, 0, 0, 2 * hid, 1), (4, 0, 0, 1, 2 * hid, 0, hid, 1),                        # bx = [b_zr ; b_h1]
                      (0, 0, ci * 9, 2, 0, 0, 2 * hid, hid * 9), (2, 0, 0, 2, 2 * hid, 0, hid, hid * 9),       # Wh = [zr's h columns ; h2]
                      (None, 0, 0, 3, 0, 0, 2 * hid, 1), (5, 0, 0, 3, 2 * hid, 0, hid, 1))                     # bh = [0 ; b_h2]
            return F.param_blocks(((3 * hid, ci, 3, 3), (3 * hid,), (3 * hid, hid, 3, 3), (3 * hid,)), blocks, params)
        Wx = torch.cat((wzr[:, :ci], self.conv_h1.weight), 0)
        bx = torch.cat((self.conv_zr.bias, self.conv_h1.bias), 0)
        Wh = torch.cat((wzr[:, ci:], self.conv_h2.weight), 0)
        bh = torch.cat((torch.zeros_like(self.conv_zr.bias), self.conv_h2.bias), 0)
        return Wx.contiguous(), bx, Wh.contiguous(), bh

    def run_sequence(self, x: Tensor, T_steps: int):
        # packed-weight cache key: the six source parameters (the regrouped cats are new tensors every call)
        src = tuple((p.data_ptr(), p._version) for conv in (self.conv_zr, self.conv_h1, self.conv_h2) for p in (conv.weight, conv.bias))
        return F.convgru_sequence(self._eng, x, T_steps, *self.regrouped(), src_key=src)


class ConvGRU(nn.Module):
    def __init__(self, input_dim: int, hidden_dim: int, kernel_size=(3, 3), n_layers: int = 1, batch_first: bool = True,
                 bias: bool = True, input_p: float = 0.2, hidden_p: float = 0.1):
        super().__init__()
        self.n_layers, self.batch_first, self.hidden_dim = n_layers, batch_first, hidden_dim
        self.input_p, self.hidden_p = input_p, hidden_p
        self.cell_list = nn.ModuleList(
            [ConvGRUCell(input_dim if i == 0 else hidden_dim, hidden_dim, kernel_size, bias) for i in range(n_layers)]
        )

    def run(self, x: Tensor, T_steps: int, n: int, input_dropout_done: bool = False):
        """x ``[T*n,h,w,Cp]`` time-major -> (last layer's states ``[T*n,h,w,hidp]``, [last state per layer])."""
        if self.training and self.input_p > 0 and not input_dropout_done:
            # sequence-consistent ("RNN") dropout on the input [RECALLED upstream]: one mask shared by all timesteps
            x = F.dropout2(x, 0.0, self.input_p, x.numel() // T_steps)
        last: List[Tensor] = []
        seq = x
        for i, cell in enumerate(self.cell_list):
            seq, h_last = cell.run_sequence(seq, T_steps)
            last.append(h_last)
            if self.training and self.hidden_p > 0 and i + 1 < self.n_layers:
                seq = F.dropout2(seq, self.hidden_p, 0.0, seq.numel())
        return seq, last


class TemporalEncoder(nn.Module):
    def __init__(self, in_channels: int, out_channels: int = 384, ks: int = 3, n_layers: int = 1):
        super().__init__()
        self.rnn = ConvGRU(in_channels, out_channels, (ks, ks), n_layers, batch_first=True)


class SelfAttention(nn.Module):
    """Parameter container of lucidrains ``SelfAttention(dim, heads)``."""

    def __init__(self, dim: int, heads: int):
        super().__init__()
        self.heads, self.dim_heads = heads, dim // heads
        self.to_q = nn.Linear(dim, dim, bias=False)
        self.to_kv = nn.Linear(dim, 2 * dim, bias=False)
        self.to_out = nn.Linear(dim, dim)


class PermuteToFrom(nn.Module):
    def __init__(self, permutation, fn):
        super().__init__()
        self.fn = fn
        self.permutation = list(permutation)


class AxialAttention(nn.Module):
    """``AxialAttention(dim, dim_index=1, heads=8, num_dimensions=2)``: attention along H plus attention along W, summed."""

    def __init__(self, dim: int, num_dimensions: int = 2, heads: int = 8, dim_heads=None, dim_index: int = 1, sum_axial_out: bool = True):
        super().__init__()
        if num_dimensions != 2 or dim_index != 1 or not sum_axial_out or dim_heads is not None:
            raise NotImplementedError("only the configuration MetNet uses is on the path")
        assert dim % heads == 0, "hidden_dim must be divisible by the number of heads (8)"
        self.dim, self.heads = dim, heads
        # upstream permutations for dim_index=1: axis H first ([0,3,2,1]), then axis W ([0,2,3,1])
        self.axial_attentions = nn.ModuleList([PermuteToFrom(p, SelfAttention(dim, heads)) for p in ([0, 3, 2, 1], [0, 2, 3, 1])])

    def run(self, x: Tensor) -> Tensor:
        """NHWC ``[n,h,w,hidp] -> [n,h,w,hidp]``: one projection GEMM, the attention core, one output GEMM."""
        a0, a1 = self.axial_attentions[0].fn, self.axial_attentions[1].fn
        hid, hidp = self.dim, x.shape[-1]

        def lanes(w: Tensor, parts: int) -> Tensor:  # [parts*hid, K] -> [parts*hidp, K] (pad each part's rows)
            if hid == hidp:
                return w
            w = w.view(parts, hid, w.shape[1])
            return torch.nn.functional.pad(w, (0, 0, 0, hidp - hid)).reshape(parts * hidp, -1)

        if hid == hidp and x.is_cuda and x.dtype == torch.float32 and not os.environ.get("SF_AXIAL_NESTED") and not os.environ.get("SF_NO_PARAM_BLOCKS"):
            return F.axial_layer(x.contiguous(), a0, a1, hid, self.heads)   # one autograd node (round 6; SF_AXIAL_NESTED=1: the nested form below)
        if hid == hidp and x.is_cuda and not os.environ.get("SF_NO_PARAM_BLOCKS"):
            # both projection matrices (and all six weight gradients) in one sf_copy_blocks launch each way
            params = (a0.to_q.weight, a0.to_kv.weight, a1.to_q.weight, a1.to_kv.weight, a0.to_out.weight, a1.to_out.weight)
            blocks = ((0, 0, 0, 0, 0, 0, hid, hid), (1, 0, 0, 0, hid, 0, 2 * hid, hid), (2, 0, 0, 0, 3 * hid, 0, hid, hid), (3, 0, 0, 0, 4 * hid, 0, 2 * hid, hid),
                      (4, 0, 0, 1, 0, 0, hid, hid), (5, 0, 0, 1, 0, hid, hid, hid))
            w_in, w_out = F.param_blocks(((6 * hid, hid), (hid, 2 * hid)), blocks, params)
        else:
            w_in = torch.cat((lanes(a0.to_q.weight, 1), lanes(a0.to_kv.weight, 2), lanes(a1.to_q.weight, 1), lanes(a1.to_kv.weight, 2)), 0)
            pad_k = lambda w: w if hid == hidp else torch.nn.functional.pad(w, (0, hidp - hid))
            w_out = torch.cat((pad_k(a0.to_out.weight), pad_k(a1.to_out.weight)), 1)  # [hid, 2*hidp]
        qkv = F.linear(x, w_in, None, 6 * hidp)
        att = F.attention_core(qkv, hid, self.heads)  # [n,h,w,2*hidp] = [axis0 | axis1]
        return F.linear(att, w_out, a0.to_out.bias + a1.to_out.bias, hidp)

    def forward(self, x: Tensor) -> Tensor:
        """``[B,C,H,W] -> [B,C,H,W]`` (module-surface form)."""
        require_device(x, "x")
        return F.nhwc_to_nchw(self.run(F.nchw_to_nhwc(x.float())), self.dim)


class MetNet(nn.Module):
    def __init__(
        self,
        image_encoder: str = "downsampler",
        input_channels: int = 12,
        sat_channels: int = 12,
        input_size: int = 256,
        output_channels: int = 12,
        hidden_dim: int = 64,
        kernel_size: int = 3,
        num_layers: int = 1,
        num_att_layers: int = 1,
        forecast_steps: int = 48,
        temporal_dropout: float = 0.2,
        space2depth_order: str = "pixel_unshuffle",
        **kwargs,
    ):
        """Keyword surface of upstream ``metnet.MetNet`` as called from reference ``pl_metnet.py:46-59``
        (``head=`` and other extras are swallowed by ``**kwargs`` as upstream does).

        ``space2depth_order`` (not an upstream keyword): channel order of the preprocessor's space-to-depth that ``conv1``'s
        weight columns are laid out for - ``"pixel_unshuffle"`` (``c*4 + dh*2 + dw``, torch's ``PixelUnshuffle``; default) or ``"einops"``
        (``(dh*2 + dw)*C + c``, the reference's in-tree ``space_to_depth``, ``satflow/models/utils.py:48-60``) - for checkpoints trained with
        either upstream variant.  The kernels' data layout does not change: the image part of ``conv1.weight`` is read through a column
        permutation (one small gather per step; its gradient scatters back through it)."""
        super().__init__()
        if space2depth_order not in ("pixel_unshuffle", "einops"):
            raise ValueError(f"space2depth_order={space2depth_order!r}: 'pixel_unshuffle' or 'einops'")
        self.space2depth_order = space2depth_order
        if image_encoder not in ("downsampler", "default"):
            raise NotImplementedError(f"image_encoder={image_encoder!r}: only the DownSampler encoder is on the hot path")
        self.forecast_steps, self.input_channels, self.output_channels = forecast_steps, input_channels, output_channels
        self.sat_channels, self.input_size, self.hidden_dim = sat_channels, input_size, hidden_dim
        self.preprocessor = MetNetPreprocessor(sat_channels=sat_channels, crop_size=input_size, use_space2depth=True, split_input=True)
        self.image_channels = input_channels - sat_channels + sat_channels * 8
        self.drop = nn.Dropout(temporal_dropout)
        encoder = DownSampler(self.image_channels + forecast_steps)
        self.image_encoder = TimeDistributed(encoder)
        self.ct = ConditionTime(forecast_steps)
        self.temporal_enc = TemporalEncoder(encoder.output_channels, hidden_dim, ks=kernel_size, n_layers=num_layers)
        self.temporal_agg = nn.Sequential(*[AxialAttention(dim=hidden_dim, dim_index=1, heads=8, num_dimensions=2) for _ in range(num_att_layers)])
        self.head = nn.Conv2d(hidden_dim, output_channels, kernel_size=(1, 1))
        # conv1 restricted to the image lanes (the one-hot lead-time lanes are folded into the first pooling)
        self._conv1 = ConvEngine([self.image_channels], 160)
        # data lane (pixel-unshuffle order: [centre crop | 2x2 mean] x (c*4 + d), then the other channels) -> column of conv1.weight in "einops" order
        sat, cols = sat_channels, list(range(self.image_channels))
        for half in range(2):
            for c in range(sat):
                for d in range(4):
                    cols[half * 4 * sat + c * 4 + d] = half * 4 * sat + d * sat + c
        self.register_buffer("_s2d_cols", torch.tensor(cols, dtype=torch.long), persistent=False)

    def forward(self, imgs: Tensor, lead_time: int = 0) -> Tensor:
        """``imgs[B,T,C,4*input_size,4*input_size] -> [B, forecast_steps, output_channels, input_size//4, input_size//4]``."""
        from .. import kernels as K

        require_device(imgs, "imgs")
        B, Tn, C, H, W = imgs.shape
        if C != self.input_channels:
            raise RuntimeError(f"expected {self.input_channels} input channels, got {C}")
        L, S = self.forecast_steps, self.input_size
        enc: DownSampler = self.image_encoder.module
        F_ = Tn * B
        st = encoder_storage_dtype()  # fp32, or bf16 in "bf16a" mode (encoder activations only)
        frames = F.metnet_preprocess(imgs.float(), self.sat_channels, S, st)  # [T*B, S, S, Cimg_p], computed once
        # conv1: its image part once per frame; ConditionTime's one-hot planes (reference layers/ConditionTime.py:22-33)
        # contribute a per-lead-time, border-aware constant that is added inside the fused first pooling
        c1 = enc.module[0]
        cimg = self.image_channels
        # the convolution is handed a fresh slice of c1.weight every call: its packed-weight cache keys on the LIVE parameters
        # (looked up now, not captured at construction - they may have been replaced by load_state_dict(assign=True) / re-assignment)
        self._conv1.key_tensors = (c1.weight, c1.bias)
        w_img = c1.weight[:, :cimg].contiguous() if self.space2depth_order == "pixel_unshuffle" else c1.weight[:, :cimg].index_select(1, self._s2d_cols)
        base = F.conv3x3(self._conv1, frames, w_img, c1.bias, out_dtype=st)  # [T*B, S, S, 160]
        if enc.capture is not None:
            enc.capture["base"] = base.detach()
        # [L*T*B, S/2, S/2, 160], image (l*F + f); in training mode with the per-lead-time sums its BatchNorm needs
        p1, p1_stats = F.leadtime_pool(base, c1.weight, cimg, L, want_stats=True) if self.training else (F.leadtime_pool(base, c1.weight, cimg, L), None)
        # rest of the DownSampler with per-lead-time BatchNorm batches; the last pooling also re-orders
        # images from [lead][time][batch] to [time][lead][batch] for the recurrent part
        rnn = self.temporal_enc.rnn
        # nn.Dropout(temporal_dropout) and the ConvGRU's sequence-consistent input dropout ride on the encoder's last pooling
        drop = (self.drop.p, rnn.input_p, Tn) if self.training else None
        feat = enc.run(None, L, pooled=p1, perm=(L, Tn), dropout=drop, out_dtype=st, pooled_stats=p1_stats)  # [T*L*B, S/4, S/4, 256]
        _, last = rnn.run(feat, Tn, L * B, input_dropout_done=True)
        a = last[-1]  # [L*B, s, s, hidp]
        for layer in self.temporal_agg:
            a = layer.run(a)
        hw = self.head.weight.view(self.output_channels, self.hidden_dim)
        o = F.linear(a, hw, self.head.bias)  # [L*B, s, s, outp]
        s, O = S // 4, self.output_channels
        return F._FromNHWC.apply(o, (B, L, O, s, s), B, L, O, s, s, (L * O * s * s, O * s * s, s * s))

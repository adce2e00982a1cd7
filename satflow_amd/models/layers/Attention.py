"""The in-tree attention layers on the HIP kernels - SURVEY 8f-4.  Mirror of reference ``satflow/models/layers/Attention.py``:
``SeparableAttn`` / ``SeparableAttnCell`` (``:7-109``), ``SelfAttention`` (``:112-170``), ``SelfAttention2d`` (``:173-223``) - same
constructors, parameter names and ``forward`` signatures on NCHW / NCTWH tensors.

Execution: the 1x1(x1) convolutions are pointwise linear maps on NHWC tokens (``sf_linear_*``, exact-fp32 MFMA), the max-poolings
``sf_maxpool3d_*``, the attention products ``sf_bmm_f32`` on exactly the (strided / flat-reinterpreted) views the reference multiplies,
the row softmax ``sf_softmax_rows_*`` and the residual ``gamma * out + x`` ``sf_axpy`` (+ ``sf_dot`` for gamma's gradient).
PyTorch only re-orders memory (``permute().contiguous()`` where the reference's ``.view`` needs a particular layout).
"""
from __future__ import annotations

import torch
import torch.nn as nn
from torch import Tensor
from torch.nn import functional as TF
from torch.nn import init

from ... import functional as F
from ... import functional_gan as FG
from ..._hip import cpad, require_device


def _tokens(x: Tensor) -> Tensor:
    """``[B, C, D0, D1, D2]`` (any strides) -> NHWC tokens ``[B, D0, D1, D2, Cp]`` (batch-major)."""
    B, C, d0, d1, d2 = x.shape
    xc = x.float().contiguous()
    # the layout kernel's image (batch, time) pair is used as (d0 index, batch): output image index = b * d0 + i0
    t = F._ToNHWC.apply(xc, d0, B, C, d1, d2, (d1 * d2, C * d0 * d1 * d2, d0 * d1 * d2))
    return t.view(B, d0, d1, d2, t.shape[-1])


def _untokens(t: Tensor, C: int) -> Tensor:
    """NHWC tokens ``[B, D0, D1, D2, Cp]`` -> contiguous ``[B, C, D0, D1, D2]``."""
    B, d0, d1, d2, cp = t.shape
    return F._FromNHWC.apply(t.reshape(B * d0, d1, d2, cp), (B, C, d0, d1, d2), d0, B, C, d1, d2, (d1 * d2, C * d0 * d1 * d2, d0 * d1 * d2))


def _project(tokens: Tensor, conv: nn.Module, out_lanes=None) -> Tensor:
    w = conv.weight
    return F.linear(tokens, w.reshape(w.shape[0], w.shape[1]), conv.bias, out_lanes=out_lanes, lowp=True)   # a 1x1 convolution under autocast


class SeparableAttn(nn.Module):
    def __init__(self, in_dim, activation=TF.relu, pooling_factor=2, padding_mode="constant", padding_value=0):
        super().__init__()
        self.model = nn.Sequential(
            SeparableAttnCell(in_dim, "T", activation, pooling_factor, padding_mode, padding_value),
            SeparableAttnCell(in_dim, "W", activation, pooling_factor, padding_mode, padding_value),
            SeparableAttnCell(in_dim, "H", activation, pooling_factor, padding_mode, padding_value),
        )

    def forward(self, x):
        return self.model(x)


class SeparableAttnCell(nn.Module):
    def __init__(self, in_dim, attn_id=None, activation=TF.relu, pooling_factor=2, padding_mode="constant", padding_value=0):
        super().__init__()
        self.attn_id = attn_id
        self.activation = activation
        self.in_dim = in_dim
        self.query_conv = nn.Conv3d(in_channels=in_dim, out_channels=in_dim // 2, kernel_size=1)
        self.key_conv = nn.Conv3d(in_channels=in_dim, out_channels=in_dim // 2, kernel_size=1)
        self.value_conv = nn.Conv3d(in_channels=in_dim, out_channels=in_dim, kernel_size=1)
        self.pooling = nn.MaxPool3d(kernel_size=(2, 1, 1), stride=(pooling_factor, 1, 1))
        self.pooling_factor = pooling_factor
        self.padding_mode = padding_mode
        self.padding_value = padding_value
        self.gamma = nn.Parameter(torch.zeros((1,)))
        self.softmax = nn.Softmax(dim=-1)

    def init_conv(self, conv, glu=True):
        init.xavier_uniform_(conv.weight)
        if conv.bias is not None:
            conv.bias.data.zero_()

    def forward(self, x):
        require_device(x, "x")
        B, C, Tn, W, H = x.size()
        assert Tn % 2 == 0 and W % 2 == 0 and H % 2 == 0, "T, W, H is not even"
        pf, d = self.pooling_factor, self.in_dim // 2
        if self.attn_id == "T":
            A, out = Tn, x
        elif self.attn_id == "W":
            A, out = W, x.transpose(2, 3)
        else:
            A, out = H, x.transpose(2, 4)
        tok = _tokens(out)                                              # [B, A, r1, r2, Cp]
        q = _untokens(_project(tok, self.query_conv), d)                # [B, d, A, r1, r2] contiguous, as the reference's conv output
        k = _untokens(FG.max_pool3(_project(tok, self.key_conv), (2, 1, 1), (pf, 1, 1)), d)
        v = _untokens(FG.max_pool3(_project(tok, self.value_conv, out_lanes=cpad(C)), (2, 1, 1), (pf, 1, 1)), C)
        # the reference's flat views (Attention.py:86-98)
        query, key, value = q.view(B, A, -1), k.view(B, -1, A // pf), v.view(B, -1, A // pf)
        score = FG.softmax_last(FG.bmm(query, key, lowp=True))                      # [B, A, A // pf]
        o = FG.bmm(value, score.transpose(2, 1), lowp=True)                         # [B, C*r1*r2, A]
        if self.attn_id == "T":
            o = o.view(B, C, W, H, Tn).permute(0, 1, 4, 2, 3)
        elif self.attn_id == "W":
            o = o.view(B, C, Tn, H, W).permute(0, 1, 2, 4, 3)
        else:
            o = o.view(B, C, Tn, W, H)
        return FG.gamma_residual(o.contiguous(), x.float().contiguous(), self.gamma)


class SelfAttention(nn.Module):
    def __init__(self, in_dim, activation=TF.relu, pooling_factor=2):
        super().__init__()
        self.activation = activation
        self.in_dim = in_dim
        self.query_conv = nn.Conv3d(in_channels=in_dim, out_channels=in_dim // 2, kernel_size=1)
        self.key_conv = nn.Conv3d(in_channels=in_dim, out_channels=in_dim // 2, kernel_size=1)
        self.value_conv = nn.Conv3d(in_channels=in_dim, out_channels=in_dim, kernel_size=1)
        self.pooling = nn.MaxPool3d(kernel_size=2, stride=pooling_factor)
        self._stride = pooling_factor
        self.pooling_factor = pooling_factor**3
        self.gamma = nn.Parameter(torch.zeros(1))
        self.softmax = nn.Softmax(dim=-1)

    def init_conv(self, conv, glu=True):
        init.xavier_uniform_(conv.weight)
        if conv.bias is not None:
            conv.bias.data.zero_()

    def forward(self, x):
        require_device(x, "x")
        if len(x.size()) == 4:
            batch_size, C, W, H = x.size()
            Tn = 1
        else:
            batch_size, C, Tn, W, H = x.size()
        assert Tn % 2 == 0 and W % 2 == 0 and H % 2 == 0, "T, W, H is not even"  # (a 4-D input has T = 1: the reference asserts too)
        B, N, s = batch_size, Tn * W * H, self._stride
        tok = _tokens(x)                                                 # [B, T, W, H, Cp]
        cp = tok.shape[-1]
        q = _project(tok, self.query_conv).view(B, N, -1)
        k = FG.max_pool3(_project(tok, self.key_conv), (2, 2, 2), (s, s, s))
        v = FG.max_pool3(_project(tok, self.value_conv, out_lanes=cp), (2, 2, 2), (s, s, s))
        nk = k.shape[1] * k.shape[2] * k.shape[3]
        if nk != N // self.pooling_factor:
            raise RuntimeError(f"shape '[{B}, -1, {N // self.pooling_factor}]' is invalid for the pooled keys ({nk} positions)")  # the reference's .view fails
        score = FG.softmax_last(FG.bmm(q, k.view(B, nk, -1).transpose(1, 2), lowp=True))   # [B, N, N / pf^3]
        out = FG.bmm(score, v.view(B, nk, cp), lowp=True).view(B, Tn, W, H, cp)
        return _untokens(FG.gamma_residual(out, tok, self.gamma), C)


class SelfAttention2d(nn.Module):
    r"""Self attention of SAGAN as the reference writes it (``Attention.py:173-223``): ``attention = softmax(key^T query)`` normalised
    over the query index, ``output = gamma * value * attention + x``."""

    def __init__(self, input_dims, output_dims=None, return_attn=False):
        output_dims = input_dims // 8 if output_dims is None else output_dims
        if output_dims == 0:
            raise Exception("The output dims corresponding to the input dims is 0. Increase the input dims to 8 or more. Else specify output_dims")
        super().__init__()
        self.query = nn.Conv2d(input_dims, output_dims, 1)
        self.key = nn.Conv2d(input_dims, output_dims, 1)
        self.value = nn.Conv2d(input_dims, input_dims, 1)
        self.gamma = nn.Parameter(torch.zeros(1))
        self.return_attn = return_attn
        self.input_dims = input_dims

    def forward(self, x):
        require_device(x, "x")
        B, C, H, W = x.shape
        n = H * W
        tok = F.nchw_to_nhwc(x.float())                                  # [B, H, W, Cp]
        cp = tok.shape[-1]
        q = _project(tok, self.query).view(B, n, -1)
        k = _project(tok, self.key).view(B, n, -1)
        v = _project(tok, self.value, out_lanes=cp).view(B, n, cp)
        attn = FG.softmax_last(FG.bmm(k, q.transpose(1, 2), lowp=True))             # attn[i, j] = key_i . query_j, softmax over j
        out = FG.bmm(attn.transpose(1, 2), v, lowp=True).view(B, H, W, cp)          # out[j] = sum_i attn[i, j] value_i
        res = F.nhwc_to_nchw(FG.gamma_residual(out, tok, self.gamma), C)
        return (res, attn) if self.return_attn else res

"""Oracle: MetNet stack as executed by ``LitMetNet`` (CPU, torch fp32, functional).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).

Pinned pieces (checked against the in-tree reference copies by
``tests/golden/make_golden.py``): ``condition_time`` (reference
``satflow/models/layers/ConditionTime.py:5-33``), ``time_distributed``
(``layers/TimeDistributed.py:21-40``), ``space_to_depth``
(``satflow/models/utils.py:48-70``).

PARITY UNPINNED for everything else: the arithmetic of ``metnet.MetNet``
(reference call site ``satflow/models/pl_metnet.py:6,46-59,65``) lives in the
un-vendored packages ``metnet>=0.0.3`` (``requirements.txt:18``, lower bound
only) and, transitively, ``axial_attention`` (lucidrains).  The functions
below restate the published upstream algorithm; the only
reference-held check is the shape/NaN test ``tests/test_models.py:42-61``.

Parameter dictionaries use the upstream ``MetNet.state_dict()`` key names
(``image_encoder.module.module.0.weight`` ... ``head.bias``).
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]


# ----------------------------------------------------------------------------------------------
# pinned helpers
# ----------------------------------------------------------------------------------------------
def condition_time(x: Tensor, fstep: int, horizon: int) -> Tensor:
    """Append ``horizon`` one-hot lead-time planes on dim 2 of ``x[B,T,C,H,W]``.

    Reference ``layers/ConditionTime.py:22-33`` (5-D mode): plane ``fstep`` is
    all ones, the others zero; ``assert i < seq_len`` (``:7``).
    """
    assert fstep < horizon
    B, T, _, H, W = x.shape
    planes = x.new_zeros(B, T, horizon, H, W)
    planes[:, :, fstep] = 1
    return torch.cat((x, planes), 2)


def time_distributed(fn: Callable[[Tensor], Tensor], x: Tensor) -> Tensor:
    """Fold time into batch, apply ``fn``, unfold (``layers/TimeDistributed.py:21-29,42-46``)."""
    B, T = x.shape[:2]
    y = fn(x.reshape(B * T, *x.shape[2:]))
    return y.reshape(B, T, *y.shape[1:])


def space_to_depth(frames: Tensor, block: int) -> Tensor:
    """Channels-last space-to-depth, channel order ``(dh dw c)`` (``models/utils.py:48-60``)."""
    B, H, W, C = frames.shape
    v = frames.reshape(B, H // block, block, W // block, block, C)
    return v.permute(0, 1, 3, 2, 4, 5).reshape(B, H // block, W // block, block * block * C)


# ----------------------------------------------------------------------------------------------
# upstream restatement (unpinned)
# ----------------------------------------------------------------------------------------------
def center_crop(x: Tensor, size: int) -> Tensor:
    """torchvision ``CenterCrop`` on the last two dims (top = round((H-size)/2))."""
    H, W = x.shape[-2:]
    top, left = int(round((H - size) / 2.0)), int(round((W - size) / 2.0))
    return x[..., top : top + size, left : left + size]


def preprocess(x: Tensor, sat_channels: int, crop_size: int, order: str = "pixel_unshuffle") -> Tensor:
    """``MetNetPreprocessor(sat_channels, crop_size, use_space2depth=True, split_input=True)``.

    ``x[B,T,C,H,W]``.  Satellite channels: PixelUnshuffle(2) (channel order
    ``c*4 + dh*2 + dw``), then [centre crop ; 2x2 mean] on the channel axis;
    remaining channels: 2x2 mean then centre crop.  Output
    ``[B,T,8*sat+(C-sat),crop,crop]``.

    ``order="einops"``: the space-to-depth channel order of the reference's own in-tree ``space_to_depth``
    (``satflow/models/utils.py:48-60``: ``"b (h dh) (w dw) c -> b h w (dh dw c)"``, channel ``(dh*2 + dw)*C + c``) - what some upstream
    versions of the preprocessor use; computed here with that very function (``space_to_depth`` above, pinned by ``metnet_layers.npz``).
    """
    B, T, C, H, W = x.shape
    if order == "einops":
        sat = space_to_depth(x[:, :, :sat_channels].reshape(B * T, sat_channels, H, W).permute(0, 2, 3, 1), 2).permute(0, 3, 1, 2)
    else:
        assert order == "pixel_unshuffle", order
        sat = F.pixel_unshuffle(x[:, :, :sat_channels].reshape(B * T, sat_channels, H, W), 2)
    parts = [center_crop(sat, crop_size), F.avg_pool2d(sat, 2)]
    if C > sat_channels:
        other = F.avg_pool2d(x[:, :, sat_channels:].reshape(B * T, C - sat_channels, H, W), 2)
        parts.append(center_crop(other, crop_size))
    out = torch.cat(parts, 1)
    return out.reshape(B, T, *out.shape[1:])


def batch_norm_train(x: Tensor, weight: Tensor, bias: Tensor, eps: float = 1e-5) -> Tensor:
    """Training-mode BatchNorm2d (biased batch statistics over N,H,W)."""
    return F.batch_norm(x, None, None, weight, bias, training=True, eps=eps)


def max_pool2(t: Tensor, routing: Tensor | None = None) -> Tensor:
    """``nn.MaxPool2d(2, stride 2)``.  ``routing`` (bool, shape of ``t``, exactly one True per 2x2 window): the window
    element to take instead of recomputing the argmax - used by the tie-aware parity tests, which inject the routing the
    implementation under test actually chose (max-pooling's gradient is discontinuous where two candidates of a window
    agree to rounding; the VALUE taken is the same to rounding either way, the gradient then follows the same path)."""
    if routing is None:
        return F.max_pool2d(t, 2)
    n, c, h, w = t.shape
    picked = (t * routing.to(t.dtype)).view(n, c, h // 2, 2, w // 2, 2)
    return picked.sum(dim=(3, 5))


def downsampler(x: Tensor, p: Params, prefix: str, bn_stats: Dict[str, Tuple[Tensor, Tensor]] | None,
                routing: Tuple[Tensor | None, Tensor | None] = (None, None)) -> Tensor:
    """``DownSampler``: conv3x3 -> maxpool2 -> BN -> conv -> BN -> conv -> BN -> conv -> maxpool2.

    No activation functions.  ``bn_stats=None`` selects training-mode batch
    statistics; otherwise ``{"3": (mean, var), ...}`` gives eval-mode running
    statistics per BatchNorm index.  ``routing``: see ``max_pool2`` (first / last pooling).
    """

    def conv(i: int, t: Tensor) -> Tensor:
        return F.conv2d(t, p[f"{prefix}.{i}.weight"], p[f"{prefix}.{i}.bias"], padding=1)

    def bn(i: int, t: Tensor) -> Tensor:
        w, b = p[f"{prefix}.{i}.weight"], p[f"{prefix}.{i}.bias"]
        if bn_stats is None:
            return batch_norm_train(t, w, b)
        mean, var = bn_stats[str(i)]
        return F.batch_norm(t, mean, var, w, b, training=False, eps=1e-5)

    t = bn(3, max_pool2(conv(0, x), routing[0]))
    t = bn(5, conv(4, t))
    t = bn(7, conv(6, t))
    return max_pool2(conv(8, t), routing[1])


def convgru_cell(x: Tensor, h: Tensor, p: Params, prefix: str, operand=None) -> Tensor:
    """``ConvGRUCell``: ``z,r = split(sigmoid(conv_zr([x;h])))``; ``n = tanh(conv_h1(x) + r*conv_h2(h))``;
    ``h' = (1-z)*n + z*h``.

    ``operand`` (tests of the bf16 kernels): a rounding applied to every convolution OPERAND (inputs and weights) and to nothing
    else - what a bf16-operand / fp32-accumulate convolution sees; the blend keeps the unrounded state."""
    op = operand if operand is not None else (lambda t: t)
    hid = h.shape[1]
    pad = p[f"{prefix}.conv_zr.weight"].shape[-1] // 2
    zr = torch.sigmoid(
        F.conv2d(op(torch.cat((x, h), 1)), op(p[f"{prefix}.conv_zr.weight"]), p[f"{prefix}.conv_zr.bias"], padding=pad)
    )
    z, r = zr[:, :hid], zr[:, hid:]
    n = torch.tanh(
        F.conv2d(op(x), op(p[f"{prefix}.conv_h1.weight"]), p[f"{prefix}.conv_h1.bias"], padding=pad)
        + r * F.conv2d(op(h), op(p[f"{prefix}.conv_h2.weight"]), p[f"{prefix}.conv_h2.bias"], padding=pad)
    )
    return (1 - z) * n + z * h


def convgru(x: Tensor, p: Params, prefix: str, num_layers: int, operand=None) -> Tuple[Tensor, List[Tensor]]:
    """Multi-layer ConvGRU over ``x[B,T,C,H,W]`` with zero initial state (dropout off).

    Returns ``(layer_output[B,T,hid,H,W] of the last layer, [last h of every layer])``.
    """
    B, T, _, H, W = x.shape
    seq = [x[:, t] for t in range(T)]
    last: List[Tensor] = []
    for layer in range(num_layers):
        cell = f"{prefix}.cell_list.{layer}"
        hid = p[f"{cell}.conv_h2.weight"].shape[0]
        h = x.new_zeros(B, hid, H, W)
        outs = []
        for t in range(T):
            h = convgru_cell(seq[t], h, p, cell, operand)
            outs.append(h)
        seq = outs
        last.append(h)
    return torch.stack(seq, 1), last


def self_attention(seq: Tensor, p: Params, prefix: str, heads: int) -> Tensor:
    """lucidrains ``SelfAttention`` on ``seq[b, t, dim]`` (q/kv without bias, out with bias)."""
    b, t, d = seq.shape
    e = d // heads
    q = seq @ p[f"{prefix}.to_q.weight"].t()
    kv = seq @ p[f"{prefix}.to_kv.weight"].t()
    k, v = kv[..., :d], kv[..., d:]
    split = lambda u: u.reshape(b, t, heads, e).transpose(1, 2)  # [b, heads, t, e]
    q, k, v = split(q), split(k), split(v)
    dots = torch.softmax((q @ k.transpose(-1, -2)) * e**-0.5, dim=-1)
    out = (dots @ v).transpose(1, 2).reshape(b, t, d)
    return out @ p[f"{prefix}.to_out.weight"].t() + p[f"{prefix}.to_out.bias"]


def axial_attention(x: Tensor, p: Params, prefix: str, heads: int = 8) -> Tensor:
    """``AxialAttention(dim, dim_index=1, heads=8, num_dimensions=2)`` on ``x[B,C,H,W]``.

    Attention 0 runs along H (one sequence per column), attention 1 along W
    (one sequence per row); the two outputs are summed (``sum_axial_out``).
    """
    B, C, H, W = x.shape
    along_h = x.permute(0, 3, 2, 1).reshape(B * W, H, C)
    a0 = self_attention(along_h, p, f"{prefix}.axial_attentions.0.fn", heads)
    a0 = a0.reshape(B, W, H, C).permute(0, 3, 2, 1)
    along_w = x.permute(0, 2, 3, 1).reshape(B * H, W, C)
    a1 = self_attention(along_w, p, f"{prefix}.axial_attentions.1.fn", heads)
    a1 = a1.reshape(B, H, W, C).permute(0, 3, 1, 2)
    return a0 + a1


def metnet_forward(
    imgs: Tensor,
    p: Params,
    *,
    sat_channels: int,
    input_size: int,
    forecast_steps: int,
    num_layers: int = 1,
    num_att_layers: int = 1,
    bn_stats: Dict[str, Tuple[Tensor, Tensor]] | None = None,
    pool_routing: Dict[Tuple[str, int], Tensor] | None = None,
    feature_scale: Dict[int, Tensor] | None = None,
    space2depth_order: str = "pixel_unshuffle",
) -> Tensor:
    """``MetNet.forward(imgs[B,T,C,H,W]) -> [B, forecast_steps, out, input_size//4, input_size//4]``.

    ``pool_routing`` (tests only): ``{("p1", lead): bool[B*T,160,S,S], ("p2", lead): bool[B*T,256,S/2,S/2]}`` - the
    max-pool routing to follow per lead time (frames in ``b*T + t`` order), see ``max_pool2``.

    Per lead time ``i`` (upstream recomputes everything per lead time, SURVEY
    3.2): preprocess -> ConditionTime(i) -> TimeDistributed(DownSampler) ->
    ConvGRU (last state of last layer) -> axial attention layers -> 1x1 head;
    stacked on dim 1.  Dropouts are identity (``temporal_dropout=0`` / eval) unless ``feature_scale`` (tests only) gives, per lead
    time, the keep-scale ``[B,T,256,s,s]`` of ``nn.Dropout(temporal_dropout)`` times the ConvGRU's sequence-consistent input dropout:
    the masks are then REPLAYED (they come from the kernels' counter-based generator), not drawn.
    """
    outs = []
    base = preprocess(imgs, sat_channels, input_size, space2depth_order)
    for i in range(forecast_steps):
        t = condition_time(base, i, forecast_steps)
        rt = (None, None) if pool_routing is None else (pool_routing.get(("p1", i)), pool_routing.get(("p2", i)))
        t = time_distributed(lambda f: downsampler(f, p, "image_encoder.module.module", bn_stats, rt), t)
        if feature_scale is not None:
            t = t * feature_scale[i]
        _, last = convgru(t, p, "temporal_enc.rnn", num_layers)
        a = last[-1]
        for layer in range(num_att_layers):
            a = axial_attention(a, p, f"temporal_agg.{layer}")
        outs.append(F.conv2d(a, p["head.weight"], p["head.bias"]))
    return torch.stack(outs, 1)


def init_params(
    *,
    input_channels: int,
    sat_channels: int,
    output_channels: int,
    hidden_dim: int,
    forecast_steps: int,
    kernel_size: int = 3,
    num_layers: int = 1,
    num_att_layers: int = 1,
    seed: int = 0,
    encoder_channels: int = 256,
) -> Params:
    """Random parameters with upstream shapes/keys (torch default-like fan-in uniform init)."""
    g = torch.Generator().manual_seed(seed)

    def uni(shape, fan_in):
        b = 1.0 / math.sqrt(fan_in)
        return (torch.rand(shape, generator=g) * 2 - 1) * b

    p: Params = {}
    cin = sat_channels * 8 + (input_channels - sat_channels) + forecast_steps
    enc = "image_encoder.module.module"
    for idx, (ci, co) in {0: (cin, 160), 4: (160, encoder_channels), 6: (encoder_channels,) * 2, 8: (encoder_channels,) * 2}.items():
        p[f"{enc}.{idx}.weight"] = uni((co, ci, 3, 3), ci * 9)
        p[f"{enc}.{idx}.bias"] = uni((co,), ci * 9)
    for idx, c in {3: 160, 5: encoder_channels, 7: encoder_channels}.items():
        p[f"{enc}.{idx}.weight"] = 1 + 0.1 * (torch.rand(c, generator=g) - 0.5)
        p[f"{enc}.{idx}.bias"] = 0.1 * (torch.rand(c, generator=g) - 0.5)
    k = kernel_size
    for layer in range(num_layers):
        ci = encoder_channels if layer == 0 else hidden_dim
        cell = f"temporal_enc.rnn.cell_list.{layer}"
        p[f"{cell}.conv_zr.weight"] = uni((2 * hidden_dim, ci + hidden_dim, k, k), (ci + hidden_dim) * k * k)
        p[f"{cell}.conv_zr.bias"] = uni((2 * hidden_dim,), (ci + hidden_dim) * k * k)
        p[f"{cell}.conv_h1.weight"] = uni((hidden_dim, ci, k, k), ci * k * k)
        p[f"{cell}.conv_h1.bias"] = uni((hidden_dim,), ci * k * k)
        p[f"{cell}.conv_h2.weight"] = uni((hidden_dim, hidden_dim, k, k), hidden_dim * k * k)
        p[f"{cell}.conv_h2.bias"] = uni((hidden_dim,), hidden_dim * k * k)
    for layer in range(num_att_layers):
        for ax in range(2):
            fn = f"temporal_agg.{layer}.axial_attentions.{ax}.fn"
            p[f"{fn}.to_q.weight"] = uni((hidden_dim, hidden_dim), hidden_dim)
            p[f"{fn}.to_kv.weight"] = uni((2 * hidden_dim, hidden_dim), hidden_dim)
            p[f"{fn}.to_out.weight"] = uni((hidden_dim, hidden_dim), hidden_dim)
            p[f"{fn}.to_out.bias"] = uni((hidden_dim,), hidden_dim)
    p["head.weight"] = uni((output_channels, hidden_dim, 1, 1), hidden_dim)
    p["head.bias"] = uni((output_channels,), hidden_dim)
    return p

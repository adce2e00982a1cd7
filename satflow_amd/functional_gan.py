"""Autograd-aware ops of the in-tree DGMR / DVD-GAN style networks and attention layers over the C ABI.

As in ``functional.py`` every ``torch.autograd.Function`` is plumbing: it allocates outputs, hands raw pointers to
libsatflow_hip.so on the current stream and wires the matching backward kernels.  Activations are NHWC fp32 ``[N,H,W,Cp]``
(``Cp`` = channels padded to 16) unless a docstring says otherwise.
"""
from __future__ import annotations

from typing import Optional, Tuple

import os

import torch

from . import functional as F
from . import kernels as K
from ._hip import NULL, SF_F32, T, check, cpad, lib, require_device, stream_ptr

Tensor = torch.Tensor


def _ws(n: int, dev) -> Tensor:
    return torch.empty(max(int(n), 1), dtype=torch.float32, device=dev)


# ----------------------------------------------------------------------------------------------
# spectral normalisation (reference layers/Normalization.py:10-62)
# ----------------------------------------------------------------------------------------------
class _SpectralNormFn(torch.autograd.Function):
    """``w_bar -> w_bar / sigma`` with the power iteration advancing ``u`` / ``v`` IN PLACE (the reference assigns ``u.data`` /
    ``v.data`` in every forward call, ``Normalization.py:25-27``).  ``u`` / ``v`` are constants of the differentiation."""

    @staticmethod
    def forward(ctx, w_bar: Tensor, u: Tensor, v: Tensor, power_iterations: int):
        wb = w_bar.contiguous()
        height = wb.shape[0]
        width = wb.numel() // height
        assert u.numel() == height and v.numel() == width and u.is_contiguous() and v.is_contiguous()
        w = torch.empty_like(wb)
        sigma = torch.empty(1, dtype=torch.float32, device=wb.device)
        ws = _ws(lib().sf_spectral_norm_workspace_floats(height, width), wb.device)
        check(lib().sf_spectral_norm_fwd(wb.data_ptr(), height, width, u.data_ptr(), v.data_ptr(), int(power_iterations), w.data_ptr(), sigma.data_ptr(),
                                         ws.data_ptr(), stream_ptr()), "sf_spectral_norm_fwd")
        # the vectors this call ended with (a later call advances the module's own copies before the backward pass runs): both in ONE launch
        uv = torch.empty(height + width, dtype=torch.float32, device=wb.device)
        F._copy_blocks([(u.detach().view(1, -1), 0, 0, height, uv, 0, 0, height + width, 1, height),
                        (v.detach().view(1, -1), 0, 0, width, uv, 0, height, height + width, 1, width)])
        ctx.save_for_backward(wb, uv, sigma)
        ctx.w_bar = w_bar   # (identity: where its gradient goes)
        return w

    @staticmethod
    def backward(ctx, g: Tensor):
        wb, uv, sigma = ctx.saved_tensors
        g = g.contiguous()
        height = wb.shape[0]
        u, v = uv[:height], uv[height:]
        # (the parameter's own gradient slice when an optimizer registered it with functional.GRAD_SINK: no add_ by autograd afterwards)
        dw, dw_ret = F.grad_out(ctx.w_bar if ctx.w_bar.is_contiguous() else wb)
        ws = _ws(512, wb.device)
        check(lib().sf_spectral_norm_bwd(g.data_ptr(), wb.data_ptr(), u.data_ptr(), v.data_ptr(), sigma.data_ptr(), height, wb.numel() // height, dw.data_ptr(),
                                         ws.data_ptr(), stream_ptr()), "sf_spectral_norm_bwd")
        return dw_ret, None, None, None


def spectral_norm_weight(w_bar: Tensor, u: Tensor, v: Tensor, power_iterations: int = 1) -> Tensor:
    require_device(w_bar, "weight_bar")
    return _SpectralNormFn.apply(w_bar, u, v, power_iterations)


# ----------------------------------------------------------------------------------------------
# pooling / up-sampling / temporal taps
# ----------------------------------------------------------------------------------------------
class _Pool2Fn(torch.autograd.Function):
    """``scale * sum`` over 2x2 windows (and ``tpool`` frames, time-major with ``nb`` images per frame) ``+ addend``."""

    @staticmethod
    def forward(ctx, x: Tensor, tpool: int, nb: int, scale: float, addend: Optional[Tensor]):
        n, h, w, c = x.shape
        assert h % 2 == 0 and w % 2 == 0 and n % (tpool * nb) == 0, (x.shape, tpool, nb)
        y = torch.empty(n // tpool, h // 2, w // 2, c, dtype=torch.float32, device=x.device)
        check(lib().sf_pool2(T(x), n // tpool, h // 2, w // 2, tpool, nb, scale, T(addend) if addend is not None else NULL, T(y), stream_ptr()), "sf_pool2")
        ctx.meta = (tpool, nb, scale, tuple(x.shape))
        return y

    @staticmethod
    def backward(ctx, g: Tensor):
        tpool, nb, scale, shape = ctx.meta
        g = g.contiguous()
        gx = None
        if ctx.needs_input_grad[0]:
            gx = torch.empty(shape, dtype=torch.float32, device=g.device)
            check(lib().sf_expand2(T(g), g.shape[0], g.shape[1], g.shape[2], tpool, nb, scale, T(gx), stream_ptr()), "sf_expand2")
        return gx, None, None, None, (g if ctx.needs_input_grad[4] else None)


class _Expand2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, scale: float):
        n, h, w, c = x.shape
        y = torch.empty(n, 2 * h, 2 * w, c, dtype=torch.float32, device=x.device)
        check(lib().sf_expand2(T(x), n, h, w, 1, 1, scale, T(y), stream_ptr()), "sf_expand2")
        ctx.scale = scale
        return y

    @staticmethod
    def backward(ctx, g: Tensor):
        g = g.contiguous()
        n, h2, w2, c = g.shape
        gx = torch.empty(n, h2 // 2, w2 // 2, c, dtype=torch.float32, device=g.device)
        check(lib().sf_pool2(T(g), n, h2 // 2, w2 // 2, 1, 1, ctx.scale, NULL, T(gx), stream_ptr()), "sf_pool2")
        return gx, None


def avg_pool2(x: Tensor, addend: Optional[Tensor] = None) -> Tensor:
    """``F.avg_pool2d(x, 2) (+ addend)``."""
    return _Pool2Fn.apply(x.contiguous(), 1, 1, 0.25, addend.contiguous() if addend is not None else None)


def avg_pool3(x: Tensor, nb: int, addend: Optional[Tensor] = None) -> Tensor:
    """``F.avg_pool3d(x, 2) (+ addend)`` on time-major frames ``[T*nb,H,W,C] -> [T/2*nb,H/2,W/2,C]``."""
    return _Pool2Fn.apply(x.contiguous(), 2, nb, 0.125, addend.contiguous() if addend is not None else None)


def upsample2(x: Tensor) -> Tensor:
    """``F.interpolate(x, scale_factor=2)`` (nearest)."""
    return _Expand2Fn.apply(x.contiguous(), 1.0)


class _TimeStack3Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, Tn: int):
        n, h, w, c = x.shape
        assert n % Tn == 0
        y = torch.empty(n, h, w, 3 * c, dtype=torch.float32, device=x.device)
        check(lib().sf_time_stack3_fwd(T(x), Tn, (n // Tn) * h * w, T(y), stream_ptr()), "sf_time_stack3_fwd")
        ctx.Tn = Tn
        return y

    @staticmethod
    def backward(ctx, g: Tensor):
        g = g.contiguous()
        n, h, w, c3 = g.shape
        gx = torch.empty(n, h, w, c3 // 3, dtype=torch.float32, device=g.device)
        check(lib().sf_time_stack3_bwd(T(g), ctx.Tn, (n // ctx.Tn) * h * w, T(gx), stream_ptr()), "sf_time_stack3_bwd")
        return gx, None


def time_stack3(x: Tensor, Tn: int) -> Tensor:
    """Time-major ``[T*nb,H,W,C] -> [T*nb,H,W,3C]``: lanes ``dt*C + c`` hold frame ``t + dt - 1`` (zeros outside the clip)."""
    return _TimeStack3Fn.apply(x.contiguous(), Tn)


# ----------------------------------------------------------------------------------------------
# convolutions whose weight is a fresh tensor every call (spectral norm): nothing is cached across calls
# ----------------------------------------------------------------------------------------------
class FreshConvEngine(F.ConvEngine):
    """Index maps of a 3x3 convolution whose weight changes with every forward call (``W / sigma`` with an advancing sigma): the
    packed images are built per call and travel with the autograd node - two forward passes of the same layer before a backward
    (discriminator on real and on generated frames) must not see each other's weights."""

    def packed(self, weight: Tensor, bias: Optional[Tensor], kind, need: Tuple[bool, ...] = ()):
        w4 = weight.reshape(weight.shape[0], weight.shape[1], 3, 3)
        if kind == "fwd":
            return K.pack_weights(w4, bias, self.fwd_map, transpose=False)
        return K.pack_weights(w4, None, self.bwd_map(need), transpose=True)


def conv_nhwc(x: Tensor, weight: Tensor, bias: Optional[Tensor], eng: Optional[F.ConvEngine] = None) -> Tensor:
    """``nn.Conv2d(k, padding=k//2)`` on NHWC ``x``: 1x1 -> ``sf_linear_*``, 3x3 -> the MFMA kernels (``eng``), else ``sf_conv2d_*``."""
    k = weight.shape[-1]
    if k == 1:
        return F.linear(x, weight.reshape(weight.shape[0], weight.shape[1]), bias, lowp=True)   # 16-bit operands in the 16-bit modes, as a Conv2d under autocast
    if k == 3:
        assert eng is not None
        return F.conv3x3(eng, x, weight, bias)
    return F.conv2d(x, weight, bias, 1, k // 2)


class _PadShift4Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor):
        n, h, w, c = x.shape
        y = torch.empty(n, h + 4, w + 4, 4 * c, dtype=torch.float32, device=x.device)
        check(lib().sf_pad_shift_stack4_fwd(T(x), n, h, w, T(y), stream_ptr()), "sf_pad_shift_stack4_fwd")
        ctx.shape = tuple(x.shape)
        return y

    @staticmethod
    def backward(ctx, g: Tensor):
        n, h, w, c = ctx.shape
        gx = torch.empty(ctx.shape, dtype=torch.float32, device=g.device)
        check(lib().sf_pad_shift_stack4_bwd(T(g.contiguous()), n, h, w, T(gx), stream_ptr()), "sf_pad_shift_stack4_bwd")
        return gx


class _CropFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, border: int):
        n, hp, wp, c = x.shape
        h, w = hp - 2 * border, wp - 2 * border
        y = torch.empty(n, h, w, c, dtype=torch.float32, device=x.device)
        check(lib().sf_border(T(x), n, h, w, border, 0, T(y), stream_ptr()), "sf_border")
        ctx.border = border
        return y

    @staticmethod
    def backward(ctx, g: Tensor):
        g = g.contiguous()
        n, h, w, c = g.shape
        b = ctx.border
        gx = torch.empty(n, h + 2 * b, w + 2 * b, c, dtype=torch.float32, device=g.device)
        check(lib().sf_border(T(g), n, h, w, b, 1, T(gx), stream_ptr()), "sf_border")
        return gx, None


class _Regroup5Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weight: Tensor, lanes: int):
        O, I = weight.shape[0], weight.shape[1]
        if not (weight.stride(3) == 1 and weight.stride(2) == 5 and weight.stride(1) == 25):   # anything but a column slice of a dense weight
            weight = weight.contiguous()
        w3 = torch.empty(O, 4 * lanes, 3, 3, dtype=torch.float32, device=weight.device)
        check(lib().sf_regroup5x5_fwd(weight.data_ptr(), weight.stride(0), O, I, lanes, w3.data_ptr(), stream_ptr()), "sf_regroup5x5_fwd")
        ctx.meta = (O, I, lanes)
        return w3

    @staticmethod
    def backward(ctx, g: Tensor):
        O, I, lanes = ctx.meta
        g5 = torch.empty(O, I, 5, 5, dtype=torch.float32, device=g.device)
        check(lib().sf_regroup5x5_bwd(g.contiguous().data_ptr(), O, I, lanes, g5.data_ptr(), stream_ptr()), "sf_regroup5x5_bwd")
        return g5, None


def regroup5x5(weight: Tensor, lanes: int) -> Tensor:
    """``[O, I, 5, 5]`` -> ``[O, 4 * lanes, 3, 3]``: the weight of the ONE 3x3 convolution over four shifted copies (``lanes`` channels each, ``I <= lanes``)
    that equals the 5x5 convolution (``sf_regroup5x5_*``: one gather each way; a column slice of a wider weight is read in place).  A function of the
    weight only, so a recurrent cell builds it once per sequence, not once per frame."""
    require_device(weight, "weight")
    return _Regroup5Fn.apply(weight.float(), lanes)


def conv5x5_as_3x3(x: Tensor, weight: Tensor, bias: Optional[Tensor], eng: "F.ConvEngine", wbatch: Optional["F.WeightGradBatch"] = None) -> Tensor:
    """``nn.Conv2d(k=5, padding=2)`` on the 3x3 MFMA kernels (``sf_pad_shift_stack4_fwd``: the input on a domain padded by 2, shifted
    four ways, stacked as channels; the 5x5 kernel as four 3x3 tiles; the interior of the result).  ``weight``: the 5x5 weight, or what
    ``regroup5x5(weight, x lanes)`` made of it.  ``eng``: a ConvEngine for ``[4 * x lanes] -> cout``."""
    w3 = regroup5x5(weight, x.shape[-1]) if weight.shape[-1] == 5 else weight
    assert w3.shape[1] == 4 * x.shape[-1] and w3.shape[-1] == 3, (w3.shape, x.shape)
    return _CropFn.apply(F.conv3x3(eng, _PadShift4Fn.apply(x.contiguous()), w3, bias, wbatch=wbatch), 2)


class _Conv5x5ShiftFn(torch.autograd.Function):
    """``nn.Conv2d(k=5, padding=2)`` on the 16-bit 3x3 kernels WITHOUT a stacked tensor (``sf_conv5x5_fwd``): ``w3 = regroup5x5(weight, x lanes)`` for the
    forward pass and the weight gradient, ``w3t = regroup5x5(flipped transposed weight, output lanes)`` for the input gradient (the input gradient of a
    5x5 convolution is a 5x5 convolution of the output gradient).  ``eng`` / ``eng_t``: ConvEngines ``[4 * x lanes] -> cout`` / ``[4 * out lanes] -> cin``
    (packed images cached on the cell's parameters).  ``wbatch``: as ``functional._ConvFn``."""

    @staticmethod
    def forward(ctx, eng, eng_t, x: Tensor, w3: Tensor, w3t: Tensor, bias: Optional[Tensor], wbatch):
        n, h, w, _ = x.shape
        ctx.wbatch = wbatch
        if wbatch is not None:
            wbatch.register()
        packed, bp = eng.packed(w3, bias, "fwd")
        y = torch.empty(n, h, w, eng.coutp, dtype=torch.float32, device=x.device)
        K.conv5x5_shift4(x, n, h, w, packed, bp, eng.fwd_map, y)
        ctx.eng, ctx.eng_t, ctx.has_bias = eng, eng_t, bias is not None
        ctx.save_for_backward(x, w3, w3t)
        ctx.bias = bias
        return y

    @staticmethod
    def backward(ctx, gy: Tensor):
        x, w3, w3t = ctx.saved_tensors
        eng, eng_t = ctx.eng, ctx.eng_t
        gy = gy.contiguous()
        n, h, w, _ = x.shape
        dx = None
        if ctx.needs_input_grad[2]:
            dx = torch.empty(n, h, w, eng_t.coutp, dtype=torch.float32, device=gy.device)
            K.conv5x5_shift4(gy, n, h, w, eng_t.packed(w3t, None, "fwd")[0], None, eng_t.fwd_map, dx)
        if not (ctx.needs_input_grad[3] or (ctx.has_bias and ctx.needs_input_grad[5])):
            return None, None, dx, None, None, None, None
        if ctx.wbatch is not None:
            if not ctx.wbatch.add(x, gy):
                return None, None, dx, None, None, None, None
            x, gy = ctx.wbatch.take()
            n = x.shape[0]
        dw3 = torch.empty(w3.shape, dtype=torch.float32, device=gy.device)
        db = torch.empty(w3.shape[0], dtype=torch.float32, device=gy.device) if ctx.has_bias else None
        K.conv5x5_shift4_bwd_weight(x, gy, n, h, w, eng.wgrad_map, dw3, db)
        return None, None, dx, dw3, None, db, None


def conv5x5_shift4_ok(x_lanes: int, out_lanes: int, eng, eng_t) -> bool:
    """Shapes ``sf_conv5x5_fwd`` / ``sf_conv5x5_bwd_weight`` take in the current compute mode: 16-bit operands, lanes in whole 32-channel weight-gradient
    tiles on both sides (the input gradient is the same call with the roles swapped), N blocks the shifted loader is compiled for."""
    from ._hip import SF_BF16, SF_F16, compute_dtype
    return (compute_dtype() in (SF_BF16, SF_F16) and x_lanes % 32 == 0 and out_lanes % 32 == 0 and eng.fwd_map.nf <= 4 and eng_t.fwd_map.nf <= 4
            and not os.environ.get("SF_CONV5_NO_SHIFT4"))


def conv5x5_shift4(x: Tensor, w3: Tensor, w3t: Tensor, bias: Optional[Tensor], eng, eng_t, wbatch=None) -> Tensor:
    return _Conv5x5ShiftFn.apply(eng, eng_t, x.contiguous(), w3, w3t, bias, wbatch)


def regroup5x5_transposed(weight: Tensor, lanes: int) -> Tensor:
    """``regroup5x5`` of the input-gradient kernel of a 5x5 convolution: ``weight [O, I, 5, 5]`` flipped in both spatial axes with O and I swapped,
    ``[I, 4 * lanes, 3, 3]`` (``lanes`` = channel lanes of the OUTPUT gradient, >= O).  No gradient flows through it."""
    return regroup5x5(weight.detach().flip(2, 3).transpose(0, 1).contiguous(), lanes)


class _S2D2Fn(torch.autograd.Function):
    """``[n,h,w,C] <-> [n,h/2,w/2,4C]`` (``sf_space_to_depth2``; a permutation: the backward pass is the other direction)."""

    @staticmethod
    def forward(ctx, x: Tensor, inverse: bool):
        n, a, b, c = x.shape
        h, w, cf = (2 * a, 2 * b, c // 4) if inverse else (a, b, c)
        y = torch.empty((n, h, w, cf) if inverse else (n, h // 2, w // 2, 4 * c), dtype=torch.float32, device=x.device)
        check(lib().sf_space_to_depth2(T(x), n, h, w, int(inverse), T(y), stream_ptr()), "sf_space_to_depth2")
        ctx.meta = (n, h, w, inverse)
        return y

    @staticmethod
    def backward(ctx, g: Tensor):
        n, h, w, inverse = ctx.meta
        g = g.contiguous()
        c = g.shape[-1]
        gx = torch.empty((n, h // 2, w // 2, 4 * c) if inverse else (n, h, w, c // 4), dtype=torch.float32, device=g.device)
        check(lib().sf_space_to_depth2(T(g), n, h, w, int(not inverse), T(gx), stream_ptr()), "sf_space_to_depth2")
        return gx, None


def space_to_depth2(x: Tensor) -> Tensor:
    """NHWC ``[n,h,w,C] -> [n,h/2,w/2,4C]``, lane ``(2 py + px) * C + c`` = pixel ``(2Y + py, 2X + px)``."""
    return _S2D2Fn.apply(x.contiguous(), False)


def depth_to_space2(x: Tensor) -> Tensor:
    return _S2D2Fn.apply(x.contiguous(), True)


class _Regroup5S2DFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, weight: Tensor, lanes: int, hp: int):
        R, I = weight.shape[0], weight.shape[1]
        if not (weight.stride(3) == 1 and weight.stride(2) == 5 and weight.stride(1) == 25):   # anything but a column slice of a dense weight
            weight = weight.contiguous()
        w3 = torch.empty(4 * R, 4 * lanes, 3, 3, dtype=torch.float32, device=weight.device)
        check(lib().sf_regroup5x5_s2d_fwd(weight.data_ptr(), weight.stride(0), R, hp, I, lanes, w3.data_ptr(), stream_ptr()), "sf_regroup5x5_s2d_fwd")
        ctx.meta = (R, I, lanes, hp)
        return w3

    @staticmethod
    def backward(ctx, g: Tensor):
        R, I, lanes, hp = ctx.meta
        g5 = torch.empty(R, I, 5, 5, dtype=torch.float32, device=g.device)
        check(lib().sf_regroup5x5_s2d_bwd(g.contiguous().data_ptr(), R, hp, I, lanes, g5.data_ptr(), stream_ptr()), "sf_regroup5x5_s2d_bwd")
        return g5, None, None


def regroup5x5_s2d(weight: Tensor, lanes: int, hp: int) -> Tensor:
    """``[R, I, 5, 5]`` (``R`` = gate blocks of ``hp`` rows) -> ``[4R, 4 * lanes, 3, 3]``: the weight of the ONE 3x3 convolution on the
    ``space_to_depth2`` domain that equals the 5x5 convolution (``sf_regroup5x5_s2d_*``); output rows gate-major, then output phase."""
    require_device(weight, "weight")
    return _Regroup5S2DFn.apply(weight.float(), lanes, hp)


# ----------------------------------------------------------------------------------------------
# 4x4 convolutions (PatchGAN, reference gan/discriminators.py:139-223) on the 3x3 MFMA kernels
# ----------------------------------------------------------------------------------------------
class _PadS2dFn(torch.autograd.Function):
    """``[n,h,w,C] -> [n,h/2+1,w/2+1,4C]``: pad by 1, fold 2x2 pixel blocks into channels (``sf_pad_s2d_fwd``)."""

    @staticmethod
    def forward(ctx, x: Tensor):
        n, h, w, c = x.shape
        y = torch.empty(n, h // 2 + 1, w // 2 + 1, 4 * c, dtype=torch.float32, device=x.device)
        check(lib().sf_pad_s2d_fwd(T(x), n, h, w, T(y), stream_ptr()), "sf_pad_s2d_fwd")
        ctx.shape = tuple(x.shape)
        return y

    @staticmethod
    def backward(ctx, g: Tensor):
        n, h, w, c = ctx.shape
        gx = torch.empty(ctx.shape, dtype=torch.float32, device=g.device)
        check(lib().sf_pad_s2d_bwd(T(g.contiguous()), n, h, w, T(gx), stream_ptr()), "sf_pad_s2d_bwd")
        return gx


def regroup4x4s2(weight: Tensor, lanes: int) -> Tensor:
    """``[O, I, 4, 4]`` (stride 2, padding 1) -> ``[O, 4 * lanes, 3, 3]``: the weight of the 3x3 convolution over ``_PadS2dFn``'s tensor that equals the
    strided convolution - tap ``(1 + a, 1 + b)`` holds ``W[:, c, 2a + dy, 2b + dx]`` for channel ``(2 dy + dx) * lanes + c``, the other five taps are
    zero.  Autograd-tracked torch ops on a parameter-sized tensor."""
    O, I = weight.shape[0], weight.shape[1]
    g = weight.reshape(O, I, 2, 2, 2, 2).permute(0, 3, 5, 1, 2, 4)          # [O, dy, dx, I, a, b]
    if lanes != I:
        g = torch.nn.functional.pad(g, (0, 0, 0, 0, 0, lanes - I))
    return torch.nn.functional.pad(g.reshape(O, 4 * lanes, 2, 2), (1, 0, 1, 0))


def regroup4x4s2_same(weight: Tensor, lanes: int) -> Tensor:
    """``[O, I, 4, 4]`` (stride 2, padding 1) -> ``[O, 4 * lanes, 3, 3]``: the weight of the 3x3 'SAME' convolution over ``space_to_depth2(x)`` - the
    UNPADDED input with its 2x2 pixel blocks folded into channels - that equals the strided convolution.  Output row i reads input rows 2i - 1 .. 2i + 2
    = the second row of block i - 1, both rows of block i, the first row of block i + 1: block tap ty in {0, 1, 2} and row parity dy carry kernel row
    ``2 ty + dy - 1`` (outside 0..3: zero), the same along the columns; the convolution's own zero padding IS the strided convolution's padding, so the
    result has exactly the strided convolution's h/2 x w/2 pixels - no padded copy of the input, no crop of the output (round 5; the padded form is
    ``regroup4x4s2``).  Autograd-tracked torch ops on a parameter-sized tensor."""
    O, I = weight.shape[0], weight.shape[1]
    g = torch.nn.functional.pad(weight, (1, 1, 1, 1)).reshape(O, I, 3, 2, 3, 2).permute(0, 3, 5, 1, 2, 4)   # [O, dy, dx, I, ty, tx]
    if lanes != I:
        g = torch.nn.functional.pad(g, (0, 0, 0, 0, 0, lanes - I))
    return g.reshape(O, 4 * lanes, 3, 3)


def conv4x4_as_3x3(x: Tensor, weight: Tensor, bias: Optional[Tensor], stride: int, eng: "F.ConvEngine") -> Tensor:
    """``nn.Conv2d(k=4, stride=1|2, padding=1)`` on the 3x3 MFMA kernels.  Stride 2 (even h, w): one 3x3 convolution over the padded input with its
    2x2 pixel blocks folded into channels (``regroup4x4s2``).  Stride 1: the 4x4 kernel is a 5x5 kernel with a zero first row and column, i.e.
    ``conv5x5_as_3x3``; its last output row and column do not exist in the reference's result.  ``eng``: a ConvEngine for ``[4 * x lanes] -> cout``."""
    n, h, w, cp = x.shape
    if stride == 2:
        if not os.environ.get("SF_CONV4_PADDED"):   # (A/B switch: the padded input + cropped output of rounds 2-4)
            return F.conv3x3(eng, space_to_depth2(x), regroup4x4s2_same(weight, cp), bias)
        y = F.conv3x3(eng, _PadS2dFn.apply(x.contiguous()), regroup4x4s2(weight, cp), bias)
        return y[:, : h // 2, : w // 2].contiguous()
    assert stride == 1
    w5 = torch.nn.functional.pad(weight, (1, 0, 1, 0))                       # offsets -2..2 with a zero at -2
    return conv5x5_as_3x3(x, w5, bias, eng)[:, : h - 1, : w - 1].contiguous()


# ----------------------------------------------------------------------------------------------
# conditional BatchNorm (+ ReLU, + nearest up-sampling)  (reference layers/Normalization.py:65-85, GResBlock.py:63-78)
# ----------------------------------------------------------------------------------------------
class _CondNormFn(torch.autograd.Function):
    """``act(gamma_n * batchnorm(x) + beta_n)`` (training-mode statistics over all images, no affine) with ``embed [N, 2*C]`` =
    ``gamma | beta`` per image; optionally ReLU and 2x nearest up-sampling of the result in the same pass."""

    @staticmethod
    def forward(ctx, x: Tensor, embed: Tensor, creal: int, running_mean: Optional[Tensor], running_var: Optional[Tensor], momentum: float, eps: float,
                training: bool, relu: bool, up: bool):
        n, h, w, C = x.shape
        dev = x.device
        embed = embed.contiguous()
        assert embed.shape == (n, 2 * creal)
        if training:
            stats = torch.empty(4, C, dtype=torch.float32, device=dev)
            sums = torch.empty(2, C, dtype=torch.float64, device=dev)
            ones, zeros = torch.ones(creal, dtype=torch.float32, device=dev), torch.zeros(creal, dtype=torch.float32, device=dev)
            check(lib().sf_batchnorm_train_fwd(T(x), n * h * w, 1, creal, ones.data_ptr(), zeros.data_ptr(), eps, momentum,
                                               running_mean.data_ptr() if running_mean is not None else None,
                                               running_var.data_ptr() if running_var is not None else None,
                                               stats[0].data_ptr(), stats[1].data_ptr(), stats[2].data_ptr(), stats[3].data_ptr(), sums.data_ptr(), NULL, SF_F32,
                                               stream_ptr()), "sf_batchnorm_train_fwd")
            mean, rstd = stats[0], stats[1]
        else:  # running statistics are constants (tiny per-channel torch ops)
            mean = torch.zeros(C, dtype=torch.float32, device=dev)
            rstd = torch.zeros(C, dtype=torch.float32, device=dev)
            mean[:creal] = running_mean
            rstd[:creal] = torch.rsqrt(running_var + eps)
        y = torch.empty(n, 2 * h if up else h, 2 * w if up else w, C, dtype=torch.float32, device=dev)
        check(lib().sf_film_act_fwd(T(x), n, h, w, mean.data_ptr(), rstd.data_ptr(), embed.data_ptr(), creal, int(relu), int(up), T(y), stream_ptr()), "sf_film_act_fwd")
        ctx.meta = (creal, training, relu, up)
        ctx.save_for_backward(x, embed, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, gy: Tensor):
        x, embed, mean, rstd = ctx.saved_tensors
        creal, training, relu, up = ctx.meta
        n, h, w, C = x.shape
        dev = x.device
        gy = gy.contiguous()
        dxhat = torch.empty_like(x)
        dembed = torch.empty_like(embed)
        ws = _ws(lib().sf_film_act_bwd_workspace_floats(n, h, w, C, creal), dev)
        check(lib().sf_film_act_bwd(T(gy), T(x), n, h, w, mean.data_ptr(), rstd.data_ptr(), embed.data_ptr(), creal, int(relu), int(up), T(dxhat), dembed.data_ptr(),
                                    ws.data_ptr(), stream_ptr()), "sf_film_act_bwd")
        dx = None
        if ctx.needs_input_grad[0]:
            if training:  # through the batch statistics: the BatchNorm backward with a unit affine map
                dx = torch.empty_like(x)
                sums = torch.empty(2, C, dtype=torch.float64, device=dev)
                coef = torch.empty(3, C, dtype=torch.float32, device=dev)
                ones = torch.ones(creal, dtype=torch.float32, device=dev)
                dummy = torch.empty(2, creal, dtype=torch.float32, device=dev)
                check(lib().sf_batchnorm_train_bwd(T(x), T(dxhat), n * h * w, 1, creal, ones.data_ptr(), mean.data_ptr(), rstd.data_ptr(), sums.data_ptr(),
                                                   coef.data_ptr(), T(dx), dummy[0].data_ptr(), dummy[1].data_ptr(), SF_F32, stream_ptr()), "sf_batchnorm_train_bwd")
            else:
                dx = dxhat * rstd  # frozen statistics: d xhat / dx = rstd (pointwise torch op; evaluation-mode fine-tuning only)
        return dx, dembed, None, None, None, None, None, None, None, None


def conditional_norm(x: Tensor, embed: Tensor, creal: int, bn: torch.nn.BatchNorm2d, training: bool, relu: bool = False, up: bool = False) -> Tensor:
    momentum = bn.momentum
    if training and bn.track_running_stats and bn.num_batches_tracked is not None:
        if momentum is None:
            momentum = -(float(bn.num_batches_tracked) + 1.0)
        bn.num_batches_tracked += 1
    use_batch = training or bn.running_mean is None
    return _CondNormFn.apply(x.contiguous(), embed, creal, bn.running_mean, bn.running_var, float(momentum if momentum is not None else 0.0), bn.eps, use_batch,
                             relu, up)


# ----------------------------------------------------------------------------------------------
# small pointwise pieces
# ----------------------------------------------------------------------------------------------
def relu(x: Tensor) -> Tensor:
    return F.leaky_relu(x, 0.0)


class _ReluSumFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor):
        n, C = x.shape[0], x.shape[-1]
        pixels = x.numel() // (n * C)
        out = torch.empty(n, C, dtype=torch.float32, device=x.device)
        check(lib().sf_relu_sum_pixels_fwd(T(x), n, pixels, out.data_ptr(), stream_ptr()), "sf_relu_sum_pixels_fwd")
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, g: Tensor):
        (x,) = ctx.saved_tensors
        n, C = x.shape[0], x.shape[-1]
        g = g.contiguous()
        gx = torch.empty_like(x)
        check(lib().sf_relu_sum_pixels_bwd(g.data_ptr(), T(x), n, x.numel() // (n * C), T(gx), stream_ptr()), "sf_relu_sum_pixels_bwd")
        return gx


def relu_sum_pixels(x: Tensor) -> Tensor:
    """``F.relu(x).view(N, C, -1).sum(2)`` on NHWC ``x`` -> ``[N, Cp]``."""
    return _ReluSumFn.apply(x.contiguous())


class _GammaResidualFn(torch.autograd.Function):
    """``gamma * o + x`` with ``gamma`` a one-element parameter."""

    @staticmethod
    def forward(ctx, o: Tensor, x: Tensor, gamma: Tensor):
        y = torch.empty_like(o)
        check(lib().sf_axpy(o.data_ptr(), x.data_ptr(), gamma.data_ptr(), 0.0, o.numel(), y.data_ptr(), stream_ptr()), "sf_axpy")
        ctx.save_for_backward(o, gamma)
        return y

    @staticmethod
    def backward(ctx, g: Tensor):
        o, gamma = ctx.saved_tensors
        g = g.contiguous()
        go = torch.empty_like(g)
        check(lib().sf_axpy(g.data_ptr(), None, gamma.data_ptr(), 0.0, g.numel(), go.data_ptr(), stream_ptr()), "sf_axpy")
        dgamma = torch.empty_like(gamma)
        ws = _ws(512, g.device)
        check(lib().sf_dot(g.data_ptr(), o.data_ptr(), g.numel(), dgamma.data_ptr(), ws.data_ptr(), stream_ptr()), "sf_dot")
        return go, g, dgamma


def gamma_residual(o: Tensor, x: Tensor, gamma: Tensor) -> Tensor:
    assert o.shape == x.shape and gamma.numel() == 1
    return _GammaResidualFn.apply(o.contiguous(), x.contiguous(), gamma)


class _AddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a: Tensor, b: Tensor):
        y = torch.empty_like(a)
        check(lib().sf_axpy(a.data_ptr(), b.data_ptr(), None, 1.0, a.numel(), y.data_ptr(), stream_ptr()), "sf_axpy")
        return y

    @staticmethod
    def backward(ctx, g: Tensor):
        return g, g


def add(a: Tensor, b: Tensor) -> Tensor:
    assert a.shape == b.shape
    return _AddFn.apply(a.contiguous(), b.contiguous())


class _TanhFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor):
        y = torch.empty_like(x)
        check(lib().sf_tanh(x.data_ptr(), None, x.numel(), y.data_ptr(), stream_ptr()), "sf_tanh")
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g: Tensor):
        (y,) = ctx.saved_tensors
        g = g.contiguous()
        gx = torch.empty_like(g)
        check(lib().sf_tanh(g.data_ptr(), y.data_ptr(), g.numel(), gx.data_ptr(), stream_ptr()), "sf_tanh")
        return gx


def tanh(x: Tensor) -> Tensor:
    return _TanhFn.apply(x.contiguous())


# ----------------------------------------------------------------------------------------------
# gate arithmetic of the generator's ConvGRU
# ----------------------------------------------------------------------------------------------
class _DvdGruGatesFn(torch.autograd.Function):
    """``(gx_zr, gh_zr, h) -> (z, rh)``: ``z = sig(.)`` is returned as the first ``hidp`` lanes of ``zr`` (kept whole for the out stage)."""

    @staticmethod
    def forward(ctx, gx: Tensor, gh: Optional[Tensor], h: Optional[Tensor], hidp: int, out_rh: Optional[Tensor] = None, gslot=None):
        shp, dev = gx.shape[:-1], gx.device
        pixels = gx.numel() // gx.shape[-1]
        zr = torch.empty(*shp, 2 * hidp, dtype=torch.float32, device=dev)
        rh = out_rh if out_rh is not None else torch.empty(*shp, hidp, dtype=torch.float32, device=dev)   # (a frame's slot: see sequence_slots)
        assert rh.shape == (*shp, hidp) and rh.dtype == torch.float32 and rh.is_contiguous()
        check(lib().sf_dvdgru_gates_fwd(T(gx), T(gh) if gh is not None else NULL, T(h) if h is not None else NULL, pixels, hidp, T(zr), T(rh), stream_ptr()),
              "sf_dvdgru_gates_fwd")
        ctx.hidp, ctx.has = hidp, (gh is not None, h is not None)
        ctx.gslot = gslot   # (GradSlots, frame): where this frame's pre-activation gradient is written
        if out_rh is not None:
            ctx.mark_dirty(out_rh)   # written in place and returned (advisor r5)
        ctx.save_for_backward(zr, h if h is not None else gx.new_empty(0))
        ctx.set_materialize_grads(False)
        return zr, rh

    @staticmethod
    def backward(ctx, dzr: Optional[Tensor], drh: Optional[Tensor]):
        # dzr: only the z half can carry a gradient (the out stage reads z from zr); it arrives as a dense [.., 2*hidp] tensor
        zr, h = ctx.saved_tensors
        hidp = ctx.hidp
        has_gh, has_h = ctx.has
        pixels = zr.numel() // (2 * hidp)
        dz = dzr.contiguous() if dzr is not None else None   # read through its row stride: no slice copy
        drh = drh.contiguous() if drh is not None else None
        dpre = ctx.gslot[0].slot(ctx.gslot[1]) if ctx.gslot is not None else torch.empty_like(zr)
        assert dpre.shape == zr.shape
        dh = torch.empty(*zr.shape[:-1], hidp, dtype=torch.float32, device=zr.device) if has_h else None
        check(lib().sf_dvdgru_gates_bwd(T(dz) if dz is not None else NULL, T(drh) if drh is not None else NULL, T(zr), T(h) if has_h else NULL, pixels, hidp, T(dpre),
                                        T(dh) if dh is not None else NULL, stream_ptr()), "sf_dvdgru_gates_bwd")
        return dpre, (dpre if has_gh else None), dh, None, None, None


class _DvdGruOutFn(torch.autograd.Function):
    """``(gx_o, gh_o, zr, h) -> h' = h (1 - z) + tanh(gx_o + gh_o) z``."""

    @staticmethod
    def forward(ctx, gx: Tensor, gh: Optional[Tensor], zr: Tensor, h: Optional[Tensor], hidp: int, out: Optional[Tensor] = None, gslot=None):
        shp, dev = gx.shape[:-1], gx.device
        pixels = gx.numel() // gx.shape[-1]
        cand = torch.empty(*shp, hidp, dtype=torch.float32, device=dev)
        hn = out if out is not None else torch.empty(*shp, hidp, dtype=torch.float32, device=dev)   # (``out``: a frame's slot of ``sequence_slots``)
        assert hn.shape == (*shp, hidp) and hn.dtype == torch.float32 and hn.is_contiguous()
        check(lib().sf_dvdgru_out_fwd(T(gx), T(gh) if gh is not None else NULL, T(zr), T(h) if h is not None else NULL, pixels, hidp, T(cand), T(hn), stream_ptr()),
              "sf_dvdgru_out_fwd")
        ctx.hidp, ctx.has, ctx.lanes = hidp, (gh is not None, h is not None), gx.shape[-1]
        ctx.gslot = gslot
        if out is not None:
            ctx.mark_dirty(out)   # written in place and returned: the Function contract (version bump, in-place checks) - advisor r5
        ctx.save_for_backward(cand, zr, h if h is not None else gx.new_empty(0))
        return hn

    @staticmethod
    def backward(ctx, dhn: Tensor):
        cand, zr, h = ctx.saved_tensors
        hidp = ctx.hidp
        has_gh, has_h = ctx.has
        pixels = cand.numel() // hidp
        dhn = dhn.contiguous()
        da = ctx.gslot[0].slot(ctx.gslot[1]) if ctx.gslot is not None else torch.empty_like(cand)
        assert da.shape == cand.shape
        dzr = torch.empty_like(zr)   # the kernel zeroes the r half: it does not reach h' through this stage
        dh = torch.empty_like(cand) if has_h else None
        check(lib().sf_dvdgru_out_bwd(T(dhn), T(cand), T(zr), T(h) if has_h else NULL, pixels, hidp, T(da), T(dzr), T(dh) if dh is not None else NULL, stream_ptr()),
              "sf_dvdgru_out_bwd")
        return da, (da if has_gh else None), dzr, dh, None, None, None


def sequence_slots(frames: int, shape, device):
    """``(buffer [frames * shape[0], ...], [slot_0, ...])``: one contiguous fp32 buffer for a sequence of per-frame results and a tensor per frame on
    its slice of the storage (own tensors, not autograd views: a kernel writes each through ``out=``).  ``assemble(buffer, results)`` then IS the
    concatenation of the per-frame results - the frames were written where ``torch.cat`` would have copied them."""
    buf = torch.empty(frames * shape[0], *shape[1:], dtype=torch.float32, device=device)
    per = buf.numel() // max(frames, 1)
    slots = [torch.empty(0, dtype=torch.float32, device=device).set_(buf.untyped_storage(), buf.storage_offset() + t * per, tuple(shape)) for t in range(frames)]
    return buf, slots


class _AssembleFn(torch.autograd.Function):
    """``torch.cat(results, 0)`` for results that already sit in consecutive slices of ``holder[0]`` (``sequence_slots``): no copy; backward = the slices."""

    @staticmethod
    def forward(ctx, holder, *results: Tensor):
        buf = holder[0]
        per = buf.numel() // len(results)
        for t, r in enumerate(results):
            if r.data_ptr() != buf.data_ptr() + 4 * t * per or r.numel() != per:
                raise RuntimeError("assemble: result %d is not slot %d of the buffer" % (t, t))
        ctx.frames = len(results)
        return buf

    @staticmethod
    def backward(ctx, g: Tensor):
        g = g.contiguous()
        return (None, *g.view(ctx.frames, g.shape[0] // ctx.frames, *g.shape[1:]).unbind(0))


def assemble(buf: Tensor, results) -> Tensor:
    return _AssembleFn.apply((buf,), *results)


class GradSlots:
    """The backward-pass counterpart of ``sequence_slots``: ONE gradient buffer for a tensor that the forward pass split into frames, allocated when the
    first frame's backward asks for its slot.  A frame's backward kernel writes its gradient into ``slot(t)``; ``split_frames``' backward then finds the
    frames' gradients already in place and returns the buffer (no ``torch.stack``), and a batched weight gradient reads them in place as well."""

    def __init__(self, frames: int, frame_shape, device) -> None:
        self.frames, self.shape, self.device, self.buf = frames, tuple(frame_shape), device, None

    def slot(self, t: int) -> Tensor:
        if self.buf is None:
            self.buf = torch.empty(self.frames * self.shape[0], *self.shape[1:], dtype=torch.float32, device=self.device)
        per = self.buf.numel() // self.frames
        return torch.empty(0, dtype=torch.float32, device=self.device).set_(self.buf.untyped_storage(), self.buf.storage_offset() + t * per, self.shape)


class _SplitFn(torch.autograd.Function):
    """``x.view(frames, n, ...).unbind(0)`` as tensors of their own on x's storage; backward: the frames' gradients where they lie in the ``GradSlots``
    buffer (written there by the frames' backward kernels), else their concatenation."""

    @staticmethod
    def forward(ctx, x: Tensor, frames: int, gslots: "GradSlots"):
        x = x.contiguous()
        per = x.numel() // frames
        shape = (x.shape[0] // frames, *x.shape[1:])
        ctx.gslots, ctx.frames, ctx.shape = gslots, frames, shape
        return tuple(torch.empty(0, dtype=x.dtype, device=x.device).set_(x.untyped_storage(), x.storage_offset() + t * per, shape) for t in range(frames))

    @staticmethod
    def backward(ctx, *gs):
        buf = ctx.gslots.buf
        per = (buf.numel() // ctx.frames) if buf is not None else 0
        if buf is not None and all(g is not None and g.is_contiguous() and g.data_ptr() == buf.data_ptr() + 4 * t * per and g.numel() == per
                                   for t, g in enumerate(gs)):
            return buf, None, None
        ref = next(g for g in gs if g is not None)
        return torch.cat([g if g is not None else torch.zeros(ctx.shape, dtype=ref.dtype, device=ref.device) for g in gs], 0), None, None


def split_frames(x: Tensor, frames: int):
    """``(per-frame tensors of time-major x, GradSlots for their gradients)``."""
    gs = GradSlots(frames, (x.shape[0] // frames, *x.shape[1:]), x.device)
    return _SplitFn.apply(x, frames, gs), gs


def dvdgru_gates(gx_zr: Tensor, gh_zr: Optional[Tensor], h: Optional[Tensor], hidp: int, out_rh: Optional[Tensor] = None, gslot=None) -> Tuple[Tensor, Tensor]:
    return _DvdGruGatesFn.apply(gx_zr.contiguous(), gh_zr.contiguous() if gh_zr is not None else None, h.contiguous() if h is not None else None, hidp, out_rh, gslot)


def dvdgru_out(gx_o: Tensor, gh_o: Optional[Tensor], zr: Tensor, h: Optional[Tensor], hidp: int, out: Optional[Tensor] = None, gslot=None) -> Tensor:
    return _DvdGruOutFn.apply(gx_o.contiguous(), gh_o.contiguous() if gh_o is not None else None, zr, h.contiguous() if h is not None else None, hidp, out, gslot)


# ----------------------------------------------------------------------------------------------
# torch.bmm / softmax on strided views
# ----------------------------------------------------------------------------------------------
_bmm_raw = K.bmm_raw


class _BmmFn(torch.autograd.Function):
    """``torch.bmm(A, B)`` for arbitrary strided 3-D views; the gradients are the same kernel on transposed views.  ``lowp``: an attention
    product - bf16 MFMA operands in the bf16 compute modes (forward and both gradients), as under the reference's autocast."""

    @staticmethod
    def forward(ctx, A: Tensor, B: Tensor, lowp: bool):
        out = torch.empty(A.shape[0], A.shape[1], B.shape[2], dtype=torch.float32, device=A.device)
        _bmm_raw(A, B, out, lowp=lowp)
        ctx.save_for_backward(A, B)
        ctx.lowp = lowp
        return out

    @staticmethod
    def backward(ctx, g: Tensor):
        A, B = ctx.saved_tensors
        dA = dB = None
        if ctx.needs_input_grad[0]:
            dA = torch.empty(A.shape, dtype=torch.float32, device=A.device)
            _bmm_raw(g, B.transpose(1, 2), dA, lowp=ctx.lowp)
        if ctx.needs_input_grad[1]:
            dB = torch.empty(B.shape, dtype=torch.float32, device=B.device)
            _bmm_raw(A.transpose(1, 2), g, dB, lowp=ctx.lowp)
        return dA, dB, None


class _FlashAttnFn(torch.autograd.Function):
    """``softmax(q k^T) v`` of the self-attention layers in the 16-bit compute modes, fused (sf_flash_attention_fwd / _bwd): what
    ``bmm(softmax_last(bmm(q, k^T, lowp)), v, lowp)`` computes, without the ``[n, HW, HW]`` score tensor."""

    @staticmethod
    def forward(ctx, q: Tensor, k: Tensor, v: Tensor, scale: float):
        out, lse = K.flash_attention_fwd(q, k, v, scale)
        ctx.save_for_backward(q, k, v, out, lse)
        ctx.scale = scale
        return out

    @staticmethod
    def backward(ctx, g: Tensor):
        q, k, v, out, lse = ctx.saved_tensors
        dq, dk, dv = K.flash_attention_bwd(q, k, v, out, lse, g, ctx.scale)
        return dq, dk, dv, None


def attention(q: Tensor, k: Tensor, v: Tensor) -> Tensor:
    """``bmm(softmax(bmm(q, k^T)), v)`` of reference ``Discriminator.py:118-124`` / ``Attention.py:206-217`` on ``[n, HW, lanes]`` operands: the fused
    kernel where it applies (16-bit compute modes, HW a multiple of 128, operand widths it is built for), the three materialised steps otherwise."""
    if K.flash_attention_ok(q, k, v):
        return _FlashAttnFn.apply(q, k, v, 1.0)
    att = softmax_last(bmm(q, k.transpose(1, 2), lowp=True))
    return bmm(att, v, lowp=True)


def bmm(A: Tensor, B: Tensor, lowp: bool = False) -> Tensor:
    """``lowp=False``: exact fp32 in every compute mode (selection matrices, dot products of embeddings)."""
    require_device(A, "A")
    return _BmmFn.apply(A, B, lowp)


class _SoftmaxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor):
        L = x.shape[-1]
        y = torch.empty_like(x)
        check(lib().sf_softmax_rows_fwd(x.data_ptr(), x.numel() // L, L, y.data_ptr(), stream_ptr()), "sf_softmax_rows_fwd")
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g: Tensor):
        (y,) = ctx.saved_tensors
        g = g.contiguous()
        L = y.shape[-1]
        dx = torch.empty_like(y)
        check(lib().sf_softmax_rows_bwd(g.data_ptr(), y.data_ptr(), y.numel() // L, L, dx.data_ptr(), stream_ptr()), "sf_softmax_rows_bwd")
        return dx


def softmax_last(x: Tensor) -> Tensor:
    """``softmax(x, dim=-1)`` of a contiguous fp32 tensor."""
    return _SoftmaxFn.apply(x.contiguous())


class _MaxPool3Fn(torch.autograd.Function):
    """``nn.MaxPool3d(kernel, stride)`` over NHWC tokens ``[B, D0, D1, D2, C]`` (``sf_maxpool3d_*``)."""

    @staticmethod
    def forward(ctx, x: Tensor, kernel: Tuple[int, int, int], stride: Tuple[int, int, int]):
        B, d0, d1, d2, C = x.shape
        o = [(d - k) // s_ + 1 for d, k, s_ in zip((d0, d1, d2), kernel, stride)]
        y = torch.empty(B, *o, C, dtype=torch.float32, device=x.device)
        check(lib().sf_maxpool3d_fwd(T(x), B, d0, d1, d2, *kernel, *stride, T(y), stream_ptr()), "sf_maxpool3d_fwd")
        ctx.meta = (kernel, stride)
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, g: Tensor):
        (x,) = ctx.saved_tensors
        kernel, stride = ctx.meta
        B, d0, d1, d2, C = x.shape
        gx = torch.empty_like(x)
        check(lib().sf_maxpool3d_bwd(T(x), T(g.contiguous()), B, d0, d1, d2, *kernel, *stride, T(gx), stream_ptr()), "sf_maxpool3d_bwd")
        return gx, None, None


def max_pool3(x: Tensor, kernel: Tuple[int, int, int], stride: Tuple[int, int, int]) -> Tensor:
    return _MaxPool3Fn.apply(x.contiguous(), tuple(kernel), tuple(stride))

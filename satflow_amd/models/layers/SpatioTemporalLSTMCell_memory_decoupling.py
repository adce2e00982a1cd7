"""ST-LSTM cell with memory decoupling (PredRNN v2) - SURVEY 8f-4.

Mirror of the reference ``SpatioTemporalLSTMCell`` (``satflow/models/layers/SpatioTemporalLSTMCell_memory_decoupling.py:13-138``): same
constructor, same parameter names (``conv_x.0.weight``, ``conv_h.0.weight``, ``conv_m.0.weight``, ``conv_o.0.weight``,
``conv_last.weight`` and, with ``layer_norm=True``, ``conv_x.1.weight / .bias`` ... of the ``nn.LayerNorm([C', width, width])`` layers -
a reference ``state_dict`` loads with ``strict=True``), same ``forward(x_t, h_t, c_t, m_t) -> (h_new, c_new, m_new, delta_c, delta_m)``
on NCHW tensors.

The four 3x3 convolutions run on the MFMA convolution kernels (``sf_conv3x3_fwd`` and its gradients; ``conv_o`` reads ``c_new`` and
``m_new`` as two sources, so ``cat`` is never formed for it), ``conv_last`` on ``sf_linear_fwd``, the layer norms (``layer_norm=True``,
reference ``:20-62``) in ``sf_layernorm_chw_*``, the gate arithmetic in the two fused pointwise stages ``sf_stlstm_gates_*`` /
``sf_stlstm_out_*``.  Any hidden width: the kernels see gate-major blocks of ``hidp = ceil16(num_hidden)`` lanes; for widths that are
not multiples of 16 the convolution weights get zero rows for the pad lanes (tiny autograd-tracked pads of the parameters).
Supported: ``filter_size=3, stride=1`` (3x3 'same' convolutions) - anything else raises ``NotImplementedError``.
"""
import torch
import torch.nn.functional as TF
from torch import Tensor, nn

from ... import functional as F
from ..._hip import T, check, cpad, lib, require_device, stream_ptr


class _LayerNormCHWFn(torch.autograd.Function):
    """``nn.LayerNorm([C', W, W])`` on NHWC activations with gate-major padded lanes (``sf_layernorm_chw_*``)."""

    @staticmethod
    def forward(ctx, x: Tensor, gamma: Tensor, beta: Tensor, gates: int, hid: int, hidp: int, eps: float):
        x = x.contiguous()
        n = x.shape[0]
        pixels = x.shape[1] * x.shape[2]
        y = torch.empty_like(x)
        partial = torch.empty(n * 32 * 2, dtype=torch.float64, device=x.device)
        g, b = gamma.contiguous(), beta.contiguous()
        check(lib().sf_layernorm_chw_fwd(T(x), n, pixels, gates, hid, hidp, g.data_ptr(), b.data_ptr(), eps, partial.data_ptr(), T(y), stream_ptr()),
              "sf_layernorm_chw_fwd")
        ctx.meta = (gates, hid, hidp, eps)
        ctx.save_for_backward(x, g, partial)
        return y

    @staticmethod
    def backward(ctx, dy: Tensor):
        x, g, partial = ctx.saved_tensors
        gates, hid, hidp, eps = ctx.meta
        dy = dy.contiguous()
        n, pixels = x.shape[0], x.shape[1] * x.shape[2]
        dx = torch.empty_like(x)
        dgamma, dbeta = torch.empty_like(g), torch.empty_like(g)
        scratch = torch.empty_like(partial)
        check(lib().sf_layernorm_chw_bwd(T(x), T(dy), n, pixels, gates, hid, hidp, g.data_ptr(), eps, partial.data_ptr(), scratch.data_ptr(), T(dx),
                                         dgamma.data_ptr(), dbeta.data_ptr(), stream_ptr()), "sf_layernorm_chw_bwd")
        return dx, dgamma, dbeta, None, None, None, None


class SpatioTemporalLSTMCell(nn.Module):
    def __init__(self, in_channel, num_hidden, width, filter_size, stride, layer_norm):
        super().__init__()
        if filter_size != 3 or stride != 1:
            raise NotImplementedError("the HIP ST-LSTM cell implements filter_size=3, stride=1 (3x3 'same' convolutions)")
        self.num_hidden = num_hidden
        self.padding = filter_size // 2
        self._forget_bias = 1.0
        self.width = width
        self.layer_norm = bool(layer_norm)
        nh = num_hidden
        self._hidp = hp = cpad(nh)

        def conv(cin, cout, k):
            return nn.Conv2d(cin, cout, kernel_size=k, stride=1, padding=k // 2, bias=False)

        def block(cin, gates):  # nn.Sequential wrappers keep the reference's parameter names ("conv_x.0.weight", "conv_x.1.weight", ...)
            mods = [conv(cin, nh * gates, 3)]
            if layer_norm:
                mods.append(nn.LayerNorm([nh * gates, width, width]))
            return nn.Sequential(*mods)

        self.conv_x = block(in_channel, 7)
        self.conv_h = block(nh, 4)
        self.conv_m = block(nh, 3)
        self.conv_o = block(nh * 2, 1)
        self.conv_last = conv(nh * 2, nh, 1)
        self._eng_x = F.ConvEngine([in_channel], 7 * hp)
        self._eng_h = F.ConvEngine([nh], 4 * hp)
        self._eng_m = F.ConvEngine([nh], 3 * hp)
        self._eng_o = F.ConvEngine([nh, nh], nh)

    def _gate_major(self, w: Tensor, gates: int) -> Tensor:
        """``[gates*nh, cin, 3, 3] -> [gates*hidp, cin, 3, 3]`` with zero rows for the pad lanes of every gate block."""
        nh, hp = self.num_hidden, self._hidp
        if hp == nh:
            return w
        return TF.pad(w.view(gates, nh, *w.shape[1:]), (0, 0, 0, 0, 0, 0, 0, hp - nh)).reshape(gates * hp, *w.shape[1:])

    def _conv_block(self, eng: F.ConvEngine, block: nn.Sequential, x: Tensor, gates: int) -> Tensor:
        w = block[0].weight
        eng.key_tensors = (w,)  # the padded weight is a fresh tensor every call: the pack cache keys on the live parameter
        y = F.conv3x3(eng, x, self._gate_major(w, gates), None)
        if self.layer_norm:
            ln = block[1]
            if (y.shape[1], y.shape[2]) != (self.width, self.width):
                raise RuntimeError(f"layer_norm=True: the cell was built for {self.width}x{self.width} maps, got {y.shape[1]}x{y.shape[2]}")
            y = _LayerNormCHWFn.apply(y, ln.weight, ln.bias, gates, self.num_hidden, self._hidp, ln.eps)
        return y

    def run(self, x: Tensor, h: Tensor, c: Tensor, m: Tensor):
        """NHWC tensors ``[N,H,W,Cp]`` -> (h', c', m', delta_c, delta_m), each ``[N,H,W,hidp]``."""
        nh, hp = self.num_hidden, self._hidp
        gx = self._conv_block(self._eng_x, self.conv_x, x, 7)
        gh = self._conv_block(self._eng_h, self.conv_h, h, 4)
        gm = self._conv_block(self._eng_m, self.conv_m, m, 3)
        c_new, m_new, mem, delta_c, delta_m, pre_o = F.stlstm_gates(gx, gh, gm, c, m, hp, self._forget_bias)
        n = x.shape[0]
        co = F.conv3x3_broadcast(self._eng_o, c_new, m_new, self.conv_o[0].weight, None, n, (0, 0), (0, 0))
        if self.layer_norm:
            ln = self.conv_o[1]
            co = _LayerNormCHWFn.apply(co, ln.weight, ln.bias, 1, nh, hp, ln.eps)
        wl = self.conv_last.weight.reshape(nh, 2 * nh)
        if hp != nh:  # mem = [c' (hidp lanes) | m' (hidp lanes)]
            wl = torch.cat((TF.pad(wl[:, :nh], (0, hp - nh)), TF.pad(wl[:, nh:], (0, hp - nh))), 1)
        last = F.linear(mem, wl, None)
        h_new = F.stlstm_out(pre_o, co, last, hp)
        return h_new, c_new, m_new, delta_c, delta_m

    def forward(self, x_t: Tensor, h_t: Tensor, c_t: Tensor, m_t: Tensor):
        for t, name in ((x_t, "x_t"), (h_t, "h_t"), (c_t, "c_t"), (m_t, "m_t")):
            require_device(t, name)
        outs = self.run(*(F.nchw_to_nhwc(t.float()) for t in (x_t, h_t, c_t, m_t)))
        return tuple(F.nhwc_to_nchw(o, self.num_hidden) for o in outs)

"""ST-LSTM cell with memory decoupling (PredRNN v2) - SURVEY 8f-4.

Mirror of the reference ``SpatioTemporalLSTMCell`` (``satflow/models/layers/SpatioTemporalLSTMCell_memory_decoupling.py:13-138``): same
constructor, same parameter names (``conv_x.0.weight``, ``conv_h.0.weight``, ``conv_m.0.weight``, ``conv_o.0.weight``,
``conv_last.weight`` - a reference ``state_dict`` loads with ``strict=True``), same ``forward(x_t, h_t, c_t, m_t) ->
(h_new, c_new, m_new, delta_c, delta_m)`` on NCHW tensors.

The four 3x3 convolutions run on the MFMA convolution kernels (``sf_conv3x3_fwd`` and its gradients; ``conv_o`` reads ``c_new`` and
``m_new`` as two sources, so ``cat`` is never formed for it), ``conv_last`` on ``sf_linear_fwd``, the gate arithmetic in the two
fused pointwise stages ``sf_stlstm_gates_*`` / ``sf_stlstm_out_*``.  Supported: ``filter_size=3, stride=1, layer_norm=False`` and a
hidden width that is a multiple of 16 (the kernels' channel padding; the gate blocks of the convolution outputs are then plain
channel ranges) - anything else raises ``NotImplementedError``.
"""
import torch
from torch import Tensor, nn

from ... import functional as F
from ..._hip import require_device


class SpatioTemporalLSTMCell(nn.Module):
    def __init__(self, in_channel, num_hidden, width, filter_size, stride, layer_norm):
        super().__init__()
        if filter_size != 3 or stride != 1:
            raise NotImplementedError("the HIP ST-LSTM cell implements filter_size=3, stride=1 (3x3 'same' convolutions)")
        if layer_norm:
            raise NotImplementedError("the HIP ST-LSTM cell implements layer_norm=False")
        if num_hidden % 16:
            raise NotImplementedError("the HIP ST-LSTM cell needs num_hidden to be a multiple of 16 (channel padding of the kernels)")
        self.num_hidden = num_hidden
        self.padding = filter_size // 2
        self._forget_bias = 1.0
        self.width = width
        nh = num_hidden

        def conv(cin, cout, k):
            return nn.Conv2d(cin, cout, kernel_size=k, stride=1, padding=k // 2, bias=False)

        # nn.Sequential wrappers keep the reference's parameter names ("conv_x.0.weight", ...)
        self.conv_x = nn.Sequential(conv(in_channel, nh * 7, 3))
        self.conv_h = nn.Sequential(conv(nh, nh * 4, 3))
        self.conv_m = nn.Sequential(conv(nh, nh * 3, 3))
        self.conv_o = nn.Sequential(conv(nh * 2, nh, 3))
        self.conv_last = conv(nh * 2, nh, 1)
        self._eng_x = F.ConvEngine([in_channel], 7 * nh)
        self._eng_h = F.ConvEngine([nh], 4 * nh)
        self._eng_m = F.ConvEngine([nh], 3 * nh)
        self._eng_o = F.ConvEngine([nh, nh], nh)

    def run(self, x: Tensor, h: Tensor, c: Tensor, m: Tensor):
        """NHWC tensors ``[N,H,W,Cp]`` -> (h', c', m', delta_c, delta_m), each ``[N,H,W,num_hidden]``."""
        nh = self.num_hidden
        gx = F.conv3x3(self._eng_x, x, self.conv_x[0].weight, None)
        gh = F.conv3x3(self._eng_h, h, self.conv_h[0].weight, None)
        gm = F.conv3x3(self._eng_m, m, self.conv_m[0].weight, None)
        c_new, m_new, mem, delta_c, delta_m, pre_o = F.stlstm_gates(gx, gh, gm, c, m, nh, self._forget_bias)
        n = x.shape[0]
        co = F.conv3x3_broadcast(self._eng_o, c_new, m_new, self.conv_o[0].weight, None, n, (0, 0), (0, 0))
        last = F.linear(mem, self.conv_last.weight.reshape(nh, 2 * nh), None)
        h_new = F.stlstm_out(pre_o, co, last, nh)
        return h_new, c_new, m_new, delta_c, delta_m

    def forward(self, x_t: Tensor, h_t: Tensor, c_t: Tensor, m_t: Tensor):
        for t, name in ((x_t, "x_t"), (h_t, "h_t"), (c_t, "c_t"), (m_t, "m_t")):
            require_device(t, name)
        outs = self.run(*(F.nchw_to_nhwc(t.float()) for t in (x_t, h_t, c_t, m_t)))
        return tuple(F.nhwc_to_nchw(o, self.num_hidden) for o in outs)

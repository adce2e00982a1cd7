"""ConvLSTM cell on the fused HIP kernel.

Module surface of reference ``satflow/models/layers/ConvLSTM.py:7-64``: same constructor,
``conv.weight [4*hid, in+hid, kh, kw]`` / ``conv.bias [4*hid]`` parameter names (so reference
checkpoints load), ``forward(input_tensor, cur_state) -> (h, c)`` on NCHW tensors and
``init_hidden``.  The arithmetic of ``forward`` (``:42-57``) is one launch of
``sf_convlstm_cell_fwd``; its autograd is ``sf_convlstm_cell_bwd_gates`` + the input-gradient
convolution + ``sf_conv3x3_bwd_weight``.

``CellEngine`` is the layout-level worker shared with the unrolled encoder-decoder
(``satflow_amd/models/conv_lstm.py``), which drives whole sequences without going through
this single-step surface.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import torch
from torch import nn

from ... import kernels as K
from ..._hip import NULL, T, cpad, generation, require_device, sfTensor
from ..utils import get_conv_layer

Tensor = torch.Tensor


class CellEngine:
    """Packed-weight cache + kernel launches for one ConvLSTM cell (NHWC, padded channels)."""

    def __init__(self, conv: nn.Module, input_dim: int, hidden_dim: int) -> None:
        self.conv, self.cin, self.hid = conv, input_dim, hidden_dim
        self.cinp, self.hidp = cpad(input_dim), cpad(hidden_dim)
        self.fwd_map = K.lstm_fwd_map(input_dim, hidden_dim)
        self.bwd_maps = {nd: K.lstm_bwd_map(input_dim, hidden_dim, nd) for nd in (False, True)}
        # K order of the weight gradient: [h ; x] when the hidden lanes fill whole 64-channel tiles of the bf16-storage kernel (see lstm_wgrad_map)
        self.h_first = cpad(hidden_dim) % 64 == 0
        self.wgrad_map = K.lstm_wgrad_map(input_dim, hidden_dim, self.h_first)
        self._key = None
        self._packed = {}

    # ---- packed weights (derived cache, refreshed when the parameters change) ----
    def _refresh(self) -> None:
        w, b = self.conv.weight, self.conv.bias
        key = (w.data_ptr(), w._version, None if b is None else (b.data_ptr(), b._version), w.device, generation())
        if key != self._key:
            self._packed = {}
            self._key = key

    def packed_fwd(self) -> Tuple[Tensor, Optional[Tensor]]:
        self._refresh()
        if "fwd" not in self._packed:
            self._packed["fwd"] = K.pack_weights(self.conv.weight, self.conv.bias, self.fwd_map, transpose=False)
        return self._packed["fwd"]

    def packed_bwd(self, need_dx: bool) -> Tensor:
        self._refresh()
        k = ("bwd", need_dx)
        if k not in self._packed:
            self._packed[k] = K.pack_weights(self.conv.weight, None, self.bwd_maps[need_dx], transpose=True)[0]
        return self._packed[k]

    # ---- launches ----
    def step(self, x: sfTensor, h_prev: Optional[Tensor], c_prev: Optional[Tensor], n: int, h: int, w: int,
             h_out: Tensor, c_out: Tensor, gates: Optional[Tensor]) -> None:
        packed, bias = self.packed_fwd()
        K.convlstm_cell_fwd(x, T(h_prev, self.hidp), T(c_prev, self.hidp), n, h, w, packed, bias, self.hidp,
                            T(h_out), T(c_out), T(gates) if gates is not None else NULL)

    def bwd_gates(self, dh: Sequence[sfTensor], dc_next: Optional[Tensor], gates: Tensor, c_prev: Optional[Tensor],
                  c_new: Tensor, dz: Tensor, dc_prev: Optional[Tensor], amax: Optional[Tensor] = None) -> None:
        """``amax`` ("f32e" mode): a ZEROED device word the kernel raises to max |dz| - handed on to ``bwd_data`` / ``bwd_weight`` with dz."""
        pixels = gates.numel() // gates.shape[-1]
        # bf16-stored gates ("bf16a"): c' is taken again from them instead of read back - 4 of the 36 bytes per element of this HBM-bound pass
        # (SF_LSTM_READ_C=1: A/B switch); with fp32-stored gates the state is read (bit-exact backward of the parity mode)
        recompute = gates.dtype == torch.bfloat16 and not os.environ.get("SF_LSTM_READ_C")
        K.convlstm_cell_bwd_gates(dh, T(dc_next, self.hidp), T(gates), T(c_prev, self.hidp), NULL if recompute else T(c_new), pixels, self.hidp,
                                  T(dz, amax=amax if dz.dtype == torch.float32 else None), T(dc_prev, self.hidp))

    def bwd_data(self, dz: Tensor, n: int, h: int, w: int, need_dx: bool, dcat: Tensor, amax: Optional[Tensor] = None) -> None:
        """dcat[.., (cinp if need_dx) + hidp] = conv(dz, W^T flipped).  ``amax`` ("f32e" mode): dz's scale word as left by ``bwd_gates``; without
        it the word is taken here (one more pass over dz, ``K.grad_operand``)."""
        gm = self.bwd_maps[need_dx]
        src = K.grad_operand_with(dz, amax) if amax is not None else K.grad_operand(dz)
        K.conv3x3(src, NULL, n, h, w, self.packed_bwd(need_dx), None, gm, T(dcat))

    def bwd_weight(self, x: sfTensor, h_prev: sfTensor, dz: sfTensor, n: int, h: int, w: int, dw: Tensor, db: Optional[Tensor],
                   accumulate: bool) -> None:
        if self.h_first:
            K.conv3x3_bwd_weight(h_prev, x, dz, n, h, w, self.wgrad_map, dw, db, accumulate)
        else:
            K.conv3x3_bwd_weight(x, h_prev, dz, n, h, w, self.wgrad_map, dw, db, accumulate)


class _CellStepFn(torch.autograd.Function):
    """One cell step on NHWC tensors ``[N,H,W,Cp]``; ``h``/``c`` may be ``None`` (zero state)."""

    @staticmethod
    def forward(ctx, eng: CellEngine, x: Tensor, h: Optional[Tensor], c: Optional[Tensor], weight: Tensor, bias: Optional[Tensor]):
        n, H, W, _ = x.shape
        new = lambda ch: torch.empty(n, H, W, ch, dtype=torch.float32, device=x.device)
        h_out, c_out, gates = new(eng.hidp), new(eng.hidp), new(4 * eng.hidp)
        eng.step(T(x), h, c, n, H, W, h_out, c_out, gates)
        ctx.eng = eng
        ctx.has_state = (h is not None, c is not None)
        ctx.need_dx = x.requires_grad
        ctx.save_for_backward(x, h if h is not None else x.new_empty(0), c if c is not None else x.new_empty(0), c_out, gates)
        return h_out, c_out

    @staticmethod
    def backward(ctx, dh_out: Tensor, dc_out: Tensor):
        eng: CellEngine = ctx.eng
        x, h, c, c_out, gates = ctx.saved_tensors
        has_h, has_c = ctx.has_state
        n, H, W, _ = x.shape
        dh_out = dh_out.contiguous() if dh_out is not None else torch.zeros_like(c_out)
        dc_out = dc_out.contiguous() if dc_out is not None else None
        dz = torch.empty_like(gates)
        dc_prev = torch.empty_like(c_out) if has_c else None
        word = K.scale_words(1, x.device)   # ("f32e": dz's scale word, raised by the gate kernel; None otherwise)
        eng.bwd_gates([T(dh_out)], dc_out, gates, c if has_c else None, c_out, dz, dc_prev, word)
        need_dx = ctx.need_dx
        dx = dh = None
        if need_dx or has_h:
            width = (eng.cinp if need_dx else 0) + eng.hidp
            dcat = torch.empty(n, H, W, width, dtype=torch.float32, device=x.device)
            eng.bwd_data(dz, n, H, W, need_dx, dcat, word)
            if need_dx:
                dx = dcat[..., : eng.cinp].contiguous()
            if has_h:
                dh = dcat[..., width - eng.hidp :].contiguous()
        dw = torch.empty_like(eng.conv.weight)
        db = torch.empty_like(eng.conv.bias) if eng.conv.bias is not None else None
        # zero state == zero contribution to dW's h-columns; feed an explicit zero tensor for the K lanes
        hsrc = T(h) if has_h else T(torch.zeros(n, H, W, eng.hidp, dtype=torch.float32, device=x.device))
        eng.bwd_weight(T(x), hsrc, K.grad_operand_with(dz, word) if word is not None else K.grad_operand(dz), n, H, W, dw, db, accumulate=False)
        return None, dx, dh, dc_prev, dw, db


class ConvLSTMCell(nn.Module):
    def __init__(self, input_dim, hidden_dim, kernel_size, bias, conv_type: str = "standard"):
        """Same arguments as reference ``layers/ConvLSTM.py:8-40``."""
        super().__init__()
        self.input_dim = input_dim
        self.hidden_dim = hidden_dim
        self.kernel_size = kernel_size
        self.padding = kernel_size[0] // 2, kernel_size[1] // 2
        self.bias = bias
        if tuple(kernel_size) != (3, 3):
            raise NotImplementedError("the HIP cell implements the 3x3 kernels the reference models use (conv_lstm.py:132-162)")
        conv2d = get_conv_layer(conv_type)
        self.conv = conv2d(
            in_channels=input_dim + hidden_dim, out_channels=4 * hidden_dim, kernel_size=self.kernel_size,
            padding=self.padding, bias=bias,
        )
        self.engine = CellEngine(self.conv, input_dim, hidden_dim)

    def forward(self, input_tensor: Tensor, cur_state: Sequence[Tensor]) -> Tuple[Tensor, Tensor]:
        """``(h_next, c_next)`` for NCHW ``input_tensor`` and ``cur_state=(h, c)`` (reference ``:42-57``)."""
        from ..conv_lstm import nchw_to_nhwc, nhwc_to_nchw  # layout autograd ops

        h_cur, c_cur = cur_state
        require_device(input_tensor, "input_tensor")
        x = nchw_to_nhwc(input_tensor)
        h = nchw_to_nhwc(h_cur)
        c = nchw_to_nhwc(c_cur)
        h2, c2 = _CellStepFn.apply(self.engine, x, h, c, self.conv.weight, self.conv.bias)
        return nhwc_to_nchw(h2, self.hidden_dim), nhwc_to_nchw(c2, self.hidden_dim)

    def init_hidden(self, batch_size, image_size):
        height, width = image_size
        dev = self.conv.weight.device
        return (
            torch.zeros(batch_size, self.hidden_dim, height, width, device=dev),
            torch.zeros(batch_size, self.hidden_dim, height, width, device=dev),
        )

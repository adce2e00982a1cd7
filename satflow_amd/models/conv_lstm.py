"""ConvLSTM encoder-decoder on the HIP kernels.

Module surface of reference ``satflow/models/conv_lstm.py``: ``ConvLSTM(input_channels,
hidden_dim, out_channels, conv_type)`` with sub-modules ``encoder_1_convlstm,
encoder_2_convlstm, decoder_1_convlstm, decoder_2_convlstm, decoder_CNN`` (``:132-169``; the
``state_dict`` keys are pinned in ``tests/golden/convlstm_state_dict_keys.txt``) and the
registered Lightning wrapper ``EncoderDecoderConvLSTM`` (``:13-91``).

Execution differs from the reference by design.  ``ConvLSTM.forward`` (reference ``:171-228``)
is ONE autograd node, ``_StackFn``: the whole 2x(T_in+T_out) unroll runs as a sequence of fused
cell launches on time-major NHWC sequence buffers (hidden/cell/gate tensors of every step stay
resident in HBM - there is room for all of them in 288 GB - so nothing is re-materialised, no
``cat``/``stack``/``permute`` copies exist, and t=0 reads no state at all instead of
allocating zeros, ``:218-221``).  Its backward walks the unroll in reverse with one pointwise
launch + one input-gradient convolution per cell step, and defers every weight gradient to a
single split-K GEMM per cell over all timesteps at the end.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Tuple, Union
import os

import torch
from torch import nn

from .. import kernels as K
from ..functional import ConvEngine, _FromNHWC, _ToNHWC, conv3x3, grad_out, nchw_to_nhwc, nhwc_to_nchw  # noqa: F401 (re-exported)
from .._hip import (NULL, SF_EPI_LINEAR, SF_EPI_SIGMOID, T, cpad, gate_storage_dtype, generation, require_device, sfTensor,
                     state_storage_dtype)
from .base import LightningModule, get_loss, register_model
from .layers.ConvLSTM import CellEngine, ConvLSTMCell

Tensor = torch.Tensor


# ----------------------------------------------------------------------------------------------
# the unrolled 4-cell encoder-decoder as one autograd node
# ----------------------------------------------------------------------------------------------
_SIDE: Dict[Any, "torch.cuda.Stream"] = {}


def _diagonal() -> bool:
    """Diagonal (wavefront) order of the two encoder cells on two streams in the FORWARD pass (default since round 5: +0.5 % with the anti-phase
    backward below, +1 % with the side-stream weight gradients on top - small, but consistent over five A/B pairs); ``SF_LSTM_DIAG=0``: serial."""
    return os.environ.get("SF_LSTM_DIAG", "1") == "1"


def _diagonal_bwd() -> bool:
    """Anti-phase diagonal order of the two encoder cells' BACKWARD on two streams (round 5); ``SF_LSTM_DIAG_BWD=0`` keeps the serial order."""
    return os.environ.get("SF_LSTM_DIAG_BWD", "1") == "1"


def _streams_allowed() -> bool:
    """The two-stream schedules run in eager mode; under hipGraph capture they fall back to the serial order unless SF_LSTM_CAPTURE_STREAMS=1 (experiment:
    fork / join of the side streams inside the capture)."""
    return not torch.cuda.is_current_stream_capturing() or bool(os.environ.get("SF_LSTM_CAPTURE_STREAMS"))


def _side_stream(dev, which: int = 0) -> "torch.cuda.Stream":
    """Second streams of the stack, per device: 0 = the diagonal partner (encoder 2 forward / encoder 1 backward), 1 = the decoder cells' weight
    gradients (their own queue: on the partner's they would sit in front of encoder 1's first gate kernel)."""
    key = (str(dev), which)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=dev)
    return _SIDE[key]

class _StackFn(torch.autograd.Function):
    """x ``[T*B,H,W,Cp]`` time-major -> decoder-2 hidden states ``[Tout*B,H,W,hidp]`` time-major.

    Dataflow of reference ``conv_lstm.py:171-196``: encoder cells 1,2 over the T input frames;
    decoder cell 3 is fed ``h2[T-1]`` at the first forecast step and ``h4[s-1]`` afterwards
    (``:185,195``); decoder cell 4 is fed ``h3[s]``.
    """

    @staticmethod
    def forward(ctx, engines: List[CellEngine], B: int, T_in: int, T_out: int, x: Tensor, *params: Tensor):
        e1, e2, d1, d2 = engines
        H, W = x.shape[1], x.shape[2]
        hidp = e1.hidp
        dev = x.device
        keep = any(ctx.needs_input_grad)  # False under no_grad / inference: no gates are written

        def seq(steps: int, ch: int, dtype=torch.float32) -> Tensor:
            return torch.empty(steps, B, H, W, ch, dtype=dtype, device=dev)

        steps = (T_in, T_in, T_out, T_out)
        # hidden states: fp32, or bf16 in "bf16a" mode (they are only ever read as bf16 MFMA operands: same results, half
        # the traffic, LDS-DMA staging); the input frames follow them so that every convolution source has one storage type
        st = state_storage_dtype()
        need_dx = x.requires_grad
        if x.dtype != st:
            x = x.to(st)
        Hs = [seq(s, hidp, st) for s in steps]
        Cs = [seq(s, hidp) for s in steps]
        # saved gates (later overwritten by dz): backward-only data, bf16 in "bf16a" mode - half of the stack's HBM traffic
        Gs = [seq(s, 4 * hidp, gate_storage_dtype()) if keep else None for s in steps]
        xs = x.view(T_in, B, H, W, x.shape[-1])

        def run(k: int, eng: CellEngine, inp: Tensor, t: int) -> None:
            eng.step(T(inp), Hs[k][t - 1] if t else None, Cs[k][t - 1] if t else None, B, H, W, Hs[k][t], Cs[k][t],
                     Gs[k][t] if keep else None)

        if _diagonal() and T_in > 1 and _streams_allowed():
            # diagonal order (reference conv_lstm.py:176-182: encoder_2 at step t only needs encoder_1 at step t): encoder 2 runs one step
            # behind encoder 1 on a second stream, so two cell launches are in flight and one's epilogue meets the other's K loop
            main, side = torch.cuda.current_stream(dev), _side_stream(dev)
            e1.packed_fwd(), e2.packed_fwd()   # packed on the main stream, before the second stream reads them
            side.wait_stream(main)
            for t in range(T_in):
                run(0, e1, xs[t], t)
                ev = torch.cuda.Event()
                ev.record(main)
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    run(1, e2, Hs[0][t], t)
            main.wait_stream(side)
        else:
            for t in range(T_in):
                run(0, e1, xs[t], t)
                run(1, e2, Hs[0][t], t)
        for s in range(T_out):
            run(2, d1, Hs[1][T_in - 1] if s == 0 else Hs[3][s - 1], s)
            run(3, d2, Hs[2][s], s)

        ctx.engines, ctx.dims = engines, (B, T_in, T_out, H, W)
        ctx.need_dx = need_dx
        if keep:
            ctx.save_for_backward(x, *Hs, *Cs, *Gs)
        return Hs[3].view(T_out * B, H, W, hidp)

    @staticmethod
    def backward(ctx, g_out: Tensor):
        e1, e2, d1, d2 = engines = ctx.engines
        B, T_in, T_out, H, W = ctx.dims
        if getattr(ctx, "consumed", False):
            # the saved gates were overwritten in place by dz through raw pointers (invisible to autograd's version check)
            raise RuntimeError("ConvLSTM stack: backward was already run on this graph; its saved gate buffers were consumed in place. "
                               "A second backward over a retained graph is not supported - run the forward again.")
        ctx.consumed = True
        saved = ctx.saved_tensors
        x, Hs, Cs, Gs = saved[0], saved[1:5], saved[5:9], saved[9:13]
        hidp = e1.hidp
        dev = x.device
        # the head's gradient: summed in fp32 by the gate kernel whatever its storage (bf16 when the head convolution's input - the bf16-stored
        # states - is: no fp32 copy of it is made)
        g_out = g_out.contiguous()
        if g_out.dtype not in (torch.float32, torch.bfloat16) or os.environ.get("SF_LSTM_GOUT_F32"):   # (A/B switch: the fp32 copy)
            g_out = g_out.float()
        g_out = g_out.view(T_out, B, H, W, hidp)
        xs = x.view(T_in, B, H, W, x.shape[-1])
        need_dx = [ctx.need_dx, True, True, True]
        # per cell: [dx (if needed) ; dh_prev] scratch of the input-gradient conv, and the dc carry
        widths = [(eng.cinp if nd else 0) + hidp for eng, nd in zip(engines, need_dx)]
        # [dx ; dh] of the input-gradient convolutions: bf16 in "bf16a" mode (what a 16-bit autocast leaves between a convolution's backward and
        # the gate arithmetic; summed in fp32 by the gate backward) - half the stores of those launches, a tenth of the gate backward's bytes
        dcat_dt = torch.float32 if os.environ.get("SF_LSTM_DCAT_F32") else gate_storage_dtype()   # (A/B switch)
        dcat = [torch.empty(B, H, W, wd, dtype=dcat_dt, device=dev) for wd in widths]
        dc = [torch.empty(B, H, W, hidp, dtype=torch.float32, device=dev) for _ in range(4)]
        dxs = torch.empty(xs.shape, dtype=torch.float32, device=dev) if ctx.need_dx else None

        # "f32e" mode: one scale word per cell and step, raised to max |dz| by the gate kernel that writes dz and read by the input-gradient convolution
        # (and, as the maximum over a cell's steps, by its weight gradient) - None in every other mode
        Tmax = max(T_in, T_out)
        words = K.scale_words(4 * Tmax, dev)
        word = (lambda k, t: words[k * Tmax + t: k * Tmax + t + 1]) if words is not None else (lambda k, t: None)

        def dx_of(k: int) -> sfTensor:  # gradient wrt the cell's layer input, left by its last bwd_data
            return T(dcat[k], engines[k].cinp, 0)

        def dh_of(k: int) -> sfTensor:  # gradient wrt the cell's previous hidden state
            return T(dcat[k], hidp, widths[k] - hidp)

        def back(k: int, t: int, last: bool, sources: List[sfTensor]) -> None:
            """Backward of cell k at step t.  dz overwrites the saved gates in place."""
            eng = engines[k]
            if not last:
                sources = sources + [dh_of(k)]
            eng.bwd_gates(sources, None if last else dc[k], Gs[k][t], Cs[k][t - 1] if t else None, Cs[k][t], Gs[k][t],
                          dc[k] if t else None, word(k, t))
            if t or need_dx[k]:
                eng.bwd_data(Gs[k][t], B, H, W, need_dx[k], dcat[k], word(k, t))

        # weight gradients: one split-K GEMM per cell over all of its timesteps (Gs now hold dz)
        zeros = torch.zeros(B, H, W, hidp, dtype=Hs[0].dtype, device=dev)
        cell_grads: List[Optional[Tuple[Optional[Tensor], Optional[Tensor]]]] = [None] * 4
        done = [False] * 4

        def wgrad_cell(k: int) -> None:
            eng = engines[k]
            # (the parameters' own gradient slices when an optimizer registered them with functional.GRAD_SINK: no add_ by autograd afterwards)
            (dw, dw_ret), (db, db_ret) = grad_out(eng.conv.weight), grad_out(eng.conv.bias)
            steps = T_in if k < 2 else T_out
            # step 0 has a zero previous state; steps 1.. read h[t-1] straight from the sequence buffer
            if k == 0:
                first_in, rest_in = xs[0], xs[1:]
            elif k == 1:
                first_in, rest_in = Hs[0][0], Hs[0][1:]
            elif k == 2:
                first_in, rest_in = Hs[1][T_in - 1], Hs[3][: T_out - 1]
            else:
                first_in, rest_in = Hs[2][0], Hs[2][1:]
            # ("f32e": the gate gradients go in with their scale word - step 0's own, the maximum of the later steps' for the launch over all of them)
            eng.bwd_weight(T(first_in), T(zeros), K.grad_operand_with(Gs[k][0], word(k, 0)), B, H, W, dw, db, False)
            if steps > 1:
                rest = words[k * Tmax + 1: k * Tmax + steps].amax().reshape(1) if words is not None else None
                eng.bwd_weight(T(rest_in), T(Hs[k][: steps - 1]), K.grad_operand_with(Gs[k][1:], rest), (steps - 1) * B, H, W, dw, db, True)
            cell_grads[k] = (dw_ret, db_ret)
            done[k] = True

        for s in range(T_out - 1, -1, -1):
            last = s == T_out - 1
            # decoder 2: head gradient + (decoder 1 consumed h4[s] as its input at step s+1)
            back(3, s, last, [T(g_out[s])] + ([] if last else [dx_of(2)]))
            back(2, s, last, [dx_of(3)])
        # The decoder cells' dz are complete here.  Round 5 (SF_LSTM_WGRAD_SIDE=0: the A/B switch back): their weight gradients - MFMA-bound - go to a second
        # stream now, next to the encoder's backward unroll whose gate kernels are HBM-bound (complementary resources; measured: +0.5 %).
        wg_side = (os.environ.get("SF_LSTM_WGRAD_SIDE", "1") == "1" and _streams_allowed())
        if wg_side:
            main_s, side_s = torch.cuda.current_stream(dev), _side_stream(dev, 1)
            side_s.wait_stream(main_s)
            with torch.cuda.stream(side_s):
                wgrad_cell(3)
                wgrad_cell(2)
        if _diagonal_bwd() and T_in > 1 and _streams_allowed():
            # Diagonal backwards, in ANTI-PHASE (round 5): encoder 1 at step t only needs encoder 2's input gradient of step t, so it runs on a
            # second stream while encoder 2 goes on to step t-1.  The round-4 form released encoder 1's step as soon as its input existed - and a
            # kernel trace (tools/trace_overlap.sh) showed the two streams in lockstep: gate kernel next to gate kernel (both HBM-bound), input-gradient
            # convolution next to input-gradient convolution (both MFMA-bound), no gain.  Here encoder 1's gate kernel of step t is held until encoder
            # 2's gate kernel of step t-1 has FINISHED, i.e. it starts together with encoder 2's convolution: HBM-bound next to MFMA-bound, and the
            # two kernels fit one CU together (the convolution's 2 x 232 registers per SIMD lane leave room for one 48-register wave of the gate
            # kernel, tools/kernel_regs.py).  Encoder 2's [dx ; dh] scratch is double-buffered by the parity of t; it may overwrite a buffer only
            # after encoder 1's gate kernel has consumed the dx in it (two steps earlier).
            main, side = torch.cuda.current_stream(dev), _side_stream(dev)
            d1pair = [dcat[1], torch.empty_like(dcat[1])]
            e1.packed_bwd(need_dx[0]), e2.packed_bwd(True)   # packed on the main stream, before the second stream reads them
            side.wait_stream(main)
            read_done: List[Optional[torch.cuda.Event]] = [None, None]
            data_done: Optional[torch.cuda.Event] = None   # encoder 2's convolution of the step encoder 1 handles next

            def enc1(t: int, gate2_done: Optional[torch.cuda.Event]) -> None:
                side.wait_event(data_done)
                if gate2_done is not None:
                    side.wait_event(gate2_done)
                with torch.cuda.stream(side):
                    eng = e1
                    lastt = t == T_in - 1
                    src = [T(d1pair[t & 1], e2.cinp, 0)] + ([] if lastt else [dh_of(0)])
                    eng.bwd_gates(src, None if lastt else dc[0], Gs[0][t], Cs[0][t - 1] if t else None, Cs[0][t], Gs[0][t], dc[0] if t else None, word(0, t))
                    read_done[t & 1] = torch.cuda.Event()
                    read_done[t & 1].record(side)
                    if t or need_dx[0]:
                        eng.bwd_data(Gs[0][t], B, H, W, need_dx[0], dcat[0], word(0, t))
                    if ctx.need_dx:
                        dxs[t].copy_(dcat[0][..., : e1.cinp])

            for t in range(T_in - 1, -1, -1):
                last = t == T_in - 1
                src = ([dx_of(2)] if last else []) + ([] if last else [T(d1pair[(t + 1) & 1], hidp, widths[1] - hidp)])
                e2.bwd_gates(src, None if last else dc[1], Gs[1][t], Cs[1][t - 1] if t else None, Cs[1][t], Gs[1][t], dc[1] if t else None, word(1, t))
                if not last:
                    g_done = torch.cuda.Event()
                    g_done.record(main)
                    enc1(t + 1, g_done)
                if read_done[t & 1] is not None:
                    main.wait_event(read_done[t & 1])
                e2.bwd_data(Gs[1][t], B, H, W, True, d1pair[t & 1], word(1, t))
                data_done = torch.cuda.Event()
                data_done.record(main)
            enc1(0, None)
            main.wait_stream(side)  # also covers the allocator: every buffer the side stream touched is free for reuse on the main stream
        else:
            for t in range(T_in - 1, -1, -1):
                last = t == T_in - 1
                # encoder 2: at the last input step its h feeds decoder 1's first step
                back(1, t, last, [dx_of(2)] if last else [])
                back(0, t, last, [dx_of(1)])
                if ctx.need_dx:
                    dxs[t].copy_(dcat[0][..., : e1.cinp])

        for k in range(4):
            if not done[k]:
                wgrad_cell(k)
        if wg_side:
            main_s.wait_stream(side_s)  # (also covers the allocator: the side stream's buffers are free for reuse on the main stream)
        grads: List[Optional[Tensor]] = [g for pair in cell_grads for g in pair]
        gx = dxs.view(x.shape) if ctx.need_dx else None
        return (None, None, None, None, gx, *grads)


class ConvLSTM(nn.Module):
    def __init__(self, input_channels, hidden_dim, out_channels, conv_type: str = "standard"):
        super().__init__()
        cell = lambda cin: ConvLSTMCell(input_dim=cin, hidden_dim=hidden_dim, kernel_size=(3, 3), bias=True, conv_type=conv_type)
        self.encoder_1_convlstm = cell(input_channels)
        self.encoder_2_convlstm = cell(hidden_dim)
        self.decoder_1_convlstm = cell(hidden_dim)
        self.decoder_2_convlstm = cell(hidden_dim)
        self.decoder_CNN = nn.Conv3d(in_channels=hidden_dim, out_channels=out_channels, kernel_size=(1, 3, 3), padding=(0, 1, 1))
        self.hidden_dim, self.out_channels, self.input_channels = hidden_dim, out_channels, input_channels
        self._head = ConvEngine([hidden_dim], out_channels)

    def cells(self) -> List[ConvLSTMCell]:
        return [self.encoder_1_convlstm, self.encoder_2_convlstm, self.decoder_1_convlstm, self.decoder_2_convlstm]

    def forward(self, x: Tensor, forecast_steps: int = 0, hidden_state=None) -> Tensor:
        """``x[B,T,C,H,W] -> [B, out, forecast_steps, H, W]`` (reference ``conv_lstm.py:205-228``).

        ``hidden_state`` is accepted and ignored, as in the reference (``:205,218-221``).
        """
        require_device(x, "x")
        B, T_in, C, H, W = x.shape
        if C != self.input_channels:
            raise RuntimeError(f"expected {self.input_channels} input channels, got {C}")
        if forecast_steps < 1 or T_in < 1:
            raise RuntimeError("ConvLSTM needs at least one input frame and forecast_steps >= 1 (torch.stack of an empty list "
                               "fails in the reference too, conv_lstm.py:198)")
        x = x.float()
        # [T*B,H,W,Cp], written directly in the storage type the stack reads its frames in (bf16 in "bf16a" mode: no fp32 copy + cast pass)
        xs = _ToNHWC.apply(x, B, T_in, C, H, W, (T_in * C * H * W, C * H * W, H * W),
                            torch.float32 if os.environ.get("SF_LSTM_X_F32") else state_storage_dtype())   # (A/B switch)
        cells = self.cells()
        params = [p for c in cells for p in (c.conv.weight, c.conv.bias)]
        hseq = _StackFn.apply([c.engine for c in cells], B, T_in, forecast_steps, xs, *params)
        y = conv3x3(self._head, hseq, self.decoder_CNN.weight, self.decoder_CNN.bias, sigmoid=True)
        O, To = self.out_channels, forecast_steps
        return _FromNHWC.apply(y, (B, O, To, H, W), B, To, O, H, W, (O * To * H * W, H * W, To * H * W))


@register_model
class EncoderDecoderConvLSTM(LightningModule):
    def __init__(
        self,
        hidden_dim: int = 64,
        input_channels: int = 12,
        out_channels: int = 1,
        forecast_steps: int = 48,
        lr: float = 0.001,
        visualize: bool = False,
        loss: Union[str, torch.nn.Module] = "mse",
        pretrained: bool = False,
        conv_type: str = "standard",
    ):
        """Same keyword surface as reference ``conv_lstm.py:15-33`` (so ``configs/model/convlstm.yaml`` loads)."""
        super().__init__()
        self.forecast_steps = forecast_steps
        self.criterion = get_loss(loss)
        self.lr = lr
        self.visualize = visualize
        self.model = ConvLSTM(input_channels, hidden_dim, out_channels, conv_type=conv_type)
        self.save_hyperparameters()

    @classmethod
    def from_config(cls, config):
        return EncoderDecoderConvLSTM(
            hidden_dim=config.get("num_hidden", 64),
            input_channels=config.get("in_channels", 12),
            out_channels=config.get("out_channels", 1),
            forecast_steps=config.get("forecast_steps", 1),
            lr=config.get("lr", 0.001),
        )

    def forward(self, x, future_seq=0, hidden_state=None):
        return self.model.forward(x, future_seq, hidden_state)

    def configure_optimizers(self):
        return torch.optim.Adam(self.parameters(), lr=self.lr)

    # -- steps: same metric names as the reference (train/loss, {train,val}/frame_{f}_loss), but the
    #    per-frame losses come from ONE reduction and ONE host sync instead of forecast_steps .item() calls
    #    (reference conv_lstm.py:66-68,80-82).
    def _frame_losses(self, y_hat: Tensor, y: Tensor) -> Tensor:
        frames = getattr(self.criterion, "last_frame_losses", None)  # produced by the fused loss kernel in the same pass
        return frames if frames is not None else ((y_hat.detach() - y) ** 2).mean(dim=(0, 2, 3, 4))

    def _log_frames(self, prefix: str, frames: Tensor) -> None:
        # logged as 0-dim device tensors: no host sync in the step (Lightning reduces them at epoch end)
        self.log_dict({f"{prefix}/frame_{f}_loss": v for f, v in enumerate(frames.unbind(0))}, on_step=False, on_epoch=True)

    def training_step(self, batch, batch_idx):
        x, y = batch
        y_hat = self(x, self.forecast_steps)
        y_hat = torch.permute(y_hat, dims=(0, 2, 1, 3, 4))
        loss = self.criterion(y_hat, y)
        self.log("train/loss", loss, on_step=True)
        self._log_frames("train", self._frame_losses(y_hat, y))
        return loss

    def validation_step(self, batch, batch_idx):
        x, y = batch
        y_hat = self(x, self.forecast_steps)
        y_hat = torch.permute(y_hat, dims=(0, 2, 1, 3, 4))
        val_loss = self.criterion(y_hat, y)
        self.log("val/loss", val_loss, on_step=True, on_epoch=True)
        self._log_frames("val", self._frame_losses(y_hat, y))
        return val_loss

    def test_step(self, batch, batch_idx):
        x, y = batch
        y_hat = self(x, self.forecast_steps)
        return self.criterion(y_hat, y)

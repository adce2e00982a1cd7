"""The ConvGRU the reference's generator imports (``satflow/models/layers/Generator.py:5``: ``from satflow.models.layers.ConvGRU import
ConvGRU``) - a module that is MISSING from the reference tree.  Restated from the DVD-GAN implementation ``Generator.py`` was taken
from (parity UNPINNED, oracle/dgmr.py): ``ConvGRU(input_size, hidden_sizes, kernel_sizes, n_layers)`` stacks ``ConvGRUCell``s
(sub-modules ``ConvGRUCell_00`` ...), ``forward(x, hidden=None)`` returns the list of the layers' new hidden states:

    z = sig(update_gate([x; h])),  r = sig(reset_gate([x; h])),  n = tanh(out_gate([x; r * h])),  h' = h (1 - z) + n z

Execution on the HIP kernels: every gate convolution is split into its x-part and its h-part (a convolution over a concatenation is
the sum of the convolutions of the parts), so nothing is concatenated; over a SEQUENCE (``run_sequence``) the x-parts of a layer
run for all frames in one launch and only the h-parts + the two fused gate stages (``sf_dvdgru_*``) are sequential.
3x3 kernels run on the MFMA convolution, and so do 5x5 kernels: a single call as ONE 3x3 convolution over four shifted copies of the padded input
(``functional_gan.conv5x5_as_3x3``), a sequence as ONE 3x3 convolution on the half-resolution layout (2x2 pixel blocks folded into channels,
``functional_gan.space_to_depth2`` / ``regroup5x5_s2d``: no padded domain, no copies, no crop; ``SF_GRU5_STACK4=1`` is the A/B switch back);
other sizes on the direct ``sf_conv2d_*``.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as TF
from torch import Tensor
from torch.nn import init

from ... import functional as F
from ... import functional_gan as FG
from ..._hip import cpad, require_device


class ConvGRUCell(nn.Module):
    def __init__(self, input_size, hidden_size, kernel_size):
        super().__init__()
        padding = kernel_size // 2
        self.input_size, self.hidden_size, self.kernel_size = input_size, hidden_size, kernel_size
        self.reset_gate = nn.Conv2d(input_size + hidden_size, hidden_size, kernel_size, padding=padding)
        self.update_gate = nn.Conv2d(input_size + hidden_size, hidden_size, kernel_size, padding=padding)
        self.out_gate = nn.Conv2d(input_size + hidden_size, hidden_size, kernel_size, padding=padding)
        for conv in (self.reset_gate, self.update_gate, self.out_gate):
            init.orthogonal_(conv.weight)
            init.constant_(conv.bias, 0.0)
        hp = cpad(hidden_size)
        self._eng = {}
        self._hp = hp

    # ---- weight views: gate-major output lanes (each gate padded to hidp rows), x-part / h-part columns ----
    def _rows(self, w: Tensor) -> Tensor:
        return TF.pad(w, (0, 0, 0, 0, 0, 0, 0, self._hp - self.hidden_size)) if self._hp != self.hidden_size else w

    def _as_tiles(self) -> bool:
        return self.kernel_size == 5 and not os.environ.get("SF_CONV5_DIRECT")

    def _engine(self, tag: str, cin: int, cout: int) -> "F.ConvEngine":
        key = (tag, cin, cout)
        if key not in self._eng:
            self._eng[key] = F.ConvEngine([cin], cout)
        # the regrouped weights are fresh tensors every call but functions of the cell's parameters only: the packed images are
        # cached on THOSE (identity + version + optimizer generation) - one pack per step, not one per frame
        self._eng[key].key_tensors = tuple(self.parameters())
        return self._eng[key]

    def shift4_ok(self, x_lanes: int) -> bool:
        """Can this 5x5 cell's four convolutions take the shifted-view route (``functional_gan.conv5x5_shift4``) on an input of ``x_lanes`` lanes?"""
        hp, hid = self._hp, self.hidden_size
        pairs = (("zr_x", x_lanes, 2 * hp, self.input_size), ("o_x", x_lanes, hp, self.input_size), ("zr_h", hp, 2 * hp, hid), ("o_h", hp, hp, hid))
        return self._as_tiles() and all(FG.conv5x5_shift4_ok(li, rows, self._engine("s4:" + tag, 4 * li, rows), self._engine("s4t:" + tag, 4 * rows, ci))
                                        for tag, li, rows, ci in pairs)

    def weights(self, x_lanes: Optional[int] = None, s2d: bool = False, shift4: bool = False):
        """The six derived weights of a call / a sequence.  5x5 kernels are regrouped HERE into the weight of the 3x3 convolution over four shifted
        copies (``x_lanes``: channel lanes of the cell's input tensor) - once per sequence, not once per frame.  ``s2d``: for tensors in the
        ``space_to_depth2`` layout instead (``run_sequence``): weights by ``regroup5x5_s2d``, biases repeated per output phase."""
        ci, hp, hid = self.input_size, self._hp, self.hidden_size
        wz, wr, wo = self._rows(self.update_gate.weight), self._rows(self.reset_gate.weight), self._rows(self.out_gate.weight)
        pb = (lambda b: TF.pad(b, (0, hp - hid))) if hp != hid else (lambda b: b)
        wzr = torch.cat((wz, wr), 0)
        W = dict(zr_x=wzr[:, :ci], zr_h=wzr[:, ci:], o_x=wo[:, :ci], o_h=wo[:, ci:],
                 b_zr=torch.cat((pb(self.update_gate.bias), pb(self.reset_gate.bias)), 0), b_o=pb(self.out_gate.bias))
        if self._as_tiles():
            xl = cpad(ci) if x_lanes is None else x_lanes
            for k, lanes in (("zr_x", xl), ("o_x", xl), ("zr_h", hp), ("o_h", hp)):
                if shift4:   # + the input-gradient kernel (flipped, transposed) of the shifted-view route, over the lanes of the output gradient
                    W[k + "_t"] = FG.regroup5x5_transposed(W[k], W[k].shape[0])
                W[k] = FG.regroup5x5_s2d(W[k], lanes, hp) if s2d else FG.regroup5x5(W[k], lanes)
            if shift4:
                W["shift4"] = True
            if s2d:
                W["b_zr"] = W["b_zr"].view(2, 1, hp).expand(2, 4, hp).reshape(-1)
                W["b_o"] = W["b_o"].view(1, hp).expand(4, hp).reshape(-1)
                W["s2d"] = True
        else:
            for k in ("zr_x", "zr_h", "o_x", "o_h"):
                W[k] = W[k].contiguous()
        return W

    def _conv(self, tag: str, x: Tensor, w: Tensor, b: Optional[Tensor], wbatch=None, s2d: bool = False, wt: Optional[Tensor] = None) -> Tensor:
        k = self.kernel_size
        if wt is not None:   # shifted-view route: w = regroup5x5(weight), wt = regroup5x5_transposed(weight)
            # the weight's identity is part of the engine key: with input lanes == hidden lanes the x-part and h-part weights have the
            # same shape, and one engine's pack cache (keyed on the cell's parameters) would hand the h-convolution the x weights
            eng, eng_t = self._engine("s4:" + tag, w.shape[1], w.shape[0]), self._engine("s4t:" + tag, wt.shape[1], wt.shape[0])
            return FG.conv5x5_shift4(x, w, wt, b, eng, eng_t, wbatch)
        if s2d:
            k = 3   # weights and tensors are in the space_to_depth2 layout: a plain 3x3 convolution
        if k == 3 or self._as_tiles():
            cin = w.shape[1]   # 5x5: already regrouped over four shifted copies of the (padded) input lanes
            key = (tag, cin, w.shape[0])
            if key not in self._eng:
                self._eng[key] = F.ConvEngine([cin], w.shape[0])
            # the regrouped weights are fresh tensors every call but functions of the cell's parameters only: the packed images are
            # cached on THOSE (identity + version + optimizer generation) - one pack per step, not one per frame
            self._eng[key].key_tensors = tuple(self.parameters())
            return F.conv3x3(self._eng[key], x, w, b, wbatch=wbatch) if k == 3 else FG.conv5x5_as_3x3(x, w, b, self._eng[key], wbatch)
        return FG.conv_nhwc(x, w, b)   # any other size (and SF_CONV5_DIRECT=1, the A/B switch): the direct fp32 convolution

    def x_parts(self, x: Tensor, W: dict):
        """x-parts of the three gates for any number of frames at once: ``(gx_zr [.., 2*hidp], gx_o [.., hidp])`` incl. the biases."""
        s2d = bool(W.get("s2d"))
        return (self._conv("zr_x", x, W["zr_x"], W["b_zr"], s2d=s2d, wt=W.get("zr_x_t")),
                self._conv("o_x", x, W["o_x"], W["b_o"], s2d=s2d, wt=W.get("o_x_t")))

    def step(self, gx_zr: Tensor, gx_o: Tensor, h: Optional[Tensor], W: dict, out: Optional[Tensor] = None, out_rh: Optional[Tensor] = None,
             gslots=(None, None)) -> Tensor:
        """``out`` / ``out_rh``: where the new state / the reset state r * h is written (a frame's slot of ``functional_gan.sequence_slots``: the inputs
        of the two state convolutions of all frames then lie back to back, and their batched weight gradients read them in place)."""
        s2d = bool(W.get("s2d"))
        hp = 4 * self._hp if s2d else self._hp   # the pointwise stages are layout-blind: [z | r] halves of 4 * hidp lanes each
        # gslots: where the backward pass writes this frame's two pre-activation gradients (functional_gan.GradSlots: the gradients of the x-parts of all
        # frames - which are also the output gradients of the state convolutions - then lie back to back)
        if h is None:  # zero state: the h-parts vanish
            zr, _ = FG.dvdgru_gates(gx_zr, None, None, hp, None, gslots[0])
            return FG.dvdgru_out(gx_o, None, zr, None, hp, out, gslots[1])
        gh_zr = self._conv("zr_h", h, W["zr_h"], None, W.get("batch_zr_h"), s2d, W.get("zr_h_t"))
        zr, rh = FG.dvdgru_gates(gx_zr, gh_zr, h, hp, out_rh, gslots[0])
        gh_o = self._conv("o_h", rh, W["o_h"], None, W.get("batch_o_h"), s2d, W.get("o_h_t"))
        return FG.dvdgru_out(gx_o, gh_o, zr, h, hp, out, gslots[1])

    def run(self, x: Tensor, h: Optional[Tensor]) -> Tensor:
        W = self.weights(x.shape[-1])
        return self.step(*self.x_parts(x, W), h, W)

    def forward(self, input_, prev_state=None):
        require_device(input_, "input")
        h = F.nchw_to_nhwc(prev_state.float()) if prev_state is not None else None
        return F.nhwc_to_nchw(self.run(F.nchw_to_nhwc(input_.float()), h), self.hidden_size)


class ConvGRU(nn.Module):
    def __init__(self, input_size, hidden_sizes, kernel_sizes, n_layers):
        super().__init__()
        self.input_size = input_size
        self.hidden_sizes = [hidden_sizes] * n_layers if not isinstance(hidden_sizes, (list, tuple)) else list(hidden_sizes)
        self.kernel_sizes = [kernel_sizes] * n_layers if not isinstance(kernel_sizes, (list, tuple)) else list(kernel_sizes)
        assert len(self.hidden_sizes) == n_layers and len(self.kernel_sizes) == n_layers
        self.n_layers = n_layers
        cells = []
        for i in range(n_layers):
            cell = ConvGRUCell(input_size if i == 0 else self.hidden_sizes[i - 1], self.hidden_sizes[i], self.kernel_sizes[i])
            name = "ConvGRUCell_" + str(i).zfill(2)
            setattr(self, name, cell)
            cells.append(getattr(self, name))
        self.cells = cells

    def run(self, x: Tensor, hidden: Optional[List[Optional[Tensor]]] = None) -> List[Tensor]:
        """One call on NHWC tensors: every layer's new hidden state."""
        hidden = hidden or [None] * self.n_layers
        out, inp = [], x
        for cell, h in zip(self.cells, hidden):
            inp = cell.run(inp, h)
            out.append(inp)
        return out

    def run_sequence(self, x: Tensor, T_frames: int, constant_input: bool) -> Tensor:
        """The generator's frame loop (reference ``Generator.py:91-117``): frame i calls the stack with frame i-1's hidden list.
        ``x``: time-major NHWC ``[T*n,H,W,Cp]`` (or ``[n,H,W,Cp]`` fed to every frame when ``constant_input``).  Returns the last
        layer's states, time-major ``[T*n,H,W,hidp]``.  Layer by layer: a layer's x-parts for all frames in one launch."""
        seq, const, folded = x, constant_input, False
        for cell in self.cells:
            # 5x5 cells run on the half-resolution layout (2x2 pixel blocks folded into channels: sf_space_to_depth2) - ONE permutation of the
            # sequence on the way in and one on the way out instead of a padded, four-times-copied input and a crop per convolution
            # ... or, in the 16-bit modes with lanes in whole 32-channel tiles, on the shifted-view route: the 3x3 kernels read the four displaced views of
            # the unchanged tensors in their loaders - no permutation either, and none of the half-resolution form's 4x weight bytes
            lanes_in = seq.shape[-1] // 4 if folded else seq.shape[-1]
            shift = not os.environ.get("SF_GRU5_STACK4") and cell.shift4_ok(lanes_in)
            fold = (not shift and cell._as_tiles() and seq.shape[1] % (1 if folded else 2) == 0 and seq.shape[2] % (1 if folded else 2) == 0
                    and not os.environ.get("SF_GRU5_STACK4"))
            if fold != folded:
                seq, folded = (FG.space_to_depth2(seq) if fold else FG.depth_to_space2(seq)), fold
            W = cell.weights(seq.shape[-1] // 4 if fold else seq.shape[-1], s2d=fold, shift4=shift)
            if cell.kernel_size in (3, 5) and not os.environ.get("SF_GRU_WGRAD_PER_FRAME"):
                # the state convolutions' weight gradients: once per sequence over all frames, not once per frame (functional.WeightGradBatch)
                W["batch_zr_h"], W["batch_o_h"] = F.WeightGradBatch(), F.WeightGradBatch()
            gx_zr, gx_o = cell.x_parts(seq, W)
            n = gx_zr.shape[0] if const else gx_zr.shape[0] // T_frames
            # per-frame views through unbind (its backward is ONE stack, not a zero-filled full-size tensor per slice)
            slots_ok = not os.environ.get("SF_GRU_CAT")
            gz = go = None
            if const:
                zr_t, o_t = [gx_zr] * T_frames, [gx_o] * T_frames
            elif slots_ok:   # frames as tensors of their own + one gradient buffer each for the way back (no torch.stack of the frames' gradients)
                (zr_t, gz), (o_t, go) = FG.split_frames(gx_zr, T_frames), FG.split_frames(gx_o, T_frames)
            else:
                zr_t, o_t = gx_zr.view(T_frames, n, *gx_zr.shape[1:]).unbind(0), gx_o.view(T_frames, n, *gx_o.shape[1:]).unbind(0)
            h, outs = None, []
            # every frame's state is written straight into its slice of the layer's output sequence (no torch.cat of the frames afterwards)
            if slots_ok:
                slot_shape = (*o_t[0].shape[:-1], 4 * cell._hp if fold else cell._hp)
                buf, slots = FG.sequence_slots(T_frames, slot_shape, o_t[0].device)
                _, rh_slots = FG.sequence_slots(max(T_frames - 1, 1), slot_shape, o_t[0].device)   # frames 1.. have a reset state
            for t in range(T_frames):
                h = cell.step(zr_t[t], o_t[t], h, W, slots[t] if slots_ok else None, rh_slots[t - 1] if slots_ok and t else None,
                              ((gz, t) if gz is not None else None, (go, t) if go is not None else None))
                outs.append(h)
            seq, const = (FG.assemble(buf, outs) if slots_ok else torch.cat(outs, 0)), False
        return FG.depth_to_space2(seq) if folded else seq

    def forward(self, x, hidden=None):
        require_device(x, "x")
        hs = [F.nchw_to_nhwc(h.float()) if h is not None else None for h in hidden] if hidden else None
        return [F.nhwc_to_nchw(o, c) for o, c in zip(self.run(F.nchw_to_nhwc(x.float()), hs), self.hidden_sizes)]

"""Oracle: ConvLSTM cell and 4-cell encoder-decoder (CPU, torch fp32, functional).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Parity PINNED against
the reference import, see ``tests/golden/make_golden.py``.

Every function takes the weights explicitly (reference ``state_dict`` layout)
so a test can feed the very same numbers to the HIP path and to this file.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def convlstm_cell(
    x: Tensor, h: Tensor, c: Tensor, weight: Tensor, bias: Optional[Tensor]
) -> Tuple[Tensor, Tensor]:
    """One ConvLSTM step.

    Follows reference ``satflow/models/layers/ConvLSTM.py:42-57``:
    ``z = conv([x ; h])`` (x channels first, ``:45``), gate order i, f, o, g
    (``:48``), ``c' = f*c + i*g`` (``:54``), ``h' = o*tanh(c')`` (``:55``).
    ``weight`` is ``[4*hid, cin+hid, kh, kw]`` (``:34-40``), "same" padding
    (``:29``).
    """
    hid = h.shape[1]
    kh, kw = weight.shape[-2:]
    z = F.conv2d(torch.cat((x, h), 1), weight, bias, padding=(kh // 2, kw // 2))
    zi, zf, zo, zg = z[:, :hid], z[:, hid : 2 * hid], z[:, 2 * hid : 3 * hid], z[:, 3 * hid :]
    c_new = torch.sigmoid(zf) * c + torch.sigmoid(zi) * torch.tanh(zg)
    h_new = torch.sigmoid(zo) * torch.tanh(c_new)
    return h_new, c_new


CELLS: Sequence[str] = (
    "encoder_1_convlstm",
    "encoder_2_convlstm",
    "decoder_1_convlstm",
    "decoder_2_convlstm",
)


def convlstm_forward(x: Tensor, forecast_steps: int, params: Dict[str, Tensor]) -> Tensor:
    """Encoder-decoder forward, ``x[B,T,C,H,W] -> [B,out,forecast_steps,H,W]``.

    Follows reference ``satflow/models/conv_lstm.py:171-228``: zero initial
    states (``:218-221`` -> ``layers/ConvLSTM.py:59-64``); encoder loop over
    the T input frames through cells 1 and 2 (``:176-182``); the decoder's
    first input is the encoder's last ``h2`` and afterwards its own ``h4`` fed
    back (``:185-196``); the ``h4`` of every forecast step are stacked on a
    new time axis behind the channel axis, pushed through
    ``Conv3d(hid->out, (1,3,3), pad (0,1,1))`` and a sigmoid (``:198-201``).

    ``params`` uses the reference's ``ConvLSTM.state_dict()`` keys, e.g.
    ``encoder_1_convlstm.conv.weight`` and ``decoder_CNN.weight``.
    """
    B, T, _, H, W = x.shape
    hid = params["encoder_1_convlstm.conv.bias"].shape[0] // 4
    zeros = lambda: torch.zeros(B, hid, H, W, dtype=x.dtype, device=x.device)
    state = {name: (zeros(), zeros()) for name in CELLS}

    def step(name: str, inp: Tensor) -> Tensor:
        hc = convlstm_cell(
            inp, *state[name], params[f"{name}.conv.weight"], params[f"{name}.conv.bias"]
        )
        state[name] = hc
        return hc[0]

    feed = None
    for t in range(T):
        feed = step("encoder_2_convlstm", step("encoder_1_convlstm", x[:, t]))
    frames = []
    for _ in range(forecast_steps):
        feed = step("decoder_2_convlstm", step("decoder_1_convlstm", feed))
        frames.append(feed)
    seq = torch.stack(frames, 2)  # [B, hid, Tout, H, W]
    y = F.conv3d(seq, params["decoder_CNN.weight"], params["decoder_CNN.bias"], padding=(0, 1, 1))
    return torch.sigmoid(y)


def training_loss(
    x: Tensor, y: Tensor, forecast_steps: int, params: Dict[str, Tensor]
) -> Tuple[Tensor, Tensor]:
    """MSE training loss + per-frame losses.

    Follows ``EncoderDecoderConvLSTM.training_step``
    (``satflow/models/conv_lstm.py:53-70``): the prediction is permuted to
    ``[B,T,C,H,W]`` (``:56``), ``loss = mse(y_hat, y)`` (``:63``), and one MSE
    per forecast frame (``:66-68``).  Returns ``(loss, frame_losses[Tout])``.
    """
    y_hat = convlstm_forward(x, forecast_steps, params).permute(0, 2, 1, 3, 4)
    loss = F.mse_loss(y_hat, y)
    frames = ((y_hat - y) ** 2).mean(dim=(0, 2, 3, 4))
    return loss, frames

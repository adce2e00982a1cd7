"""GPU parity at BASELINE.json's full sizes (fp32 mode): forward of one sample of configs[1] and configs[2] against the
CPU oracle, plus size-independent properties of the full-batch kernels (linearity of the convolution, gradient of a
linear functional).  The oracle side runs on the host cores (seconds per sample)."""
import pytest
import torch

from conftest import assert_close

pytestmark = pytest.mark.gpu


def test_convlstm_cfg2_forward_fullsize(device):
    """configs[1]: EncoderDecoderConvLSTM 12ch 128x128, T=12 -> 6, hidden 64, one sample, vs oracle/convlstm.py."""
    from oracle import convlstm as O
    from satflow_amd.models import EncoderDecoderConvLSTM

    torch.manual_seed(1234)
    m = EncoderDecoderConvLSTM(hidden_dim=64, input_channels=12, out_channels=12, forecast_steps=6).to(device)
    x = torch.rand(1, 12, 12, 128, 128, generator=torch.Generator().manual_seed(1234))
    torch.set_num_threads(min(32, torch.get_num_threads()))
    params = {k: v.detach().cpu() for k, v in m.model.state_dict().items()}
    ref = O.convlstm_forward(x, 6, params)
    with torch.no_grad():
        out = m(x.to(device), 6)
    assert out.shape == (1, 12, 6, 128, 128)
    assert_close(out, ref, "cfg2 prediction")


def test_metnet_cfg3_forward_fullsize(device):
    """configs[2]: LitMetNet 12ch 256x256, T=24 -> 12 lead times, hidden 64, one sample, training-mode BatchNorm
    (statistics per lead-time call), dropout off, vs oracle/metnet.py."""
    from oracle import metnet as M
    from satflow_amd.models import LitMetNet

    torch.manual_seed(1234)
    m = LitMetNet(input_channels=12, sat_channels=12, input_size=64, output_channels=12, hidden_dim=64, forecast_steps=12,
                  temporal_dropout=0.0).to(device)
    m.model.temporal_enc.rnn.input_p = 0.0
    m.train()
    x = torch.randn(1, 24, 12, 256, 256, generator=torch.Generator().manual_seed(1234))
    torch.set_num_threads(min(32, torch.get_num_threads()))
    P = {k: v.detach().cpu() for k, v in m.model.state_dict().items() if v.dtype == torch.float32 and "running" not in k}
    with torch.no_grad():
        ref = M.metnet_forward(x, P, sat_channels=12, input_size=64, forecast_steps=12)
        out = m(x.to(device))
    assert out.shape == (1, 12, 12, 16, 16)
    assert_close(out, ref, "cfg3 prediction")


def test_conv_linearity_and_adjoint_fullbatch(device):
    """Properties at the bench's batch (2304 images of 32x32, 256 -> 256 channels), fp32 mode:
    conv(a*x1 + x2) == a*conv(x1) + conv(x2) - (a+1)*bias-free part, and <conv(x), g> == <x, conv^T(g)> (dgrad is the adjoint)."""
    from satflow_amd import kernels as K
    from satflow_amd._hip import NULL, T
    from satflow_amd.functional import ConvEngine

    n, C, H, W = 2304, 256, 32, 32
    g = torch.Generator(device="cpu").manual_seed(5)
    eng = ConvEngine([C], C)
    w = (torch.randn(C, C, 3, 3, generator=g) * 0.02).to(device)
    fwd, _ = K.pack_weights(w, None, eng.fwd_map, False)
    bwd, _ = K.pack_weights(w, None, eng.bwd_map((True,)), True)
    x1 = torch.randn(n, H, W, C, device=device)
    x2 = torch.randn(n, H, W, C, device=device)
    conv = lambda x, pk, gm: (lambda y: (K.conv3x3(T(x), NULL, n, H, W, pk, None, gm, T(y)), y)[1])(torch.empty(n, H, W, C, device=device))
    y1, y2 = conv(x1, fwd, eng.fwd_map), conv(x2, fwd, eng.fwd_map)
    y12 = conv(2.5 * x1 + x2, fwd, eng.fwd_map)
    lin = (y12 - (2.5 * y1 + y2)).abs().max() / y12.abs().max()
    assert float(lin) < 1e-5, float(lin)
    gy = torch.randn(n, H, W, C, device=device)
    gx = conv(gy, bwd, eng.bwd_map((True,)))
    lhs, rhs = (y1.double() * gy.double()).sum(), (x1.double() * gx.double()).sum()
    # both sides are sums of 6e8 products with heavy cancellation: bound the mismatch by the Cauchy-Schwarz scale
    scale = float(y1.double().norm() * gy.double().norm())
    assert abs(float(lhs - rhs)) <= 1e-6 * scale, (float(lhs), float(rhs), scale)


def test_wgrad_bf16_storage_adjoint_fullbatch(device):
    """The all-bf16-storage weight-gradient kernel (LDS-DMA ring + transposing LDS reads) at the bench's batch (2304 images of
    32x32, 256 -> 256 channels): <dW, V> == <conv(x; V), gy> for a random direction V, i.e. the weight gradient is the
    adjoint of the (oracle-checked) forward kernel in its weight argument; db == sum of gy.  bf16-representable operands, so
    both sides see identical MFMA inputs and differ only by fp32 summation order."""
    import satflow_amd
    from satflow_amd import kernels as K
    from satflow_amd._hip import NULL, T
    from satflow_amd.functional import ConvEngine

    satflow_amd.set_compute_dtype("bf16a")
    try:
        n, C, H, W = 2304, 256, 32, 32
        g = torch.Generator(device="cpu").manual_seed(7)
        eng = ConvEngine([C], C)
        x = torch.randn(n, H, W, C, device=device).bfloat16()
        gy = torch.randn(n, H, W, C, device=device).bfloat16()
        V = (torch.randn(C, C, 3, 3, generator=g) * 0.05).bfloat16().float().to(device)
        dw, db = torch.empty(C, C, 3, 3, device=device), torch.empty(C, device=device)
        K.conv3x3_bwd_weight(T(x), NULL, T(gy), n, H, W, eng.wgrad_map, dw, db, False)
        packed, _ = K.pack_weights(V, None, eng.fwd_map, False)
        y = torch.empty(n, H, W, C, device=device)  # fp32 output: the pairing below must not see a bf16 rounding of conv(x; V)
        K.conv3x3(T(x), NULL, n, H, W, packed, None, eng.fwd_map, T(y))
        lhs = float((dw.double() * V.double()).sum())
        rhs = float((y.double() * gy.double()).sum())
        scale = float(dw.double().norm() * V.double().norm())
        assert abs(lhs - rhs) <= 1e-5 * scale, (lhs, rhs, scale)
        assert_close(db, gy.float().sum(dim=(0, 1, 2)), "db", grad=True)
        # accumulate=True adds into dW
        dw2 = dw.clone()
        K.conv3x3_bwd_weight(T(x), NULL, T(gy), n, H, W, eng.wgrad_map, dw2, None, True)
        assert_close(dw2, 2 * dw, "accumulated dW", grad=True)
    finally:
        satflow_amd.set_compute_dtype("f32")

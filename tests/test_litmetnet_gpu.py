"""GPU parity of the LitMetNet Lightning surface (reference satflow/models/pl_metnet.py:67-124): dict batches
through `_combine_data_sources`, `training_step` / `validation_step` with their metric names, and the gradient wrt the input
images (sf_metnet_preprocess_bwd).  The oracle side is oracle/metnet.py (PARITY UNPINNED for the MetNet arithmetic)."""
import pytest
import torch
import torch.nn.functional as TF

from conftest import assert_close
from oracle import metnet as M

pytestmark = pytest.mark.gpu


def _g(seed):
    return torch.Generator().manual_seed(seed)


CFG = dict(input_channels=5, sat_channels=4, input_size=8, output_channels=2, hidden_dim=16, forecast_steps=3)


def _lit(device, dropout=0.0):
    from satflow_amd.models import LitMetNet

    torch.manual_seed(7)
    m = LitMetNet(**CFG, temporal_dropout=dropout)
    m.model.temporal_enc.rnn.input_p = 0.0
    with torch.no_grad():
        for name, p in m.model.named_parameters():
            if name.endswith("bias"):
                p.copy_(0.1 * torch.randn(p.shape, generator=_g(3)))
    P = {k: v.detach().clone() for k, v in m.model.state_dict().items() if v.dtype == torch.float32 and "running" not in k}
    return m.to(device), P


def _dict_batch():
    from satflow_amd.models.pl_metnet import SATELLITE_DATA, TOPOGRAPHIC_DATA

    # the reference concatenates on dim 1 and MetNet reads dim 1 as time (pl_metnet.py:100 vs metnet's [B,T,C,H,W]); mirrored literally:
    # dim 1 = 2 "satellite" + 1 repeated "topographic" entries = 3 timesteps, dim 2 = 5 channels
    sat = torch.randn(2, 2, 5, 32, 32, generator=_g(11)).double()  # float64 on purpose: the step casts with .float()
    topo = torch.randn(2, 1, 32, 32, generator=_g(12)).double()
    y = torch.randn(2, 3, 2, 2, 2, generator=_g(13)).double()
    return ({SATELLITE_DATA: sat, TOPOGRAPHIC_DATA: topo}, {SATELLITE_DATA: y})


def _combined(x):
    from satflow_amd.models.pl_metnet import SATELLITE_DATA, TOPOGRAPHIC_DATA

    sat, topo = x[SATELLITE_DATA], x[TOPOGRAPHIC_DATA]
    return torch.cat([sat, topo.unsqueeze(2).expand(-1, -1, sat.shape[2], -1, -1)], 1).float()


def test_training_step_dict_batch(device):
    from satflow_amd.models.pl_metnet import SATELLITE_DATA

    m, P = _lit(device)
    x, y = _dict_batch()
    xin = _combined(x)
    assert xin.shape == (2, 3, 5, 32, 32)
    Pr = {k: v.clone().requires_grad_() for k, v in P.items()}
    ref = M.metnet_forward(xin, Pr, sat_channels=4, input_size=8, forecast_steps=3)
    yt = y[SATELLITE_DATA].float()
    ref_loss = TF.mse_loss(ref, yt)
    ref_loss.backward()
    m.train()
    xd = {k: v.to(device) for k, v in x.items()}
    yd = {k: v.to(device) for k, v in y.items()}
    loss = m.training_step((xd, yd), 0)
    loss.backward()
    assert_close(loss, ref_loss, "train/loss", rtol=1e-5, atol=1e-7)
    assert set(m.logged) == {"train/loss"} | {f"train/frame_{f}_loss" for f in range(3)}
    for f in range(3):  # reference: criterion(y_hat[:, f], y[:, f]) per forecast frame (pl_metnet.py:121-123)
        assert_close(m.logged[f"train/frame_{f}_loss"], TF.mse_loss(ref[:, f], yt[:, f]), f"train/frame_{f}_loss", rtol=1e-5, atol=1e-7)
    for k, p in m.model.named_parameters():
        assert_close(p.grad, Pr[k].grad, f"d{k}", grad=True)


def test_validation_step_eval_mode(device):
    """Lightning runs validation_step under model.eval(): running-statistics BatchNorm, dropouts off; metric names val/*."""
    from satflow_amd.models.pl_metnet import SATELLITE_DATA

    m, P = _lit(device, dropout=0.3)
    sd = m.model.state_dict()
    stats = {}
    for i in ("3", "5", "7"):
        pre = f"image_encoder.module.module.{i}"
        with torch.no_grad():
            sd[f"{pre}.running_mean"].copy_(0.3 * torch.randn(sd[f"{pre}.running_mean"].shape, generator=_g(int(i))))
            sd[f"{pre}.running_var"].copy_(0.5 + torch.rand(sd[f"{pre}.running_var"].shape, generator=_g(10 + int(i))))
        stats[i] = (sd[f"{pre}.running_mean"].cpu().clone(), sd[f"{pre}.running_var"].cpu().clone())
    x, y = _dict_batch()
    with torch.no_grad():
        ref = M.metnet_forward(_combined(x), P, sat_channels=4, input_size=8, forecast_steps=3, bn_stats=stats)
    yt = y[SATELLITE_DATA].float()
    m.eval()
    with torch.no_grad():
        loss = m.validation_step(({k: v.to(device) for k, v in x.items()}, {k: v.to(device) for k, v in y.items()}), 0)
    assert_close(loss, TF.mse_loss(ref, yt), "val/loss", rtol=1e-5, atol=1e-7)
    assert set(m.logged) == {"val/loss"} | {f"val/frame_{f}_loss" for f in range(3)}
    for f in range(3):
        assert_close(m.logged[f"val/frame_{f}_loss"], TF.mse_loss(ref[:, f], yt[:, f]), f"val/frame_{f}_loss", rtol=1e-5, atol=1e-7)
    # running statistics untouched by validation
    for i in ("3", "5", "7"):
        assert torch.equal(m.model.state_dict()[f"image_encoder.module.module.{i}.running_mean"].cpu(), stats[i][0])


def test_tuple_batch_and_convlstm_validation_names(device):
    """Plain (x, y) tensor batches (the path bench.py drives) and EncoderDecoderConvLSTM.validation_step / test_step
    (conv_lstm.py:72-91): val/loss + val/frame_{f}_loss, test_step compares the UNPERMUTED prediction as the reference does."""
    from oracle import convlstm as O
    from satflow_amd.models import EncoderDecoderConvLSTM

    torch.manual_seed(2)
    m = EncoderDecoderConvLSTM(hidden_dim=8, input_channels=4, out_channels=4, forecast_steps=4).to(device)
    x, y = torch.randn(2, 3, 4, 16, 16, generator=_g(1)), torch.rand(2, 4, 4, 16, 16, generator=_g(2))
    params = {k: v.detach().cpu() for k, v in m.model.state_dict().items()}
    ref_loss, ref_frames = O.training_loss(x, y, 4, params)
    with torch.no_grad():
        val = m.validation_step((x.to(device), y.to(device)), 0)
    assert_close(val, ref_loss, "val/loss", rtol=1e-5, atol=1e-7)
    assert set(m.logged) == {"val/loss"} | {f"val/frame_{f}_loss" for f in range(4)}
    assert_close(torch.stack([m.logged[f"val/frame_{f}_loss"] for f in range(4)]), ref_frames, "val frame losses", rtol=1e-5, atol=1e-7)
    with torch.no_grad():  # out_channels == forecast_steps so that the reference's un-permuted comparison is shape-valid
        tst = m.test_step((x.to(device), y.to(device)), 0)
    assert_close(tst, TF.mse_loss(O.convlstm_forward(x, 4, params), y), "test_step loss", rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("B,T,C,sat,raw", [(2, 3, 13, 12, 64), (1, 2, 5, 4, 32), (1, 1, 12, 12, 128), (1, 2, 6, 2, 32)])
def test_preprocess_backward(device, B, T, C, sat, raw):
    """sf_metnet_preprocess_bwd vs autograd through the oracle's preprocessor."""
    from satflow_amd.models.metnet import MetNetPreprocessor

    x = torch.randn(B, T, C, raw, raw, generator=_g(1))
    xr = x.clone().requires_grad_()
    ref = M.preprocess(xr, sat, raw // 4)
    cot = torch.randn(ref.shape, generator=_g(2))
    (ref * cot).sum().backward()
    xd = x.to(device).requires_grad_()
    out = MetNetPreprocessor(sat, raw // 4)(xd)
    (out * cot.to(device)).sum().backward()
    assert_close(out, ref, "preprocess", rtol=1e-6, atol=1e-6)
    assert_close(xd.grad, xr.grad, "d(imgs)", rtol=1e-6, atol=1e-6)


def test_metnet_input_gradient(device):
    """Gradient wrt the input images through the whole network (the reference's autograd produces it; round 1 dropped it)."""
    from parity_util import gpu_pool_routing
    from satflow_amd.models import MetNet

    torch.manual_seed(0)
    net = MetNet(**CFG, temporal_dropout=0.0)
    net.temporal_enc.rnn.input_p = 0.0
    P = {k: v.detach().clone() for k, v in net.state_dict().items() if v.dtype == torch.float32 and "running" not in k}
    net = net.to(device).train()
    x = torch.randn(2, 2, 5, 32, 32, generator=_g(5))
    cot = torch.randn(2, 3, 2, 2, 2, generator=_g(6))
    net.image_encoder.module.capture = {}
    xd = x.to(device).requires_grad_()
    out = net(xd)
    (out * cot.to(device)).sum().backward()
    routing = gpu_pool_routing(net, 2, 2)
    xr = x.clone().requires_grad_()
    ref = M.metnet_forward(xr, P, sat_channels=4, input_size=8, forecast_steps=3, pool_routing=routing)
    (ref * cot).sum().backward()
    assert_close(out, ref, "out")
    assert xd.grad is not None
    assert_close(xd.grad, xr.grad, "d(imgs)", grad=True, force_rel=True)

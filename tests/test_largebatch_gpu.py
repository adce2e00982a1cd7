"""GPU: the batch sizes the scaling runs execute besides the benchmarked B = 8 (tools/run_scale.sh: strong scaling splits a global
batch of 64 over the ranks, so N = 1 runs B = 64 and N = 2 runs B = 32; > 2^31 elements per activation tensor at B = 64).  Size-
independent properties only: in eval mode (running statistics: samples are independent) a large batch must reproduce its chunks; a
training step in the benchmarked mode (bf16a, temporal dropout 0.2) must give a finite loss and finite gradients on every parameter."""
import pytest
import torch

import satflow_amd
from conftest import rel_l2

pytestmark = pytest.mark.gpu


def _model(device, dropout):
    from satflow_amd.models import LitMetNet

    torch.manual_seed(1234)
    return LitMetNet(input_channels=12, sat_channels=12, input_size=64, output_channels=12, hidden_dim=64, forecast_steps=12,
                     temporal_dropout=dropout).to(device)


def _free_gib():
    import gc

    gc.collect()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()   # blocks cached for earlier tests of the same process (the full-size DGMR step) are free memory, not used memory
    free, _ = torch.cuda.mem_get_info()
    return free / 2**30


@pytest.mark.parametrize("B", [16, 64])
def test_metnet_large_batch_eval_reproduces_its_chunks(device, B):
    if B == 64 and _free_gib() < 150:
        pytest.skip("needs about 120 GiB of free HBM")
    satflow_amd.set_compute_dtype("bf16a")
    try:
        m = _model(device, 0.0).eval()
        g = torch.Generator().manual_seed(B)
        x = torch.randn(B, 24, 12, 256, 256, generator=g)
        with torch.no_grad():
            big = m(x.to(device))
            assert big.shape == (B, 12, 12, 16, 16) and torch.isfinite(big).all()
            for lo in (0, B - 8):  # the first and the last chunk (the last one sits behind 2^31 elements in the large tensors)
                part = m(x[lo:lo + 8].to(device))
                assert torch.equal(part, big[lo:lo + 8]), f"chunk {lo}: rel L2 {rel_l2(part, big[lo:lo + 8]):.3e}"
    finally:
        satflow_amd.set_compute_dtype("f32")


@pytest.mark.parametrize("B", [16, 64])
def test_metnet_large_batch_training_step_is_finite(device, B):
    if B == 64 and _free_gib() < 200:
        pytest.skip("needs about 170 GiB of free HBM")
    from satflow_amd.optim import FlatAdam

    satflow_amd.set_compute_dtype("bf16a")
    try:
        m = _model(device, 0.2).train()
        opt = FlatAdam(m.parameters(), lr=1e-3)
        g = torch.Generator().manual_seed(B + 1)
        x = torch.randn(B, 24, 12, 256, 256, generator=g).to(device)
        y = torch.randn(B, 12, 12, 16, 16, generator=g).to(device)
        opt.zero_grad()
        loss = m.training_step((x, y), 0)
        loss.backward()
        assert torch.isfinite(loss) and 0.1 < float(loss) < 10.0, float(loss)
        for k, p in m.named_parameters():
            assert p.grad is not None and torch.isfinite(p.grad).all(), k
            assert float(p.grad.abs().max()) > 0 or k.endswith("bias"), k
        opt.step()
        assert all(torch.isfinite(p).all() for p in m.parameters())
    finally:
        satflow_amd.set_compute_dtype("f32")

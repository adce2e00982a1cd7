import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# Parity gate of BASELINE.json north_star for the fp32 path.
RTOL, ATOL = 1e-4, 1e-5


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a HIP device (MI355X); run with -m gpu on the GPU box")


@pytest.fixture(scope="session")
def device():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda:0")


def rel_l2(actual, expected) -> float:
    import torch

    a = torch.as_tensor(actual).detach().float().cpu()
    e = torch.as_tensor(expected).detach().float().cpu()
    return float((a - e).norm() / (e.norm() + 1e-30))


def assert_close(actual, expected, name, rtol=RTOL, atol=ATOL, grad=False, force_rel=False):
    """allclose with a readable report (also checks relative L2 so tiny tensors cannot hide).

    ``grad=True``: the tensor is a gradient whose entries are sums of 10^2..10^5 O(1) products
    (|ref|max up to ~10^2 with the random cotangents of the goldens).  Both sides carry fp32
    summation noise proportional to that scale, so ``atol`` is applied to the tensor normalised
    by ``max(1, |ref|max)``; ``rtol`` is unchanged.
    """
    import torch

    a = torch.as_tensor(actual).detach().float().cpu()
    e = torch.as_tensor(expected).detach().float().cpu()
    assert a.shape == e.shape, f"{name}: shape {tuple(a.shape)} != {tuple(e.shape)}"
    assert torch.isfinite(a).all(), f"{name}: non-finite values"
    err = (a - e).abs()
    if grad:
        atol = atol * max(1.0, float(e.abs().max()))
    bound = atol + rtol * e.abs()
    bad = err > bound
    rel_l2 = (a - e).norm() / (e.norm() + 1e-30)
    assert not bad.any(), (
        f"{name}: {int(bad.sum())}/{a.numel()} elements outside rtol={rtol} atol={atol}; "
        f"max abs err {err.max():.3e} at |ref| {e.flatten()[err.argmax()].abs():.3e}; rel L2 {rel_l2:.3e}"
    )
    if force_rel:  # tensors whose entries all sit below atol (dx under default init): the elementwise check is vacuous
        assert float(e.norm()) > 0 and rel_l2 <= 10 * rtol, f"{name}: relative L2 error {rel_l2:.3e} (|ref|max {e.abs().max():.3e})"
    if float(e.norm()) > 100 * atol * e.numel() ** 0.5:  # skip for tensors that are numerically zero (e.g. d(bias) in front of BatchNorm)
        assert rel_l2 <= 10 * rtol, f"{name}: relative L2 error {rel_l2:.3e}"

"""GPU, RCCL ("nccl" backend), one rank: the gradient-exchange code that bench.py runs at N > 1 - process-group
initialisation on the HIP device, the bucketed asynchronous all-reduce launched from the autograd hooks, the wait in
`step()` and the fused Adam - executed on a 1-GPU box.  A single rank has nothing to exchange, so FlatAdam's
MIN_EXCHANGE_WORLD is lowered to 1 for this test: SUM over one rank is the identity, and a step with the exchange
(overlapped, and as a single all-reduce) must leave the gradient a step without it leaves.  What this covers
that the gloo world-2 tests cannot: stream ordering between the HIP kernels writing the flat gradient and RCCL's
stream reading it.  Runs in a child process so that the process group does not leak into the other tests."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

CHILD = r"""
import os, sys, torch, torch.distributed as dist
import satflow_amd
from satflow_amd.models import MetNet, EncoderDecoderConvLSTM
from satflow_amd.optim import FlatAdam
from satflow_amd.functional import mse_loss_with_frames

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
satflow_amd.set_compute_dtype("f32")

def run(kind, exchange, overlap):
    FlatAdam.MIN_EXCHANGE_WORLD = 1 if exchange else 2
    torch.manual_seed(0)
    if kind == "metnet":
        m = MetNet(input_channels=5, sat_channels=4, input_size=8, output_channels=2, hidden_dim=16, forecast_steps=3, temporal_dropout=0.0).to(dev)
        x = torch.randn(2, 4, 5, 32, 32, device=dev, generator=torch.Generator(dev).manual_seed(1))
        y = torch.randn(2, 3, 2, 2, 2, device=dev, generator=torch.Generator(dev).manual_seed(2))
    else:
        m = EncoderDecoderConvLSTM(hidden_dim=16, input_channels=4, out_channels=1, forecast_steps=2).to(dev)
        x = torch.randn(2, 3, 4, 32, 32, device=dev, generator=torch.Generator(dev).manual_seed(1))
        y = torch.randn(2, 1, 2, 32, 32, device=dev, generator=torch.Generator(dev).manual_seed(2))
    m.train()  # seeded identically, so the dropout masks agree; the runs differ only by the order of a few float atomics
    opt = FlatAdam(m.parameters(), lr=1e-3, overlap=overlap)
    assert opt.exchange == exchange and opt.overlap == (exchange and overlap)
    launched, g0 = 0, None
    for it in range(3):
        opt.zero_grad()
        out = m(x, 2) if kind == "convlstm" else m(x)
        assert out.shape == y.shape
        loss = mse_loss_with_frames(out.contiguous(), y, frame_dim=2 if kind == "convlstm" else 1)[0]
        loss.backward()
        launched += len(opt._work)
        opt.allreduce_grads()
        if it == 0:
            g0 = opt.flat_g.clone()
        opt.step()
    torch.cuda.synchronize()
    assert torch.isfinite(opt.flat_p).all()
    return g0, launched

for kind in ("convlstm", "metnet"):
    ref, _ = run(kind, False, False)
    one, _ = run(kind, True, False)
    ovl, launched = run(kind, True, True)
    assert launched > 0, "no slice was launched from the autograd hooks"
    # the exchanged gradient of the first step (parameters are not compared: Adam turns the sign of a ~1e-9 gradient,
    # e.g. of a convolution bias in front of BatchNorm, into a +-lr step)
    tol = 1e-5 * float(ref.abs().max()) + 1e-7
    assert float(ref.abs().max()) > 1e-4 and torch.isfinite(ref).all()
    assert torch.allclose(ref, one, rtol=0, atol=tol), kind + ": single all-reduce changed the gradient"
    assert torch.allclose(ref, ovl, rtol=0, atol=tol), kind + ": overlapped all-reduce changed the gradient"
dist.barrier()
dist.destroy_process_group()
print("RCCL_OK")
"""


def test_rccl_single_rank_exchange_matches_no_exchange():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env["PYTHONPATH"] = root + os.pathsep + env.get("PYTHONPATH", "")
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]

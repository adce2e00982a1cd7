"""CPU, world_size 2, gloo: the data-parallel gradient exchange of FlatAdam (one SUM all-reduce of the flat
gradient, mean folded into the update) is equivalent to averaging per-rank gradients of a sharded minibatch."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from satflow_amd.models import EncoderDecoderConvLSTM
    from satflow_amd.optim import FlatAdam

    torch.manual_seed(0)  # identical replicas
    m = EncoderDecoderConvLSTM(hidden_dim=8, input_channels=4, out_channels=1, forecast_steps=2)
    opt = FlatAdam(m.parameters(), lr=1e-3)
    # parameters and gradients are views into the flat buffers
    for p, off in zip(opt.params, opt.offsets):
        assert p.data_ptr() == opt.flat_p.data_ptr() + 4 * off and p.grad.data_ptr() == opt.flat_g.data_ptr() + 4 * off
    assert list(m.state_dict().keys())[0] == "model.encoder_1_convlstm.conv.weight"
    opt.zero_grad()
    g = torch.Generator().manual_seed(100 + rank)  # rank-specific "gradient" (stands in for the shard's backward)
    for p in opt.params:
        p.grad.add_(torch.randn(p.shape, generator=g))
    local = opt.flat_g.clone()
    opt.allreduce_grads()
    gathered = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(gathered, local)
    expect = sum(gathered)
    ok = torch.allclose(opt.flat_g, expect, rtol=0, atol=1e-6) and opt.world == world
    # the optimizer itself has no CPU path
    try:
        opt.step()
        ok = False
    except RuntimeError as e:
        ok = ok and "no CPU optimizer path" in str(e)
    out[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_flat_gradient_allreduce_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        out = mgr.dict()
        procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
        assert dict(out) == {0: True, 1: True}


def _worker_overlap(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from satflow_amd.optim import FlatAdam

    torch.manual_seed(0)  # identical replicas
    net = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Tanh(), torch.nn.Linear(16, 16), torch.nn.Tanh(), torch.nn.Linear(16, 3),
                              torch.nn.Linear(3, 2))
    unused = torch.nn.Parameter(torch.ones(5))  # never receives a gradient: its slice must still be reduced (as zeros)
    opt = FlatAdam(list(net.parameters()) + [unused], lr=1e-3, overlap=True, buckets=3)
    ok = opt.overlap and len(opt._bucket_range) == 3 and opt._bucket_range[0][0] == 0 and opt._bucket_range[-1][1] == opt.numel
    ok = ok and all(a[1] == b[0] for a, b in zip(opt._bucket_range, opt._bucket_range[1:]))
    for step in range(2):  # two steps: the launch bookkeeping must reset
        opt.zero_grad()
        x = torch.randn(4, 6, generator=torch.Generator().manual_seed(10 * step + rank))  # this rank's shard
        net(x).square().sum().backward()  # hooks fire here: slices are reduced while the backward is still running
        launched = sorted(opt._work)
        ok = ok and len(launched) >= 2  # every slice made only of `net` parameters went out during backward
        opt.allreduce_grads()
        opt.allreduce_grads()  # idempotent until the next zero_grad
        # reference: plain autograd on both shards
        ref = torch.nn.Sequential(*[type(m)(*((m.in_features, m.out_features) if isinstance(m, torch.nn.Linear) else ())) for m in net])
        ref.load_state_dict(net.state_dict())
        for r in range(world):
            xr = torch.randn(4, 6, generator=torch.Generator().manual_seed(10 * step + r))
            ref(xr).square().sum().backward()
        for p, q in zip(net.parameters(), ref.parameters()):
            ok = ok and torch.allclose(p.grad, q.grad, rtol=1e-5, atol=1e-6)
        ok = ok and float(unused.grad.abs().max()) == 0.0
    out[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_overlapped_bucketed_allreduce_world2():
    """`FlatAdam(overlap=True)`: gradient slices are all-reduced from autograd hooks while the backward pass runs (the
    reference's Lightning-DDP behaviour, configs/trainer/ddp.yaml:4-5); result == sum of the per-rank gradients."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        out = mgr.dict()
        procs = [ctx.Process(target=_worker_overlap, args=(r, world, port, out)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
        assert dict(out) == {0: True, 1: True}

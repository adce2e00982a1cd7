"""CPU, world_size 2, gloo: the data-parallel gradient exchange of FlatAdam (one SUM all-reduce of the flat
gradient, mean folded into the update) is equivalent to averaging per-rank gradients of a sharded minibatch."""
import os
import socket

import pytest

from conftest import ROOT
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


def _worker_accum(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from satflow_amd.optim import FlatAdam

    torch.manual_seed(rank)  # DIFFERENT replicas: the constructor must broadcast rank 0's parameters and buffers (as DDP does)
    net = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.BatchNorm1d(16), torch.nn.Linear(16, 3))
    with torch.no_grad():
        net[1].running_mean.fill_(float(rank + 1))
    opt = FlatAdam(net.parameters(), lr=1e-2, overlap=True, buckets=2, buffers=list(net.buffers()))
    flat = [opt.flat_p.clone(), net[1].running_mean.clone()]
    for t in flat:
        both = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(both, t)
        ok = torch.equal(both[0], both[1])
        if not ok:
            out[rank] = "broadcast"
            return
    ok = float(net[1].running_mean[0]) == 1.0
    # gradient accumulation: two micro-batches, only the second backward exchanges
    xs = [torch.randn(4, 6, generator=torch.Generator().manual_seed(7 * k + rank)) for k in range(2)]
    opt.zero_grad()
    with opt.no_sync():
        net(xs[0]).square().sum().backward()
    ok = ok and not opt._work  # nothing was launched inside no_sync
    net(xs[1]).square().sum().backward()
    opt.allreduce_grads()
    got = opt.flat_g.clone()
    # reference: sum over ranks of the locally accumulated gradient (computed without hooks)
    opt.disable_overlap()
    opt.zero_grad()
    net[1].running_mean.copy_(flat[1])
    for x in xs:
        net(x).square().sum().backward()
    local = opt.flat_g.clone()
    both = [torch.zeros_like(local) for _ in range(world)]
    dist.all_gather(both, local)
    ok = ok and torch.allclose(got, both[0] + both[1], rtol=1e-5, atol=1e-6)
    out[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_and_gradient_accumulation_world2():
    """FlatAdam broadcasts rank 0's parameters/buffers at construction and supports accumulation via no_sync()."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        out = mgr.dict()
        procs = [ctx.Process(target=_worker_accum, args=(r, world, port, out)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
        assert dict(out) == {0: True, 1: True}


def test_second_backward_without_no_sync_raises():
    """a second backward onto an already exchanged slice must not silently add local gradients to reduced sums."""
    from satflow_amd.optim import FlatAdam

    net = torch.nn.Linear(3, 2)
    opt = FlatAdam(net.parameters(), lr=1e-3)
    # emulate an active 2-rank exchange without a process group: hooks + a fake "work" entry
    opt.exchange = opt.overlap = True
    opt._bucket_of, opt._bucket_range, opt._count = [0, 0], [[0, opt.numel]], [2]
    opt._pending, opt._work = [0], {0: object()}
    hook = opt._make_hook(0)
    with pytest.raises(RuntimeError, match="no_sync"):
        hook(None)
    with opt.no_sync():
        hook(None)  # accumulation passes do not touch the bookkeeping


def test_flat_adam_is_a_torch_optimizer_with_state():
    """lr schedulers drive it (the reference steps warm-up/cosine per step, pl_metnet.py:71-77); moments survive a checkpoint."""
    from satflow_amd.models.pl_metnet import LinearWarmupCosineAnnealingLR
    from satflow_amd.optim import FlatAdam

    net = torch.nn.Linear(3, 2)
    opt = FlatAdam(net.parameters(), lr=1e-3)
    assert isinstance(opt, torch.optim.Optimizer)
    sched = LinearWarmupCosineAnnealingLR(opt, warmup_epochs=10, max_epochs=100)
    assert opt.lr == 0.0  # warm-up starts at 0
    opt.flat_m.fill_(0.5), opt.flat_v.fill_(0.25)
    opt.t = 7
    sd = opt.state_dict()
    opt2 = FlatAdam(torch.nn.Linear(3, 2).parameters(), lr=5.0)
    opt2.load_state_dict(sd)
    assert opt2.t == 7 and opt2.lr == opt.lr and torch.equal(opt2.flat_m, opt.flat_m) and torch.equal(opt2.flat_v, opt.flat_v)
    del sched


def _worker_bench(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    import bench

    res = bench.main(["--gpus", str(world), "--steps", "3", "--warmup", "1", "--workload", "stub", "--scaling", "strong", "--global-batch", "64"])
    out[rank] = None if res is None else {k: res[k] for k in ("n_gpus", "steps", "warmup", "scaling", "unit", "higher_is_better")} | {
        "per_gpu_batch": res["config"]["per_gpu_batch"], "global_batch": res["config"]["global_batch"], "nranks": res["comm"]["nranks"],
        "ok": res["value"] > 0 and abs(res["value"] - 3 * 64 / (res["ms_per_step"] * 3e-3)) < 1e-6 * res["value"]}


def test_bench_main_launch_plumbing_world2():
    """bench.py's own main(): env-driven rank/world, process-group setup, warm-up + timed loop between barriers, MAX over ranks,
    strong-scaling batch split, one JSON record from rank 0 - on gloo with the stub workload, so that the first real N-GPU
    run cannot die on a flag."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        out = mgr.dict()
        procs = [ctx.Process(target=_worker_bench, args=(r, world, port, out)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(180)
            assert p.exitcode == 0
        res = dict(out)
    assert res[1] is None  # only rank 0 reports
    assert res[0] == {"n_gpus": 2, "steps": 3, "warmup": 1, "scaling": "strong", "unit": "samples/s", "higher_is_better": True,
                      "per_gpu_batch": 32, "global_batch": 64, "nranks": 2, "ok": True}


def test_bench_refuses_world_mismatch(monkeypatch):
    import bench

    monkeypatch.setenv("WORLD_SIZE", "4")
    with pytest.raises(SystemExit, match="WORLD_SIZE=4"):
        bench.main(["--gpus", "8", "--workload", "stub"])
    monkeypatch.setenv("WORLD_SIZE", "1")
    with pytest.raises(SystemExit, match="torch.distributed.run"):
        bench.main(["--gpus", "2", "--workload", "stub"])


def test_bench_launches_itself_when_no_launcher_did():
    """`python bench.py --gpus 2` with WORLD_SIZE unset (how the driver may call the scaling legs, VERDICT r5 missing 6): bench.py starts
    torch.distributed.run as a CHILD process before any GPU call, rank 0's JSON line comes back on stdout and the exit code is the launcher's."""
    import json
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "stub", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, env=env, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]   # the JSON line and NOTHING else on stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["comm"]["nranks"] == 2 and out["config"]["parallelism"] == "dp2"
    # a failing rank's exit code comes back through the launcher
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "stub", "--scaling", "strong", "--global-batch", "63"],
                       capture_output=True, text=True, env=env, timeout=300, cwd=ROOT)
    assert r.returncode != 0


def test_bench_stdout_is_the_json_line_only():
    """`python bench.py` owns the process's stdout for the ONE JSON line: whatever else writes to fd 1 - Python prints, C stdio of a library (RCCL's version
    banner, flushed at exit and therefore BEHIND the line) - lands on stderr.  Checked with a child that writes to fd 1 both ways around a stub run."""
    import json
    import subprocess
    import sys

    code = ("import os, sys, runpy; sys.argv = ['bench.py', '--workload', 'stub', '--steps', '2', '--warmup', '1']; "
            "import atexit; atexit.register(lambda: os.write(1, b'banner at exit\\n')); "
            "runpy.run_path(os.path.join(%r, 'bench.py'), run_name='__main__')" % ROOT)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1, r.stdout[-2000:]
    assert json.loads(lines[0])["n_gpus"] == 1
    assert "banner at exit" in r.stderr

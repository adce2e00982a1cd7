#!/usr/bin/env python3
"""Headline benchmark: training-step throughput (samples/s, ms/step) of the SatFlow hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload metnet|convlstm] [--batch B]

One process per GPU (for N > 1 launch with torch.distributed.run; RANK/LOCAL_RANK/WORLD_SIZE come
from the environment, backend "nccl" == RCCL over xGMI).  A step is one full optimisation step on
one synthetic minibatch: forward, MSE loss, backward, gradient all-reduce (N > 1), fused Adam.
Data parallel with a fixed per-GPU batch => "scaling": "weak"; `value` is the whole-job samples/s.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline     -- the dominant kernel, timed live with HIP events on its own stream
  cpu_baseline -- the CPU oracle (oracle/, a port of the reference algorithm pinned against the
                  reference's own outputs) timed on this host's cores on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_F32_TFLOPS = 157.3  # MI355X fp32 matrix/vector peak (MI355X_MICROARCH.md)
PEAK_HBM_GBPS = 8000.0


def event_time(fn, iters: int, warm: int = 3) -> float:
    """Average seconds per call of `fn`, HIP events on the current stream (where the kernels are launched)."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def time_cpu(one, what: str, budget_s: float = 25.0) -> dict:
    """Time `one()` (one CPU sample) on the host cores within a bounded budget (about 10-30 s)."""
    threads = min(os.cpu_count() or 1, 32)  # more threads than this only adds contention at these conv sizes
    torch.set_num_threads(threads)
    t0 = time.perf_counter()
    one()  # warm-up (oneDNN primitive creation)
    warm = time.perf_counter() - t0
    n, dt = 0, warm
    if warm < budget_s / 2:
        t0 = time.perf_counter()
        while n < 3 and (time.perf_counter() - t0) + dt < budget_s:
            one()
            n += 1
            dt = (time.perf_counter() - t0) / n
    return {"value": 1.0 / dt, "unit": "samples/s", "cores": threads, "kind": "port",
            "sample": f"{what}; {n if n else 1} timed step(s){'' if n else ' (the warm-up itself: budget exhausted)'} "
                      f"on {threads} threads of {os.cpu_count()} host cores, torch {torch.__version__} CPU, {dt*1e3:.0f} ms/sample"}


# ----------------------------------------------------------------------------------------------
# workloads
# ----------------------------------------------------------------------------------------------
class ConvLSTMWorkload:
    """BASELINE.json configs[1]: EncoderDecoderConvLSTM 12ch 128x128, T=12->6, hidden 64 (fp32 parity path)."""

    name = "convlstm_cfg2"

    def __init__(self, dev, batch: int, rank: int):
        from satflow_amd.models import EncoderDecoderConvLSTM
        from satflow_amd.optim import FlatAdam

        self.B, self.T, self.C, self.H, self.W, self.hid, self.fs, self.out = batch, 12, 12, 128, 128, 64, 6, 12
        torch.manual_seed(1234)  # same initial weights on every rank
        self.model = EncoderDecoderConvLSTM(hidden_dim=self.hid, input_channels=self.C, out_channels=self.out,
                                            forecast_steps=self.fs).to(dev)
        g = torch.Generator(device="cpu").manual_seed(1234 + rank)  # per-rank data shard
        self.x = torch.rand(self.B, self.T, self.C, self.H, self.W, generator=g).to(dev)
        self.y = torch.rand(self.B, self.fs, self.out, self.H, self.W, generator=g).to(dev)
        self.opt = FlatAdam(self.model.parameters(), lr=self.model.lr)
        self.dev = dev

    def step(self):
        self.opt.zero_grad()
        loss = self.model.training_step((self.x, self.y), 0)
        loss.backward()
        self.opt.step()
        return loss

    def config(self, world):
        return {"workload": "EncoderDecoderConvLSTM 12ch 128x128 T=12->6 hidden=64 out=12 (BASELINE configs[1])",
                "per_gpu_batch": self.B, "global_batch": self.B * world, "parallelism": f"dp{world}",
                "step": "fwd + mse + bwd + allreduce + adam"}

    def roofline(self):
        """Dominant kernel: the fused 128->256 ConvLSTM cell step (3 of the 4 cells, 24 of 36 launches)."""
        from satflow_amd._hip import T
        eng = self.model.model.encoder_2_convlstm.engine
        B, H, W, hid = self.B, self.H, self.W, self.hid
        mk = lambda c: torch.randn(B, H, W, c, device=self.dev)
        x, h, c, ho, co, g = mk(hid), mk(hid), mk(hid), mk(hid), mk(hid), mk(4 * hid)
        t = event_time(lambda: eng.step(T(x), h, c, B, H, W, ho, co, g), iters=20)
        flops = 2 * 9 * (hid + hid) * 4 * hid * H * W * B
        alg_bytes = (hid + 2 * hid + 2 * hid) * H * W * B * 4 + 9 * 2 * hid * 4 * hid * 4
        return {"bound": "mfma", "achieved": flops / t / 1e12, "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
                "frac": flops / t / 1e12 / PEAK_F32_TFLOPS, "traffic": None,
                "kernel": "conv3x3_f32_kernel<4,LSTM> (sf_convlstm_cell_fwd, 128->256 ch, 128x128, B=%d)" % B,
                "launch_us": t * 1e6, "algorithmic_flops": flops, "algorithmic_bytes": alg_bytes,
                "hbm_gbps_algorithmic": alg_bytes / t / 1e9, "hbm_frac_algorithmic": alg_bytes / t / 1e9 / PEAK_HBM_GBPS,
                "note": "fp32 parity mode: exact-f32 MFMA, bound by the 157.3 TF fp32 matrix pipe (intensity 461 F/B >> ridge 20 F/B)"}

    def cpu_baseline(self):
        from oracle import convlstm as O  # checker/baseline only

        params = {k: v.detach().cpu().clone().requires_grad_() for k, v in self.model.model.state_dict().items()}
        x, y = self.x[:1].cpu(), self.y[:1].cpu()

        def one():
            loss, _ = O.training_loss(x, y, self.fs, params)
            loss.backward()

        return time_cpu(one, "oracle fwd+bwd (no optimizer), B=1 of the same workload, fp32")


def build_workload(name: str, dev, batch: int, rank: int):
    if name == "convlstm":
        return ConvLSTMWorkload(dev, batch, rank)
    if name == "metnet":
        from bench_metnet import MetNetWorkload  # noqa: WPS433

        return MetNetWorkload(dev, batch, rank)
    raise SystemExit(f"unknown workload {name}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=os.environ.get("SF_WORKLOAD", "convlstm"))
    ap.add_argument("--batch", type=int, default=8, help="per-GPU batch (weak scaling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    wl = build_workload(args.workload, dev, args.batch, rank)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        wl.step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = wl.step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    final_loss = float(loss.item())

    if rank == 0:
        samples = args.steps * wl.B * world
        out = {
            "metric": "samples/sec + per-step ms, training step (fwd+bwd+optimizer)",
            "value": samples / elapsed, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic (seeded uniform/normal tensors of the BASELINE shape, random-init weights)",
            "config": wl.config(world), "final_loss": final_loss,
        }
        out["roofline"] = wl.roofline()
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = wl.cpu_baseline()
            out["cpu_baseline"]["gpu_speedup"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Flat-buffer training state: parameters, gradients and Adam moments in three contiguous fp32
buffers, so that the optimizer is one launch (``sf_adam_step``) and data-parallel training needs
one RCCL all-reduce over xGMI per step (reference: Lightning DDP, ``configs/trainer/ddp.yaml:4-5``,
bucketed NCCL all-reduce + ``torch.optim.Adam``, ``conv_lstm.py:48-51``).

``state_dict`` compatibility is untouched: each ``nn.Parameter`` keeps its name/shape and is a
view into the flat buffer.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist

from ._hip import bump_generation, check, lib, stream_ptr


def _round_up(n: int, k: int) -> int:
    return (n + k - 1) // k * k


class FlatAdam:
    """Adam over a module's parameters, flattened.  ``step()`` also performs the data-parallel
    gradient mean when a process group is given (SUM all-reduce, 1/world folded into the update)."""

    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 process_group: Optional["dist.ProcessGroup"] = None, distributed: Optional[bool] = None) -> None:
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        assert self.params, "no parameters"
        dev = self.params[0].device
        self.lr, self.betas, self.eps = lr, betas, eps
        self.offsets, total = [], 0
        for p in self.params:
            assert p.dtype == torch.float32 and p.device == dev
            self.offsets.append(total)
            total += _round_up(p.numel(), 4)  # keep every view 16-byte aligned
        self.numel = total
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(total, dtype=torch.float32, device=dev)
        for p, off in zip(self.params, self.offsets):
            n = p.numel()
            self.flat_p[off : off + n].copy_(p.data.reshape(-1))
            p.data = self.flat_p[off : off + n].view(p.shape)
            p.grad = self.flat_g[off : off + n].view(p.shape)
        self.t = 0
        self.group = process_group
        self.distributed = dist.is_available() and dist.is_initialized() if distributed is None else distributed
        self.world = dist.get_world_size(process_group) if self.distributed else 1

    def zero_grad(self) -> None:
        self.flat_g.zero_()
        for p, off in zip(self.params, self.offsets):  # re-attach if autograd replaced .grad
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * off:
                p.grad = self.flat_g[off : off + p.numel()].view(p.shape)

    def allreduce_grads(self) -> None:
        """One SUM all-reduce of the whole gradient (4-9 MB: latency-bound, one bucket)."""
        if self.distributed and self.world > 1:
            dist.all_reduce(self.flat_g, op=dist.ReduceOp.SUM, group=self.group)

    def step(self) -> None:
        if not self.flat_p.is_cuda:
            raise RuntimeError("FlatAdam.step: parameters are not on a HIP device; there is no CPU optimizer path")
        self.allreduce_grads()
        self.t += 1
        check(
            lib().sf_adam_step(self.flat_p.data_ptr(), self.flat_g.data_ptr(), self.flat_m.data_ptr(), self.flat_v.data_ptr(),
                               self.numel, self.lr, self.betas[0], self.betas[1], self.eps, self.t, 1.0 / self.world, stream_ptr()),
            "sf_adam_step",
        )
        # the update went through raw pointers: invalidate the packed-weight caches explicitly
        bump_generation()

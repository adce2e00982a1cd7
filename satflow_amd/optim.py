"""Flat-buffer training state: parameters, gradients and Adam moments in three contiguous fp32
buffers, so that the optimizer is one launch (``sf_adam_step``) and data-parallel training needs
one RCCL all-reduce over xGMI per step (reference: Lightning DDP, ``configs/trainer/ddp.yaml:4-5``,
bucketed NCCL all-reduce + ``torch.optim.Adam``, ``conv_lstm.py:48-51``).

``state_dict`` compatibility is untouched: each ``nn.Parameter`` keeps its name/shape and is a
view into the flat buffer.
"""
from __future__ import annotations

import contextlib
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist

from ._hip import bump_generation, check, check_device_errors, lib, stream_ptr


def _round_up(n: int, k: int) -> int:
    return (n + k - 1) // k * k


class FlatAdam(torch.optim.Optimizer):
    """Adam over a module's parameters, flattened.  ``step()`` also performs the data-parallel
    gradient mean when a process group is given (SUM all-reduce, 1/world folded into the update).

    A ``torch.optim.Optimizer``: ``param_groups[0]["lr"]`` is what ``step()`` uses, so torch lr schedulers (the reference
    steps a warm-up/cosine schedule every step, ``pl_metnet.py:71-77``) drive it; ``state_dict()`` carries the moments.
    Like DDP it broadcasts rank 0's parameters (and the given ``buffers``, e.g. BatchNorm running statistics) at
    construction, and ``no_sync()`` defers the exchange over gradient-accumulation backward passes."""

    # Smallest group size that exchanges gradients.  2 in production (a single rank has nothing to exchange); the
    # single-GPU RCCL test lowers it to 1 so that the hook / async all-reduce / stream-ordering path runs on a 1-GPU box.
    MIN_EXCHANGE_WORLD = 2

    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 process_group: Optional["dist.ProcessGroup"] = None, distributed: Optional[bool] = None,
                 overlap: bool = False, buckets: int = 3, buffers: Optional[Iterable[torch.Tensor]] = None,
                 check_errors_every: int = 100) -> None:
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        assert self.params, "no parameters"
        dev = self.params[0].device
        super().__init__(self.params, dict(lr=lr, betas=betas, eps=eps))
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
        # every `check_errors_every` steps (0 = never) step() reads the kernels' sticky error words (one synchronisation:
        # _hip.check_device_errors) and raises - a failed in-launch hand-off has already turned the loss into NaN by then
        self.check_errors_every = int(check_errors_every)
        self.group = process_group
        self.distributed = dist.is_available() and dist.is_initialized() if distributed is None else distributed
        self.world = dist.get_world_size(process_group) if self.distributed else 1
        # Overlap of the gradient exchange with the backward pass (reference: Lightning DDP's bucketed all-reduce fired from
        # autograd hooks).  The flat gradient is cut at parameter boundaries into `buckets` contiguous slices of about
        # equal size; parameters sit in the buffer in module order, the backward pass finishes them in reverse order, so the
        # LAST slice (head / attention / recurrent cells) completes first and its all-reduce runs on RCCL's stream while
        # the encoder's backward is still computing.  A slice is launched when every parameter in it has received its
        # gradient (post-accumulate hooks); `step()` waits for the launched ones and reduces whatever was not launched.
        self.exchange = self.distributed and self.world >= self.MIN_EXCHANGE_WORLD
        if self.distributed and self.world > 1:
            # replicas start identical (what DDP's constructor does): rank 0's parameters and buffers win
            dist.broadcast(self.flat_p, src=dist.get_global_rank(process_group, 0) if process_group is not None else 0, group=process_group)
            for b in (buffers or []):
                dist.broadcast(b, src=dist.get_global_rank(process_group, 0) if process_group is not None else 0, group=process_group)
        self._sync = True        # False inside no_sync(): backward passes only accumulate
        self._reduced = False    # a syncing backward has already launched / completed this step's exchange
        self.overlap = bool(overlap) and self.exchange
        self._bucket_of: List[int] = []
        self._bucket_range: List[List[int]] = []
        self._pending: List[int] = []
        self._work: dict = {}
        self._done = False
        if self.overlap:
            nb = max(1, min(int(buckets), len(self.params)))
            target, b, lo = total / nb, 0, 0
            for i, (p, off) in enumerate(zip(self.params, self.offsets)):
                end = off + _round_up(p.numel(), 4)
                self._bucket_of.append(b)
                last = i + 1 == len(self.params)
                if last or (end >= (b + 1) * target and b + 1 < nb):
                    self._bucket_range.append([lo, end])
                    lo, b = end, b + 1
            self._count = [self._bucket_of.count(k) for k in range(len(self._bucket_range))]
            self._pending = list(self._count)
            self._hooks = [p.register_post_accumulate_grad_hook(self._make_hook(self._bucket_of[i])) for i, p in enumerate(self.params)]
        elif self.exchange:
            # one all-reduce after the backward pass: a backward that lands AFTER the exchange (explicit allreduce_grads(), then
            # another backward, then step()) would add rank-local gradients to the reduced sum - refuse it like the overlap path
            self._hooks = [p.register_post_accumulate_grad_hook(self._guard_hook) for p in self.params]
        # Gradient sink (functional.GradSink): the weight-gradient kernels write a parameter's gradient straight into its slice of flat_g
        # instead of returning a tensor that autograd adds to it (one small launch per parameter and step).  Not with an exchanging
        # process group: there the post-accumulate hooks are what launches / guards the all-reduce, and they fire from AccumulateGrad.
        self.direct_grads = not self.exchange
        if self.direct_grads:
            from .functional import GRAD_SINK
            for p in self.params:
                GRAD_SINK.register(self, p, p.grad)

    def disable_overlap(self) -> None:
        """Back to one all-reduce after the backward pass (removes the autograd hooks)."""
        for h in getattr(self, "_hooks", []):
            h.remove()
        self._hooks, self._work, self.overlap = [], {}, False

    # hyper-parameters live in param_groups[0] (what torch schedulers and Lightning edit); step() reads them from there
    @property
    def betas(self):
        return self.param_groups[0]["betas"]

    @betas.setter
    def betas(self, value) -> None:
        self.param_groups[0]["betas"] = tuple(value)

    @property
    def eps(self) -> float:
        return self.param_groups[0]["eps"]

    @eps.setter
    def eps(self, value: float) -> None:
        self.param_groups[0]["eps"] = float(value)

    @property
    def lr(self) -> float:
        return self.param_groups[0]["lr"]

    @lr.setter
    def lr(self, value: float) -> None:
        self.param_groups[0]["lr"] = value

    @contextlib.contextmanager
    def no_sync(self):
        """Gradient accumulation (DDP's ``no_sync``): backward passes inside only add into the flat gradient; the next
        backward outside exchanges the accumulated sum."""
        old, self._sync = self._sync, False
        try:
            yield
        finally:
            self._sync = old

    def _guard_hook(self, _param) -> None:
        if self._reduced and self._sync:
            raise RuntimeError("FlatAdam: a backward pass reached the gradient buffer after this step's all-reduce; the local gradients "
                               "would be added to the already reduced sum.  Call step() / zero_grad() first, or accumulate inside "
                               "`optimizer.no_sync()` before the exchanging backward.")

    def _make_hook(self, k: int):
        def hook(_param) -> None:
            if not self._sync:
                return
            if self._pending[k] <= 0 or k in self._work:
                # the slice was already exchanged by an earlier backward of this step: adding local gradients on top of the
                # reduced sum would be silently wrong (and different per rank)
                raise RuntimeError("FlatAdam(overlap=True): a second backward pass reached an already exchanged gradient slice. "
                                   "Wrap all but the last backward of a step in `optimizer.no_sync()`, or call zero_grad().")
            self._pending[k] -= 1
            if self._pending[k] == 0:
                lo, hi = self._bucket_range[k]
                self._work[k] = dist.all_reduce(self.flat_g[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

        return hook

    def zero_grad(self, set_to_none: bool = False) -> None:  # noqa: ARG002 (gradients are views of the flat buffer)
        self.flat_g.zero_()
        self._reduced = False
        if self.direct_grads:
            from .functional import GRAD_SINK
            GRAD_SINK.reopen(self)
        if self.overlap:
            self._pending, self._work, self._done = list(self._count), {}, False
        for p, off in zip(self.params, self.offsets):  # re-attach if autograd replaced .grad
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + 4 * off:
                p.grad = self.flat_g[off : off + p.numel()].view(p.shape)

    def allreduce_grads(self) -> None:
        """SUM all-reduce of the gradient: one call over the whole flat buffer (4-9 MB: latency-bound), or - with
        ``overlap`` - completion of the per-slice reductions the backward pass already launched."""
        if not self.exchange:
            return
        if not self.overlap:
            if self._reduced:  # idempotent within a step (step() calls it again after an explicit call)
                return
            self._reduced = True
            dist.all_reduce(self.flat_g, op=dist.ReduceOp.SUM, group=self.group)
            return
        if self._done:
            return
        self._done = True
        for k, (lo, hi) in enumerate(self._bucket_range):
            if k in self._work:
                self._work[k].wait()
            else:  # some parameter of the slice got no gradient this step (unused): reduce it now
                dist.all_reduce(self.flat_g[lo:hi], op=dist.ReduceOp.SUM, group=self.group)
        self._work = {}

    # ---- checkpoint / resume: Adam moments, step count and lr travel with the optimizer state ----
    def state_dict(self) -> dict:
        # not torch.optim's {"state", "param_groups"} layout (the moments are two flat buffers, not per-parameter entries);
        # "param_groups" carries the hyper-parameters in torch's form for tooling that reads lr / betas from it
        g = self.param_groups[0]
        return {"flat_m": self.flat_m.clone(), "flat_v": self.flat_v.clone(), "t": self.t, "lr": g["lr"], "betas": tuple(g["betas"]),
                "eps": g["eps"], "numel": self.numel,
                "param_groups": [{k: v for k, v in g.items() if k != "params"} | {"params": list(range(len(self.params)))}]}

    def load_state_dict(self, state: dict) -> None:
        if int(state["numel"]) != self.numel:
            raise ValueError(f"FlatAdam.load_state_dict: {state['numel']} elements saved, {self.numel} here")
        self.flat_m.copy_(state["flat_m"])
        self.flat_v.copy_(state["flat_v"])
        self.t, self.lr = int(state["t"]), float(state["lr"])
        self.betas, self.eps = tuple(state["betas"]), float(state["eps"])

    def step(self, closure=None) -> None:
        if closure is not None:
            raise RuntimeError("FlatAdam.step: closures are not supported")
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
        if self.check_errors_every > 0 and self.t % self.check_errors_every == 0 and not torch.cuda.is_current_stream_capturing():
            check_device_errors()

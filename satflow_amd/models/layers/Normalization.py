"""SpectralNorm and ConditionalNorm on the HIP kernels - SURVEY 8f-3.

Mirror of reference ``satflow/models/layers/Normalization.py``: same constructors, same parameter registration (``SpectralNorm``
moves the wrapped module's ``weight`` to ``weight_bar`` and adds the non-trainable ``weight_u`` / ``weight_v`` PARAMETERS, ``:44-62``,
so a reference ``state_dict`` loads with ``strict=True``), same statefulness (every forward call advances ``u`` and ``v``).

The power iteration, ``sigma`` and ``W / sigma`` are one entry point (``sf_spectral_norm_fwd``) with a matching backward; the
conditional BatchNorm takes its statistics from ``sf_batchnorm_train_fwd`` and applies scale / shift per image - fused with the ReLU
and the nearest up-sampling that follow it in ``GResBlock`` - in ``sf_film_act_*``.
"""
from __future__ import annotations

import torch
from torch import Tensor, nn
from torch.nn import Parameter

from ... import functional as F
from ... import functional_gan as FG
from ..._hip import cpad, require_device


def l2normalize(v, eps=1e-12):
    return v / (v.norm() + eps)


class SpectralNorm(nn.Module):
    def __init__(self, module, name="weight", power_iterations=1):
        super().__init__()
        self.module = module
        self.name = name
        self.power_iterations = power_iterations
        if not self._made_params():
            self._make_params()
        self._eng = None

    # ---- parameter bookkeeping exactly as the reference (:33-62) ----
    def _made_params(self):
        return all(hasattr(self.module, self.name + s) for s in ("_u", "_v", "_bar"))

    def _make_params(self):
        w = getattr(self.module, self.name)
        height = w.data.shape[0]
        width = w.view(height, -1).data.shape[1]
        u = Parameter(w.data.new(height).normal_(0, 1), requires_grad=False)
        v = Parameter(w.data.new(width).normal_(0, 1), requires_grad=False)
        u.data = l2normalize(u.data)
        v.data = l2normalize(v.data)
        w_bar = Parameter(w.data)
        del self.module._parameters[self.name]
        self.module.register_parameter(self.name + "_u", u)
        self.module.register_parameter(self.name + "_v", v)
        self.module.register_parameter(self.name + "_bar", w_bar)

    # ---- the normalised weight (advances u / v; differentiable wrt weight_bar) ----
    def compute_weight(self) -> Tensor:
        u = getattr(self.module, self.name + "_u")
        v = getattr(self.module, self.name + "_v")
        w_bar = getattr(self.module, self.name + "_bar")
        require_device(w_bar, self.name + "_bar")
        w = FG.spectral_norm_weight(w_bar, u.data, v.data, self.power_iterations)
        object.__setattr__(self.module, self.name, w)  # what `setattr(self.module, self.name, ...)` leaves behind (:31)
        return w

    def _update_u_v(self):
        self.compute_weight()

    def conv_engine(self, cins, cout):
        """3x3 index maps of the wrapped convolution (packed weights are never cached: the weight changes with every call)."""
        key = (tuple(cins), cout)
        if self._eng is None or self._eng[0] != key:
            self._eng = (key, FG.FreshConvEngine(list(cins), cout))
        return self._eng[1]

    def run(self, x: Tensor) -> Tensor:
        """The wrapped ``nn.Conv2d(k, padding=k//2)`` on NHWC ``x``."""
        m = self.module
        w = self.compute_weight()
        eng = self.conv_engine([m.in_channels], m.out_channels) if w.shape[-1] == 3 else None
        return FG.conv_nhwc(x, w, m.bias, eng)

    def forward(self, *args):
        """Reference surface: the wrapped module applied with the normalised weight (NCHW in / out for convolutions)."""
        m = self.module
        if isinstance(m, nn.Conv2d):
            if m.stride != (1, 1) or m.padding != (m.kernel_size[0] // 2, m.kernel_size[1] // 2) or m.kernel_size[0] != m.kernel_size[1]:
                raise NotImplementedError("SpectralNorm(nn.Conv2d): the HIP path implements square 'same' convolutions with stride 1")
            (x,) = args
            require_device(x, "input")
            return F.nhwc_to_nchw(self.run(F.nchw_to_nhwc(x.float())), m.out_channels)
        if isinstance(m, nn.Linear):
            (x,) = args
            require_device(x, "input")
            w = self.compute_weight()
            lead = x.shape[:-1]
            xp = torch.nn.functional.pad(x.float().reshape(-1, x.shape[-1]), (0, cpad(x.shape[-1]) - x.shape[-1]))
            return F.linear(xp, w, m.bias)[:, : m.out_features].reshape(*lead, m.out_features)
        if isinstance(m, nn.Embedding):
            (idx,) = args
            return self.compute_weight()[idx]  # a row gather of the normalised table
        raise NotImplementedError(f"SpectralNorm({type(m).__name__}): Conv2d / Conv3d (inside the discriminators), Linear and Embedding are implemented")


class ConditionalNorm(nn.Module):
    def __init__(self, in_channel, n_condition=96):
        super().__init__()
        self.in_channel = in_channel
        self.bn = nn.BatchNorm2d(self.in_channel, affine=False)
        self.embed = nn.Linear(n_condition, self.in_channel * 2)
        self.embed.weight.data[:, : self.in_channel].normal_(1, 0.02)
        self.embed.weight.data[:, self.in_channel:].zero_()

    def embedding(self, class_id: Tensor) -> Tensor:
        """``embed(class_id)`` -> ``[N, 2*C]`` = gamma | beta (``sf_linear_fwd`` on the zero-padded condition)."""
        k = class_id.shape[-1]
        cp = torch.nn.functional.pad(class_id.float(), (0, cpad(k) - k))
        return F.linear(cp, self.embed.weight, self.embed.bias)[:, : 2 * self.in_channel]

    def run(self, x: Tensor, class_id: Tensor, relu: bool = False, up: bool = False, embed_rows=None) -> Tensor:
        """NHWC ``x``; ``relu`` / ``up``: the ReLU and nearest 2x up-sampling GResBlock applies next, in the same pass.
        ``embed_rows`` (0/1 matrix ``[N, conditions]``): image i uses the condition row its row selects (the generator repeats its
        conditions; a product instead of an index gather: the gather's backward is a scatter-add whose atomics' order varies from run to run)."""
        e = self.embedding(class_id)
        if embed_rows is not None:  # a 0/1 selection matrix [N, conditions]: rows of e gathered by a product - deterministic in both directions
            e = FG.bmm(embed_rows.unsqueeze(0), e.unsqueeze(0))[0]
        return FG.conditional_norm(x, e, self.in_channel, self.bn, self.training, relu, up)

    def forward(self, x, class_id):
        require_device(x, "x")
        return F.nhwc_to_nchw(self.run(F.nchw_to_nhwc(x.float()), class_id), self.in_channel)

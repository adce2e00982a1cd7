"""Normalisation-layer factory and weight initialisation of the GAN networks (surface of reference ``satflow/models/gan/common.py``).

Semantics restated, not transcribed (reference ``:7-31`` norm factory, ``:34-72`` initialisation, ``:75-88`` ``init_net``):
* a module whose class name contains ``Conv`` or ``Linear`` and that owns a ``weight`` gets it re-drawn by the chosen scheme and its
  bias zeroed; a ``BatchNorm2d`` gets ``weight ~ N(1, gain)``, ``bias = 0``; everything else is left alone;
* the norm factory returns a constructor taking the channel count.
"""
from __future__ import annotations

import functools
from typing import Callable, Dict

import torch
from torch import nn
from torch.nn import init

# scheme name -> in-place initialiser of a weight tensor given the gain (the four the reference accepts)
_WEIGHT_SCHEMES: Dict[str, Callable[[torch.Tensor, float], None]] = {
    "normal": lambda w, gain: init.normal_(w, 0.0, gain),
    "xavier": lambda w, gain: init.xavier_normal_(w, gain=gain),
    "kaiming": lambda w, gain: init.kaiming_normal_(w, a=0, mode="fan_in"),
    "orthogonal": lambda w, gain: init.orthogonal_(w, gain=gain),
}

# norm name -> constructor(channels); "instance" / "none" as the reference builds them (InstanceNorm2d without affine parameters
# or running statistics; an identity).  The HIP discriminators run BatchNorm2d and identity; InstanceNorm2d has no kernel yet.
_NORM_FACTORIES: Dict[str, Callable[[int], nn.Module]] = {
    "batch": functools.partial(nn.BatchNorm2d, affine=True, track_running_stats=True),
    "instance": functools.partial(nn.InstanceNorm2d, affine=False, track_running_stats=False),
    "none": lambda _channels: nn.Identity(),
}


def get_norm_layer(norm_type: str = "instance"):
    try:
        return _NORM_FACTORIES[norm_type]
    except KeyError:
        raise NotImplementedError("normalization layer [%s] is not found" % norm_type) from None


def _is_affine_map(m: nn.Module) -> bool:
    name = type(m).__name__
    w = getattr(m, "weight", None)
    if isinstance(w, nn.parameter.UninitializedParameter):  # a LazyLinear before its first batch: it initialises itself when it materialises
        return False
    return w is not None and ("Conv" in name or "Linear" in name)


def init_weights(net: nn.Module, init_type: str = "normal", init_gain: float = 0.02) -> None:
    scheme = _WEIGHT_SCHEMES.get(init_type)
    for m in net.modules():  # what ``net.apply`` visits, in the same (post-order-independent) set
        if _is_affine_map(m):
            if scheme is None:
                raise NotImplementedError("initialization method [%s] is not implemented" % init_type)
            with torch.no_grad():
                scheme(m.weight, init_gain)
                if getattr(m, "bias", None) is not None:
                    m.bias.zero_()
        elif "BatchNorm2d" in type(m).__name__ and m.weight is not None:
            with torch.no_grad():
                m.weight.normal_(1.0, init_gain)
                m.bias.zero_()


def init_net(net: nn.Module, init_type: str = "normal", init_gain: float = 0.02) -> nn.Module:
    init_weights(net, init_type, init_gain=init_gain)
    return net


def cal_gradient_penalty(netD, real_data, fake_data, device, type="mixed", constant=1.0, lambda_gp=10.0):
    """Reference ``gan/common.py:87-133`` (WGAN-GP penalty; defined there but called by no model).  ``lambda_gp <= 0`` returns ``(0.0, None)``
    as the reference does.  The penalty itself differentiates THROUGH a gradient (``create_graph=True``): the HIP discriminators'
    autograd nodes implement first derivatives only, so it raises instead of silently computing something else."""
    if lambda_gp > 0.0:
        raise NotImplementedError("cal_gradient_penalty: second-order gradients through the HIP discriminator kernels are not implemented "
                                  "(no reference model calls it; GANLoss('wgangp') itself is available)")
    return 0.0, None

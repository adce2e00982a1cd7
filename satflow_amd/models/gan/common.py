"""Normalisation-layer factory and weight initialisation of the GAN networks (surface of reference ``satflow/models/gan/common.py``)."""
from __future__ import annotations

import functools

import torch
from torch.nn import init


def get_norm_layer(norm_type: str = "instance"):
    """``batch`` -> affine BatchNorm2d with running statistics (reference ``:17-18``); the HIP path implements that one."""
    if norm_type == "batch":
        return functools.partial(torch.nn.BatchNorm2d, affine=True, track_running_stats=True)
    if norm_type in ("instance", "none"):
        raise NotImplementedError(f"norm={norm_type!r}: the HIP discriminator implements the shipped configuration (norm: 'batch', "
                                  "configs/model/cloudgan_convlstm.yaml:11)")
    raise NotImplementedError("normalization layer [%s] is not found" % norm_type)


def init_weights(net, init_type: str = "normal", init_gain: float = 0.02) -> None:
    """Reference ``:34-72``: conv / linear weights ~ N(0, gain) (or xavier / kaiming / orthogonal), biases 0; BatchNorm2d weight ~ N(1, gain)."""

    def init_func(m):
        classname = m.__class__.__name__
        if hasattr(m, "weight") and (classname.find("Conv") != -1 or classname.find("Linear") != -1):
            if init_type == "normal":
                init.normal_(m.weight.data, 0.0, init_gain)
            elif init_type == "xavier":
                init.xavier_normal_(m.weight.data, gain=init_gain)
            elif init_type == "kaiming":
                init.kaiming_normal_(m.weight.data, a=0, mode="fan_in")
            elif init_type == "orthogonal":
                init.orthogonal_(m.weight.data, gain=init_gain)
            else:
                raise NotImplementedError("initialization method [%s] is not implemented" % init_type)
            if hasattr(m, "bias") and m.bias is not None:
                init.constant_(m.bias.data, 0.0)
        elif classname.find("BatchNorm2d") != -1:
            init.normal_(m.weight.data, 1.0, init_gain)
            init.constant_(m.bias.data, 0.0)

    net.apply(init_func)


def init_net(net, init_type: str = "normal", init_gain: float = 0.02):
    init_weights(net, init_type, init_gain=init_gain)
    return net

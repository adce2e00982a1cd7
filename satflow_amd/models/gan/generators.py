"""Generator factory (surface of reference ``satflow/models/gan/generators.py:12-69``): on the hot path the generator is a module
handed in by the caller (CloudGAN passes its ConvLSTM, ``cloudgan.py:88-99``); the named ResNet / U-Net generators are other
model families (out of scope)."""
from __future__ import annotations

from typing import Union

import torch

from .common import get_norm_layer, init_net


def define_generator(input_nc, output_nc, ngf, netG: Union[str, torch.nn.Module], norm="batch", use_dropout=False, init_type="normal",
                     init_gain=0.02):
    get_norm_layer(norm_type=norm)
    if isinstance(netG, torch.nn.Module):
        net = netG
    elif netG in ("resnet_9blocks", "resnet_6blocks", "unet_128", "unet_256"):
        raise NotImplementedError(f"generator {netG!r} is another model family (not on the hot path); pass a module, e.g. the ConvLSTM")
    else:
        raise NotImplementedError("Generator model name [%s] is not recognized" % netG)
    return init_net(net, init_type, init_gain)

"""PatchGAN discriminator and GAN objective on the HIP kernels (surface of reference ``satflow/models/gan/discriminators.py``).

``NLayerDiscriminator`` keeps the reference's ``nn.Sequential`` (``model.0.weight`` ... ``model.11.bias``: the ``state_dict`` keys of
reference ``:139-223``) as its parameter container; the arithmetic - 4x4 convolutions with stride 2 / 1, training-mode BatchNorm2d,
LeakyReLU(0.2) - runs in ``sf_conv2d_*``, ``sf_batchnorm_*`` and ``sf_leaky_relu``.  ``run`` takes a batch that is the
concatenation of several reference calls (one per forecast timestep): BatchNorm statistics and running-statistics updates are
per call (``groups``), in call order.
"""
from __future__ import annotations

import functools

import torch
from torch import nn

from ... import functional as F
from ..._hip import require_device
from ..utils import get_conv_layer
from .common import get_norm_layer, init_net


def define_discriminator(input_nc, ndf, netD, n_layers_D=3, norm="batch", init_type="normal", init_gain=0.02, conv_type: str = "standard"):
    """Reference ``:11-67``: ``basic`` = 3-layer PatchGAN, ``n_layers`` = PatchGAN with ``n_layers_D`` layers."""
    norm_layer = get_norm_layer(norm_type=norm)
    if netD == "basic":
        net = NLayerDiscriminator(input_nc, ndf, n_layers=3, norm_layer=norm_layer, conv_type=conv_type)
    elif netD == "n_layers":
        net = NLayerDiscriminator(input_nc, ndf, n_layers_D, norm_layer=norm_layer, conv_type=conv_type)
    elif netD in ("pixel", "enhanced"):
        raise NotImplementedError(f"discriminator {netD!r}: the HIP path implements the PatchGAN the shipped CloudGAN-ConvLSTM config uses "
                                  "(discriminator_model: 'basic', configs/model/cloudgan_convlstm.yaml:13)")
    else:
        raise NotImplementedError("Discriminator model name [%s] is not recognized" % netD)
    return init_net(net, init_type, init_gain)


class GANLoss(nn.Module):
    """Reference ``:70-136``.  ``vanilla`` = BCE-with-logits against the real / fake label, fused with its gradient (``sf_bce_logits_loss``)."""

    def __init__(self, gan_mode, target_real_label=1.0, target_fake_label=0.0):
        super().__init__()
        self.register_buffer("real_label", torch.tensor(target_real_label))
        self.register_buffer("fake_label", torch.tensor(target_fake_label))
        self.gan_mode = gan_mode
        if gan_mode != "vanilla":
            if gan_mode in ("lsgan", "wgangp"):
                raise NotImplementedError(f"gan mode {gan_mode}: the HIP path implements 'vanilla' (loss: 'vanilla', cloudgan_convlstm.yaml:15)")
            raise NotImplementedError("gan mode %s not implemented" % gan_mode)

    def labels(self, target_is_real: bool) -> float:
        return float(self.real_label if target_is_real else self.fake_label)

    def __call__(self, prediction, target_is_real):
        """``prediction [N,1,h,w]`` (NCHW, as the reference's discriminator returns it) -> scalar loss."""
        logits = F.nchw_to_nhwc(prediction)
        loss, _ = F.bce_logits_groups(logits, self.labels(target_is_real), self.labels(target_is_real), 1, prediction.shape[1])
        return loss

    def grouped(self, logits_nhwc, real_even: bool, real_odd: bool, groups: int):
        """Loss over ``groups`` concatenated discriminator calls on NHWC logits: (mean over all calls, per-call means)."""
        return F.bce_logits_groups(logits_nhwc, self.labels(real_even), self.labels(real_odd), groups, 1)


class NLayerDiscriminator(nn.Module):
    """Defines a PatchGAN discriminator (reference ``:139-223``)."""

    def __init__(self, input_nc, ndf=64, n_layers=3, norm_layer=nn.BatchNorm2d, conv_type: str = "standard"):
        super().__init__()
        if type(norm_layer) == functools.partial:
            use_bias = norm_layer.func == nn.InstanceNorm2d
        else:
            use_bias = norm_layer == nn.InstanceNorm2d
        if conv_type != "standard":
            raise NotImplementedError("the HIP discriminator implements conv_type='standard' (the antialiased variant needs antialiased_cnns.BlurPool)")
        conv2d = get_conv_layer(conv_type)
        kw, padw = 4, 1
        sequence = [conv2d(input_nc, ndf, kernel_size=kw, stride=2, padding=padw), nn.LeakyReLU(0.2, True)]
        nf_mult = 1
        for n in range(1, n_layers):
            nf_mult_prev, nf_mult = nf_mult, min(2**n, 8)
            sequence += [conv2d(ndf * nf_mult_prev, ndf * nf_mult, kernel_size=kw, stride=2, padding=padw, bias=use_bias),
                         norm_layer(ndf * nf_mult), nn.LeakyReLU(0.2, True)]
        nf_mult_prev, nf_mult = nf_mult, min(2**n_layers, 8)
        sequence += [conv2d(ndf * nf_mult_prev, ndf * nf_mult, kernel_size=kw, stride=1, padding=padw, bias=use_bias),
                     norm_layer(ndf * nf_mult), nn.LeakyReLU(0.2, True)]
        sequence += [conv2d(ndf * nf_mult, 1, kernel_size=kw, stride=1, padding=padw)]
        self.model = nn.Sequential(*sequence)

    def run(self, x_nhwc: torch.Tensor, groups: int = 1) -> torch.Tensor:
        """NHWC ``[N,H,W,Cp] -> [N,h,w,16]`` patch logits in lane 0; N = ``groups`` reference calls concatenated."""
        mods = list(self.model)
        y, i = x_nhwc, 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, nn.Conv2d):
                fuse = i + 1 < len(mods) and isinstance(mods[i + 1], nn.LeakyReLU)  # conv -> LeakyReLU without a norm in between
                y = F.conv2d(y, m.weight, m.bias, m.stride[0], m.padding[0], mods[i + 1].negative_slope if fuse else 1.0)
                i += 2 if fuse else 1
            elif isinstance(m, nn.BatchNorm2d):
                y = F.batchnorm(y, m, groups if self.training else 1, self.training)
                i += 1
            elif isinstance(m, nn.LeakyReLU):
                y = F.leaky_relu(y, m.negative_slope)
                i += 1
            else:
                raise NotImplementedError(type(m).__name__)
        return y

    def forward(self, input):
        """Standard forward on NCHW input (one reference call)."""
        require_device(input, "input")
        return F.nhwc_to_nchw(self.run(F.nchw_to_nhwc(input.float()), 1), 1)

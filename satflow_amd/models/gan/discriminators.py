"""PatchGAN discriminator and GAN objective on the HIP kernels (surface of reference ``satflow/models/gan/discriminators.py``).

``NLayerDiscriminator`` keeps the reference's ``nn.Sequential`` (``model.0.weight`` ... ``model.11.bias``: the ``state_dict`` keys of
reference ``:139-223``) as its parameter container; the arithmetic - 4x4 convolutions with stride 2 / 1, training-mode BatchNorm2d,
LeakyReLU(0.2) - runs in ``sf_conv2d_*``, ``sf_batchnorm_*`` and ``sf_leaky_relu``.  ``run`` takes a batch that is the
concatenation of several reference calls (one per forecast timestep): BatchNorm statistics and running-statistics updates are
per call (``groups``), in call order.
"""
from __future__ import annotations

import functools
import os

import torch
from torch import nn

from ... import functional as F
from ... import functional_gan as FG
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
    elif netD == "pixel":
        net = PixelDiscriminator(input_nc, ndf, norm_layer=norm_layer, conv_type=conv_type)
    elif netD == "enhanced":
        net = CloudGANDiscriminator(input_channels=input_nc, num_filters=ndf, num_stages=3, conv_type=conv_type)
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
        if gan_mode not in F.GAN_MODES:
            raise NotImplementedError("gan mode %s not implemented" % gan_mode)

    def labels(self, target_is_real: bool) -> float:
        """The label as a host number (a kernel argument).  The buffers are device scalars as in the reference; reading one is a device-to-host
        copy - a synchronisation on every loss call and not capturable into a hipGraph - so the value is cached until the buffer is written
        (``load_state_dict`` / ``.to()`` / in-place edits change its version or identity)."""
        buf = self.real_label if target_is_real else self.fake_label
        key = (id(buf), buf._version, buf.data_ptr())
        cache = self.__dict__.setdefault("_label_cache", {})
        hit = cache.get(target_is_real)
        if hit is None or hit[0] != key:
            hit = cache[target_is_real] = (key, float(buf))
        return hit[1]

    def __call__(self, prediction, target_is_real):
        """``prediction [N,1,h,w]`` (NCHW, as the reference's discriminator returns it) -> scalar loss."""
        if prediction.dim() == 2:  # [N, 1] scores (CloudGANDiscriminator): one pixel per sample
            prediction = prediction.reshape(prediction.shape[0], prediction.shape[1], 1, 1)
        logits = F.nchw_to_nhwc(prediction)
        loss, _ = F.bce_logits_groups(logits, self.labels(target_is_real), self.labels(target_is_real), 1, prediction.shape[1], self.gan_mode)
        return loss

    def grouped(self, logits_nhwc, real_even: bool, real_odd: bool, groups: int):
        """Loss over ``groups`` concatenated discriminator calls on NHWC logits: (mean over all calls, per-call means)."""
        return F.bce_logits_groups(logits_nhwc, self.labels(real_even), self.labels(real_odd), groups, 1, self.gan_mode)


def instance_norm(x: torch.Tensor, m: nn.InstanceNorm2d) -> torch.Tensor:
    """``nn.InstanceNorm2d`` without affine parameters or running statistics (what ``get_norm_layer("instance")`` builds, reference
    ``gan/common.py:19-22``) on NHWC ``x``: per (image, channel) statistics = the training-mode BatchNorm kernels with one group per
    image and a unit affine map."""
    if m.affine or m.track_running_stats:
        raise NotImplementedError("InstanceNorm2d with affine parameters / running statistics (the reference factory builds neither)")
    n, c = x.shape[0], m.num_features
    key = (c, str(x.device))
    if key not in _UNIT_AFFINE:
        _UNIT_AFFINE[key] = (torch.ones(c, device=x.device), torch.zeros(c, device=x.device))
    ones, zeros = _UNIT_AFFINE[key]
    return F._BatchNormTrainFn.apply(x, ones, zeros, None, None, n, m.eps, 0.0, None)


_UNIT_AFFINE = {}


def _on_mfma(m: nn.Conv2d, x: torch.Tensor) -> bool:
    """4x4 kernels with padding 1 and stride 1 or 2 (even maps) run as 3x3 convolutions; SF_CONV4_DIRECT=1: the direct fp32 kernel (A/B switch)."""
    if os.environ.get("SF_CONV4_DIRECT") or m.kernel_size != (4, 4) or m.padding != (1, 1) or m.dilation != (1, 1) or m.groups != 1:
        return False
    if m.stride == (2, 2):
        return x.shape[1] % 2 == 0 and x.shape[2] % 2 == 0 and x.shape[1] >= 2 and x.shape[2] >= 2
    return m.stride == (1, 1) and x.shape[1] >= 2 and x.shape[2] >= 2


def _run_sequence(mods, x_nhwc: torch.Tensor, groups: int, training: bool) -> torch.Tensor:
    """Shared executor of the discriminators' ``nn.Sequential``s: Conv2d (+ fused LeakyReLU), BatchNorm2d / InstanceNorm2d / Identity."""
    y, i = x_nhwc, 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, nn.Conv2d):
            fuse = i + 1 < len(mods) and isinstance(mods[i + 1], nn.LeakyReLU)  # conv -> LeakyReLU without a norm in between
            slope = mods[i + 1].negative_slope if fuse else 1.0
            if m.kernel_size == (1, 1) and m.stride == (1, 1) and m.padding == (0, 0):
                y = F.linear(y, m.weight.view(m.out_channels, m.in_channels), m.bias)
                if fuse:
                    y = F.leaky_relu(y, slope)
            elif _on_mfma(m, y):
                # 4x4 kernels (PatchGAN) as 3x3 convolutions on the matrix cores (functional_gan.conv4x4_as_3x3); the packed weights are cached
                # on the layer's own parameters
                cin = 4 * y.shape[-1]
                eng = m.__dict__.setdefault("_sf_eng", {})
                if cin not in eng:
                    eng[cin] = F.ConvEngine([cin], m.out_channels)
                eng[cin].key_tensors = (m.weight,) if m.bias is None else (m.weight, m.bias)
                y = FG.conv4x4_as_3x3(y, m.weight, m.bias, m.stride[0], eng[cin])
                if fuse:
                    y = F.leaky_relu(y, slope)
            else:
                y = F.conv2d(y, m.weight, m.bias, m.stride[0], m.padding[0], slope)
            i += 2 if fuse else 1
        elif isinstance(m, nn.BatchNorm2d):
            y = F.batchnorm(y, m, groups if training else 1, training)
            i += 1
        elif isinstance(m, nn.InstanceNorm2d):
            y = instance_norm(y, m)
            i += 1
        elif isinstance(m, nn.LeakyReLU):
            y = F.leaky_relu(y, m.negative_slope)
            i += 1
        elif isinstance(m, nn.Identity):
            i += 1
        else:
            raise NotImplementedError(type(m).__name__)
    return y


class PixelDiscriminator(nn.Module):
    """Defines a 1x1 PatchGAN discriminator (pixelGAN), reference ``gan/discriminators.py:228-262``: three 1x1 convolutions
    (``sf_linear_*``) with a norm layer and LeakyReLU(0.2) in between."""

    def __init__(self, input_nc, ndf=64, norm_layer=nn.BatchNorm2d, conv_type: str = "standard"):
        super().__init__()
        if type(norm_layer) == functools.partial:
            use_bias = norm_layer.func == nn.InstanceNorm2d
        else:
            use_bias = norm_layer == nn.InstanceNorm2d
        if conv_type != "standard":
            raise NotImplementedError("the HIP discriminators implement conv_type='standard'")
        conv2d = get_conv_layer(conv_type)
        self.net = nn.Sequential(
            conv2d(input_nc, ndf, kernel_size=1, stride=1, padding=0), nn.LeakyReLU(0.2, True),
            conv2d(ndf, ndf * 2, kernel_size=1, stride=1, padding=0, bias=use_bias), norm_layer(ndf * 2), nn.LeakyReLU(0.2, True),
            conv2d(ndf * 2, 1, kernel_size=1, stride=1, padding=0, bias=use_bias),
        )

    def run(self, x_nhwc: torch.Tensor, groups: int = 1) -> torch.Tensor:
        return _run_sequence(list(self.net), x_nhwc, groups, self.training)

    def forward(self, input):
        require_device(input, "input")
        return F.nhwc_to_nchw(self.run(F.nchw_to_nhwc(input.float()), 1), 1)


class CloudGANBlock(nn.Module):
    """Reference ``gan/discriminators.py:265-283``: 3x3 convolution WITHOUT padding, ReLU, 2x2 max-pooling (floor)."""

    def __init__(self, input_channels, conv_type: str = "standard"):
        super().__init__()
        if conv_type != "standard":
            raise NotImplementedError("the HIP discriminators implement conv_type='standard' (the antialiased variant needs antialiased_cnns.BlurPool)")
        conv2d = get_conv_layer(conv_type)
        self.conv = conv2d(input_channels, input_channels * 2, kernel_size=(3, 3))
        self.relu = torch.nn.ReLU()
        self.pool = torch.nn.MaxPool2d(kernel_size=(2, 2), stride=2)
        self.blurpool = torch.nn.Identity()
        self._eng = F.ConvEngine([input_channels], input_channels * 2)

    def run(self, x: torch.Tensor) -> torch.Tensor:
        from ... import functional_gan as FG

        # the unpadded convolution = the interior of the 'same' convolution (MFMA kernel) - one border pixel cut off
        y = FG._CropFn.apply(F.conv3x3(self._eng, x, self.conv.weight, self.conv.bias), 1)
        y = F.leaky_relu(y, 0.0)
        n, h, w, c = y.shape
        return FG.max_pool3(y.view(n, 1, h, w, c), (1, 2, 2), (1, 2, 2)).view(n, h // 2, w // 2, c)

    def forward(self, x):
        require_device(x, "x")
        return F.nhwc_to_nchw(self.run(F.nchw_to_nhwc(x.float())), self.conv.out_channels)


class CloudGANDiscriminator(nn.Module):
    """Reference ``gan/discriminators.py:286-312`` (``discriminator_model="enhanced"``, CloudGAN's constructor default): 1x1 convolution,
    ``num_stages`` CloudGANBlocks, flatten, one linear score per sample.  ``fc`` is the reference's ``LazyLinear``: it takes its input
    width from the first batch; its weight keeps the reference's (C, H, W) flattening order in the ``state_dict`` and is re-ordered to
    the kernels' (H, W, C) on the fly."""

    def __init__(self, input_channels: int = 12, num_filters: int = 64, num_stages: int = 3, conv_type: str = "standard"):
        super().__init__()
        conv2d = get_conv_layer(conv_type)
        self.conv_1 = conv2d(input_channels, num_filters, kernel_size=1, stride=1, padding=0)
        stages = []
        for _ in range(num_stages):
            stages.append(CloudGANBlock(num_filters, conv_type))
            num_filters = num_filters * 2
        self.stages = torch.nn.Sequential(*stages)
        self.flatten = torch.nn.Flatten()
        self.fc = torch.nn.LazyLinear(1)  # Real/Fake
        self._out_channels = num_filters

    def run(self, x_nhwc: torch.Tensor, groups: int = 1) -> torch.Tensor:
        """NHWC ``[N,H,W,Cp] -> [N,1,1,16]`` scores in lane 0 (``groups`` is irrelevant: no batch statistics)."""
        y = F.linear(x_nhwc, self.conv_1.weight.view(self.conv_1.out_channels, self.conv_1.in_channels), self.conv_1.bias)
        for blk in self.stages:
            y = blk.run(y)
        n, h, w, cp = y.shape
        c = self._out_channels
        if isinstance(self.fc.weight, nn.parameter.UninitializedParameter):  # LazyLinear: materialise as torch would on its first call
            with torch.no_grad():
                self.fc.initialize_parameters(torch.empty(1, c * h * w, device=y.device))
                self.fc.in_features = c * h * w
                self.fc.reset_parameters()
            self.fc.__class__ = self.fc.cls_to_become
        if self.fc.in_features != c * h * w:
            raise RuntimeError(f"CloudGANDiscriminator: fc was built for {self.fc.in_features} features, this input gives {c * h * w}")
        wk = torch.nn.functional.pad(self.fc.weight.view(1, c, h, w).permute(0, 2, 3, 1), (0, cp - c)).reshape(1, h * w * cp)
        return F.linear(y.reshape(n, h * w * cp), wk, self.fc.bias).view(n, 1, 1, -1)

    def forward(self, x):
        require_device(x, "x")
        return self.run(F.nchw_to_nhwc(x.float()))[:, 0, 0, :1]


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
        return _run_sequence(list(self.model), x_nhwc, groups, self.training)

    def forward(self, input):
        """Standard forward on NCHW input (one reference call)."""
        require_device(input, "input")
        return F.nhwc_to_nchw(self.run(F.nchw_to_nhwc(input.float()), 1), 1)

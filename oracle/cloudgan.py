"""Oracle: CloudGAN with the ConvLSTM generator and the PatchGAN discriminator (CPU, torch fp32, functional).

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Parity PINNED against the reference import
(``tests/golden/make_golden.py::cloudgan_cases``): the losses of both optimizer steps and the logits reproduce
``satflow.models.cloudgan.CloudGAN`` to 1e-6.

Parameters are passed explicitly in the reference's ``state_dict`` layouts: the generator's ``ConvLSTM.state_dict()`` keys
(``oracle.convlstm``) and the discriminator's ``NLayerDiscriminator.state_dict()`` keys (``model.0.weight`` ...).
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

from . import convlstm as O

Tensor = torch.Tensor


def patch_discriminator(x: Tensor, p: Dict[str, Tensor], n_layers: int = 3, eps: float = 1e-5) -> Tensor:
    """``NLayerDiscriminator.forward`` in training mode (reference ``gan/discriminators.py:139-223``): conv4x4 s2 + LeakyReLU(0.2);
    (n_layers - 1) x [conv4x4 s2 (no bias) + BatchNorm2d (batch statistics) + LeakyReLU]; conv4x4 s1 + BatchNorm2d + LeakyReLU;
    conv4x4 s1 -> 1 channel of patch logits.  Sequential indices: 0 conv, 2 conv, 3 bn, 5 conv, 6 bn, ... as in the reference."""
    y = F.leaky_relu(F.conv2d(x, p["model.0.weight"], p["model.0.bias"], stride=2, padding=1), 0.2)
    idx = 2
    for n in range(1, n_layers + 1):
        stride = 2 if n < n_layers else 1
        y = F.conv2d(y, p[f"model.{idx}.weight"], None, stride=stride, padding=1)
        y = F.batch_norm(y, None, None, p[f"model.{idx + 1}.weight"], p[f"model.{idx + 1}.bias"], training=True, eps=eps)
        y = F.leaky_relu(y, 0.2)
        idx += 3
    return F.conv2d(y, p[f"model.{idx}.weight"], p[f"model.{idx}.bias"], stride=1, padding=1)


def bce_logits(pred: Tensor, real: bool) -> Tensor:
    """``GANLoss("vanilla")`` (reference ``:70-136``): BCE-with-logits against an all-ones / all-zeros target."""
    return F.binary_cross_entropy_with_logits(pred, torch.ones_like(pred) if real else torch.zeros_like(pred))


def generator_step(images: Tensor, future: Tensor, gen: Dict[str, Tensor], disc: Dict[str, Tensor], forecast_steps: int, lambda_l1: float,
                   n_layers: int = 3) -> Tuple[Tensor, List[Tensor]]:
    """``CloudGAN.train_per_timestep(..., optimizer_idx=0)`` (reference ``cloudgan.py:137-157``): returns (g_loss, per-frame lambda * L1)."""
    generated = O.convlstm_forward(images, forecast_steps, gen)  # [B, C, F, H, W]
    total, l1s = 0, []
    for i in range(forecast_steps):
        fake = generated[:, :, i]
        gan = bce_logits(patch_discriminator(fake, disc, n_layers), True)
        l1 = F.l1_loss(fake, future[:, i]) * lambda_l1
        l1s.append(l1)
        total = total + gan + l1
    return total / forecast_steps, l1s


def discriminator_step(images: Tensor, future: Tensor, gen: Dict[str, Tensor], disc: Dict[str, Tensor], forecast_steps: int,
                       n_layers: int = 3) -> Tuple[Tensor, List[Tensor]]:
    """``optimizer_idx=1`` (reference ``cloudgan.py:160-189``): mean over timesteps of (BCE(D(real_i), 1) + BCE(D(fake_i), 0)) / 2."""
    generated = O.convlstm_forward(images, forecast_steps, gen)
    total, per = 0, []
    for i in range(forecast_steps):
        real_loss = bce_logits(patch_discriminator(future[:, i], disc, n_layers), True)
        fake_loss = bce_logits(patch_discriminator(generated[:, :, i], disc, n_layers), False)
        d = (real_loss + fake_loss) / 2
        per.append(d)
        total = total + d
    return total / forecast_steps, per

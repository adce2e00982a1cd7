"""CloudGAN with the ConvLSTM generator on the HIP kernels (surface of reference ``satflow/models/cloudgan.py``; config
``configs/model/cloudgan_convlstm.yaml``): the first in-tree CALLER of the fused ConvLSTM stack with a second network around it
(SURVEY 8f-2).  Same constructor keywords, ``training_step(batch, batch_idx, optimizer_idx)``, ``validation_step``, metric names
and optimizer setup.

Execution differs from the reference by design: with ``condition_time`` the reference loops over the forecast timesteps in Python,
calling the discriminator and both losses once per timestep (``:137-189``).  Here the frames of all timesteps are ONE time-major
NHWC batch: one discriminator pass whose BatchNorm layers keep the reference's per-call statistics (and running-statistics order)
through group-wise reduction, one fused loss kernel per term that also returns the per-timestep values the reference logs.
"""
from __future__ import annotations

from collections import OrderedDict

import torch

from .. import functional as F
from .base import LightningModule, get_loss_l1
from .conv_lstm import ConvLSTM
from .gan import GANLoss, define_discriminator, define_generator
from .layers import ConditionTime


class CloudGAN(LightningModule):  # (not registered in the reference either: cloudgan.py:16 has no @register_model)
    def __init__(
        self,
        forecast_steps: int = 48,
        input_channels: int = 12,
        lr: float = 0.0002,
        beta1: float = 0.5,
        beta2: float = 0.999,
        num_filters: int = 64,
        generator_model: str = "runet",
        norm: str = "batch",
        use_dropout: bool = False,
        discriminator_model: str = "enhanced",
        discriminator_layers: int = 0,
        loss: str = "vanilla",
        scheduler: str = "plateau",
        lr_epochs: int = 10,
        lambda_l1: float = 100.0,
        l1_loss: str = "l1",
        channels_per_timestep: int = 12,
        condition_time: bool = False,
        pretrained: bool = False,
    ):
        """Same keyword surface as reference ``cloudgan.py:17-38``; accepts exactly the keys of ``cloudgan_convlstm.yaml``."""
        super().__init__()
        self.lr, self.b1, self.b2 = lr, beta1, beta2
        self.loss = loss
        self.lambda_l1 = lambda_l1
        self.lr_epochs = lr_epochs
        self.lr_method = scheduler
        self.forecast_steps = forecast_steps
        self.input_channels = input_channels
        self.output_channels = forecast_steps * channels_per_timestep
        self.channels_per_timestep = channels_per_timestep
        self.condition_time = condition_time
        if condition_time:
            self.ct = ConditionTime(forecast_steps)
        if generator_model != "convlstm":
            raise NotImplementedError(f"generator_model={generator_model!r}: the hot path is the ConvLSTM generator "
                                      "(generator_model: 'convlstm', cloudgan_convlstm.yaml:9); R2U_Net is another model family")
        self.recurrent = True
        generator = ConvLSTM(input_channels, hidden_dim=num_filters, out_channels=self.channels_per_timestep)
        self.generator = define_generator(input_channels, self.output_channels, num_filters, generator, norm, use_dropout)
        self.flatten_generator = False  # reference :100-105 compares the MODULE with "convlstm": always False there too
        if not condition_time:
            raise NotImplementedError("condition_time=False feeds the discriminator cat(images, generated) on dim 1 of 5-D tensors, which the "
                                      "reference's own Conv2d discriminator rejects; the shipped config sets condition_time: True")
        self.discriminator = define_discriminator(self.channels_per_timestep, num_filters, discriminator_model, discriminator_layers, norm)
        self.criterionGAN = GANLoss(loss)
        self.criterionL1 = get_loss_l1(l1_loss)
        self.save_hyperparameters()

    # ---- NHWC plumbing ----
    def _frames(self, t5: torch.Tensor, time_dim: int) -> torch.Tensor:
        """5-D NCHW-side tensor with time on ``time_dim`` (1: ``[B,T,C,H,W]``, 2: ``[B,C,T,H,W]``) -> time-major NHWC ``[T*B,H,W,Cp]``."""
        if time_dim == 1:
            B, Tn, C, H, W = t5.shape
            strides = (Tn * C * H * W, C * H * W, H * W)
        else:
            B, C, Tn, H, W = t5.shape
            strides = (C * Tn * H * W, H * W, Tn * H * W)
        return F._ToNHWC.apply(t5.float().contiguous(), B, Tn, C, H, W, strides)

    def forward(self, x, **kwargs):
        return self.generator.forward(x, **kwargs)

    # ---- steps ----
    def train_per_timestep(self, images, future_images, optimizer_idx: int, batch_idx: int):
        """Reference ``:121-189`` (condition_time): mean over the forecast timesteps of the per-timestep losses."""
        Fs, C, lam = self.forecast_steps, self.channels_per_timestep, self.lambda_l1
        generated = self(images, forecast_steps=Fs)          # [B, C, F, H, W]
        fake = self._frames(generated, 2)                     # [F*B, H, W, Cp], frame i = rows i*B .. (i+1)*B
        if optimizer_idx == 0:
            logits = self.discriminator.run(fake, Fs)         # one BatchNorm batch per timestep, as F separate reference calls
            gan_loss, _ = self.criterionGAN.grouped(logits, True, True, Fs)
            l1, l1_frames = F.l1_loss_groups(fake, self._frames(future_images, 1), Fs, C)
            for i, v in enumerate(l1_frames.unbind(0)):
                self.log(f"train/frame_{i}_l1_loss", v * lam)
            g_loss = gan_loss + l1 * lam                     # == mean_i (gan_i + lam * l1_i)
            tqdm_dict = {"g_loss": g_loss}
            self.log_dict({"train/g_loss": g_loss})
            return OrderedDict({"loss": g_loss, "progress_bar": tqdm_dict, "log": tqdm_dict})
        if optimizer_idx == 1:
            real = self._frames(future_images, 1)
            B = images.shape[0]
            # reference call order per timestep: D(real_i), then D(fake_i)  ->  groups 2i, 2i + 1
            both = torch.stack((real.view(Fs, B, *real.shape[1:]), fake.view(Fs, B, *fake.shape[1:])), 1).reshape(2 * Fs * B, *real.shape[1:])
            logits = self.discriminator.run(both, 2 * Fs)
            d_loss, per_call = self.criterionGAN.grouped(logits, True, False, 2 * Fs)  # mean over 2F calls == mean_i (real_i + fake_i) / 2
            for i, v in enumerate(per_call.view(Fs, 2).mean(1).unbind(0)):
                self.log(f"train/frame_{i}_d_loss", v)
            tqdm_dict = {"d_loss": d_loss}
            self.log_dict({"train/d_loss": d_loss})
            return OrderedDict({"loss": d_loss, "progress_bar": tqdm_dict, "log": tqdm_dict})
        raise ValueError(f"optimizer_idx {optimizer_idx}")

    def training_step(self, batch, batch_idx, optimizer_idx):
        images, future_images = batch
        return self.train_per_timestep(images, future_images, optimizer_idx, batch_idx)

    def val_per_timestep(self, images, future_images, batch_idx):
        """Reference ``:271-313``."""
        Fs, C, lam = self.forecast_steps, self.channels_per_timestep, self.lambda_l1
        generated = self(images, forecast_steps=Fs)
        fake, real = self._frames(generated, 2), self._frames(future_images, 1)
        B = images.shape[0]
        # reference call order per timestep: D(fake_i) [generator term], D(real_i), D(fake_i) again
        logits_fake = self.discriminator.run(fake, Fs)
        logits_real = self.discriminator.run(real, Fs)
        gan_loss, _ = self.criterionGAN.grouped(logits_fake, True, True, Fs)
        l1, l1_frames = F.l1_loss_groups(fake, real, Fs, C)
        real_loss, real_i = self.criterionGAN.grouped(logits_real, True, True, Fs)
        fake_loss, fake_i = self.criterionGAN.grouped(logits_fake, False, False, Fs)
        for i in range(Fs):
            self.log(f"val/frame_{i}_d_loss", (real_i[i] + fake_i[i]) / 2)
            self.log(f"val/frame_{i}_l1_loss", l1_frames[i] * lam)
        g_loss = gan_loss + l1 * lam
        d_loss = (real_loss + fake_loss) / 2
        loss = g_loss + d_loss
        tqdm_dict = {"loss": loss}
        self.log_dict({"val/d_loss": d_loss, "val/g_loss": g_loss, "val/loss": d_loss + g_loss})
        return OrderedDict({"val/discriminator_loss": d_loss, "val/generator_loss": g_loss, "progress_bar": tqdm_dict, "log": tqdm_dict})

    def validation_step(self, batch, batch_idx):
        images, future_images = batch
        return self.val_per_timestep(images, future_images, batch_idx)

    def configure_optimizers(self):
        """Reference ``:323-352``."""
        from torch.optim import lr_scheduler

        from .pl_metnet import LinearWarmupCosineAnnealingLR

        opt_g = torch.optim.Adam(self.generator.parameters(), lr=self.lr, betas=(self.b1, self.b2))
        opt_d = torch.optim.Adam(self.discriminator.parameters(), lr=self.lr, betas=(self.b1, self.b2))
        if self.lr_method == "plateau":
            g_s = lr_scheduler.ReduceLROnPlateau(opt_g, mode="min", factor=0.2, threshold=0.01, patience=10)
            d_s = lr_scheduler.ReduceLROnPlateau(opt_d, mode="min", factor=0.2, threshold=0.01, patience=10)
        elif self.lr_method == "cosine":
            g_s = lr_scheduler.CosineAnnealingLR(opt_g, T_max=self.lr_epochs, eta_min=0)
            d_s = lr_scheduler.CosineAnnealingLR(opt_d, T_max=self.lr_epochs, eta_min=0)
        elif self.lr_method == "warmup":
            g_s = LinearWarmupCosineAnnealingLR(opt_g, warmup_epochs=self.lr_epochs, max_epochs=100)
            d_s = LinearWarmupCosineAnnealingLR(opt_d, warmup_epochs=self.lr_epochs, max_epochs=100)
        else:
            return NotImplementedError("learning rate policy is not implemented")
        return [opt_g, opt_d], [g_s, d_s]

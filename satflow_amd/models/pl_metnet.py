"""LitMetNet: the registered Lightning wrapper around MetNet (surface of reference ``satflow/models/pl_metnet.py``)."""
from __future__ import annotations

import math
from typing import Any, Dict

import torch
from torch import nn

from .base import BaseModel, get_loss, register_model
from .metnet import MetNet

# batch-dict keys of nowcasting_dataset.consts (reference pl_metnet.py:7); the package is not part of this build
SATELLITE_DATA, TOPOGRAPHIC_DATA, NWP_DATA = "sat_data", "topo_data", "nwp"

head_to_module = {"identity": nn.Identity()}


class LinearWarmupCosineAnnealingLR(torch.optim.lr_scheduler.LambdaLR):
    """Closed form of ``pl_bolts`` LinearWarmupCosineAnnealingLR(warmup_epochs, max_epochs) with
    ``warmup_start_lr=0``, ``eta_min=0`` as configured at reference ``pl_metnet.py:71`` (stepped per step, ``:77``)."""

    def __init__(self, optimizer, warmup_epochs: int, max_epochs: int):
        def factor(epoch: int) -> float:
            if epoch < warmup_epochs:
                return epoch / max(1, warmup_epochs - 1) if warmup_epochs > 1 else 1.0
            return 0.5 * (1.0 + math.cos(math.pi * (epoch - warmup_epochs) / max(1, max_epochs - warmup_epochs)))

        super().__init__(optimizer, factor)


@register_model
class LitMetNet(BaseModel):
    def __init__(
        self,
        image_encoder: str = "downsampler",
        input_channels: int = 12,
        sat_channels: int = 12,
        input_size: int = 256,
        output_channels: int = 12,
        hidden_dim: int = 64,
        kernel_size: int = 3,
        num_layers: int = 1,
        num_att_layers: int = 1,
        head: str = "identity",
        forecast_steps: int = 48,
        temporal_dropout: float = 0.2,
        lr: float = 0.001,
        pretrained: bool = False,
        visualize: bool = False,
        loss: str = "mse",
    ):
        """Same keyword surface as reference ``pl_metnet.py:17-35``; accepts exactly the keys of ``configs/model/metnet.yaml``."""
        super().__init__()
        self.forecast_steps = forecast_steps
        self.input_channels = input_channels
        self.lr = lr
        self.pretrained = pretrained
        self.visualize = visualize
        self.output_channels = output_channels
        self.criterion = get_loss(loss, channel=output_channels, nonnegative_ssim=True, convert_range=True)
        self.model = MetNet(
            image_encoder=image_encoder, input_channels=input_channels, sat_channels=sat_channels, input_size=input_size,
            output_channels=output_channels, hidden_dim=hidden_dim, kernel_size=kernel_size, num_layers=num_layers,
            num_att_layers=num_att_layers, head=head_to_module[head], forecast_steps=forecast_steps,
            temporal_dropout=temporal_dropout,
        )
        self.save_hyperparameters()

    def forward(self, imgs, **kwargs) -> Any:
        return self.model(imgs)

    def configure_optimizers(self):
        optimizer = torch.optim.Adam(self.parameters(), lr=self.lr)
        scheduler = LinearWarmupCosineAnnealingLR(optimizer, warmup_epochs=10, max_epochs=100)
        return {"optimizer": optimizer, "lr_scheduler": {"scheduler": scheduler, "interval": "step", "frequency": 1, "name": None}}

    def _combine_data_sources(self, x: Dict[str, torch.Tensor]) -> torch.Tensor:
        """Concatenate satellite, time-repeated topographic and NWP data on dim 1 (reference ``:90-107``)."""
        timesteps = x[SATELLITE_DATA].shape[2]
        topo = x[TOPOGRAPHIC_DATA].unsqueeze(2).expand(-1, -1, timesteps, -1, -1)
        nwp = x.get(NWP_DATA, [])
        to_concat = [x[SATELLITE_DATA], topo] + ([nwp] if isinstance(nwp, torch.Tensor) else list(nwp))
        return torch.cat(to_concat, dim=1).float()

    def _train_or_validate_step(self, batch, batch_idx, is_training: bool = True):
        """Reference ``:109-124`` with the per-frame ``.item()`` loop replaced by one reduction (metric names unchanged)."""
        x, y = batch
        if isinstance(x, dict):
            y = y[SATELLITE_DATA].float()
            x = self._combine_data_sources(x)
        y_hat = self(x)
        tag = "train" if is_training else "val"
        loss = self.criterion(y_hat, y)
        self.log(f"{tag}/loss", loss, prog_bar=True)
        frames = getattr(self.criterion, "last_frame_losses", None)  # from the fused loss kernel, same pass
        if frames is None:
            frames = ((y_hat.detach() - y) ** 2).mean(dim=tuple(d for d in range(y.dim()) if d != 1))
        self.log_dict({f"{tag}/frame_{f}_loss": v for f, v in enumerate(frames.unbind(0))})
        return loss

    def training_step(self, batch, batch_idx):
        return self._train_or_validate_step(batch, batch_idx, is_training=True)

    def validation_step(self, batch, batch_idx):
        return self._train_or_validate_step(batch, batch_idx, is_training=False)

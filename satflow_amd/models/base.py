"""Model registry, LightningModule surface and loss factory for the hot-path models.

The reference gets these from un-vendored packages: ``register_model / create_model /
get_model / list_models / BaseModel`` and ``get_loss`` from ``nowcasting_utils``
(reference ``satflow/models/__init__.py:1``, ``conv_lstm.py:7-8``, ``pl_metnet.py:8-9``) and
``LightningModule`` from ``pytorch_lightning``.  This file supplies the same call surface
(SURVEY.md 8b): names are the class ``__name__``; ``create_model`` forwards ``pretrained``.
If ``pytorch_lightning`` is importable its ``LightningModule`` is used, otherwise a shim
with the method names the models and a Trainer-like loop rely on.
"""
from __future__ import annotations

import inspect
from typing import Any, Callable, Dict, List, Type, Union

import torch
from torch import nn

try:  # pragma: no cover - not installed in the build image
    from pytorch_lightning import LightningModule as _LightningModule
except Exception:  # noqa: BLE001

    class _LightningModule(nn.Module):
        """Minimal stand-in: hyper-parameter capture + metric log sink."""

        def __init__(self) -> None:
            super().__init__()
            self.hparams: Dict[str, Any] = {}
            self.logged: Dict[str, Any] = {}

        def save_hyperparameters(self, *args: Any, **kwargs: Any) -> None:
            frame = inspect.currentframe().f_back
            init = getattr(type(self), "__init__")
            names = [n for n in inspect.signature(init).parameters if n != "self"]
            self.hparams = {n: frame.f_locals[n] for n in names if n in frame.f_locals}

        def log(self, name: str, value: Any, **kwargs: Any) -> None:
            self.logged[name] = value

        def log_dict(self, values: Dict[str, Any], **kwargs: Any) -> None:
            self.logged.update(values)


LightningModule = _LightningModule

_REGISTRY: Dict[str, Type[nn.Module]] = {}


def register_model(cls: Type[nn.Module]) -> Type[nn.Module]:
    """Class decorator: register under ``cls.__name__`` (reference ``conv_lstm.py:13``, ``pl_metnet.py:15``)."""
    _REGISTRY[cls.__name__] = cls
    return cls


def list_models() -> List[str]:
    return sorted(_REGISTRY)


def get_model(name: str) -> Type[nn.Module]:
    if name not in _REGISTRY:
        raise KeyError(f"unknown model {name!r}; registered: {list_models()}")
    return _REGISTRY[name]


def create_model(name: str, pretrained: bool = False, **kwargs: Any) -> nn.Module:
    """``create_model(name, pretrained=False, **kw)`` (reference ``tests/test_models.py:64-76``).

    ``hf_hub:`` names need network access (skipped in the reference's own tests, ``:79-102``).
    """
    if name.startswith("hf_hub:"):
        raise RuntimeError("hf_hub checkpoints need network access, which this build does not have")
    return get_model(name)(pretrained=pretrained, **kwargs)


class BaseModel(LightningModule):
    """Placeholder for ``nowcasting_utils.models.base.BaseModel`` (``pl_metnet.py:16``)."""


class MSELoss(nn.Module):
    """``nn.MSELoss()`` (mean reduction) on the fused HIP kernel: loss, its gradient and the per-frame losses of dim 1
    in one pass (``sf_mse_loss``).  ``last_frame_losses`` holds the per-frame means of the latest call (device tensor)."""

    def __init__(self) -> None:
        super().__init__()
        self.last_frame_losses = None

    def forward(self, pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        from ..functional import mse_loss_with_frames

        if pred.dim() < 2:
            pred, target = pred.reshape(1, -1), target.reshape(1, -1)
        loss, frames = mse_loss_with_frames(pred, target, frame_dim=1)
        self.last_frame_losses = frames
        return loss


class L1Loss(nn.Module):
    """``nn.L1Loss()`` (mean) on the fused HIP kernel (``sf_l1_loss``) for same-shape NCHW-side tensors."""

    def forward(self, pred: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        from ..functional import l1_loss_groups

        p2 = pred.float().contiguous().view(1, -1)
        t2 = target.float().contiguous().view(1, -1)
        n = p2.shape[1]
        pad = (-n) % 8  # the kernels want 16-byte aligned rows
        if pad:
            p2, t2 = torch.nn.functional.pad(p2, (0, pad)), torch.nn.functional.pad(t2, (0, pad))
        loss, _ = l1_loss_groups(p2, t2, 1, n)
        return loss


def get_loss_l1(name: str = "l1") -> nn.Module:
    """The L1 term of CloudGAN (``get_loss(l1_loss, ...)``, reference ``cloudgan.py:118``): only ``"l1"`` is on the path."""
    if name in ("l1", "L1"):
        return L1Loss()
    raise ValueError(f"l1_loss {name!r} is outside the hot-path scope (only 'l1')")


def get_loss(loss: Union[str, nn.Module, Callable] = "mse", **kwargs: Any) -> nn.Module:
    """``nowcasting_utils.models.loss.get_loss``: only ``"mse"`` is on the hot path."""
    if isinstance(loss, nn.Module):
        return loss
    if callable(loss) and not isinstance(loss, str):
        return loss
    if loss in ("mse", "MSE"):
        return MSELoss()
    raise ValueError(f"loss {loss!r} is outside the hot-path scope (only 'mse'; see DESIGN.md)")

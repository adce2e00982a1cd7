"""Model surface (mirrors reference ``satflow/models/__init__.py:1-7`` for the hot-path models)."""
from .base import BaseModel, create_model, get_loss, get_model, list_models, register_model
from .conv_lstm import ConvLSTM, EncoderDecoderConvLSTM
from .cloudgan import CloudGAN
from .metnet import MetNet
from .pl_metnet import LitMetNet

__all__ = [
    "BaseModel", "create_model", "get_model", "list_models", "register_model", "get_loss",
    "ConvLSTM", "EncoderDecoderConvLSTM", "CloudGAN", "MetNet", "LitMetNet",
]

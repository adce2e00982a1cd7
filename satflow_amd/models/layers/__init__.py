"""Layer surface (mirrors reference ``satflow/models/layers/__init__.py:1-6`` for the hot-path layers)."""
from .ConvLSTM import ConvLSTMCell

__all__ = ["ConvLSTMCell"]

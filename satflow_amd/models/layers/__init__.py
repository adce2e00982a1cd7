"""Layer surface (mirrors reference ``satflow/models/layers/__init__.py:1-6`` for the hot-path layers)."""
from .ConditionTime import ConditionTime
from .ConvLSTM import ConvLSTMCell
from .SpatioTemporalLSTMCell_memory_decoupling import SpatioTemporalLSTMCell
from .TimeDistributed import TimeDistributed

__all__ = ["ConditionTime", "ConvLSTMCell", "SpatioTemporalLSTMCell", "TimeDistributed"]

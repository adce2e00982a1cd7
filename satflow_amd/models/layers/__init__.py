"""Layer surface (mirrors reference ``satflow/models/layers/__init__.py:1-6`` for the hot-path layers, plus the in-tree DGMR /
attention pieces of SURVEY 8f-3 / 8f-4, importable by their reference module paths ``satflow_amd.models.layers.<Module>``)."""
from .Attention import SelfAttention, SelfAttention2d, SeparableAttn, SeparableAttnCell
from .ConditionTime import ConditionTime
from .ConvGRU import ConvGRU, ConvGRUCell
from .ConvLSTM import ConvLSTMCell
from .Discriminator import SpatialDiscriminator, TemporalDiscriminator
from .Generator import Generator
from .GResBlock import GResBlock
from .Normalization import ConditionalNorm, SpectralNorm
from .SpatioTemporalLSTMCell_memory_decoupling import SpatioTemporalLSTMCell
from .TimeDistributed import TimeDistributed

__all__ = ["ConditionTime", "ConvLSTMCell", "SpatioTemporalLSTMCell", "TimeDistributed", "SelfAttention", "SelfAttention2d", "SeparableAttn",
           "SeparableAttnCell", "ConvGRU", "ConvGRUCell", "SpatialDiscriminator", "TemporalDiscriminator", "Generator", "GResBlock",
           "ConditionalNorm", "SpectralNorm"]

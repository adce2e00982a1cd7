"""Minimal Hydra-style ``_target_`` instantiation for the hot-path model configs.

The reference instantiates models with ``hydra.utils.instantiate(config.model)``
(``satflow/experiments/train.py:44-45``) from ``satflow/configs/model/*.yaml``; its tests do
``Cls(**yaml_without__target_)`` (``tests/test_models.py:43-45``).  Hydra/omegaconf are not part of
this build, so this is the small subset needed for "configs load unchanged": same YAML keys ->
same constructor kwargs, ``satflow.`` targets resolved inside ``satflow_amd``, plus aliases for
the targets that are stale in the reference itself.
"""
from __future__ import annotations

import importlib
from typing import Any, Dict, Mapping

import yaml

# stale / moved targets of the shipped configs -> the class that actually implements them
TARGET_ALIASES = {
    "satflow.models.metnet.MetNet": "satflow_amd.models.pl_metnet.LitMetNet",  # configs/model/metnet.yaml:1
    "satflow.models.pl_metnet.LitMetNet": "satflow_amd.models.pl_metnet.LitMetNet",
    "satflow.models.conv_lstm.EncoderDecoderConvLSTM": "satflow_amd.models.conv_lstm.EncoderDecoderConvLSTM",
}


def resolve_target(target: str):
    path = TARGET_ALIASES.get(target, target)
    if path.startswith("satflow."):
        path = "satflow_amd." + path[len("satflow."):]
    module, _, name = path.rpartition(".")
    return getattr(importlib.import_module(module), name)


def load_config(path: str) -> Dict[str, Any]:
    with open(path, "r") as f:
        return yaml.load(f, Loader=yaml.FullLoader)


def instantiate(config: Mapping[str, Any], **overrides: Any):
    """``hydra.utils.instantiate`` for a flat model config: pops ``_target_``, passes the rest as kwargs."""
    cfg = dict(config)
    cfg.update(overrides)
    target = cfg.pop("_target_")
    return resolve_target(target)(**cfg)

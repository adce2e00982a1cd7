"""The SF_TEST_POISON debugging mode of the suite does what it says (tests/conftest.py::_poison_allocator)."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_poison_mode_fills_fresh_allocations_with_nan(device):
    if not os.environ.get("SF_TEST_POISON"):
        pytest.skip("SF_TEST_POISON not set")
    for shape in [(1000,), (64, 64, 160), (8, 128, 128, 256)]:
        t = torch.empty(shape, device=device)
        assert torch.isnan(t).all(), shape

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# Parity gate of BASELINE.json north_star for the fp32 path.
RTOL, ATOL = 1e-4, 1e-5


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a HIP device (MI355X); run with -m gpu on the GPU box")


# Tests that hold the fp32 parity gate (rtol 1e-4 / atol 1e-5) and run convolutions: each runs twice, in "f32" (exact-fp32 MFMA) and in "f32e"
# (fp32-equivalent products from three fp16 products, satflow_amd/_hip.py) - UNCHANGED bodies and tolerances.
FP32_GATED = {
    "test_convlstm_gpu.py": ("test_conv3x3_vs_oracle", "test_cell_golden", "test_model_golden", "test_model_vs_oracle_no_grad", "test_training_trajectory_golden"),
    "test_metnet_gpu.py": ("test_convgru_sequence", "test_metnet_train_step_vs_oracle", "test_metnet_eval_and_reference_shape_pin"),
    "test_litmetnet_gpu.py": ("test_training_step_dict_batch", "test_validation_step_eval_mode", "test_metnet_input_gradient"),
    "test_fullsize_gpu.py": ("test_convlstm_cfg2_forward_fullsize", "test_metnet_cfg3_forward_fullsize", "test_conv_linearity_and_adjoint_fullbatch"),
    "test_baseline_size_gpu.py": ("test_cfg2_convlstm_train_step_fullsize_f32", "test_cfg3_metnet_train_step_fullsize_f32",
                                  "test_cfg3_metnet_train_step_fullsize_f32_with_dropout", "test_config0_convgru_two_layers"),
    "test_cloudgan_gpu.py": ("test_patch_discriminator_golden", "test_cloudgan_training_steps_golden", "test_discriminator_variants_match_reference"),
    "test_stlstm_gpu.py": ("test_stlstm_cell_matches_reference_golden_fp32",),
    "test_dgmr_gpu.py": ("test_gresblock_matches_reference", "test_discriminator_matches_reference", "test_generator_matches_oracle",
                         "test_convgru_stack_matches_oracle", "test_conv5x5_on_the_3x3_kernels", "test_dgmr_gan_step_matches_oracle", "test_dgmr_configs4_size"),
    "test_attention_gpu.py": ("test_attention_layer_matches_reference",),
}


def pytest_generate_tests(metafunc):
    names = FP32_GATED.get(os.path.basename(str(metafunc.module.__file__)), ())
    if metafunc.function.__name__ in names:
        metafunc.fixturenames.append("fp32_mode")
        metafunc.parametrize("fp32_mode", ["f32", "f32e"], indirect=True)


@pytest.fixture
def fp32_mode(request):
    """The mode of this parametrisation (set by ``_apply_fp32_mode``); a test may take it as an argument to name the mode in its messages."""
    return request.param


@pytest.fixture(autouse=True)
def _apply_fp32_mode(request):
    """Sets the compute mode of an FP32_GATED parametrisation for the duration of the test.  AUTOUSE, reading the parameter from the call spec: a
    fixture that ``pytest_generate_tests`` merely appends to ``metafunc.fixturenames`` is parametrised (the ids show ``[f32e]``) but never set up unless the
    test function also takes it as an argument - the first version of this file ran every "f32e" case in "f32" (round 6; found because the published
    parity records of the two modes were identical to the last digit).  ``test_the_f32e_cases_really_run_in_f32e`` guards it."""
    cs = getattr(request.node, "callspec", None)
    mode = cs.params.get("fp32_mode") if cs is not None else None
    if mode is None:
        yield
        return
    import satflow_amd

    satflow_amd.set_compute_dtype(mode)
    yield
    satflow_amd.set_compute_dtype("f32")


@pytest.fixture(autouse=True)
def _poison_allocator(request):
    """SF_TEST_POISON=1 (a debugging mode of the GPU suite, off by default): before every test the caching allocator's free blocks are filled with NaN bit patterns,
    so a kernel that READS memory nobody wrote (a `torch.empty` buffer, pad lanes, a ragged tile) computes on NaN instead of on whatever the previous
    test left there - which on a warm box is usually plausible data of the same shape, and on a fresh one zeros.  Found nothing to fix when it was added (round 6);
    kept because a dependence of that kind shows up as a once-in-twenty-runs failure otherwise."""
    if not os.environ.get("SF_TEST_POISON") or request.node.get_closest_marker("gpu") is None:
        yield
        return
    import torch

    if torch.cuda.is_available():
        torch.cuda.synchronize()
        free, _ = torch.cuda.mem_get_info()
        big = torch.full((int(free * 0.6) // 4,), float("nan"), dtype=torch.float32, device="cuda")       # the large pool: later allocations are carved from this block
        small = [torch.full((255 * 1024,), float("nan"), dtype=torch.float32, device="cuda") for _ in range(128)]   # < 1 MiB each: the small pool's segments
        torch.cuda.synchronize()
        del big, small
    yield


@pytest.fixture(scope="session")
def device():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda:0")


def rel_l2(actual, expected) -> float:
    import torch

    a = torch.as_tensor(actual).detach().float().cpu()
    e = torch.as_tensor(expected).detach().float().cpu()
    return float((a - e).norm() / (e.norm() + 1e-30))


def assert_close(actual, expected, name, rtol=RTOL, atol=ATOL, grad=False, force_rel=False):
    """allclose with a readable report (also checks relative L2 so tiny tensors cannot hide).

    ``grad=True``: the tensor is a gradient whose entries are sums of 10^2..10^5 O(1) products
    (|ref|max up to ~10^2 with the random cotangents of the goldens).  Both sides carry fp32
    summation noise proportional to that scale, so ``atol`` is applied to the tensor normalised
    by ``max(1, |ref|max)``; ``rtol`` is unchanged.
    """
    import torch

    a = torch.as_tensor(actual).detach().float().cpu()
    e = torch.as_tensor(expected).detach().float().cpu()
    assert a.shape == e.shape, f"{name}: shape {tuple(a.shape)} != {tuple(e.shape)}"
    assert torch.isfinite(a).all(), f"{name}: non-finite values"
    err = (a - e).abs()
    if grad:
        atol = atol * max(1.0, float(e.abs().max()))
    bound = atol + rtol * e.abs()
    bad = err > bound
    rel_l2 = (a - e).norm() / (e.norm() + 1e-30)
    assert not bad.any(), (
        f"{name}: {int(bad.sum())}/{a.numel()} elements outside rtol={rtol} atol={atol}; "
        f"max abs err {err.max():.3e} at |ref| {e.flatten()[err.argmax()].abs():.3e}; rel L2 {rel_l2:.3e}"
    )
    if force_rel:  # tensors whose entries all sit below atol (dx under default init): the elementwise check is vacuous
        assert float(e.norm()) > 0 and rel_l2 <= 10 * rtol, f"{name}: relative L2 error {rel_l2:.3e} (|ref|max {e.abs().max():.3e})"
    if float(e.norm()) > 100 * atol * e.numel() ** 0.5:  # skip for tensors that are numerically zero (e.g. d(bias) in front of BatchNorm)
        assert rel_l2 <= 10 * rtol, f"{name}: relative L2 error {rel_l2:.3e}"

"""CPU: host logic - C-ABI symbol table, registry/config surface, state_dict keys, GEMM index maps, loud failure."""
import os
import re

import pytest
import torch

from conftest import GOLDEN, ROOT


def test_abi_exports_every_declared_symbol():
    from satflow_amd import _hip

    header = open(os.path.join(ROOT, "include", "satflow_hip.h")).read()
    declared = set(re.findall(r"\b(sf_[a-z0-9_]+)\s*\(", header))
    assert declared, "no symbols parsed"
    L = _hip.lib()
    for name in declared:
        assert hasattr(L, name), f"libsatflow_hip.so does not export {name}"
    assert declared == set(_hip.PROTOTYPES), declared ^ set(_hip.PROTOTYPES)
    assert L.sf_abi_version() == _hip.ABI_VERSION


def test_registry_and_state_dict_keys():
    from satflow_amd.models import EncoderDecoderConvLSTM, create_model, get_model, list_models

    assert "EncoderDecoderConvLSTM" in list_models()
    assert get_model("EncoderDecoderConvLSTM") is EncoderDecoderConvLSTM
    for name in list_models():  # reference tests/test_models.py:64-76
        create_model(name)
    m = create_model("EncoderDecoderConvLSTM", pretrained=False, hidden_dim=8, input_channels=4)
    want = open(os.path.join(GOLDEN, "convlstm_state_dict_keys.txt")).read().split()
    assert list(m.state_dict().keys()) == want
    assert m.model.encoder_1_convlstm.conv.weight.shape == (32, 12, 3, 3)
    assert m.model.decoder_CNN.weight.shape == (1, 8, 1, 3, 3)
    assert isinstance(m.configure_optimizers(), torch.optim.Adam)
    with pytest.raises(KeyError):
        get_model("nope")


def test_configs_load_unchanged():
    from satflow_amd.config import instantiate, load_config
    from satflow_amd.models import EncoderDecoderConvLSTM

    cfg = load_config(os.path.join(GOLDEN, "configs", "convlstm.yaml"))
    m = instantiate(cfg)
    assert isinstance(m, EncoderDecoderConvLSTM)
    assert m.forecast_steps == 24 and m.lr == 1e-4 and m.model.input_channels == 17
    cfg.pop("_target_")
    assert EncoderDecoderConvLSTM(**cfg).hparams["hidden_dim"] == 64  # reference tests/test_models.py:43-45 pattern
    with pytest.raises(NotImplementedError):
        instantiate(load_config(os.path.join(GOLDEN, "configs", "convlstm_coord.yaml")))
    with pytest.raises(ValueError):
        EncoderDecoderConvLSTM(conv_type="bogus")


def test_gemm_maps_cover_every_weight_once():
    from satflow_amd import kernels as K

    for cin, hid in [(4, 8), (12, 64), (3, 5), (64, 64), (17, 40)]:
        f = K.lstm_fwd_map(cin, hid)
        assert sorted(i for i in f.nmap if i >= 0) == list(range(4 * hid))
        assert sorted(i for i in f.kmap if i >= 0) == list(range(cin + hid))
        assert f.Np % 128 == 0 and f.Kp % 16 == 0
        # all four gates of a hidden channel sit at the same column of consecutive 32-lane fragments
        for nb in range(f.Np // 128):
            for j in range(32):
                col = [f.nmap[nb * 128 + g * 32 + j] for g in range(4)]
                if col[0] >= 0:
                    assert [c - col[0] for c in col] == [0, hid, 2 * hid, 3 * hid]
        for need_dx in (False, True):
            b = K.lstm_bwd_map(cin, hid, need_dx)
            want = list(range(cin + hid)) if need_dx else list(range(cin, cin + hid))
            assert sorted(i for i in b.nmap if i >= 0) == want
            assert sorted(i for i in b.kmap if i >= 0) == list(range(4 * hid))
            assert b.Np % (32 * b.nf) == 0
    assert K.choose_nf(160) == 5 and K.choose_nf(256) == 4 and K.choose_nf(96) == 3 and K.choose_nf(16) == 1


def test_cpu_input_fails_loudly():
    from satflow_amd.models import ConvLSTM

    net = ConvLSTM(4, 8, 1)
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        net(torch.randn(1, 2, 4, 16, 16), 2)


def test_product_never_imports_oracle():
    import subprocess
    import sys

    code = "import sys; import satflow_amd.models, satflow_amd.kernels, satflow_amd.config; assert not any(m.split('.')[0]=='oracle' for m in sys.modules)"
    subprocess.run([sys.executable, "-c", code], check=True, cwd=ROOT)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "satflow_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f"{f} imports the oracle"


def test_tensor_descriptor_matches_header():
    """ctypes `sfTensor` mirrors include/satflow_hip.h field for field (incl. the storage dtype), and `T()` fills byte offsets
    and the storage type from the torch tensor."""
    import ctypes as C
    import re

    import torch
    from satflow_amd import _hip

    hdr = open(os.path.join(ROOT, "include", "satflow_hip.h")).read()
    body = re.search(r"typedef struct \{(.*?)\} sfTensor;", hdr, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"(?:const float\*|void\*|int32_t)\s+(\w+);", body)
    assert fields == [f[0] for f in _hip.sfTensor._fields_], (fields, _hip.sfTensor._fields_)
    assert C.sizeof(_hip.sfTensor) == 40  # 8-byte pointer + 5 x int32 (padded to the pointer's alignment) + the amax pointer (ABI 8)
    a = torch.zeros(2, 3, 32, dtype=torch.float32)
    b = torch.zeros(2, 3, 32, dtype=torch.bfloat16)
    ta, tb = _hip.T(a, c=16, offset=16), _hip.T(b, c=16, offset=16)
    assert (ta.dtype, tb.dtype) == (_hip.SF_F32, _hip.SF_BF16)
    assert ta.ptr - a.data_ptr() == 64 and tb.ptr - b.data_ptr() == 32
    assert (ta.c, ta.stride, tb.c, tb.stride) == (16, 32, 16, 32)


def test_compute_modes():
    import torch
    import satflow_amd
    from satflow_amd import _hip

    try:
        for name, (comp, enc) in {"f32": (_hip.SF_F32, torch.float32), "bf16": (_hip.SF_BF16, torch.float32),
                                  "bf16a": (_hip.SF_BF16, torch.bfloat16), "f32e": (_hip.SF_F32E, torch.float32)}.items():
            satflow_amd.set_compute_dtype(name)
            assert _hip.compute_dtype() == comp and _hip.encoder_storage_dtype() == enc and satflow_amd.compute_dtype_name() == name
        import pytest
        with pytest.raises(KeyError):
            satflow_amd.set_compute_dtype("fp8")
    finally:
        satflow_amd.set_compute_dtype("f32")


def test_standalone_layers_match_reference_goldens():
    """The stand-alone `ConditionTime` / `TimeDistributed` modules (views only, device-agnostic) against the tensors the
    reference's own modules produced (tests/golden/metnet_layers.npz; layers/ConditionTime.py:22-33, TimeDistributed.py:21-40)."""
    import numpy as np
    import torch

    from conftest import GOLDEN
    from satflow_amd.models.layers import ConditionTime, TimeDistributed
    from satflow_amd.models.utils import space_to_depth

    G = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(GOLDEN, "metnet_layers.npz")).items()}
    assert torch.equal(ConditionTime(7)(G["x5"], 2), G["ct5"])
    assert torch.equal(ConditionTime(5, ch_dim=3, num_dims=4)(G["x4"], 4), G["ct4"])
    with pytest.raises(AssertionError):
        ConditionTime(3)(G["x5"], 3)  # `assert i < seq_len` (ConditionTime.py:7)
    conv = torch.nn.Conv2d(4, 3, 3, padding=1)
    with torch.no_grad():
        conv.weight.copy_(G["td_weight"]), conv.bias.copy_(G["td_bias"])
        assert torch.allclose(TimeDistributed(conv)(G["x5"]), G["td"], rtol=0, atol=1e-6)
        assert torch.allclose(TimeDistributed(conv, low_mem=True)(G["x5"]), G["td_low"], rtol=0, atol=1e-6)
    s2d = space_to_depth(G["s4"].numpy(), spatial_block_size=2)  # numpy in, numpy out, as the reference accepts (utils.py:48-50)
    assert isinstance(s2d, np.ndarray) and np.array_equal(s2d, G["s2d"].numpy())
    assert torch.equal(space_to_depth(G["s4"], spatial_block_size=2), G["s2d"])


def test_warmup_cosine_schedule_closed_form():
    """LinearWarmupCosineAnnealingLR(warmup_epochs=10, max_epochs=100), stepped per optimizer step (pl_metnet.py:70-77), against
    the closed form published by pl_bolts: lr = base*e/(W-1) for e < W, else base/2 * (1 + cos(pi (e-W)/(M-W)))."""
    import math

    import torch

    from satflow_amd.models import LitMetNet

    m = LitMetNet(input_channels=5, sat_channels=4, input_size=8, output_channels=2, hidden_dim=16, forecast_steps=3, lr=1e-3)
    cfg = m.configure_optimizers()
    opt, sched = cfg["optimizer"], cfg["lr_scheduler"]["scheduler"]
    assert isinstance(opt, torch.optim.Adam) and cfg["lr_scheduler"]["interval"] == "step" and cfg["lr_scheduler"]["frequency"] == 1
    table = {0: 0.0, 1: 1e-3 / 9, 5: 5e-3 / 9, 9: 1e-3, 10: 1e-3, 55: 0.5e-3, 100: 0.0}
    for e in range(101):
        lr = opt.param_groups[0]["lr"]
        want = 1e-3 * e / 9 if e < 10 else 0.5e-3 * (1 + math.cos(math.pi * (e - 10) / 90))
        assert abs(lr - want) < 1e-12, (e, lr, want)
        if e in table:
            assert abs(lr - table[e]) < 1e-9, (e, lr)
        opt.step()
        sched.step()


def test_combine_data_sources_matches_reference_semantics():
    """`_combine_data_sources` (pl_metnet.py:90-107): satellite, time-repeated topography (einops 'b c h w -> b c t h w') and the
    optional NWP list concatenated on dim 1, as float."""
    import torch

    from satflow_amd.models import LitMetNet
    from satflow_amd.models.pl_metnet import NWP_DATA, SATELLITE_DATA, TOPOGRAPHIC_DATA

    m = LitMetNet(input_channels=5, sat_channels=4, input_size=8, output_channels=2, hidden_dim=16, forecast_steps=3)
    sat = torch.randn(2, 3, 4, 6, 6).double()
    topo = torch.randn(2, 1, 6, 6).double()
    nwp = [torch.randn(2, 2, 4, 6, 6).double()]
    out = m._combine_data_sources({SATELLITE_DATA: sat, TOPOGRAPHIC_DATA: topo})
    assert out.dtype == torch.float32 and out.shape == (2, 4, 4, 6, 6)
    assert torch.equal(out[:, :3], sat.float())
    for t in range(4):
        assert torch.equal(out[:, 3, t], topo[:, 0].float())
    out2 = m._combine_data_sources({SATELLITE_DATA: sat, TOPOGRAPHIC_DATA: topo, NWP_DATA: nwp})
    assert out2.shape == (2, 6, 4, 6, 6) and torch.equal(out2[:, 4:], nwp[0].float())


def test_cloudgan_surface_config_and_state_dict_keys():
    """configs/model/cloudgan_convlstm.yaml loads unchanged into CloudGAN; its state_dict keys are the reference's
    (pinned file written by tests/golden/make_golden.py from the reference import)."""
    from satflow_amd.config import instantiate, load_config
    from satflow_amd.models import CloudGAN

    cfg = load_config(os.path.join(GOLDEN, "configs", "cloudgan_convlstm.yaml"))
    m = instantiate(cfg)
    assert isinstance(m, CloudGAN) and m.forecast_steps == 24 and m.lambda_l1 == 1 and m.condition_time
    assert m.generator.hidden_dim == 32 and m.discriminator.model[0].weight.shape == (32, 12, 4, 4)
    opts, scheds = m.configure_optimizers()
    assert len(opts) == 2 and all(isinstance(o, torch.optim.Adam) and o.defaults["betas"] == (0.5, 0.999) for o in opts) and len(scheds) == 2
    small = CloudGAN(forecast_steps=2, input_channels=3, num_filters=8, generator_model="convlstm", discriminator_model="basic",
                     channels_per_timestep=3, condition_time=True)
    want = open(os.path.join(GOLDEN, "cloudgan_state_dict_keys.txt")).read().split()
    assert list(small.state_dict().keys()) == want
    with pytest.raises(NotImplementedError):
        CloudGAN(generator_model="runet")


def test_sanitizer_harness_covers_every_entry_point():
    """tests/sanitize/harness.c (run under host ASan + UBSan by tools/sanitize_host.sh; log in profiles/) must exercise the argument
    validation of EVERY int-returning sf_* entry of the header; SF_RUN_SANITIZER=1 re-runs the sanitizer build here (about a minute)."""
    import subprocess

    from satflow_amd import _hip

    src = open(os.path.join(ROOT, "tests", "sanitize", "harness.c")).read()
    entries = [n for n, (res, _) in _hip.PROTOTYPES.items() if res is not None and n not in ("sf_abi_version", "sf_last_error_string") and not n.endswith(("_bytes", "_elems", "_floats", "_tiles", "_supported"))]
    missing = [n for n in entries if f"REFUSED({n}(" not in src]
    assert not missing, f"entry points without a refusal case in the sanitizer harness: {missing}"
    log = open(os.path.join(ROOT, "profiles", "r05_host_asan_ubsan.log")).read()
    assert "0 not refused" in log and "exit code 0" in log and "ERROR: AddressSanitizer" not in log and "runtime error" not in log
    if os.environ.get("SF_RUN_SANITIZER"):
        r = subprocess.run(["bash", os.path.join(ROOT, "tools", "sanitize_host.sh"), "/tmp/sf_host_sanitize.log"], capture_output=True, text=True, timeout=1200)
        assert "0 not refused" in r.stdout and "exit code 0" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_weight_grad_batch_protocol():
    """functional.WeightGradBatch (the once-per-sequence weight gradient of a recurrent cell's state convolution): the application whose
    backward completes the set emits the gradient of ALL applications, a second backward through a retained graph works again, and a
    backward pass that ends with applications missing raises instead of dropping the gradient.  Host logic only - a stand-in autograd
    function with the protocol of ``_ConvFn`` (register in forward; add / take in backward)."""
    from satflow_amd.functional import WeightGradBatch

    class Scale(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w, batch):
            ctx.batch = batch
            batch.register()
            ctx.save_for_backward(x, w)
            return x * w

        @staticmethod
        def backward(ctx, g):
            x, w = ctx.saved_tensors
            if not ctx.batch.add(x, g):
                return g * w, None, None
            xs, gs = ctx.batch.take()
            return g * w, (xs * gs).sum().reshape(1), None

    w = torch.tensor([1.5], requires_grad=True)
    h0 = torch.arange(3.0)
    batch, x, outs = WeightGradBatch(), h0, []
    for _ in range(4):
        x = Scale.apply(x, w, batch)
        outs.append(x)
    loss = sum(o.sum() for o in outs)
    loss.backward(retain_graph=True)
    wr, xr, lr = torch.tensor([1.5], requires_grad=True), h0, 0
    for _ in range(4):
        xr = xr * wr
        lr = lr + xr.sum()
    lr.backward()
    assert torch.allclose(w.grad, wr.grad)
    w.grad = None
    loss.backward()
    assert torch.allclose(w.grad, wr.grad)
    batch, x, outs = WeightGradBatch(), h0, []
    for _ in range(3):
        x = Scale.apply(x, w, batch)
        outs.append(x)
    with pytest.raises(RuntimeError, match="WeightGradBatch"):
        outs[1].sum().backward()


def test_conv4x4_weight_embeddings_on_cpu():
    """The host-side algebra of functional_gan.conv4x4_as_3x3, checked with torch's own convolutions on the CPU (no kernel involved):
    * stride 2: pad by 1, fold 2x2 pixel blocks into channels, a 3x3 'same' convolution with ``regroup4x4s2``'s weight, first h/2 x w/2 outputs;
    * stride 1: the 4x4 kernel as a 5x5 kernel with a zero first row / column, last output row / column dropped."""
    import torch.nn.functional as TF

    from satflow_amd import functional_gan as FG

    g = torch.Generator().manual_seed(3)
    n, C, O, h, w, cp = 2, 5, 7, 8, 10, 8
    x, W, b = torch.randn(n, C, h, w, generator=g), torch.randn(O, C, 4, 4, generator=g), torch.randn(O, generator=g)
    ref = TF.conv2d(x, W, b, stride=2, padding=1)
    xp = TF.pad(TF.pad(x.permute(0, 2, 3, 1), (0, cp - C)), (0, 0, 1, 1, 1, 1))          # NHWC, lanes padded, border of 1
    H2, W2 = h // 2 + 1, w // 2 + 1
    ys = torch.cat([xp[:, dy:dy + 2 * H2:2, dx:dx + 2 * W2:2, :] for dy in (0, 1) for dx in (0, 1)], -1)   # what sf_pad_s2d_fwd writes
    out = TF.conv2d(ys.permute(0, 3, 1, 2), FG.regroup4x4s2(W, cp), b, padding=1)[:, :, : h // 2, : w // 2]
    assert torch.allclose(out, ref, rtol=1e-5, atol=1e-5)
    # round 5: the unpadded form - space_to_depth2 of the input itself, the 3x3 convolution's own zero padding is the strided convolution's
    xq = TF.pad(x.permute(0, 2, 3, 1), (0, cp - C))
    xs = torch.cat([xq[:, dy::2, dx::2, :] for dy in (0, 1) for dx in (0, 1)], -1)                         # what sf_space_to_depth2 writes
    out2 = TF.conv2d(xs.permute(0, 3, 1, 2), FG.regroup4x4s2_same(W, cp), b, padding=1)
    assert out2.shape == ref.shape and torch.allclose(out2, ref, rtol=1e-5, atol=1e-5)
    ref1 = TF.conv2d(x, W, b, stride=1, padding=1)
    out1 = TF.conv2d(x, TF.pad(W, (1, 0, 1, 0)), b, padding=2)[:, :, : h - 1, : w - 1]
    assert torch.allclose(out1, ref1, rtol=1e-5, atol=1e-5)


def test_hot_kernels_do_not_spill():
    """Register spills of the hot kernel instantiations: every kernel of the listed sources is compiled to gfx950 ISA (hipcc -S,
    device only) and its `.amdhsa_private_segment_fixed_size` (scratch bytes per lane) is compared with a budget: 0 for everything that is not listed,
    the recorded value for the known ones - all of which keep their scratch accesses OUT of the K loop (the spilled values are per-chunk staging
    offsets / post-loop flush operands).  A change that makes any instantiation spill more fails here, on the CPU."""
    import concurrent.futures as cf
    import re
    import subprocess
    import tempfile

    csrc = os.path.join(ROOT, "satflow_amd", "csrc")
    budgets = {  # substring of the mangled kernel name -> scratch bytes allowed
        "conv3x3_wgrad_bf16_dma.hip": {"wgrad_bf16_dma_kernelILb1ELb1ELi0E": 32,       # partial-slab addresses of the group-boundary flush (off the tile loop's path)
                                       "wgrad_bf16_dma_kernelILb1ELb1ELi1E": 104,      # 2:4-sparse: + the prologue's fragment bases; the smfmac loop touches no scratch
                                       "wgrad_bf16_dma_kernelILb1ELb1ELi2E": 72,       # ... operand from the pooled gradient
                                       "wgrad_pooled8_kernel": 184},                   # ... on 8-row tiles: prologue / flush only - the tile loop has NO scratch
                                                                                       # access (a reload there is a vector-memory operation and breaks the counted waits)
        "conv3x3_bf16.hip": {"conv3x3_bf16_kernelILi8ELi5E": 40},                        # NF = 5: 160 accumulators, staging offsets reloaded once per chunk
        "conv3x3_bf16_persist.hip": {},
        "conv3x3_bf16_persist4.hip": {},
        "convgru_seq.hip": {"convgru_seq_fwd_kernelILi2ELb1ELb0E": 64},                  # the one-workgroup-per-map fallback (n > CUs / 2)
        "conv3x3_wgrad_bf16.hip": {},
        "conv3x3_f32e.hip": {"conv3x3_bf16_kernelILi8ELi5E": 40},                        # the same kernels as the SF_F32E mode (three fp16 products per fp32 product)
        "conv3x3_wgrad_f32e.hip": {},
    }

    isa = {}

    def scratch(src):
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "k.s")
            r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", os.path.join(csrc, src), "-o", out],
                               capture_output=True, text=True)
            assert r.returncode == 0, r.stderr[-2000:]
            txt = open(out).read()
        if src == "conv3x3_wgrad_bf16_dma.hip":
            isa["txt"] = txt
        return src, {m.group(1): int(m.group(2)) for m in re.finditer(r"\.amdhsa_kernel (\S+).*?\.amdhsa_private_segment_fixed_size (\d+)", txt, re.S)}

    with cf.ThreadPoolExecutor(max_workers=6) as ex:
        results = dict(ex.map(scratch, budgets))
    bad = []
    for src, kernels in results.items():
        assert kernels, src
        for name, sc in kernels.items():
            allowed = max([v for k, v in budgets[src].items() if k in name], default=0)
            if sc > allowed:
                bad.append(f"{src}: {name} uses {sc} bytes of scratch per lane (budget {allowed})")
    assert not bad, "\n".join(bad)
    # PLACEMENT, not only bytes.  The weight-gradient kernels that spill wait for their LDS-DMA pieces BY COUNT; a scratch access
    # is a vector-memory operation, so one between two matrix instructions of the tile loop puts `s_waitcnt vmcnt(0)` there and drains the ring (DESIGN.md
    # section 7, round 5).  Checked on the ISA: (a) no basic block that issues a matrix instruction touches scratch, in any of the spilling instantiations;
    # (b) in the kernel the benchmark runs (wgrad_pooled8_kernel) the WHOLE span from its first to its last matrix instruction - every block of the tile
    # loop - holds neither a scratch access nor a `vmcnt(0)`.  (The dense kernels' loop blocks do hold one `vmcnt(0)` each: the rendezvous in front of
    # their `s_barrier`, by design.)
    mat = re.compile(r"v_s?mfmac?_")
    checked = 0
    for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:", isa["txt"], re.S | re.M):
        name, body = m.group(1), m.group(2)
        if not ("wgrad_pooled8_kernel" in name or "wgrad_bf16_dma_kernelILb1ELb1E" in name):
            continue
        checked += 1
        lines = [ln.strip() for ln in body.splitlines() if ln.strip() and not ln.strip().startswith(";")]
        block, has_mat, has_scr = "entry", False, False
        for ln in lines + [".LBBend_0:"]:
            if re.match(r"\.LBB\w+:", ln):
                assert not (has_mat and has_scr), f"{name}: basic block {block} issues matrix instructions AND touches scratch"
                block, has_mat, has_scr = ln, False, False
            has_mat |= bool(mat.match(ln))
            has_scr |= ln.startswith("scratch_")
        if "wgrad_pooled8_kernel" in name:
            idx = [i for i, ln in enumerate(lines) if mat.match(ln)]
            span = lines[idx[0]:idx[-1] + 1]
            offenders = [ln for ln in span if ln.startswith("scratch_") or re.match(r"s_waitcnt.*vmcnt\(0\)", ln)]
            assert len(idx) >= 18 and not offenders, f"{name}: {offenders[:4]} inside the tile loop"
    assert checked == 4, checked


def test_the_f32e_cases_really_run_in_f32e():
    """conftest.FP32_GATED runs every fp32-gated GPU test in "f32" and "f32e".  The mode must actually be SET for test functions that do not take the
    fixture as an argument (round 6: it was not, for a while): a throw-away test module collected with this conftest reports the mode it ran in."""
    import subprocess
    import sys
    import tempfile
    import textwrap

    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "conftest.py"), "w").write(open(os.path.join(ROOT, "tests", "conftest.py")).read().replace(
            '"test_stlstm_gpu.py": (', '"test_probe_modes.py": ("test_reports_mode",), "test_stlstm_gpu.py": ('))
        open(os.path.join(d, "test_probe_modes.py"), "w").write(textwrap.dedent("""
            import satflow_amd
            def test_reports_mode():
                print("RAN_IN", satflow_amd.compute_dtype_name())
            def test_ungated():
                print("UNGATED_IN", satflow_amd.compute_dtype_name())
        """))
        r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-s", "-p", "no:cacheprovider", d], capture_output=True, text=True, cwd=d,
                           env=dict(os.environ, PYTHONPATH=ROOT))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        assert sorted(re.findall(r"RAN_IN (\w+)", r.stdout)) == ["f32", "f32e"], r.stdout[-1500:]
        assert re.findall(r"UNGATED_IN (\w+)", r.stdout) == ["f32"]


def test_no_barrier_is_reached_from_an_lds_write_without_a_wait():
    """ISA of every kernel (hipcc -S for gfx950, tools/scan_barrier_waits.py): walking backwards from each s_barrier through straight-line code there is an
    `s_waitcnt lgkmcnt(0)` before any ds_write.  hipcc drops the LDS wait of __syncthreads()'s release fence when it believes it can; where it did (the
    SF_SPLIT3 build of the convolution, round 6) a wave of another SIMD read a staged piece before the write had landed - once in ~1000 launches.  The kernels
    now wait explicitly in front of those barriers; this test keeps it that way for code the compiler touches later."""
    import concurrent.futures as cf
    import glob
    import importlib.util

    spec = importlib.util.spec_from_file_location("scan_barrier_waits", os.path.join(ROOT, "tools", "scan_barrier_waits.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    files = sorted(glob.glob(os.path.join(ROOT, "satflow_amd", "csrc", "*.hip")))
    with cf.ThreadPoolExecutor(max_workers=7) as ex:
        results = dict(zip(files, ex.map(mod.scan, files)))
    bad = []
    for f, r in results.items():
        assert r is not None, f"{f} did not compile"
        total, hits = r
        for k, v in hits.items():
            bad.append(f"{os.path.basename(f)}: {k[:100]}: {v} barrier(s) behind an un-waited LDS write")
    assert not bad, "\n".join(bad[:20])

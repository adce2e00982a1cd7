"""Scratch resets inside captured hipGraphs.  Round 5 found that a memset node (hipMemsetAsync captured into a graph) writes its value correctly on
the FIRST launch of the graph and another pattern from the second launch on (ROCm 7.2, gfx950: 256 zero bytes came back as 0x3f800000 words).  The
library had eight such resets - the zero page of the two-source weight gradient, the loss sums, the ConvGRU mailboxes - and every replay test replayed
ONCE.  They are kernels now (``sf_fill_async``, csrc/error.hip); these tests replay three times, with the scratch dirtied in between, and compare every
replay with the eager result.
"""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _capture(fn):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    return g, out


def test_two_source_weight_gradient_replays(device):
    """sf_conv3x3_bwd_weight on two bf16 sources whose first is not a whole number of 64-channel tiles (the ConvLSTM cell: x 16 + h 32 lanes) fetches its
    out-of-image halo pieces from a zero page at the head of the workspace, reset by every call."""
    import satflow_amd
    from satflow_amd import _hip
    from satflow_amd._hip import T, check, lib, stream_ptr
    from satflow_amd.functional import ConvEngine

    satflow_amd.set_compute_dtype("bf16a")
    try:
        B, H, W = 2, 64, 64
        eng = ConvEngine([16, 32], 128)
        g_ = torch.Generator(device="cpu").manual_seed(1)
        x = torch.randn(B, H, W, 16, generator=g_).to(device).bfloat16()
        hh = torch.randn(B, H, W, 32, generator=g_).to(device).bfloat16()
        dz = torch.randn(B, H, W, 128, generator=g_).to(device).bfloat16()
        nbytes = lib().sf_conv3x3_bwd_weight_workspace_bytes(128, 48, B, H, W)
        ws = torch.zeros(nbytes // 4 + 1, device=device)
        dw, db = torch.empty(128, 48, 3, 3, device=device), torch.empty(128, device=device)
        nmap, kmap = eng.wgrad_map.tables(device)

        def run():
            check(lib().sf_conv3x3_bwd_weight(T(x), T(hh), T(dz), B, H, W, nmap.data_ptr(), kmap.data_ptr(), 128, 48, dw.data_ptr(), db.data_ptr(), 0,
                                              ws.data_ptr(), nbytes, _hip.compute_dtype(), stream_ptr()), "sf_conv3x3_bwd_weight")

        run()
        torch.cuda.synchronize()
        ref_w, ref_b = dw.clone(), db.clone()
        assert torch.isfinite(ref_w).all() and ref_w.abs().max() > 1
        graph, _ = _capture(run)
        for r in range(3):
            ws[:64] = 5.0        # whatever the previous user of the block left there
            dw.zero_()
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(dw, ref_w) and torch.equal(db, ref_b), f"replay {r}"
            assert int((ws[:64] != 0).sum()) == 0, f"zero page after replay {r}"
    finally:
        satflow_amd.set_compute_dtype("f32")


def test_loss_sums_replay(device):
    """sf_mse_loss / sf_pair_loss / sf_gan_loss accumulate into double sums that every call resets."""
    from satflow_amd import functional as F

    g_ = torch.Generator(device="cpu").manual_seed(2)
    pred = torch.randn(2, 6, 4, 32, 32, generator=g_).to(device)
    target = torch.randn(2, 6, 4, 32, 32, generator=g_).to(device)
    logits = torch.randn(4, 16, 16, 8, generator=g_).to(device)
    other = torch.randn(4, 16, 16, 8, generator=g_).to(device)

    def run():
        return (F.mse_loss_with_frames(pred, target)[0], F.mse_loss_with_frames(pred, target)[1], F.l1_loss_groups(logits, other, 2, 3)[0],
                F.bce_logits_groups(logits, 1.0, 0.0, 2)[0], F.bce_logits_groups(logits, 1.0, 0.0, 2, mode="lsgan")[1])

    with torch.no_grad():
        ref = [t.clone() for t in run()]
        torch.cuda.synchronize()
        assert abs(float(ref[0]) - float(((pred - target) ** 2).mean())) < 1e-5
        graph, outs = _capture(run)
        for r in range(3):
            graph.replay()
            torch.cuda.synchronize()
            for k, (o, e) in enumerate(zip(outs, ref)):
                assert torch.equal(o, e), f"replay {r}, output {k}: {o} vs {e}"


def test_what_a_captured_memset_does(device, record_property):
    """The observation itself, kept as a record (no assertion on the runtime's behaviour - a fixed runtime passes too): hipMemsetAsync of 256 zero bytes
    captured into a graph, buffer dirtied before every launch."""
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
    hip.hipMemsetAsync.restype = ctypes.c_int
    buf = torch.full((64,), 5.0, device=device)

    def run():
        assert hip.hipMemsetAsync(buf.data_ptr(), 0, 256, torch.cuda.current_stream().cuda_stream) == 0

    graph, _ = _capture(run)
    seen = []
    for _ in range(3):
        buf.fill_(5.0)
        graph.replay()
        torch.cuda.synchronize()
        seen.append(sorted(set(buf.view(torch.int32).tolist())))
    record_property("memset_node_words_per_replay", str(seen))
    print("captured hipMemsetAsync(0, 256 B): 32-bit words in the buffer after replay 0, 1, 2:", [[hex(w & 0xFFFFFFFF) for w in s] for s in seen])

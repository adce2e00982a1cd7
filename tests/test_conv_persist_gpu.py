"""GPU: the persistent bf16 convolution kernel (conv3x3_bf16_persist.hip: one workgroup per CU walks several (tile, N block) items,
the next item's first K chunk requested during the current item's last) against the one-item-per-workgroup kernel it replaces for
large launches.  Same arithmetic, same K order, same epilogues: a launch over N images (persistent: >= 1024 tiles) must equal, BIT FOR
BIT, two launches over the halves (below the threshold: one-item kernel); the per-tile BatchNorm statistics to fp32 noise (LDS atomics)."""
import pytest
import torch

import satflow_amd

pytestmark = pytest.mark.gpu


@pytest.fixture()
def bf16_mode():
    satflow_amd.set_compute_dtype("bf16")
    yield
    satflow_amd.set_compute_dtype("f32")


@pytest.mark.parametrize("n,cin,cout,H,W", [(520, 32, 64, 32, 32), (400, 160, 256, 40, 24), (600, 16, 96, 17, 33), (2040, 256, 128, 32, 16)])
@pytest.mark.parametrize("stats", [False, True])
def test_persistent_kernel_matches_one_item_kernel(device, bf16_mode, n, cin, cout, H, W, stats):
    from satflow_amd import kernels as K
    from satflow_amd._hip import NULL, T, cpad, lib
    from satflow_amd.functional import ConvEngine

    g = torch.Generator().manual_seed(n + cin)
    eng = ConvEngine([cin], cout)
    w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).to(device)
    b = torch.randn(cout, generator=g).to(device)
    packed, bp = K.pack_weights(w, b, eng.fwd_map, False)
    x = torch.randn(n, H, W, cpad(cin), generator=g).to(device).to(torch.bfloat16)
    tiles = int(lib().sf_conv3x3_stats_tiles(H, W))
    assert tiles * n >= 1024 and tiles * (n // 2) < 1024  # whole launch persistent, halves not

    def run(xs):
        m = xs.shape[0]
        y = torch.full((m, H, W, eng.coutp), float("nan"), device=device).to(torch.bfloat16)
        st = torch.full((m * tiles, eng.fwd_map.Np, 2), float("nan"), device=device) if stats else None
        K.conv3x3(T(xs), NULL, m, H, W, packed, bp, eng.fwd_map, T(y), stats=st)
        return y, st

    y, st = run(x)
    h = n // 2
    ya, sa = run(x[:h].contiguous())
    yb, sb = run(x[h:].contiguous())
    torch.cuda.synchronize()
    assert torch.isfinite(y[..., :cout].float()).all()
    assert torch.equal(y[:h, ..., :cout], ya[..., :cout]) and torch.equal(y[h:, ..., :cout], yb[..., :cout])
    if stats:  # per-tile sums: the lanes add into LDS with atomics (order varies run to run, in either kernel): fp32 noise only
        ref = torch.cat((sa, sb))[:, :cout]
        assert torch.allclose(st[:, :cout], ref, rtol=1e-5, atol=1e-4 * float(ref.abs().max()))


@pytest.mark.parametrize("dtype,out_dtype", [(torch.bfloat16, torch.bfloat16), (torch.bfloat16, torch.float32), (torch.float32, torch.float32)])
@pytest.mark.parametrize("perm,drop", [(None, None), ((3, 2), None), ((3, 2), (0.2, 0.1, 6 * 4 * 3 * 32))])  # period: 6 pooled images per timestep
def test_maxpool_with_recorded_routing_matches_recomputed_argmax(device, monkeypatch, dtype, out_dtype, perm, drop):
    """sf_maxpool2_route_fwd/bwd (the forward pass records which window element every channel took; the backward pass reads that
    instead of the input) against the argmax-recomputing kernels: same pooled tensor, same gradient, bit for bit - including ties
    (first maximum in row-major window order), the image permutation and the fused dropouts."""
    from satflow_amd import functional as F

    g = torch.Generator().manual_seed(21)
    n, h, w, c = 12, 8, 6, 32
    x = torch.randint(-3, 4, (n, h, w, c), generator=g).float()  # small integers: many exact ties
    x = x.to(device).to(dtype)
    gy = torch.randn(n, h // 2, w // 2, c, generator=g).to(device).to(out_dtype)

    def run(recompute):
        if recompute:
            monkeypatch.setenv("SF_POOL_RECOMPUTE", "1")
        else:
            monkeypatch.delenv("SF_POOL_RECOMPUTE", raising=False)
        torch.manual_seed(5)  # the dropout seeds come from the torch RNG
        xi = x.clone().requires_grad_(True)
        y = F.maxpool2(xi, perm, out_dtype=out_dtype, dropout=drop)
        y.backward(gy)
        return y.detach(), xi.grad

    y1, g1 = run(False)
    y0, g0 = run(True)
    assert torch.equal(y1, y0) and torch.equal(g1, g0)
    assert float(g1.float().abs().sum()) > 0


# ----------------------------------------------------------------------------------------------------------------------
# The one-wave-per-SIMD persistent kernel (conv3x3_bf16_persist4.hip): 128-channel N blocks, >= 3 K chunks, no per-channel bias
# (plain, or the folded BatchNorm's grouped weights + border-class bias table).  Bit-identical to the one-item kernel.
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n,cin,cout,H,W", [(520, 48, 128, 32, 32), (400, 160, 256, 40, 24), (1040, 256, 256, 32, 16), (500, 64, 384, 33, 17), (2304, 256, 256, 32, 32)])
def test_one_wave_per_simd_kernel_matches_one_item_kernel(device, bf16_mode, n, cin, cout, H, W):
    """No bias: a launch over n images (>= 1024 tiles: persistent, one wave per SIMD) == two launches over the halves (one-item kernel)."""
    from satflow_amd import kernels as K
    from satflow_amd._hip import NULL, T, cpad, lib
    from satflow_amd.functional import ConvEngine

    g = torch.Generator().manual_seed(n + cin + 1)
    eng = ConvEngine([cin], cout)
    assert eng.fwd_map.nf == 4
    w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).to(device)
    packed, _ = K.pack_weights(w, None, eng.fwd_map, False)
    x = torch.randn(n, H, W, cpad(cin), generator=g).to(device).to(torch.bfloat16)
    tiles = int(lib().sf_conv3x3_stats_tiles(H, W))
    assert tiles * n >= 1024 and (n > 2000 or tiles * (n // 2) < 1024)

    def run(xs):
        m = xs.shape[0]
        y = torch.full((m, H, W, eng.coutp), float("nan"), device=device).to(torch.bfloat16)
        K.conv3x3(T(xs), NULL, m, H, W, packed, None, eng.fwd_map, T(y))
        return y

    y = run(x)
    parts = 2 if n <= 2000 else 6   # every part below the persistent kernels' threshold
    step = n // parts
    torch.cuda.synchronize()
    assert torch.isfinite(y[..., :cout].float()).all()
    for i in range(parts):
        yi = run(x[i * step:(i + 1) * step].contiguous())
        assert torch.equal(y[i * step:(i + 1) * step, ..., :cout], yi[..., :cout]), f"part {i} differs"
    assert float(y[..., :cout].float().abs().max()) > 0.1


@pytest.mark.parametrize("n,groups,cin,cout,H,W", [(1040, 4, 160, 256, 32, 32), (2304, 12, 256, 256, 32, 32), (1080, 6, 96, 128, 40, 24)])
def test_one_wave_per_simd_kernel_folded_batchnorm(device, bf16_mode, n, groups, cin, cout, H, W):
    """Grouped weights + border-class bias table (sf_conv3x3_fwd_folded): the persistent launch over all groups == one launch per group
    (each below the persistent threshold: the one-item kernel), bit for bit."""
    from satflow_amd import kernels as K
    from satflow_amd._hip import T, cpad, lib
    from satflow_amd.functional import ConvEngine

    g = torch.Generator().manual_seed(n + cin + groups)
    eng = ConvEngine([cin], cout)
    gm = eng.fwd_map
    assert gm.nf == 4
    w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).to(device)
    b = torch.randn(cout, generator=g).to(device)
    scale = (0.5 + torch.rand(groups, gm.Kp, generator=g)).to(device)
    shift = torch.randn(groups, gm.Kp, generator=g).to(device)
    packed, tab = K.conv3x3_fold_pack(w, b, gm, scale, shift)
    x = torch.randn(n, H, W, cpad(cin), generator=g).to(device).to(torch.bfloat16)
    tiles = int(lib().sf_conv3x3_stats_tiles(H, W))
    ipg = n // groups
    assert tiles * n >= 1024 and tiles * ipg < 1024
    y = torch.full((n, H, W, eng.coutp), float("nan"), device=device).to(torch.bfloat16)
    K.conv3x3_folded(T(x), n, H, W, packed, tab, gm, T(y))
    torch.cuda.synchronize()
    assert torch.isfinite(y[..., :cout].float()).all()
    img = gm.Np * gm.Kp * 9
    for gi in range(groups):
        yg = torch.full((ipg, H, W, eng.coutp), float("nan"), device=device).to(torch.bfloat16)
        K.conv3x3_folded(T(x[gi * ipg:(gi + 1) * ipg].contiguous()), ipg, H, W, packed[gi * img:(gi + 1) * img], tab[gi:gi + 1].contiguous(), gm, T(yg))
        assert torch.equal(y[gi * ipg:(gi + 1) * ipg, ..., :cout], yg[..., :cout]), f"group {gi} differs"


@pytest.mark.parametrize("n,groups,cin,cout,H,W", [(1040, 4, 160, 256, 32, 32), (2304, 12, 256, 256, 32, 32), (1080, 6, 96, 128, 40, 24)])
def test_one_wave_per_simd_kernel_statistics(device, bf16_mode, n, groups, cin, cout, H, W):
    """The statistics-emitting launch of the one-wave-per-SIMD kernel (sf_conv3x3_fwd_folded with a stats buffer: the DownSampler's conv2 /
    conv3 in bf16a mode): the outputs are bit-identical to the launch without statistics, and every tile's (sum, sum of squares) row equals
    the sums of the STORED bf16 values over the tile's valid pixels (float64 on the device; fp32 summation order differs: 1e-5 of the
    tile's sum of squares), ragged tiles included (40 x 24)."""
    from satflow_amd import kernels as K
    from satflow_amd._hip import T, cpad, lib
    from satflow_amd.functional import ConvEngine

    g = torch.Generator().manual_seed(7 * n + cin + groups)
    eng = ConvEngine([cin], cout)
    gm = eng.fwd_map
    assert gm.nf == 4
    w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).to(device)
    b = torch.randn(cout, generator=g).to(device)
    scale = (0.5 + torch.rand(groups, gm.Kp, generator=g)).to(device)
    shift = torch.randn(groups, gm.Kp, generator=g).to(device)
    packed, tab = K.conv3x3_fold_pack(w, b, gm, scale, shift)
    x = torch.randn(n, H, W, cpad(cin), generator=g).to(device).to(torch.bfloat16)
    tiles = int(lib().sf_conv3x3_stats_tiles(H, W))
    y0 = torch.full((n, H, W, eng.coutp), float("nan"), device=device).to(torch.bfloat16)
    K.conv3x3_folded(T(x), n, H, W, packed, tab, gm, T(y0))
    y1 = torch.full_like(y0, float("nan"))
    st = torch.full((n * tiles, gm.Np, 2), float("nan"), device=device)
    K.conv3x3_folded(T(x), n, H, W, packed, tab, gm, T(y1), stats=st)
    torch.cuda.synchronize()
    assert torch.equal(y0[..., :cout], y1[..., :cout])
    # tiles: 32 rows x 16 columns, row-major over (tile row, tile column) inside an image
    tx, ty = (W + 15) // 16, (H + 31) // 32
    assert tiles == tx * ty
    yd = y1[..., :cout].double()
    ref = torch.zeros(n, ty, tx, cout, 2, dtype=torch.float64, device=device)
    for j in range(ty):
        for i in range(tx):
            blk = yd[:, 32 * j:32 * j + 32, 16 * i:16 * i + 16]
            ref[:, j, i, :, 0] = blk.sum((1, 2))
            ref[:, j, i, :, 1] = (blk * blk).sum((1, 2))
    got = st.view(n, ty, tx, gm.Np, 2)[..., :cout, :].double()
    assert torch.isfinite(got).all()
    scale_sq = ref[..., 1].abs().max().clamp_min(1.0)
    err = (got - ref).abs().max() / scale_sq
    assert float(err) < 1e-5, float(err)


@pytest.mark.parametrize("n,groups,cin,cout,H,W", [(1040, 4, 256, 160, 32, 32), (2304, 12, 256, 256, 32, 32), (1080, 6, 128, 96, 40, 24), (1056, 6, 384, 64, 33, 17),
                                                   (1040, 4, 160, 256, 32, 32), (1080, 6, 224, 96, 40, 24), (1056, 6, 96, 64, 33, 17)])
def test_one_wave_per_simd_kernel_batchnorm_backward(device, bf16_mode, n, groups, cin, cout, H, W):
    """sf_conv3x3_bwd_data_bn on the one-wave-per-SIMD kernel (MODE 2: dx = A * conv^T(dout, W) + B * x + K in the epilogue, x read at the
    store's offsets and permuted back to the accumulator layout): the persistent launch over all groups == one launch per group (each below
    the persistent threshold: the 8-wave one-item kernel), bit for bit - ragged tiles, one to three N blocks, 4 to 24 K chunks."""
    from satflow_amd import kernels as K
    from satflow_amd._hip import T, cpad, lib
    from satflow_amd.functional import ConvEngine

    g = torch.Generator().manual_seed(3 * n + cin + groups)
    eng = ConvEngine([cin], cout)
    if eng.bwd_map((True,)).nf != 4:
        # input widths that are not whole 128-channel N blocks (160, 224, 96: the planner gives them NF = 5 / 1 / 3 and the 8-wave kernel): forced onto the
        # 128-channel blocks here - the last N block then has one to three valid channel fragments, and MODE 2 reads x in 128-byte lines = fragment PAIRS, so a
        # pair can be half valid (its second fragment is loaded from the pad lanes or the next pixel and never stored) or wholly past the tensor (sentinel)
        eng._bwd_maps[(True,)] = K._finish(K._padded(cin), K._padded(cout), nf=4)
    gm = eng.bwd_map((True,))
    w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).to(device)
    packed_t = eng.packed(w, None, "bwd", (True,))[0]
    gy = torch.randn(n, H, W, eng.coutp, generator=g).to(device).to(torch.bfloat16)
    gy[..., cout:] = 0
    x = torch.randn(n, H, W, cpad(cin), generator=g).to(device).to(torch.bfloat16)
    coef = torch.randn(groups, 3, cpad(cin), generator=g).to(device).contiguous()
    tiles = int(lib().sf_conv3x3_stats_tiles(H, W))
    ipg = n // groups
    assert tiles * n >= 1024 and tiles * ipg < 1024 and gm.nf == 4
    dx = torch.full((n, H, W, cpad(cin)), float("nan"), device=device).to(torch.bfloat16)
    K.conv3x3_bwd_data_bn(T(gy), n, H, W, packed_t, gm, T(x), coef, T(dx))
    torch.cuda.synchronize()
    assert torch.isfinite(dx[..., :cin].float()).all()
    for gi in range(groups):
        sl = slice(gi * ipg, (gi + 1) * ipg)
        dg = torch.full((ipg, H, W, cpad(cin)), float("nan"), device=device).to(torch.bfloat16)
        K.conv3x3_bwd_data_bn(T(gy[sl].contiguous()), ipg, H, W, packed_t, gm, T(x[sl].contiguous()), coef[gi:gi + 1].contiguous(), T(dg))
        assert torch.equal(dx[sl][..., :cin], dg[..., :cin]), f"group {gi} differs"
    assert float(dx[..., :cin].float().abs().max()) > 0.1


@pytest.mark.parametrize("n,cin,cout,H,W,groups,perm,drop", [
    (512, 256, 256, 32, 32, 4, (4, 16), (0.2, 0.1)),      # MetNet's conv4 shape family: two N blocks, image permutation, both dropouts
    (1024, 48, 128, 32, 16, 2, None, None),               # the minimum K (3 chunks), one tile per image, one N block
    (256, 64, 384, 34, 30, 2, (2, 8), (0.0, 0.3)),        # ragged tiles (34 x 30), three N blocks
    (288, 160, 128, 64, 64, 12, (12, 3), (0.25, 0.0)),    # 12 BatchNorm groups, several tile rows per image
])
def test_pooling_in_the_convolution_epilogue(device, bf16_mode, n, cin, cout, H, W, groups, perm, drop):
    """sf_conv3x3_fwd_folded_pool (the one-wave-per-SIMD kernel with window-major pixel fragments and the 2x2 max-pooling in its epilogue) against the
    two launches it replaces, sf_conv3x3_fwd_folded + sf_maxpool2_route_fwd: the pooled tensor AND the routing record bit for bit (ties between
    values that differ only below bf16 precision included: the epilogue pools the ROUNDED values), with the image permutation, and with MetNet's two
    dropout masks applied by sf_dropout2_bf16 against the masks fused into the pooling kernel."""
    from satflow_amd import kernels as K
    from satflow_amd._hip import T, cpad, lib
    from satflow_amd.functional import ConvEngine

    g = torch.Generator().manual_seed(n + cin + cout)
    eng = ConvEngine([cin], cout)
    gm = eng.fwd_map
    w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.05).to(device)
    b = torch.randn(cout, generator=g).to(device)
    scale = (0.5 + torch.rand(groups, gm.Kp, generator=g)).to(device)
    shift = torch.randn(groups, gm.Kp, generator=g).to(device)
    packed, tab = K.conv3x3_fold_pack(w, b, gm, scale, shift)
    # coarse values: many windows hold values that tie after the bf16 rounding
    x = (torch.randint(-4, 5, (n, H, W, cpad(cin)), generator=g).float() * 0.25).to(device).to(torch.bfloat16)
    x[..., cin:] = 0
    assert K.conv3x3_folded_pool_supported(n, H, W, gm, eng.coutp, groups), "the pooled-epilogue kernel must take this shape"
    y = torch.full((n, H, W, eng.coutp), float("nan"), device=device).to(torch.bfloat16)
    K.conv3x3_folded(T(x), n, H, W, packed, tab, gm, T(y))
    d5 = None
    if drop is not None:
        per = (n // (perm[1] if perm else 1)) * (H // 2) * (W // 2) * eng.coutp
        d5 = (drop[0], drop[1], per, 1234567, 7654321)
    ref, ref_route = K.maxpool2_route_fwd(y, perm, torch.bfloat16, d5)
    got, route = K.conv3x3_folded_pool(T(x), n, H, W, packed, tab, gm, eng.coutp, perm, d5, device)
    torch.cuda.synchronize()
    assert torch.isfinite(got[..., :cout].float()).all()
    assert torch.equal(got[..., :cout], ref[..., :cout]), float((got[..., :cout].float() - ref[..., :cout].float()).abs().max())
    # routing: 2 bits per channel, 16-bit word per channel octet; only the real channels' words are defined
    words = cout // 8
    assert torch.equal(route[..., :words], ref_route[..., :words]), int((route[..., :words] != ref_route[..., :words]).sum())
    ties = int(((y[:, 0::2, 0::2, :cout] == y[:, 0::2, 1::2, :cout]) & (y[:, 0::2, 0::2, :cout] >= y[:, 1::2, 0::2, :cout])
                & (y[:, 0::2, 0::2, :cout] >= y[:, 1::2, 1::2, :cout])).sum())
    print(f"   {ties} windows whose maximum is tied in the top row")
    # and the backward of the pair through the recorded routing: the gradient lands where the unfused path puts it
    gy = torch.randn(got.shape, generator=g).to(device).to(torch.bfloat16)
    d1 = K.maxpool2_route_bwd(route, gy, tuple(y.shape), torch.bfloat16, perm, d5)
    d2 = K.maxpool2_route_bwd(ref_route, gy, tuple(y.shape), torch.bfloat16, perm, d5)
    assert torch.equal(d1[..., :cout], d2[..., :cout])

"""Typed Python entry points over the C ABI (one function per symbol of include/satflow_hip.h).

Everything here takes/returns torch CUDA tensors in the kernels' layout: NHWC, fp32, channel
count padded to ``SF_CPAD``.  PyTorch only owns the memory and the stream; all arithmetic is in
libsatflow_hip.so.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import os

import torch

from . import _hip
from ._hip import NULL, SF_EPI_LINEAR, SF_EPI_SIGMOID, SF_F32, T, check, cpad, lib, stream_ptr, sfTensor

Tensor = torch.Tensor


# ----------------------------------------------------------------------------------------------
# GEMM index maps: how padded N / K lanes of a kernel map onto rows / columns of the reference's
# OIHW weight.  Built on the host once per module, kept as tiny device tables.
# ----------------------------------------------------------------------------------------------
def choose_nf(lanes: int) -> int:
    """N fragments (32 lanes each) per workgroup: fewest total fragments, then fewest blocks."""
    best = None
    for nf in range(1, 6):
        nblk = -(-lanes // (32 * nf))
        key = (nblk * nf, nblk)
        if best is None or key < best[0]:
            best = (key, nf)
    return best[1]


@dataclass
class GemmMap:
    nmap: List[int]  # len Np (multiple of 32*nf)
    kmap: List[int]  # len Kp (multiple of 16)
    nf: int
    n_lanes: int  # lanes actually written by the kernel (out.c)
    _dev: Dict[torch.device, Tuple[Tensor, Tensor]] = field(default_factory=dict, repr=False)

    @property
    def Np(self) -> int:
        return len(self.nmap)

    @property
    def Kp(self) -> int:
        return len(self.kmap)

    def tables(self, device) -> Tuple[Tensor, Tensor]:
        device = torch.device(device)
        if device not in self._dev:
            self._dev[device] = (
                torch.tensor(self.nmap, dtype=torch.int32, device=device),
                torch.tensor(self.kmap, dtype=torch.int32, device=device),
            )
        return self._dev[device]


def _padded(real: int, base: int = 0) -> List[int]:
    return [base + i if i < real else -1 for i in range(cpad(real))]


def _finish(nlanes_map: List[int], kmap: List[int], nf: Optional[int] = None) -> GemmMap:
    lanes = len(nlanes_map)
    nf = nf or choose_nf(lanes)
    Np = -(-lanes // (32 * nf)) * 32 * nf
    return GemmMap(nlanes_map + [-1] * (Np - lanes), kmap, nf, lanes)


def linear_map(cins: Sequence[int], cout: int) -> GemmMap:
    """Plain conv over the channel concatenation of ``cins`` sources."""
    kmap, base = [], 0
    for c in cins:
        kmap += _padded(c, base)
        base += c
    return _finish(_padded(cout), kmap)


def linear_bwd_map(cins: Sequence[int], cout: int, need: Sequence[bool]) -> GemmMap:
    """Input-gradient conv: K = padded out channels, N = padded input lanes of the needed sources."""
    nm, base = [], 0
    for c, nd in zip(cins, need):
        if nd:
            nm += _padded(c, base)
        base += c
    return _finish(nm, _padded(cout))


def lstm_fwd_map(cin: int, hid: int) -> GemmMap:
    """ConvLSTM gate conv: each N block = 4 gates x 32 hidden channels (gate order i,f,o,g,
    reference layers/ConvLSTM.py:48), so one wave holds all four gates of a hidden channel."""
    hidp = cpad(hid)
    nblk = -(-hidp // 32)
    nmap = []
    for nb in range(nblk):
        for g in range(4):
            for j in range(32):
                hc = nb * 32 + j
                nmap.append(g * hid + hc if hc < hid else -1)
    return GemmMap(nmap, _padded(cin) + _padded(hid, cin), 4, nblk * 128)


def lstm_dz_kmap(hid: int) -> List[int]:
    hidp = cpad(hid)
    return [g * hid + j if j < hid else -1 for g in range(4) for j in range(hidp)]


def lstm_bwd_map(cin: int, hid: int, need_dx: bool) -> GemmMap:
    """dz [4*hidp gate-major] -> [dx (if needed) ; dh_prev]."""
    nm = (_padded(cin) if need_dx else []) + _padded(hid, cin)
    return _finish(nm, lstm_dz_kmap(hid))


def lstm_wgrad_map(cin: int, hid: int, h_first: bool = False) -> GemmMap:
    """Weight-gradient index maps: dout lanes = dz (gate-major), K lanes = [x ; h] - or [h ; x] (``h_first``: the order the bf16-storage
    kernel likes when the hidden width is a multiple of its 64-channel tile: every tile then has ONE source, and the narrow x tile at the
    end takes the 256 x 32 slab instead of multiplying three quarters of zeros)."""
    if h_first:
        return GemmMap(lstm_dz_kmap(hid), _padded(hid, cin) + _padded(cin), 0, 4 * cpad(hid))
    return GemmMap(lstm_dz_kmap(hid), _padded(cin) + _padded(hid, cin), 0, 4 * cpad(hid))


# ----------------------------------------------------------------------------------------------
# thin wrappers
# ----------------------------------------------------------------------------------------------
def pack_weights(weight: Tensor, bias: Optional[Tensor], gm: GemmMap, transpose: bool) -> Tuple[Tensor, Optional[Tensor]]:
    """OIHW fp32 -> packed MFMA B image (+ packed bias).  sf_conv3x3_pack_weights."""
    _hip.require_device(weight, "weight")
    w = weight.detach().contiguous()
    O, I = w.shape[0], w.shape[1]
    assert w.shape[2:] == (3, 3), "only 3x3 kernels are on the path"
    nmap, kmap = gm.tables(w.device)
    dt = _hip.compute_dtype()
    # ("f32e": three virtual K chunks per real one - the fp16 parts [lo'(w), hi(w)] of every chunk, then hi(w) of all)
    packed = torch.empty(gm.Np * gm.Kp * (27 if dt == _hip.SF_F32E else 9),
                         dtype={_hip.SF_BF16: torch.bfloat16, _hip.SF_F16: torch.float16, _hip.SF_F32E: torch.float16}.get(dt, torch.float32), device=w.device)
    bp = torch.empty(gm.Np, dtype=torch.float32, device=w.device) if (bias is not None and not transpose) else None
    check(
        lib().sf_conv3x3_pack_weights(
            w.data_ptr(), O, I, nmap.data_ptr(), gm.Np, kmap.data_ptr(), gm.Kp, gm.nf, int(transpose), packed.data_ptr(),
            bias.detach().contiguous().data_ptr() if bp is not None else None, bp.data_ptr() if bp is not None else None,
            dt, stream_ptr(),
        ),
        "sf_conv3x3_pack_weights",
    )
    return packed, bp


def grad_operand(t: Tensor, acc: Optional[Tensor] = None, reset_acc: bool = False, **kw) -> sfTensor:
    """``T(t)`` for a GRADIENT tensor that is about to be an MFMA operand (the source of an input-gradient convolution, ``dout`` of a weight gradient).
    In "f32e" mode the descriptor carries the tensor's ``sf_amax`` word (one HBM pass here), through which the kernels scale it into fp16's range; every
    other mode: plain ``T(t)``.  ``acc``: a second word that accumulates the maximum over several calls (a recurrent cell's per-step gate gradients, for
    the one weight-gradient launch over all steps: ``grad_operand_with(t, acc)``)."""
    if _hip.compute_dtype() != _hip.SF_F32E or t.dtype != torch.float32:
        return T(t, **kw)
    tag = getattr(t, "_sf_amax", None)   # left by the kernel that WROTE t (``tag_amax``): no pass over it here
    if tag is not None and acc is None and tag[1] == t._version and tag[2] == t.data_ptr():
        return grad_operand_with(t, tag[0], **kw)
    word = torch.empty(1, dtype=torch.float32, device=t.device)
    d = T(t, **kw)
    check(lib().sf_amax(d, t.numel() // t.shape[-1], word.data_ptr(), acc.data_ptr() if acc is not None else None, int(reset_acc), stream_ptr()), "sf_amax")
    d.amax = word.data_ptr()
    d._keep = word   # the word lives as long as the descriptor (stream-ordered reuse by the caching allocator is safe: same stream)
    return d


def scale_words(n: int, device) -> Optional[Tensor]:
    """``n`` zeroed scale words for kernels that raise them while they write a gradient (``sf_convlstm_cell_bwd_gates`` with ``dz.amax``) - "f32e" mode
    only, else None.  One fill launch; never a memset node (torch.zeros fills with a kernel)."""
    if _hip.compute_dtype() != _hip.SF_F32E:
        return None
    return torch.zeros(n, dtype=torch.float32, device=device)


def tag_amax(t: Tensor, word: Optional[Tensor]) -> Tensor:
    """Attach the scale word a kernel raised WHILE IT WROTE ``t`` (``sf_batchnorm_train_bwd`` with ``dx.amax``) to the tensor object; ``grad_operand`` of the
    consumer then skips its ``sf_amax`` pass.  The tag travels with the Python object through the autograd engine (torch keeps a tensor's Python object, with
    its attributes, alive with the tensor) and is void once the tensor has been written in place (version counter) or is another tensor (a view, a sum of
    gradients: those have no tag).  ``word`` None (not "f32e"): nothing."""
    if word is not None:
        t._sf_amax = (word, t._version, t.data_ptr())
    return t


def grad_operand_with(t: Tensor, word: Optional[Tensor], **kw) -> sfTensor:
    """``T(t)`` carrying an amax word that was accumulated earlier (``grad_operand(..., acc=word)``); plain ``T(t)`` without a word."""
    d = T(t, **kw)
    if word is not None and _hip.compute_dtype() == _hip.SF_F32E:
        d.amax = word.data_ptr()
        d._keep = word
    return d


def conv3x3(src0: sfTensor, src1: sfTensor, n: int, h: int, w: int, packed: Tensor, bias_packed: Optional[Tensor],
            gm: GemmMap, out: sfTensor, epilogue: int = SF_EPI_LINEAR, stats: Optional[Tensor] = None) -> None:
    """``stats`` ([n * sf_conv3x3_stats_tiles(h, w), Np, 2] fp32): also emit the per-tile output statistics (sf_conv3x3_fwd_stats)."""
    if stats is not None:
        assert epilogue == SF_EPI_LINEAR and stats.is_contiguous() and stats.dtype == torch.float32
        check(
            lib().sf_conv3x3_fwd_stats(src0, src1, n, h, w, packed.data_ptr(), bias_packed.data_ptr() if bias_packed is not None else None,
                                       gm.Np, gm.nf, out, stats.data_ptr(), _hip.compute_dtype(), stream_ptr()),
            "sf_conv3x3_fwd_stats",
        )
        return
    if epilogue == SF_EPI_LINEAR and src1.ptr is None and not src0.idiv:
        # few small images with many input channels (a recurrent cell's state convolution): input channels sliced over workgroups
        nbytes = lib().sf_conv3x3_fwd_splitk_workspace_bytes(n, h, w, gm.Np, gm.nf, src0.c, _hip.compute_dtype())
        if nbytes:
            ws = torch.empty(nbytes // 4, dtype=torch.float32, device=packed.device)
            check(lib().sf_conv3x3_fwd_splitk(src0, n, h, w, packed.data_ptr(), bias_packed.data_ptr() if bias_packed is not None else None, gm.Np, gm.nf,
                                              out, ws.data_ptr(), nbytes, _hip.compute_dtype(), stream_ptr()), "sf_conv3x3_fwd_splitk")
            return
    check(
        lib().sf_conv3x3_fwd(src0, src1, n, h, w, packed.data_ptr(), bias_packed.data_ptr() if bias_packed is not None else None,
                             gm.Np, gm.nf, epilogue, out, _hip.compute_dtype(), stream_ptr()),
        "sf_conv3x3_fwd",
    )


def convlstm_cell_fwd(x: sfTensor, h_prev: sfTensor, c_prev: sfTensor, n: int, h: int, w: int, packed: Tensor,
                      bias_packed: Optional[Tensor], hidp: int, h_out: sfTensor, c_out: sfTensor, gates: sfTensor) -> None:
    check(
        lib().sf_convlstm_cell_fwd(x, h_prev, c_prev, n, h, w, packed.data_ptr(),
                                   bias_packed.data_ptr() if bias_packed is not None else None, hidp, h_out, c_out, gates,
                                   _hip.compute_dtype(), stream_ptr()),
        "sf_convlstm_cell_fwd",
    )


def convlstm_cell_bwd_gates(dh: Sequence[sfTensor], dc_next: sfTensor, gates: sfTensor, c_prev: sfTensor, c_new: sfTensor,
                            pixels: int, hidp: int, dz: sfTensor, dc_prev: sfTensor) -> None:
    """``dz.amax`` set (``T(dz, amax=word)``, word zeroed by the caller): the launch raises the word to max |dz| ("f32e" mode's scale word)."""
    dh = list(dh) + [NULL] * (3 - len(dh))
    check(
        lib().sf_convlstm_cell_bwd_gates(dh[0], dh[1], dh[2], dc_next, gates, c_prev, c_new, pixels, hidp, dz, dc_prev, SF_F32,
                                         stream_ptr()),
        "sf_convlstm_cell_bwd_gates",
    )


def conv5x5_shift4(x: Tensor, n: int, h: int, w: int, packed: Tensor, bias_packed: Optional[Tensor], gm: "GemmMap", out: Tensor) -> None:
    """A 5x5 'same' convolution of fp32-stored NHWC ``x`` on the 16-bit 3x3 kernels: four shifted views of ``x`` read in place, ``packed`` = the packed
    image of ``regroup5x5(weight, x lanes)`` for ``Kp = 4 * x lanes``.  sf_conv5x5_fwd (split over the virtual channels for few small images)."""
    dt = _hip.compute_dtype()
    nbytes = lib().sf_conv5x5_fwd_workspace_bytes(n, h, w, gm.Np, gm.nf, x.shape[-1], dt)
    ws = torch.empty(nbytes // 4 + 4, dtype=torch.float32, device=x.device) if nbytes else None
    check(lib().sf_conv5x5_fwd(T(x), n, h, w, packed.data_ptr(), bias_packed.data_ptr() if bias_packed is not None else None, gm.Np, gm.nf, T(out),
                               ws.data_ptr() if ws is not None else None, nbytes, dt, stream_ptr()), "sf_conv5x5_fwd")


def conv5x5_shift4_bwd_weight(x: Tensor, dout: Tensor, n: int, h: int, w: int, gm: "GemmMap", dw: Tensor, db: Optional[Tensor]) -> None:
    """dW3 ``[O, 4 * x lanes, 3, 3]`` (and db) of ``conv5x5_shift4``.  sf_conv5x5_bwd_weight."""
    nmap, kmap = gm.tables(dw.device)
    nbytes = lib().sf_conv5x5_bwd_weight_workspace_bytes(dout.shape[-1], x.shape[-1], n, h, w)
    ws = torch.empty(nbytes // 4 + 1, dtype=torch.float32, device=dw.device)
    assert dw.is_contiguous() and (db is None or db.is_contiguous())
    check(lib().sf_conv5x5_bwd_weight(T(x), T(dout), n, h, w, nmap.data_ptr(), kmap.data_ptr(), dw.shape[0], dw.shape[1], dw.data_ptr(),
                                      db.data_ptr() if db is not None else None, 0, ws.data_ptr(), nbytes, _hip.compute_dtype(), stream_ptr()),
          "sf_conv5x5_bwd_weight")


def conv3x3_bwd_weight(src0: sfTensor, src1: sfTensor, dout: sfTensor, n: int, h: int, w: int, gm: GemmMap, dw: Tensor,
                       db: Optional[Tensor], accumulate: bool) -> None:
    """dW/db of a 3x3 conv into the reference's OIHW gradient tensors.  sf_conv3x3_bwd_weight."""
    dev = dw.device
    nmap, kmap = gm.tables(dev)
    nbytes = lib().sf_conv3x3_bwd_weight_workspace_bytes(dout.c, src0.c + src1.c, n, h, w)
    ws = torch.empty(nbytes // 4 + 1, dtype=torch.float32, device=dev)
    O, I = dw.shape[0], dw.shape[1]
    assert dw.is_contiguous() and (db is None or db.is_contiguous())
    check(
        lib().sf_conv3x3_bwd_weight(src0, src1, dout, n, h, w, nmap.data_ptr(), kmap.data_ptr(), O, I, dw.data_ptr(),
                                    db.data_ptr() if db is not None else None, int(accumulate), ws.data_ptr(), nbytes,
                                    _hip.compute_dtype(), stream_ptr()),
        "sf_conv3x3_bwd_weight",
    )


def conv3x3_fold_pack(weight: Tensor, bias: Optional[Tensor], gm: GemmMap, scale: Tensor, shift: Tensor) -> Tuple[Tensor, Tensor]:
    """Per-group packed weights ``W * scale_g`` and the border-class bias table of a folded BatchNorm.  sf_conv3x3_fold_pack."""
    w = weight.detach().contiguous()
    groups = scale.shape[0]
    assert scale.shape == (groups, gm.Kp) and shift.shape == scale.shape and scale.is_contiguous() and shift.is_contiguous()
    nmap, kmap = gm.tables(w.device)
    packed = torch.empty(groups * gm.Np * gm.Kp * 9, dtype=torch.bfloat16, device=w.device)
    tab = torch.empty(groups, 9, gm.Np, dtype=torch.float32, device=w.device)
    check(lib().sf_conv3x3_fold_pack(w.data_ptr(), w.shape[0], w.shape[1], nmap.data_ptr(), gm.Np, kmap.data_ptr(), gm.Kp, gm.nf,
                                     bias.detach().contiguous().data_ptr() if bias is not None else None, scale.data_ptr(), shift.data_ptr(),
                                     groups, packed.data_ptr(), tab.data_ptr(), _hip.SF_BF16, stream_ptr()), "sf_conv3x3_fold_pack")
    return packed, tab


def conv3x3_folded(src: sfTensor, n: int, h: int, w: int, packed: Tensor, tab: Tensor, gm: GemmMap, out: sfTensor,
                   stats: Optional[Tensor] = None) -> None:
    check(lib().sf_conv3x3_fwd_folded(src, n, h, w, packed.data_ptr(), tab.data_ptr(), gm.Np, gm.nf, tab.shape[0], out,
                                      stats.data_ptr() if stats is not None else None, _hip.SF_BF16, stream_ptr()), "sf_conv3x3_fwd_folded")


def conv3x3_folded_pool_supported(n: int, h: int, w: int, gm: GemmMap, cout_lanes: int, groups: int) -> bool:
    """Does the pooled-epilogue kernel take this shape (sf_conv3x3_fwd_folded_pool_supported; SF_NO_POOL_FUSE=1: A/B switch)?"""
    return (not os.environ.get("SF_NO_POOL_FUSE")
            and bool(lib().sf_conv3x3_fwd_folded_pool_supported(n, h, w, gm.Np, gm.nf, gm.Kp, cout_lanes, groups)))


def conv3x3_folded_pool(src: sfTensor, n: int, h: int, w: int, packed: Tensor, tab: Tensor, gm: GemmMap, cout_lanes: int,
                        perm: Optional[Tuple[int, int]], drop, device) -> Tuple[Tensor, Tensor]:
    """Folded convolution + 2x2 max-pooling in ONE launch (sf_conv3x3_fwd_folded_pool) -> (pooled bf16 [n,h/2,w/2,c] in the permuted image order,
    routing record for ``maxpool2_route_bwd``).  ``drop = (p1, p2, period, seed1, seed2)``: the two dropout masks of ``maxpool2_route_fwd`` applied to the
    pooled tensor in place (sf_dropout2_bf16: same masks, same rounding - one multiplication in fp32, one rounding to bf16)."""
    y = torch.empty(n, h // 2, w // 2, cout_lanes, dtype=torch.bfloat16, device=device)
    route = torch.empty(n, h // 2, w // 2, cout_lanes // 8, dtype=torch.int16, device=device)
    pl, pt = perm or (0, 0)
    check(lib().sf_conv3x3_fwd_folded_pool(src, n, h, w, packed.data_ptr(), tab.data_ptr(), gm.Np, gm.nf, tab.shape[0], T(y), pl, pt, route.data_ptr(),
                                           _hip.SF_BF16, stream_ptr()), "sf_conv3x3_fwd_folded_pool")
    if drop is not None and (drop[0] > 0 or drop[1] > 0):
        check(lib().sf_dropout2_bf16(y.data_ptr(), y.numel(), drop[0], drop[1], drop[2], drop[3], drop[4], y.data_ptr(), stream_ptr()), "sf_dropout2_bf16")
    return y, route


def conv3x3_fold_supported(n: int, h: int, w: int, gm: GemmMap, groups: int, stats: bool) -> bool:
    """Shapes sf_conv3x3_fwd_folded takes (the two-images-per-workgroup kernel of small images has one weight stream)."""
    table_lds = ((gm.Kp + groups) * 9 + 2 * groups * gm.Kp) * 4  # sf_conv3x3_fold_pack's table kernel stages scale / shift of all groups
    return h >= 2 and w >= 2 and n % groups == 0 and (h > 16 or stats or gm.nf < 4 or n < 512) and table_lds <= 64 * 1024


def conv3x3_bwd_weight_pooled_supported(np_: int, kp: int, n: int, h: int, w: int, groups: int) -> bool:
    """Will ``conv3x3_bwd_weight_folded(..., pooled=)`` build its sparse operand from the pooled gradient (whole 4 x 16 tiles, 128-channel tiles)?"""
    return (not os.environ.get("SF_NO_WGRAD_POOLED") and h % 4 == 0 and w % 16 == 0 and np_ % 128 == 0
            and bool(lib().sf_conv3x3_bwd_weight_folded_sparse24_supported(np_, kp, n, h, w, groups)))


def conv3x3_bwd_weight_folded(src: sfTensor, dout: sfTensor, n: int, h: int, w: int, gm: GemmMap, scale: Tensor, shift: Tensor,
                              dw: Tensor, db: Optional[Tensor], bn: Optional[Tuple[Tensor, Tensor, Tensor, Tensor]] = None,
                              pooled_gradient: bool = False, pooled: Optional[Tuple[Tensor, Tensor, Optional[Tuple[int, int]]]] = None) -> None:
    """sf_conv3x3_bwd_weight_folded: dW/db of a convolution behind a folded BatchNorm, from the un-normalised input.
    ``bn = (weight OIHW, mean, rstd, sums[groups,2,C] float64)``: also fills ``sums`` with the BatchNorm backward's two reductions.
    ``pooled_gradient``: ``dout`` is the output of ``maxpool2_route_bwd`` (one non-zero per 2x2 window and channel): where the shape allows, dout is the
    sparse operand of the 2:4 structured-sparse matrix instruction (sf_conv3x3_bwd_weight_folded_sparse24; SF_NO_WGRAD_SPARSE=1: A/B switch).
    ``pooled = (pooled gradient, routing record, (perm_l, perm_t) or None)``: the inputs of that ``maxpool2_route_bwd`` call - the kernel then builds the
    sparse operand from them instead of reading ``dout`` (SF_NO_WGRAD_POOLED=1: A/B switch)."""
    dev = dw.device
    fn, fname = lib().sf_conv3x3_bwd_weight_folded, "sf_conv3x3_bwd_weight_folded"
    extra: tuple = ()
    if pooled_gradient and lib().sf_conv3x3_bwd_weight_folded_sparse24_supported(dout.c, src.c, n, h, w, scale.shape[0]):
        fn, fname = lib().sf_conv3x3_bwd_weight_folded_sparse24, "sf_conv3x3_bwd_weight_folded_sparse24"
        if pooled is not None and pooled[0].dtype == torch.bfloat16 and pooled[0].is_contiguous() and pooled[1].is_contiguous():
            pl, pt = pooled[2] or (0, 0)
            extra = (T(pooled[0]), pooled[1].data_ptr(), pl, pt)
        else:
            extra = (NULL, None, 0, 0)
    nmap, kmap = gm.tables(dev)
    groups = scale.shape[0]
    nbytes = lib().sf_conv3x3_bwd_weight_folded_workspace_bytes(dout.c, src.c, n, h, w, groups)
    ws = torch.empty(nbytes // 8 + 1, dtype=torch.float64, device=dev)
    wgt = mean = rstd = sums = None
    if bn is not None:
        wgt, mean, rstd, sums = bn
        wgt = wgt.detach().contiguous()
        assert sums.dtype == torch.float64 and sums.is_contiguous() and mean.is_contiguous() and rstd.is_contiguous()
    check(fn(src, dout, n, h, w, nmap.data_ptr(), kmap.data_ptr(), dw.shape[0], dw.shape[1], scale.data_ptr(),
             shift.data_ptr(), groups, dw.data_ptr(), db.data_ptr() if db is not None else None, 0,
             wgt.data_ptr() if bn is not None else None, mean.data_ptr() if bn is not None else None,
             rstd.data_ptr() if bn is not None else None, sums.data_ptr() if bn is not None else None, *extra,
             ws.data_ptr(), nbytes, _hip.SF_BF16, stream_ptr()), fname)


def conv3x3_bwd_data_bn(dout: sfTensor, n: int, h: int, w: int, packed_t: Tensor, gm: GemmMap, x: sfTensor, coef: Tensor, dx: sfTensor) -> None:
    """sf_conv3x3_bwd_data_bn: ``dx = A * conv^T(dout, W) + B * x + K`` (BatchNorm backward in the convolution's epilogue)."""
    check(lib().sf_conv3x3_bwd_data_bn(dout, n, h, w, packed_t.data_ptr(), gm.Np, gm.nf, x, coef.data_ptr(), coef.shape[0], dx, _hip.SF_BF16,
                                       stream_ptr()), "sf_conv3x3_bwd_data_bn")


def to_nhwc(src: Tensor, nb: int, nt: int, c: int, h: int, w: int, strides: Tuple[int, int, int], cp: Optional[int] = None,
            out_dtype=torch.float32) -> Tensor:
    """NCHW-side tensor -> time-major NHWC ``[nt*nb, h, w, cp]`` (pad lanes zero), stored as fp32 or bf16.  sf_nchw_to_nhwc."""
    _hip.require_device(src, "input")
    cp = cp or cpad(c)
    dst = torch.empty(nt * nb, h, w, cp, dtype=out_dtype, device=src.device)
    check(lib().sf_nchw_to_nhwc(src.data_ptr(), *strides, nb, nt, c, h, w, T(dst), SF_F32, stream_ptr()), "sf_nchw_to_nhwc")
    return dst


def from_nhwc(src: Tensor, nb: int, nt: int, c: int, h: int, w: int, dst: Tensor, strides: Tuple[int, int, int]) -> Tensor:
    """Time-major NHWC -> NCHW-side tensor ``dst`` addressed with ``strides=(b,t,c)``.  sf_nhwc_to_nchw."""
    check(lib().sf_nhwc_to_nchw(T(src), nb, nt, c, h, w, dst.data_ptr(), *strides, SF_F32, stream_ptr()), "sf_nhwc_to_nchw")
    return dst


# ----------------------------------------------------------------------------------------------
# MetNet stack
# ----------------------------------------------------------------------------------------------
def gru_rec_map(hid: int) -> GemmMap:
    """Recurrent ConvGRU conv h -> [z_h | r_h | h2]: each N block = 3 maps x 32 hidden channels."""
    hidp = cpad(hid)
    nblk = -(-hidp // 32)
    nmap = []
    for nb in range(nblk):
        for g in range(3):
            for j in range(32):
                hc = nb * 32 + j
                nmap.append(g * hid + hc if hc < hid else -1)
    return GemmMap(nmap, _padded(hid), 3, nblk * 96)


def gate_major(hid: int, gates: int) -> List[int]:
    """Padded gate-major lanes [g*hidp + j] -> rows g*hid + j."""
    hidp = cpad(hid)
    return [g * hid + j if j < hid else -1 for g in range(gates) for j in range(hidp)]


def custom_map(nlanes: List[int], klanes: List[int], nf: Optional[int] = None) -> GemmMap:
    return _finish(list(nlanes), list(klanes), nf)


def metnet_preprocess(imgs: Tensor, sat: int, crop: int, out_dtype=torch.float32) -> Tensor:
    """imgs[B,T,C,H,W] -> frames [T*B, crop, crop, Cp] (time-major), stored as ``out_dtype``.  sf_metnet_preprocess_fwd."""
    _hip.require_device(imgs, "imgs")
    imgs = imgs.contiguous()
    B, Tn, C, H, W = imgs.shape
    cp = cpad(8 * sat + (C - sat))
    out = torch.empty(Tn * B, crop, crop, cp, dtype=out_dtype, device=imgs.device)
    check(lib().sf_metnet_preprocess_fwd(imgs.data_ptr(), B, Tn, C, sat, H, W, crop, T(out), SF_F32, stream_ptr()), "sf_metnet_preprocess_fwd")
    return out


def maxpool2_fwd(x: Tensor, perm: Optional[Tuple[int, int]] = None, out_dtype=None, drop=None) -> Tensor:
    """``drop = (p1, p2, period, seed1, seed2)``: the encoder-output dropouts fused into the pooling (sf_maxpool2_dropout_fwd)."""
    n, h, w, c = x.shape
    y = torch.empty(n, h // 2, w // 2, c, dtype=out_dtype or x.dtype, device=x.device)
    pl, pt = perm or (0, 0)
    if drop is None:
        check(lib().sf_maxpool2_fwd(T(x), n, h, w, T(y), pl, pt, SF_F32, stream_ptr()), "sf_maxpool2_fwd")
    else:
        check(lib().sf_maxpool2_dropout_fwd(T(x), n, h, w, T(y), pl, pt, *drop, SF_F32, stream_ptr()), "sf_maxpool2_dropout_fwd")
    return y


def maxpool2_route_fwd(x: Tensor, perm: Optional[Tuple[int, int]] = None, out_dtype=None, drop=None) -> Tuple[Tensor, Tensor]:
    """The pooling that also records the routing (2 bits per channel, uint16 per channel octet) for ``maxpool2_route_bwd``:
    the backward pass then needs neither x nor a second read of it.  sf_maxpool2_route_fwd."""
    n, h, w, c = x.shape
    y = torch.empty(n, h // 2, w // 2, c, dtype=out_dtype or x.dtype, device=x.device)
    route = torch.empty(n, h // 2, w // 2, c // 8, dtype=torch.int16, device=x.device)
    pl, pt = perm or (0, 0)
    d = drop if drop is not None else (0.0, 0.0, 0, 0, 0)
    check(lib().sf_maxpool2_route_fwd(T(x), n, h, w, T(y), pl, pt, *d, route.data_ptr(), SF_F32, stream_ptr()), "sf_maxpool2_route_fwd")
    return y, route


def maxpool2_route_bwd(route: Tensor, gy: Tensor, shape, dtype, perm: Optional[Tuple[int, int]] = None, drop=None, masked: Optional[Tensor] = None) -> Tensor:
    """``masked``: a tensor like ``gy`` that receives the pooled gradient after the dropout masks (the operand source of the pooled sparse weight gradient)."""
    n, h, w, c = shape
    gx = torch.empty(shape, dtype=dtype, device=gy.device)
    pl, pt = perm or (0, 0)
    d = drop if drop is not None else (0.0, 0.0, 0, 0, 0)
    assert masked is None or (masked.shape == gy.shape and masked.dtype == gy.dtype and masked.is_contiguous() and gy.is_contiguous())
    word = scale_words(1, gy.device) if dtype == torch.float32 else None   # "f32e": gx is a convolution's gradient operand - its scale word comes with it
    check(lib().sf_maxpool2_route_bwd(route.data_ptr(), T(gy), n, h, w, T(gx, amax=word), pl, pt, *d, masked.data_ptr() if masked is not None else None, SF_F32,
                                      stream_ptr()), "sf_maxpool2_route_bwd")
    return tag_amax(gx, word)


def maxpool2_bwd(x: Tensor, gy: Tensor, perm: Optional[Tuple[int, int]] = None, drop=None) -> Tensor:
    n, h, w, c = x.shape
    gx = torch.empty_like(x)
    pl, pt = perm or (0, 0)
    word = scale_words(1, gy.device) if gx.dtype == torch.float32 else None
    if drop is None:
        check(lib().sf_maxpool2_bwd(T(x), T(gy), n, h, w, T(gx, amax=word), pl, pt, SF_F32, stream_ptr()), "sf_maxpool2_bwd")
    else:
        check(lib().sf_maxpool2_dropout_bwd(T(x), T(gy), n, h, w, T(gx, amax=word), pl, pt, *drop, SF_F32, stream_ptr()), "sf_maxpool2_dropout_bwd")
    return tag_amax(gx, word)


def linear_fwd(x: Tensor, W: Tensor, bias: Optional[Tensor], out_lanes: int, lowp: bool = False) -> Tensor:
    """y[..., n] = x[..., :] @ W[n, :] + b.  x [..., Kp], W [N, Kp] -> y [..., out_lanes].  sf_linear_fwd.  ``lowp``: in a 16-bit compute mode the
    operands are rounded to the mode's type (a 1x1 convolution under the reference's autocast); exact fp32 products otherwise."""
    rows = x.numel() // x.shape[-1]
    assert W.shape[1] == x.shape[-1] and W.is_contiguous()
    y = torch.empty(*x.shape[:-1], out_lanes, dtype=torch.float32, device=x.device)
    dt = _hip.exact_dtype() if lowp and not os.environ.get("SF_LINEAR_F32") else SF_F32   # (SF_LINEAR_F32=1: A/B switch; "f32e": exact fp32)
    check(lib().sf_linear_fwd(T(x), rows, W.data_ptr(), W.shape[0], bias.data_ptr() if bias is not None else None, T(y), dt,
                              stream_ptr()), "sf_linear_fwd")
    return y


def linear_bwd_weight(dy: Tensor, x: Tensor, N: int, want_bias: bool, out: Optional[Tuple[Optional[Tensor], Optional[Tensor]]] = None) -> Tuple[Tensor, Optional[Tensor]]:
    """``out = (dW [N, K] or None, db [N] or None)``: contiguous fp32 destinations to write instead of fresh tensors."""
    rows = x.numel() // x.shape[-1]
    K = x.shape[-1]
    nbytes = lib().sf_linear_bwd_weight_workspace_bytes(N, K, rows)
    ws = torch.empty(nbytes // 4 + 1, dtype=torch.float32, device=x.device)
    dW = out[0] if out is not None and out[0] is not None else torch.empty(N, K, dtype=torch.float32, device=x.device)
    db = (out[1] if out is not None and out[1] is not None else torch.empty(N, dtype=torch.float32, device=x.device)) if want_bias else None
    assert dW.is_contiguous() and dW.numel() == N * K and (db is None or (db.is_contiguous() and db.numel() == N))
    check(lib().sf_linear_bwd_weight(T(dy), N, T(x), rows, dW.data_ptr(), db.data_ptr() if db is not None else None, ws.data_ptr(),
                                     nbytes, SF_F32, stream_ptr()), "sf_linear_bwd_weight")
    return dW, db


def bmm_raw(A: Tensor, B: Tensor, out: Tensor, alpha: float = 1.0, beta: float = 0.0, lowp: bool = False) -> None:
    """``out[b] = alpha * A[b] @ B[b] + beta * out[b]`` for 3-D fp32 views with arbitrary strides.  sf_bmm_f32 (exact fp32 MFMA), or - ``lowp`` in a
    16-bit compute mode - sf_bmm_bf16 / sf_bmm_f16 (operands rounded to the mode's 16-bit type as torch.bmm's are under autocast)."""
    assert A.dim() == B.dim() == out.dim() == 3 and A.dtype == B.dtype == out.dtype == torch.float32
    bsz, M, Kd = A.shape
    N = B.shape[2]
    assert B.shape[0] == bsz and B.shape[1] == Kd and out.shape == (bsz, M, N), (A.shape, B.shape, out.shape)
    sa, sb, sc = A.stride(), B.stride(), out.stride()
    dt = _hip.compute_dtype() if lowp else _hip.SF_F32
    fn, name = {_hip.SF_BF16: (lib().sf_bmm_bf16, "sf_bmm_bf16"), _hip.SF_F16: (lib().sf_bmm_f16, "sf_bmm_f16")}.get(dt, (lib().sf_bmm_f32, "sf_bmm_f32"))
    check(fn(A.data_ptr(), sa[0], sa[1], sa[2], B.data_ptr(), sb[0], sb[1], sb[2], out.data_ptr(), sc[0], sc[1], sc[2], bsz, M, N, Kd,
             alpha, beta, stream_ptr()), name)


def flash_attention_ok(q: Tensor, k: Tensor, v: Tensor) -> bool:
    """Shapes sf_flash_attention_* takes in the current compute mode: a 16-bit mode, [batch, n, lanes] fp32 with contiguous rows, n a multiple of 128,
    q / k 16 or 32 lanes wide, v 32 / 64 / 128 / 256."""
    if _hip.compute_dtype() not in (_hip.SF_BF16, _hip.SF_F16) or q.dim() != 3:
        return False
    b, n, dqk = q.shape
    return (n % 128 == 0 and dqk in (16, 32) and k.shape == q.shape and v.shape[:2] == (b, n) and v.shape[2] in (32, 64, 128, 256)
            and all(t.dtype == torch.float32 and t.stride(2) == 1 and t.stride(1) % 4 == 0 and t.stride(0) == n * t.stride(1) and t.data_ptr() % 16 == 0 for t in (q, k, v)))


def flash_attention_fwd(q: Tensor, k: Tensor, v: Tensor, scale: float) -> Tuple[Tensor, Tensor]:
    """``softmax(scale * q k^T) v`` without the score matrix; returns (out, log-sum-exp per query).  sf_flash_attention_fwd."""
    b, n, dqk = q.shape
    dv = v.shape[2]
    out = torch.empty(b, n, dv, dtype=torch.float32, device=q.device)
    lse = torch.empty(b, n, dtype=torch.float32, device=q.device)
    check(lib().sf_flash_attention_fwd(q.data_ptr(), q.stride(1), k.data_ptr(), k.stride(1), v.data_ptr(), v.stride(1), b, n, dqk, dv, scale, out.data_ptr(),
                                       out.stride(1), lse.data_ptr(), _hip.compute_dtype(), stream_ptr()), "sf_flash_attention_fwd")
    return out, lse


def flash_attention_bwd(q: Tensor, k: Tensor, v: Tensor, out: Tensor, lse: Tensor, dout: Tensor, scale: float) -> Tuple[Tensor, Tensor, Tensor]:
    """Gradients of flash_attention_fwd (probabilities recomputed from ``lse``).  sf_flash_attention_bwd."""
    b, n, dqk = q.shape
    dv = v.shape[2]
    dout = dout.contiguous()
    dq, dk, dvg = torch.zeros_like(q), torch.zeros_like(k), torch.empty(b, n, dv, dtype=torch.float32, device=q.device)
    delta = torch.empty(b, n, dtype=torch.float32, device=q.device)
    check(lib().sf_flash_attention_bwd(q.data_ptr(), q.stride(1), k.data_ptr(), k.stride(1), v.data_ptr(), v.stride(1), out.data_ptr(), out.stride(1), lse.data_ptr(),
                                       dout.data_ptr(), dout.stride(1), b, n, dqk, dv, scale, dq.data_ptr(), dq.stride(1), dk.data_ptr(), dk.stride(1),
                                       dvg.data_ptr(), dvg.stride(1), delta.data_ptr(), _hip.compute_dtype(), stream_ptr()), "sf_flash_attention_bwd")
    return dq, dk, dvg


def linear_bwd_weight_any(dy: Tensor, x: Tensor, N: int, want_bias: bool) -> Tuple[Tensor, Optional[Tensor]]:
    """``linear_bwd_weight`` for any K: the split-K MFMA kernel up to K = 256, beyond that (the 1x1 convolutions of the DGMR
    discriminators, up to 2048 channels) the strided batched product over row slices plus a product with a ones vector that sums
    the slices (both sf_bmm_f32)."""
    Kd = x.shape[-1]
    if Kd <= 256:
        return linear_bwd_weight(dy, x, N, want_bias)
    rows = x.numel() // Kd
    lanes = dy.shape[-1]
    S = 1
    for cand in (64, 32, 16, 8, 4, 2):
        if rows % cand == 0 and rows // cand >= 64:
            S = cand
            break
    per = rows // S
    dy3, x3 = dy.reshape(S, per, lanes), x.reshape(S, per, Kd)
    part = torch.empty(S, N, Kd, dtype=torch.float32, device=x.device)
    bmm_raw(dy3[..., :N].transpose(1, 2), x3, part)
    ones = torch.ones(1, 1, max(S, per), dtype=torch.float32, device=x.device)
    dW = torch.empty(1, 1, N * Kd, dtype=torch.float32, device=x.device)
    bmm_raw(ones[..., :S], part.view(1, S, N * Kd), dW)
    db = None
    if want_bias:
        pb = torch.empty(S, 1, N, dtype=torch.float32, device=x.device)
        bmm_raw(ones[..., :per].expand(S, 1, per), dy3[..., :N], pb)
        db = torch.empty(1, 1, N, dtype=torch.float32, device=x.device)
        bmm_raw(ones[..., :S], pb.view(1, S, N), db)
        db = db.view(N)
    return dW.view(N, Kd), db


def attention_core_fwd(qkv: Tensor, hid: int, heads: int) -> Tensor:
    n, h, w, c6 = qkv.shape
    hidp = c6 // 6
    att = torch.empty(n, h, w, 2 * hidp, dtype=torch.float32, device=qkv.device)
    check(lib().sf_axial_attention_core_fwd(T(qkv), n, h, w, hid, hidp, heads, T(att), SF_F32, stream_ptr()), "sf_axial_attention_core_fwd")
    return att


def attention_core_bwd(qkv: Tensor, datt: Tensor, hid: int, heads: int) -> Tensor:
    n, h, w, c6 = qkv.shape
    hidp = c6 // 6
    dqkv = torch.empty_like(qkv)
    check(lib().sf_axial_attention_core_bwd(T(qkv), T(datt), n, h, w, hid, hidp, heads, T(dqkv), SF_F32, stream_ptr()),
          "sf_axial_attention_core_bwd")
    return dqkv


def convgru_step_fwd(gx: sfTensor, h_prev: Optional[Tensor], n: int, h: int, w: int, packed: Tensor, bias_packed: Optional[Tensor],
                     hidp: int, h_out: Tensor, gates: Optional[Tensor]) -> None:
    check(lib().sf_convgru_step_fwd(gx, T(h_prev, hidp), n, h, w, packed.data_ptr(), bias_packed.data_ptr() if bias_packed is not None else None,
                                    hidp, T(h_out), T(gates) if gates is not None else NULL, _hip.compute_dtype(), stream_ptr()), "sf_convgru_step_fwd")


def convgru_seq_supported(h: int, w: int, hidp: int) -> bool:
    """The persistent sequence kernel: SF_BF16 kernels, maps of at most 16x16 pixels (one workgroup per map), hidp <= 64."""
    return _hip.compute_dtype() == _hip.SF_BF16 and h <= 16 and w <= 16 and 16 <= hidp <= 64


def convgru_seq_fwd(gx: Tensor, h0: Optional[Tensor], Tn: int, n: int, h: int, w: int, packed: Tensor, bias_packed: Optional[Tensor], hidp: int,
                    hs: Tensor, gates: Optional[Tensor]) -> Optional[Tensor]:
    """All Tn recurrent steps in one launch (sf_convgru_seq_fwd): gx ``[Tn*n,h,w,3*hidp]``, hs ``[Tn,n,h,w,hidp]``.
    With a workspace the library may split every map over two workgroups (boundary rows exchanged inside the launch).  The
    workspace is the cached one of ``_hip.sticky_workspace``: its first word is the sticky error word that
    ``satflow_amd.check_device_errors()`` reads."""
    nbytes = int(lib().sf_convgru_seq_fwd_workspace_bytes(n, h, hidp))
    ws = _hip.sticky_workspace("convgru_seq_fwd", (n, h, hidp), nbytes, gx.device)
    check(lib().sf_convgru_seq_fwd(T(gx), T(h0, hidp), Tn, n, h, w, packed.data_ptr(), bias_packed.data_ptr() if bias_packed is not None else None,
                                   hidp, T(hs), T(gates) if gates is not None else NULL, ws.data_ptr() if ws is not None else None, nbytes,
                                   _hip.compute_dtype(), stream_ptr()), "sf_convgru_seq_fwd")
    return ws


def convgru_seq_bwd_supported(h: int, w: int, hidp: int, gates: Tensor) -> bool:
    """The persistent backward kernel: as the forward one, hidp 32 or 64, bf16-stored gates ("bf16a" mode)."""
    return convgru_seq_supported(h, w, hidp) and hidp in (32, 64) and gates.dtype == torch.bfloat16


def convgru_seq_bwd(g_seq: Optional[Tensor], g_last: Optional[Tensor], gates: Tensor, hs: Tensor, Tn: int, n: int, h: int, w: int,
                    packed_t: Tensor, hidp: int, dgx: Tensor, dgh: Tensor) -> Optional[Tensor]:
    """The whole backward time loop in one launch (sf_convgru_seq_bwd); returns the split kernel's workspace (first word = sticky
    error word, see ``convgru_seq_fwd``) or None."""
    nbytes = int(lib().sf_convgru_seq_bwd_workspace_bytes(n, h, hidp))
    ws = _hip.sticky_workspace("convgru_seq_bwd", (n, h, hidp), nbytes, gates.device)
    check(lib().sf_convgru_seq_bwd(T(g_seq, hidp), T(g_last, hidp), T(gates), T(hs), Tn, n, h, w, packed_t.data_ptr(), hidp, T(dgx), T(dgh),
                                   ws.data_ptr() if ws is not None else None, nbytes, _hip.SF_BF16, stream_ptr()), "sf_convgru_seq_bwd")
    return ws


def convgru_bwd_gates(dh: Sequence[sfTensor], gates: Tensor, h_prev: Optional[Tensor], hidp: int, dgx: Tensor, dgh: Tensor,
                      dh_direct: Optional[Tensor], amax_gx: Optional[Tensor] = None, amax_gh: Optional[Tensor] = None) -> None:
    """``amax_gx`` / ``amax_gh`` ("f32e" mode): device words the launch raises to max |dgx| / max |dgh| (the caller zeroes them; a word may be shared by
    the launches of a sequence: it then ends at the sequence's maximum)."""
    dh = list(dh) + [NULL] * (3 - len(dh))
    pixels = gates.numel() // gates.shape[-1]
    f32 = dgx.dtype == torch.float32
    check(lib().sf_convgru_bwd_gates(dh[0], dh[1], dh[2], T(gates), T(h_prev, hidp), pixels, hidp, T(dgx, amax=amax_gx if f32 else None),
                                     T(dgh, amax=amax_gh if f32 else None), T(dh_direct, hidp), SF_F32, stream_ptr()), "sf_convgru_bwd_gates")

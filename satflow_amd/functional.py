"""Autograd-aware ops over the C ABI, all on time-major NHWC fp32 tensors with padded channels.

Each ``torch.autograd.Function`` here is plumbing only: it allocates outputs, hands raw pointers
to libsatflow_hip.so on the current stream and wires the matching backward kernels.  No tensor
arithmetic happens in PyTorch on the hot path except where a comment says so.
"""
from __future__ import annotations

import contextlib
import os
from typing import List, Optional, Sequence, Tuple

import torch

from . import kernels as K
from ._hip import NULL, SF_EPI_LINEAR, SF_EPI_SIGMOID, SF_F32, T, check, cpad, generation, lib, require_device, sfTensor, stream_ptr

Tensor = torch.Tensor


# ----------------------------------------------------------------------------------------------
# gradient sink: weight-gradient kernels write straight into the optimizer's flat gradient buffer
# ----------------------------------------------------------------------------------------------
class GradSink:
    """Where the backward kernels of this module put a PARAMETER's gradient when an optimizer has registered a destination for it
    (``optim.FlatAdam``: the parameter's view of the flat gradient buffer): the kernel writes there and the ``Function`` hands autograd ``None``
    for that input - the separate small ``add_`` launch autograd's AccumulateGrad would otherwise issue per parameter and step is never launched.

    Same result as autograd's accumulation, by construction:
    * a destination is handed out ONCE per parameter between two ``zero_grad()`` calls, and only during the first backward pass after
      ``zero_grad()`` (the buffer is zero then: overwriting = accumulating); every later request - a second consumer of the parameter,
      a second backward pass of a gradient-accumulation loop - gets ``None`` and the caller returns its gradient to autograd as before;
    * autograd runs a parameter's AccumulateGrad node after ALL functions that consume the parameter, so an ``add_`` from another
      consumer always lands on top of the written value, never under it;
    * parameters no optimizer registered (tests on bare modules, frozen networks) are not in the table;
    * ANY backward pass that reaches a registered parameter closes its owner until the next ``zero_grad()`` - also a pass that reaches it only through
      plain autograd operations (a tensor hook on every registered parameter arms the same end-of-pass callback; advisor r5: without it
      ``(3 * p).sum().backward()`` followed by a sink-aware backward overwrote the accumulated gradient);
    * a parameter that was frozen after registration (``requires_grad_(False)``) or whose gradient the Function was not asked for
      (``needs_input_grad``) gets no destination: nothing is written into the optimizer's buffer for it.
    ``SF_NO_GRAD_SINK=1`` / ``no_grad_sink()``: every gradient goes through autograd (use around ``torch.autograd.grad`` on registered parameters:
    that call must not touch ``.grad``, and the engine does not tell a Function which of the two it is running under)."""

    def __init__(self) -> None:
        self.table: dict = {}      # (data_ptr, numel) of the parameter -> (gradient view, owner)
        self.taken: dict = {}      # owner -> keys handed out since its zero_grad()
        self.closed: set = set()   # owners whose first backward pass after zero_grad() is over
        self._armed = False
        self._pass_owners: set = set()
        self.off = bool(os.environ.get("SF_NO_GRAD_SINK"))

    def register(self, owner, param: Tensor, grad_view: Tensor) -> None:
        import weakref

        self.table[(param.data_ptr(), param.numel())] = (grad_view, id(owner), weakref.ref(owner), param.untyped_storage().data_ptr())
        oid = id(owner)

        def reached(_g, oid=oid):   # autograd produced a gradient for this parameter in the running pass: the owner's buffer is no longer "zero"
            if oid in self.taken:
                self._arm(oid)
            return None

        if param.requires_grad:
            param.register_hook(reached)
        if id(owner) not in self.taken:
            self.taken[id(owner)] = set()
            weakref.finalize(owner, self._drop_owner, id(owner))   # (the table holds views of the owner's gradient buffer: let go of them with the owner)

    def _drop_owner(self, oid: int) -> None:
        self.table = {k: v for k, v in self.table.items() if v[1] != oid}
        self.taken.pop(oid, None)
        self.closed.discard(oid)

    def unregister(self, owner) -> None:
        self._drop_owner(id(owner))

    def reopen(self, owner) -> None:
        """``zero_grad()`` of ``owner``: its buffer is zero again.  Also drops a pass that never ended (a backward that raised leaves the engine's final
        callbacks unrun: the armed flag would stay set for the rest of the process)."""
        self.taken[id(owner)] = set()
        self.closed.discard(id(owner))
        self._pass_owners.discard(id(owner))
        self._armed = False

    def _arm(self, oid: int) -> None:
        self._pass_owners.add(oid)
        if not self._armed:
            self._armed = True
            torch.autograd.Variable._execution_engine.queue_callback(self._end_of_backward)

    def _end_of_backward(self) -> None:
        self.closed |= self._pass_owners
        self._pass_owners, self._armed = set(), False

    def dest(self, param: Optional[Tensor]) -> Optional[Tensor]:
        if param is None or self.off or not self.table or not param.is_contiguous() or not param.requires_grad:
            return None
        key = (param.data_ptr(), param.numel())
        e = self.table.get(key)
        if e is None:
            return None
        if e[2]() is None or param.untyped_storage().data_ptr() != e[3]:
            # the optimizer is gone, or this tensor merely sits at an address one of its parameters had: not ours
            del self.table[key]
            return None
        if e[1] in self.closed or key in self.taken[e[1]]:
            return None
        self.taken[e[1]].add(key)
        self._arm(e[1])
        return e[0]


GRAD_SINK = GradSink()


@contextlib.contextmanager
def no_grad_sink():
    """Every gradient through autograd inside the block (``torch.autograd.grad`` on parameters an optimizer registered: see ``GradSink``)."""
    prev, GRAD_SINK.off = GRAD_SINK.off, True
    try:
        yield
    finally:
        GRAD_SINK.off = prev


def grad_out(param: Optional[Tensor], shape=None, zero: bool = False, needed: bool = True) -> Tuple[Tensor, Optional[Tensor]]:
    """``(tensor the gradient kernel writes, value the Function returns for that input)`` for parameter ``param`` (fp32; ``shape``: the
    kernel's view of it, e.g. OIHW with the taps split).  Call only from inside ``backward``.  ``needed`` = ``ctx.needs_input_grad`` of that
    input: when autograd did not ask for the gradient the kernel writes into scratch and the Function returns ``None`` (never the optimizer's
    buffer: a parameter frozen after the optimizer registered it must stay untouched)."""
    dst = GRAD_SINK.dest(param) if needed else None
    if dst is None:
        t = (torch.zeros if zero else torch.empty)(tuple(shape) if shape is not None else param.shape, dtype=torch.float32, device=param.device)
        return t, ((t.reshape(param.shape) if shape is not None else t) if needed else None)
    return (dst.view(tuple(shape)) if shape is not None else dst), None


# ----------------------------------------------------------------------------------------------
# parameters regrouped into the matrices the fused kernels take (one launch each way: sf_copy_blocks)
# ----------------------------------------------------------------------------------------------
def _copy_blocks(blocks) -> None:
    """``blocks``: [(src tensor or None, src_row0, src_col0, src_cols_total, dst tensor, dst_row0, dst_col0, dst_cols_total, rows, cols)] on fp32
    tensors seen as 2-D ``[shape[0], numel / shape[0]]``."""
    from ._hip import sfBlock

    if not blocks:
        return
    arr = (sfBlock * len(blocks))()
    for k, (src, sr, sc, sw, dst, dr, dc, dw, rows, cols, *tr) in enumerate(blocks):   # (an 11th entry True: the block lands transposed at (dr, dc))
        arr[k].src = (src.data_ptr() + 4 * (sr * sw + sc)) if src is not None else None
        arr[k].dst = dst.data_ptr() + 4 * (dr * dw + dc)
        arr[k].rows, arr[k].cols, arr[k].src_stride, arr[k].dst_stride = rows, cols, sw, dw
        arr[k].transpose = 1 if tr and tr[0] else 0
    check(lib().sf_copy_blocks(arr, len(blocks), stream_ptr()), "sf_copy_blocks")


class _ParamBlocksFn(torch.autograd.Function):
    """Outputs assembled from two-dimensional blocks of parameters (``torch.cat`` of slices, in one launch).  ``shapes``: the outputs' shapes;
    ``blocks``: ``(param index or None = zeros, src_row0, src_col0, out index, dst_row0, dst_col0, rows, cols)`` with every tensor seen as
    ``[shape[0], numel / shape[0]]``; together the blocks must cover every output exactly once.  Backward: the gradient blocks go back where they
    came from, straight into the parameters' gradient slices when an optimizer registered them (``GRAD_SINK``)."""

    @staticmethod
    def forward(ctx, shapes, blocks, *params):
        dev = params[0].device
        params = tuple(p.contiguous() for p in params)
        outs = tuple(torch.empty(tuple(sh), dtype=torch.float32, device=dev) for sh in shapes)
        width = lambda t: t.numel() // t.shape[0]
        area = [0] * len(outs)
        table = []
        for pi, sr, sc, oi, dr, dc, rows, cols in blocks:
            src = params[pi] if pi is not None else None
            assert src is None or (src.dtype == torch.float32 and sr + rows <= src.shape[0] and sc + cols <= width(src)), (pi, src.shape if src is not None else None)
            assert dr + rows <= outs[oi].shape[0] and dc + cols <= width(outs[oi]), (oi, outs[oi].shape, dr, dc, rows, cols)
            area[oi] += rows * cols
            table.append((src, sr, sc, width(src) if src is not None else 0, outs[oi], dr, dc, width(outs[oi]), rows, cols))
        assert all(a == o.numel() for a, o in zip(area, outs)), "the blocks must cover every output exactly once"
        _copy_blocks(table)
        ctx.blocks, ctx.nout = tuple(blocks), len(outs)
        ctx.save_for_backward(*params)
        return outs

    @staticmethod
    def backward(ctx, *gouts):
        params = ctx.saved_tensors
        width = lambda t: t.numel() // t.shape[0]
        gouts = [g.contiguous() if g is not None else None for g in gouts]
        covered = [0] * len(params)
        for pi, sr, sc, oi, dr, dc, rows, cols in ctx.blocks:
            if pi is not None and gouts[oi] is not None:
                covered[pi] += rows * cols
        dests, rets = [], []
        for k, p in enumerate(params):
            if not ctx.needs_input_grad[2 + k]:
                dests.append(None), rets.append(None)
                continue
            d, r = grad_out(p, zero=covered[k] != p.numel())   # (a sink destination is zero where nothing is written)
            dests.append(d), rets.append(r)
        table = []
        for pi, sr, sc, oi, dr, dc, rows, cols in ctx.blocks:
            if pi is None or dests[pi] is None or gouts[oi] is None:
                continue
            table.append((gouts[oi], dr, dc, width(gouts[oi]), dests[pi], sr, sc, width(params[pi]), rows, cols))
        _copy_blocks(table)
        return (None, None, *rets)


def param_blocks(shapes, blocks, params):
    return _ParamBlocksFn.apply(tuple(tuple(s) for s in shapes), tuple(blocks), *params)


# ----------------------------------------------------------------------------------------------
# layout ops: NCHW-side tensors <-> time-major NHWC
# ----------------------------------------------------------------------------------------------
class _ToNHWC(torch.autograd.Function):
    """``src`` addressed as [nb][nt] images with element strides ``(sb, st, sc)`` -> ``[nt*nb,H,W,Cp]``."""

    @staticmethod
    def forward(ctx, src: Tensor, nb: int, nt: int, c: int, h: int, w: int, strides: Tuple[int, int, int], out_dtype=torch.float32):
        src = src.contiguous()
        ctx.meta = (src.shape, nb, nt, c, h, w, strides)
        return K.to_nhwc(src, nb, nt, c, h, w, strides, out_dtype=out_dtype)

    @staticmethod
    def backward(ctx, g: Tensor):
        shape, nb, nt, c, h, w, strides = ctx.meta
        out = torch.empty(shape, dtype=torch.float32, device=g.device)
        K.from_nhwc(g.contiguous().float(), nb, nt, c, h, w, out, strides)
        return out, None, None, None, None, None, None, None


class _FromNHWC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src: Tensor, shape: Tuple[int, ...], nb: int, nt: int, c: int, h: int, w: int, strides: Tuple[int, int, int]):
        ctx.meta = (nb, nt, c, h, w, strides, src.shape[-1])
        out = torch.empty(shape, dtype=torch.float32, device=src.device)
        return K.from_nhwc(src.contiguous(), nb, nt, c, h, w, out, strides)

    @staticmethod
    def backward(ctx, g: Tensor):
        nb, nt, c, h, w, strides, cp = ctx.meta
        return K.to_nhwc(g.contiguous(), nb, nt, c, h, w, strides, cp), None, None, None, None, None, None, None


def nchw_to_nhwc(x: Tensor) -> Tensor:
    """``[N,C,H,W] -> [N,H,W,Cp]``."""
    n, c, h, w = x.shape
    return _ToNHWC.apply(x, n, 1, c, h, w, (c * h * w, 0, h * w))


def nhwc_to_nchw(x: Tensor, c: int) -> Tensor:
    """``[N,H,W,Cp] -> [N,c,H,W]``."""
    n, h, w, _ = x.shape
    return _FromNHWC.apply(x, (n, c, h, w), n, 1, c, h, w, (c * h * w, 0, h * w))


class _PreprocessFn(torch.autograd.Function):
    """MetNetPreprocessor: ``imgs[B,T,C,H,W] -> frames [T*B, crop, crop, Cp]`` (time-major NHWC), differentiable wrt ``imgs``."""

    @staticmethod
    def forward(ctx, imgs: Tensor, sat: int, crop: int, out_dtype):
        ctx.meta = (imgs.shape, sat, crop)
        return K.metnet_preprocess(imgs, sat, crop, out_dtype)

    @staticmethod
    def backward(ctx, g: Tensor):
        (B, Tn, C, H, W), sat, crop = ctx.meta
        g = g.contiguous()
        dimgs = torch.empty(B, Tn, C, H, W, dtype=torch.float32, device=g.device)
        check(lib().sf_metnet_preprocess_bwd(T(g), B, Tn, C, sat, H, W, crop, dimgs.data_ptr(), SF_F32, stream_ptr()), "sf_metnet_preprocess_bwd")
        return dimgs, None, None, None


def metnet_preprocess(imgs: Tensor, sat: int, crop: int, out_dtype=torch.float32) -> Tensor:
    return _PreprocessFn.apply(imgs.contiguous(), sat, crop, out_dtype)


# ----------------------------------------------------------------------------------------------
# 3x3 convolution (one or two channel-concatenated sources, optional image-index remap per source)
# ----------------------------------------------------------------------------------------------
class ConvEngine:
    """Index maps + packed-weight cache of one ``nn.Conv2d(k=3, padding=1)`` over ``cat(sources)``."""

    def __init__(self, cins: Sequence[int], cout: int) -> None:
        self.cins, self.cout = list(cins), cout
        self.coutp = cpad(cout)
        self.fwd_map = K.linear_map(self.cins, cout)
        self.wgrad_map = K.GemmMap(K._padded(cout), list(self.fwd_map.kmap), 0, self.coutp)
        self._bwd_maps = {}
        self._key, self._packed = None, {}
        # Parameters a DERIVED weight (slice / cat: a fresh tensor every call) was built from.  When set, the cache keys on
        # their identity + version instead of on the derived tensor's (whose version is always 0 and whose address the
        # caching allocator may hand back unchanged after the source changed).
        self.key_tensors: Optional[Tuple[Tensor, ...]] = None

    def bwd_map(self, need: Tuple[bool, ...]) -> K.GemmMap:
        if need not in self._bwd_maps:
            self._bwd_maps[need] = K.linear_bwd_map(self.cins, self.cout, need)
        return self._bwd_maps[need]

    def packed(self, weight: Tensor, bias: Optional[Tensor], kind, need: Tuple[bool, ...] = ()):
        src = self.key_tensors if self.key_tensors is not None else (weight, bias)
        key = (tuple(None if t is None else (t.data_ptr(), t._version) for t in src), generation())
        if key != self._key:
            self._key, self._packed = key, {}
        k = (kind, need)
        if k not in self._packed:
            w4 = weight.reshape(weight.shape[0], weight.shape[1], 3, 3)
            if kind == "fwd":
                self._packed[k] = K.pack_weights(w4, bias, self.fwd_map, transpose=False)
            else:
                self._packed[k] = K.pack_weights(w4, None, self.bwd_map(need), transpose=True)
        return self._packed[k]


class WeightGradBatch:
    """Weight gradient of ONE convolution weight applied several times (a recurrent cell's state convolution over the frames of a sequence),
    computed once over all applications instead of once per application: every application's backward pass hands in its (input, output
    gradient) pair and returns no weight gradient, the one that completes the set runs the weight-gradient kernel over the concatenation.
    Per-frame launches on a few small images are bound by WRITING the gradient (75 MB for a 2048 -> 1024 kernel), and autograd then adds the
    per-frame results pairwise.  If a backward pass ends with applications missing (an output that did not reach the loss) it raises
    instead of dropping the gradient."""

    def __init__(self) -> None:
        self.uses = 0
        self.xs: List[Tensor] = []
        self.gys: List[Tensor] = []
        self._armed = False

    def register(self) -> None:
        self.uses += 1

    def add(self, x: Tensor, gy: Tensor) -> bool:
        if not self._armed:
            self._armed = True
            torch.autograd.Variable._execution_engine.queue_callback(self._end_of_backward)
        self.xs.append(x)
        self.gys.append(gy)
        return len(self.xs) == self.uses

    def take(self) -> Tuple[Tensor, Tensor]:
        """(inputs, output gradients) of all applications, concatenated along the image axis in ONE order (a weight gradient sums over images: any).  Inputs
        that lie back to back in one storage - a recurrent layer's states written into ``functional_gan.sequence_slots`` - are read in place: the pairs
        are ordered by the inputs' addresses and, if those are consecutive, the input is a tensor on that storage span instead of a copy of it."""
        xs, gys = self.xs, self.gys
        self.xs, self.gys = [], []
        order = sorted(range(len(xs)), key=lambda i: xs[i].data_ptr())
        xs, gys = [xs[i] for i in order], [gys[i] for i in order]
        esize = xs[0].element_size()
        consecutive = len(xs) > 1 and all(
            a.is_contiguous() and b.is_contiguous() and a.dtype == b.dtype and a.shape[1:] == b.shape[1:]
            and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr() and a.data_ptr() + a.numel() * esize == b.data_ptr()
            for a, b in zip(xs, xs[1:]))
        if consecutive:
            x = torch.empty(0, dtype=xs[0].dtype, device=xs[0].device).set_(xs[0].untyped_storage(), xs[0].storage_offset(),
                                                                             (sum(t.shape[0] for t in xs), *xs[0].shape[1:]))
        else:
            x = torch.cat(xs, 0)
        gsize = gys[0].element_size()
        g_consecutive = len(gys) > 1 and all(
            a.is_contiguous() and b.is_contiguous() and a.dtype == b.dtype and a.shape[1:] == b.shape[1:]
            and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr() and a.data_ptr() + a.numel() * gsize == b.data_ptr()
            for a, b in zip(gys, gys[1:]))
        if g_consecutive:   # (the frames' gradients written into functional_gan.GradSlots: in place too)
            g = torch.empty(0, dtype=gys[0].dtype, device=gys[0].device).set_(gys[0].untyped_storage(), gys[0].storage_offset(),
                                                                              (sum(t.shape[0] for t in gys), *gys[0].shape[1:]))
        else:
            g = torch.cat(gys, 0)
        return x, g

    def _end_of_backward(self) -> None:
        left, self._armed = len(self.xs), False
        self.xs, self.gys = [], []
        if left:
            raise RuntimeError(f"WeightGradBatch: the backward pass ended with {left} of {self.uses} applications reported - an application's "
                               "output did not reach the loss; its weight gradient cannot be batched")


class _ConvFn(torch.autograd.Function):
    """``y = act(conv3x3(cat(x0, x1)) + b)``: x_i ``[N_i,H,W,C_ip]`` -> ``[n,H,W,Coutp]``.

    ``remap_i = (idiv, imod)``: kernel image j reads image ``(j // idiv) % imod`` of source i, so a
    source with fewer images is broadcast (MetNet's lead-time axis) instead of copied.  Gradients
    are produced only for non-remapped sources (the remapped ones here are inputs / constants).
    """

    @staticmethod
    def forward(ctx, eng: ConvEngine, x0: Tensor, x1: Optional[Tensor], weight: Tensor, bias: Optional[Tensor], n: int,
                remap0: Tuple[int, int], remap1: Tuple[int, int], sigmoid: bool, out_dtype=None, stats: Optional[Tensor] = None,
                wbatch: Optional[WeightGradBatch] = None):
        H, W = x0.shape[1], x0.shape[2]
        ctx.wbatch = wbatch
        if wbatch is not None:
            assert x1 is None and remap0 == (0, 0) and not sigmoid, "a batched weight gradient takes plain single-source applications"
            wbatch.register()
        packed, bp = eng.packed(weight, bias, "fwd")
        y = torch.empty(n, H, W, eng.coutp, dtype=out_dtype or torch.float32, device=x0.device)
        s0 = T(x0, idiv=remap0[0], imod=remap0[1])
        s1 = T(x1, idiv=remap1[0], imod=remap1[1]) if x1 is not None else NULL
        K.conv3x3(s0, s1, n, H, W, packed, bp, eng.fwd_map, T(y), SF_EPI_SIGMOID if sigmoid else SF_EPI_LINEAR, stats)
        ctx.eng, ctx.sigmoid, ctx.n, ctx.remaps = eng, sigmoid, n, (remap0, remap1)
        ctx.has = (x1 is not None, bias is not None)
        ctx.save_for_backward(x0, x1 if x1 is not None else x0.new_empty(0), y if sigmoid else x0.new_empty(0), weight)
        ctx.bias = bias  # only to keep the packed-weight cache key stable in backward
        return y

    @staticmethod
    def backward(ctx, gy: Tensor):
        eng: ConvEngine = ctx.eng
        x0, x1, y, weight = ctx.saved_tensors
        has_x1, has_bias = ctx.has
        n, (remap0, remap1) = ctx.n, ctx.remaps
        H, W = x0.shape[1], x0.shape[2]
        gy = gy.contiguous()
        if ctx.sigmoid:
            if gy.dtype == torch.float32 and y.dtype == torch.float32 and y.is_contiguous() and gy.numel() % 4 == 0:
                g2 = torch.empty_like(gy)
                check(lib().sf_sigmoid_bwd(gy.data_ptr(), y.data_ptr(), gy.numel(), g2.data_ptr(), stream_ptr()), "sf_sigmoid_bwd")
                gy = g2
            else:
                gy = gy * y * (1.0 - y)  # pointwise torch op on the small head tensor only
        if x0.dtype == torch.bfloat16 and gy.dtype != torch.bfloat16:
            # bf16-stored sources take a bf16-stored output gradient (weight-gradient kernel).  Exact: both kernels below
            # round the gradient to bf16 when they build their MFMA operands anyway.
            gy = gy.to(torch.bfloat16)
        need = (ctx.needs_input_grad[1], has_x1 and ctx.needs_input_grad[2])
        if (need[0] and remap0 != (0, 0)) or (need[1] and remap1 != (0, 0)):
            raise RuntimeError("gradient wrt a broadcast (image-remapped) convolution source is not implemented")
        d0 = d1 = None
        gyT = K.grad_operand(gy)   # ("f32e": with the gradient's amax word - one pass here serves the input AND the weight gradient)
        if any(need):
            needk = need if has_x1 else need[:1]
            gm = eng.bwd_map(tuple(needk))
            lanes = sum(cpad(c) for c, nd in zip(eng.cins, needk) if nd)
            dcat = torch.empty(n, H, W, lanes, dtype=x0.dtype, device=gy.device)  # a gradient is stored like its tensor
            K.conv3x3(gyT, NULL, n, H, W, eng.packed(weight, ctx.bias, "bwd", tuple(needk))[0], None, gm, T(dcat))
            if need[0] and need[1]:
                c0p = cpad(eng.cins[0])
                d0, d1 = dcat[..., :c0p].contiguous(), dcat[..., c0p:].contiguous()
            elif need[0]:
                d0 = dcat
            else:
                d1 = dcat
        if not (ctx.needs_input_grad[3] or (has_bias and ctx.needs_input_grad[4])):
            return None, d0, d1, None, None, None, None, None, None, None, None, None  # frozen parameters (a GAN's other network): no weight gradient
        if ctx.wbatch is not None:
            if not ctx.wbatch.add(x0, gy):
                return None, d0, d1, None, None, None, None, None, None, None, None, None   # a later application's backward completes the set
            x0, gy = ctx.wbatch.take()
            n = x0.shape[0]
            gyT = K.grad_operand(gy)
        dw4, dw_ret = grad_out(weight, (weight.shape[0], weight.shape[1], 3, 3), needed=ctx.needs_input_grad[3])
        db, db_ret = grad_out(ctx.bias, needed=ctx.needs_input_grad[4]) if has_bias else (None, None)
        s0 = T(x0, idiv=remap0[0], imod=remap0[1])
        s1 = T(x1, idiv=remap1[0], imod=remap1[1]) if has_x1 else NULL
        K.conv3x3_bwd_weight(s0, s1, gyT, n, H, W, eng.wgrad_map, dw4, db, accumulate=False)
        return None, d0, d1, dw_ret, db_ret, None, None, None, None, None, None, None


class ConvStats:
    """Per-tile output statistics of a convolution (see ``sf_conv3x3_fwd_stats``), handed to the BatchNorm behind it."""

    def __init__(self, n: int, h: int, w: int, np_: int, device) -> None:
        self.tiles = int(lib().sf_conv3x3_stats_tiles(h, w))
        self.np, self.n = np_, n
        self.data = torch.empty(n * self.tiles, np_, 2, dtype=torch.float32, device=device)


_WBATCH_SCOPE: List[Optional[dict]] = [None]


@contextlib.contextmanager
def batched_weight_grads():
    """Scope for the FORWARD pass of a cell unrolled over time: every plain ``conv3x3`` application of the same ``ConvEngine`` (= the same layer)
    inside it shares one ``WeightGradBatch``, so the backward pass computes each layer's weight gradient once over all time steps instead of
    once per step (T launches writing the whole gradient each, then T - 1 adds by autograd).  The weight handed to the applications must be
    the same function of the same parameters every time (a cell's layer: yes).  Every application's output must reach the loss
    (``WeightGradBatch`` raises at the end of a backward pass otherwise)."""
    prev, _WBATCH_SCOPE[0] = _WBATCH_SCOPE[0], {}
    try:
        yield
    finally:
        _WBATCH_SCOPE[0] = prev


def conv3x3(eng: ConvEngine, x: Tensor, weight: Tensor, bias: Optional[Tensor], sigmoid: bool = False, out_dtype=None,
            want_stats: Optional[bool] = None, wbatch: Optional[WeightGradBatch] = None):
    """``out_dtype=torch.bfloat16`` stores the result as bf16 (SF_BF16 kernels only; "bf16a" encoder mode).
    ``want_stats`` not None: returns ``(y, stats)`` with ``stats`` a ``ConvStats`` for ``batchnorm(..., stats=)`` when it is
    true and the bf16 or f32e kernels run, else None."""
    if want_stats is None:
        scope = _WBATCH_SCOPE[0]
        if wbatch is None and scope is not None and not sigmoid and torch.is_grad_enabled() and weight.requires_grad:
            wbatch = scope.setdefault(id(eng), WeightGradBatch())
        return _ConvFn.apply(eng, x, None, weight, bias, x.shape[0], (0, 0), (0, 0), sigmoid, out_dtype, None, wbatch)
    from ._hip import SF_BF16, SF_F32E, compute_dtype

    st = None
    if want_stats and not sigmoid and (compute_dtype() == SF_BF16 or (compute_dtype() == SF_F32E and not os.environ.get("SF_F32E_NO_STATS"))):
        st = ConvStats(x.shape[0], x.shape[1], x.shape[2], eng.fwd_map.Np, x.device)
    y = _ConvFn.apply(eng, x, None, weight, bias, x.shape[0], (0, 0), (0, 0), sigmoid, out_dtype, st.data if st is not None else None)
    return y, st


def conv3x3_broadcast(eng: ConvEngine, x0: Tensor, x1: Tensor, weight: Tensor, bias: Optional[Tensor], n: int,
                      remap0: Tuple[int, int], remap1: Tuple[int, int]) -> Tensor:
    return _ConvFn.apply(eng, x0, x1, weight, bias, n, remap0, remap1, False)


# ----------------------------------------------------------------------------------------------
# max pooling / batch norm
# ----------------------------------------------------------------------------------------------
class _MaxPoolFn(torch.autograd.Function):
    """2x2 max-pooling; when a gradient will be needed the forward pass records the routing (2 bits per element) instead of keeping
    the input alive: the backward pass reads the routing, not the input (``sf_maxpool2_route_fwd/bwd``; SF_POOL_RECOMPUTE=1: the
    argmax-recomputing kernels)."""

    @staticmethod
    def forward(ctx, x: Tensor, perm: Optional[Tuple[int, int]], out_dtype=None, drop=None):
        ctx.perm, ctx.drop = perm, drop
        if ctx.needs_input_grad[0] and x.shape[-1] % 8 == 0 and not os.environ.get("SF_POOL_RECOMPUTE"):
            y, route = K.maxpool2_route_fwd(x, perm, out_dtype, drop)
            ctx.meta = (tuple(x.shape), x.dtype)
            ctx.save_for_backward(route)
            return y
        ctx.meta = None
        ctx.save_for_backward(x)
        return K.maxpool2_fwd(x, perm, out_dtype, drop)

    @staticmethod
    def backward(ctx, gy: Tensor):
        (t,) = ctx.saved_tensors
        if ctx.meta is not None:
            return K.maxpool2_route_bwd(t, gy.contiguous(), ctx.meta[0], ctx.meta[1], ctx.perm, ctx.drop), None, None, None
        return K.maxpool2_bwd(t, gy.contiguous(), ctx.perm, ctx.drop), None, None, None


def _draw_seeds() -> Tuple[int, int]:
    seeds = torch.randint(0, 2**62, (2,), dtype=torch.int64).tolist()  # host RNG: follows torch.manual_seed, no device sync
    return seeds[0], seeds[1]


def maxpool2(x: Tensor, perm: Optional[Tuple[int, int]] = None, out_dtype=None, dropout: Optional[Tuple[float, float, int]] = None) -> Tensor:
    """``out_dtype=torch.float32`` on a bf16 input is the encoder's exit back to fp32 storage (max is exact in either type).
    ``dropout=(p1, p2, period)``: ``dropout2(maxpool2(x), p1, p2, period)`` in the same pass (fresh seeds from the torch RNG)."""
    drop = None
    if dropout is not None and (dropout[0] > 0 or dropout[1] > 0):
        drop = (float(dropout[0]), float(dropout[1]), int(dropout[2]), *_draw_seeds())
    return _MaxPoolFn.apply(x, perm, out_dtype, drop)


class _LeadTimePoolFn(torch.autograd.Function):
    """``pooled[l*F + f] = maxpool2(base[f] + P_l(w1))`` for all lead times l (see ``sf_leadtime_pool_fwd``)."""

    @staticmethod
    def forward(ctx, base: Tensor, w1: Tensor, cimg: int, L: int, stats: Optional[Tensor] = None):
        Fr, H, W, C = base.shape
        w1 = w1.contiguous()
        ws = torch.empty(lib().sf_leadtime_pool_workspace_floats(L, C), dtype=torch.float32, device=base.device)
        out = torch.empty(L * Fr, H // 2, W // 2, C, dtype=base.dtype, device=base.device)
        if stats is not None:  # also the per-workgroup sums the BatchNorm behind the pooling needs
            check(lib().sf_leadtime_pool_fwd_stats(T(base), Fr, H, W, w1.data_ptr(), w1.shape[0], w1.shape[1], cimg, L, ws.data_ptr(), T(out),
                                                   stats.data_ptr(), SF_F32, stream_ptr()), "sf_leadtime_pool_fwd_stats")
        else:
            check(lib().sf_leadtime_pool_fwd(T(base), Fr, H, W, w1.data_ptr(), w1.shape[0], w1.shape[1], cimg, L, ws.data_ptr(), T(out), SF_F32,
                                             stream_ptr()), "sf_leadtime_pool_fwd")
        ctx.meta = (cimg, L)
        ctx.save_for_backward(base, w1)
        return out

    @staticmethod
    def backward(ctx, g: Tensor):
        base, w1 = ctx.saved_tensors
        cimg, L = ctx.meta
        Fr, H, W, C = base.shape
        g = g.contiguous()
        ws = torch.empty(lib().sf_leadtime_pool_workspace_floats(L, C), dtype=torch.float32, device=base.device)
        dbase = torch.empty_like(base)
        dw1, dw1_ret = grad_out(w1, zero=True)  # only the one-hot columns cimg..cimg+L-1 are written (a sink destination is zero already)
        check(lib().sf_leadtime_pool_bwd(T(base), T(g), Fr, H, W, w1.data_ptr(), w1.shape[0], w1.shape[1], cimg, L, ws.data_ptr(), T(dbase),
                                         dw1.data_ptr(), SF_F32, stream_ptr()), "sf_leadtime_pool_bwd")
        return dbase, dw1_ret, None, None, None


class PoolStats:
    """Per-workgroup output statistics of the lead-time pooling (``sf_leadtime_pool_fwd_stats``) in the record layout of
    ``ConvStats``: group (lead time) l owns records ``[l * tiles, (l + 1) * tiles)``."""

    def __init__(self, L: int, C: int, device) -> None:
        self.tiles = int(lib().sf_leadtime_pool_stats_tiles())
        self.np, self.n = C, L
        self.data = torch.empty(L * self.tiles, C, 2, dtype=torch.float32, device=device)


def leadtime_pool(base: Tensor, w1: Tensor, cimg: int, L: int, want_stats: bool = False):
    """``want_stats``: returns ``(pooled, PoolStats)`` for the BatchNorm behind the pooling (L <= 12 and an LDS-sized border table,
    else ``(pooled, None)``)."""
    if not want_stats:
        return _LeadTimePoolFn.apply(base, w1, cimg, L)
    C = base.shape[-1]
    st = PoolStats(L, C, base.device) if L <= 12 and L * 11 * C * 4 <= 160 * 1024 else None
    return _LeadTimePoolFn.apply(base, w1, cimg, L, st.data if st is not None else None), st


class _BatchNormTrainFn(torch.autograd.Function):
    """Training-mode BatchNorm2d with ``groups`` independent statistics sets (one per lead time)."""

    @staticmethod
    def forward(ctx, x: Tensor, gamma: Tensor, beta: Tensor, running_mean: Optional[Tensor], running_var: Optional[Tensor],
                groups: int, eps: float, momentum: float, conv_stats=None):
        C = x.shape[-1]
        creal = gamma.shape[0]
        pixels = x.numel() // C
        assert pixels % groups == 0
        dev = x.device
        stats = torch.empty(4, groups, C, dtype=torch.float32, device=dev)  # mean, rstd, scale, shift
        sums = torch.empty(groups, 2, C, dtype=torch.float64, device=dev)
        y = torch.empty_like(x)
        rm = running_mean.data_ptr() if running_mean is not None else None
        rv = running_var.data_ptr() if running_var is not None else None
        stat_ptrs = (stats[0].data_ptr(), stats[1].data_ptr(), stats[2].data_ptr(), stats[3].data_ptr())
        if conv_stats is None:
            check(lib().sf_batchnorm_train_fwd(T(x), pixels // groups, groups, creal, gamma.data_ptr(), beta.data_ptr(), eps, momentum, rm, rv,
                                               *stat_ptrs, sums.data_ptr(), T(y), SF_F32, stream_ptr()), "sf_batchnorm_train_fwd")
        else:  # statistics from the producing convolution's epilogue: x is not read a first time
            assert conv_stats.n % groups == 0 and conv_stats.np >= C
            check(lib().sf_batchnorm_train_fwd_stats(T(x), pixels // groups, groups, creal, gamma.data_ptr(), beta.data_ptr(), eps, momentum, rm, rv,
                                                     *stat_ptrs, sums.data_ptr(), conv_stats.data.data_ptr(),
                                                     conv_stats.tiles * (conv_stats.n // groups), conv_stats.np, T(y), SF_F32, stream_ptr()),
                  "sf_batchnorm_train_fwd_stats")
        ctx.groups, ctx.creal = groups, creal
        ctx.beta = beta  # (identity only: where its gradient goes)
        ctx.save_for_backward(x, gamma, stats)
        return y

    @staticmethod
    def backward(ctx, gy: Tensor):
        x, gamma, stats = ctx.saved_tensors
        C = x.shape[-1]
        pixels = x.numel() // C
        gy = gy.contiguous()
        dx = torch.empty_like(x)
        sums = torch.empty(ctx.groups, 2, C, dtype=torch.float64, device=x.device)
        (dgamma, dgamma_ret), (dbeta, dbeta_ret) = grad_out(gamma), grad_out(ctx.beta)
        coef = torch.empty(ctx.groups, 3, C, dtype=torch.float32, device=x.device)
        # "f32e": dx is the gradient operand of the convolution in front of this BatchNorm - its scale word is raised by the apply pass itself
        word = K.scale_words(1, x.device) if dx.dtype == torch.float32 else None
        check(lib().sf_batchnorm_train_bwd(T(x), T(gy), pixels // ctx.groups, ctx.groups, ctx.creal, gamma.data_ptr(), stats[0].data_ptr(),
                                           stats[1].data_ptr(), sums.data_ptr(), coef.data_ptr(), T(dx, amax=word), dgamma.data_ptr(), dbeta.data_ptr(),
                                           SF_F32, stream_ptr()), "sf_batchnorm_train_bwd")
        return K.tag_amax(dx, word), dgamma_ret, dbeta_ret, None, None, None, None, None, None


class _BNConvFn(torch.autograd.Function):
    """``conv3x3(batchnorm_train(x))`` with the normalisation folded into the convolution (bf16-stored activations, SF_BF16 kernels):
    ``conv(a_g x + b_g) = conv_{W a_g}(x) + T_g[border class]`` - the normalised tensor is never written.  Backward: the plain input
    gradient (unscaled W) through the BatchNorm backward; the weight gradient from the un-normalised ``x``
    (``sf_conv3x3_bwd_weight_folded``)."""

    @staticmethod
    def forward(ctx, eng: ConvEngine, x: Tensor, gamma: Tensor, beta: Tensor, running_mean: Optional[Tensor], running_var: Optional[Tensor],
                groups: int, eps: float, momentum: float, in_stats, weight: Tensor, bias: Optional[Tensor], out_dtype, out_stats: Optional[Tensor],
                pool=None):
        """``pool = (perm, drop)``: the 2x2 max-pooling behind the convolution (and MetNet's two dropouts, ``drop = (p1, p2, period, seed1, seed2)`` or
        None) taken in the convolution's epilogue - the result is the POOLED tensor, the backward pass first routes its gradient back
        (``batchnorm_conv3x3_maxpool`` checks that the launch qualifies)."""
        n, H, W, C = x.shape
        creal = gamma.shape[0]
        pixels = n * H * W
        dev = x.device
        gm = eng.fwd_map
        assert gm.Kp == C and x.dtype == torch.bfloat16 and pixels % groups == 0
        stats = torch.empty(4, groups, C, dtype=torch.float32, device=dev)  # mean, rstd, scale, shift
        sums = torch.empty(groups, 2, C, dtype=torch.float64, device=dev)
        rm = running_mean.data_ptr() if running_mean is not None else None
        rv = running_var.data_ptr() if running_var is not None else None
        stat_ptrs = (stats[0].data_ptr(), stats[1].data_ptr(), stats[2].data_ptr(), stats[3].data_ptr())
        if in_stats is None:
            check(lib().sf_batchnorm_train_fwd(T(x), pixels // groups, groups, creal, gamma.data_ptr(), beta.data_ptr(), eps, momentum, rm, rv,
                                               *stat_ptrs, sums.data_ptr(), NULL, SF_F32, stream_ptr()), "sf_batchnorm_train_fwd")
        else:
            assert in_stats.n % groups == 0 and in_stats.np >= C
            check(lib().sf_batchnorm_train_fwd_stats(T(x), pixels // groups, groups, creal, gamma.data_ptr(), beta.data_ptr(), eps, momentum, rm, rv,
                                                     *stat_ptrs, sums.data_ptr(), in_stats.data.data_ptr(), in_stats.tiles * (in_stats.n // groups),
                                                     in_stats.np, NULL, SF_F32, stream_ptr()), "sf_batchnorm_train_fwd_stats")
        w4 = weight.reshape(weight.shape[0], weight.shape[1], 3, 3)
        packed, tab = K.conv3x3_fold_pack(w4, bias, gm, stats[2], stats[3])
        ctx.eng, ctx.groups, ctx.creal, ctx.has_bias = eng, groups, creal, bias is not None
        ctx.bias, ctx.beta = bias, beta
        ctx.pool = pool
        if pool is not None:
            assert out_stats is None
            y, route = K.conv3x3_folded_pool(T(x), n, H, W, packed, tab, gm, eng.coutp, pool[0], pool[1], dev)
            ctx.save_for_backward(x, gamma, stats, weight, route)
            return y
        y = torch.empty(n, H, W, eng.coutp, dtype=out_dtype or torch.bfloat16, device=dev)
        K.conv3x3_folded(T(x), n, H, W, packed, tab, gm, T(y), out_stats)
        ctx.save_for_backward(x, gamma, stats, weight)
        return y

    @staticmethod
    def backward(ctx, gy: Tensor):
        eng: ConvEngine = ctx.eng
        if ctx.pool is not None:   # the gradient of the POOLED tensor: dropout masks + routing first (sf_maxpool2_route_bwd reads the recorded codes)
            x, gamma, stats, weight, route = ctx.saved_tensors
            gy = gy.contiguous()
            if gy.dtype != torch.bfloat16:
                gy = gy.to(torch.bfloat16)
            # (without dropout in the routing kernel the weight gradient can build its sparse operand from the pooled gradient + codes themselves)
            # (... after the dropout masks, which the routing kernel applies on the way: it leaves the masked pooled gradient behind as well)
            dropped = ctx.pool[1] is not None and (ctx.pool[1][0] > 0 or ctx.pool[1][1] > 0)
            use_pooled = K.conv3x3_bwd_weight_pooled_supported(eng.coutp, x.shape[-1], x.shape[0], x.shape[1], x.shape[2], ctx.groups)
            masked = torch.empty_like(gy) if dropped and use_pooled else None
            pooled = (masked if masked is not None else gy, route, ctx.pool[0]) if use_pooled else None
            gy = K.maxpool2_route_bwd(route, gy, (x.shape[0], x.shape[1], x.shape[2], eng.coutp), torch.bfloat16, ctx.pool[0], ctx.pool[1], masked)
        else:
            x, gamma, stats, weight = ctx.saved_tensors
            pooled = None
        n, H, W, C = x.shape
        groups = ctx.groups
        gy = gy.contiguous()
        if gy.dtype != torch.bfloat16:
            gy = gy.to(torch.bfloat16)
        dev = gy.device
        dw4, dw_ret = grad_out(weight, (weight.shape[0], weight.shape[1], 3, 3))
        db, db_ret = grad_out(ctx.bias) if ctx.has_bias else (None, None)
        dx = torch.empty_like(x)
        sums = torch.empty(groups, 2, C, dtype=torch.float64, device=dev)
        (dgamma, dgamma_ret), (dbeta, dbeta_ret) = grad_out(gamma), grad_out(ctx.beta)
        coef = torch.empty(groups, 3, C, dtype=torch.float32, device=dev)
        gm = eng.bwd_map((True,))
        packed_t = eng.packed(weight, ctx.bias, "bwd", (True,))[0]
        if os.environ.get("SF_BN_BWD_PASSES"):  # A/B switch: the three-kernel form (input gradient, reduction pass, apply pass)
            dn = torch.empty(n, H, W, C, dtype=x.dtype, device=dev)
            K.conv3x3(T(gy), NULL, n, H, W, packed_t, None, gm, T(dn))
            check(lib().sf_batchnorm_train_bwd(T(x), T(dn), (n * H * W) // groups, groups, ctx.creal, gamma.data_ptr(), stats[0].data_ptr(),
                                               stats[1].data_ptr(), sums.data_ptr(), coef.data_ptr(), T(dx), dgamma.data_ptr(), dbeta.data_ptr(),
                                               SF_F32, stream_ptr()), "sf_batchnorm_train_bwd")
            K.conv3x3_bwd_weight_folded(T(x), T(gy), n, H, W, eng.wgrad_map, stats[2], stats[3], dw4, db)
        else:
            # The weight gradient first: its per-group partial results also give the BatchNorm backward's two reductions
            # (sum dn = W . V_g, sum dn * x = W . dWraw_g), so d(normalised input) = conv^T(gy, W) is never materialised - the input
            # gradient convolution applies dx = A dn + B x + K in its epilogue.
            w4 = weight.reshape(weight.shape[0], weight.shape[1], 3, 3)
            K.conv3x3_bwd_weight_folded(T(x), T(gy), n, H, W, eng.wgrad_map, stats[2], stats[3], dw4, db, bn=(w4, stats[0], stats[1], sums),
                                        pooled_gradient=ctx.pool is not None, pooled=pooled)   # (gy then comes out of maxpool2_route_bwd just above)
            check(lib().sf_batchnorm_train_bwd_coef(sums.data_ptr(), (n * H * W) // groups, groups, C, ctx.creal, gamma.data_ptr(), stats[0].data_ptr(),
                                                    stats[1].data_ptr(), coef.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), SF_F32, stream_ptr()),
                  "sf_batchnorm_train_bwd_coef")
            K.conv3x3_bwd_data_bn(T(gy), n, H, W, packed_t, gm, T(x), coef, T(dx))
        return None, dx, dgamma_ret, dbeta_ret, None, None, None, None, None, None, dw_ret, db_ret, None, None, None


def bn_fold_enabled() -> bool:
    """The BatchNorm -> Conv2d fold runs with bf16-stored activations ("bf16a") unless SF_NO_BN_FOLD is set (A/B switch)."""
    from ._hip import SF_BF16, compute_dtype

    return compute_dtype() == SF_BF16 and not os.environ.get("SF_NO_BN_FOLD")


def batchnorm_conv3x3(x: Tensor, bn: torch.nn.BatchNorm2d, groups: int, in_stats: Optional["ConvStats"], eng: ConvEngine, weight: Tensor,
                      bias: Optional[Tensor], out_dtype=None, want_stats: bool = False):
    """Training-mode ``conv3x3(bn(x))`` -> ``(y, stats)`` (``stats``: ConvStats of y when ``want_stats``).  Folded (no normalised
    tensor) when x is bf16-stored and the shape allows, else the two separate ops."""
    n, H, W, _ = x.shape
    if not (x.dtype == torch.bfloat16 and bn_fold_enabled() and K.conv3x3_fold_supported(n, H, W, eng.fwd_map, groups, want_stats)):
        y = batchnorm(x, bn, groups, True, in_stats)
        return conv3x3(eng, y, weight, bias, out_dtype=out_dtype, want_stats=want_stats)
    momentum = bn.momentum
    if bn.track_running_stats and bn.num_batches_tracked is not None:
        if momentum is None:
            momentum = -(float(bn.num_batches_tracked) + 1.0)
        bn.num_batches_tracked += groups
    st = ConvStats(n, H, W, eng.fwd_map.Np, x.device) if want_stats else None
    y = _BNConvFn.apply(eng, x, bn.weight, bn.bias, bn.running_mean, bn.running_var, groups, bn.eps, momentum if momentum is not None else 0.0,
                        in_stats, weight, bias, out_dtype, st.data if st is not None else None)
    return y, st


def batchnorm_conv3x3_maxpool(x: Tensor, bn: torch.nn.BatchNorm2d, groups: int, in_stats: Optional["ConvStats"], eng: ConvEngine, weight: Tensor,
                              bias: Optional[Tensor], perm: Optional[Tuple[int, int]] = None, out_dtype=None,
                              dropout: Optional[Tuple[float, float, int]] = None) -> Tensor:
    """Training-mode ``dropout2(maxpool2(conv3x3(bn(x))))`` (the DownSampler's conv4 -> pooling pair with MetNet's two dropouts behind it): ONE
    convolution launch with the pooling in its epilogue when the folded one-wave-per-SIMD kernel takes the shape (bf16-stored x and result),
    otherwise ``batchnorm_conv3x3`` + ``maxpool2``.  Same values either way: the maximum of the bf16-rounded convolution results, the same routing
    record for the backward pass, the same dropout masks."""
    n, H, W, _ = x.shape
    fused = (x.dtype == torch.bfloat16 and (out_dtype or x.dtype) == torch.bfloat16 and bn_fold_enabled() and torch.is_grad_enabled()
             and K.conv3x3_fold_supported(n, H, W, eng.fwd_map, groups, False)
             and K.conv3x3_folded_pool_supported(n, H, W, eng.fwd_map, eng.coutp, groups))
    if not fused:
        y, _ = batchnorm_conv3x3(x, bn, groups, in_stats, eng, weight, bias, out_dtype=x.dtype)
        return maxpool2(y, perm, out_dtype=out_dtype, dropout=dropout)
    momentum = bn.momentum
    if bn.track_running_stats and bn.num_batches_tracked is not None:
        if momentum is None:
            momentum = -(float(bn.num_batches_tracked) + 1.0)
        bn.num_batches_tracked += groups
    drop = None
    if dropout is not None and (dropout[0] > 0 or dropout[1] > 0):
        drop = (float(dropout[0]), float(dropout[1]), int(dropout[2]), *_draw_seeds())
    return _BNConvFn.apply(eng, x, bn.weight, bn.bias, bn.running_mean, bn.running_var, groups, bn.eps, momentum if momentum is not None else 0.0,
                           in_stats, weight, bias, torch.bfloat16, None, (perm, drop))


class _BatchNormEvalFn(torch.autograd.Function):
    """Eval-mode BatchNorm2d (running statistics are constants): ``y = a*x + b``; backward ``dx = a*dy``, ``dgamma``, ``dbeta``."""

    @staticmethod
    def forward(ctx, x: Tensor, gamma: Tensor, beta: Tensor, running_mean: Tensor, running_var: Tensor, eps: float):
        C = x.shape[-1]
        ab = torch.empty(2, C, dtype=torch.float32, device=x.device)
        y = torch.empty_like(x)
        check(lib().sf_batchnorm_eval_fwd(T(x), x.numel() // C, gamma.shape[0], gamma.data_ptr(), beta.data_ptr(), eps, running_mean.data_ptr(),
                                          running_var.data_ptr(), ab[0].data_ptr(), ab[1].data_ptr(), T(y), SF_F32, stream_ptr()),
              "sf_batchnorm_eval_fwd")
        ctx.eps = eps
        ctx.save_for_backward(x, gamma, running_mean, running_var)
        return y

    @staticmethod
    def backward(ctx, gy: Tensor):
        x, gamma, rm, rv = ctx.saved_tensors
        C = x.shape[-1]
        gy = gy.contiguous()
        if gy.dtype != x.dtype:
            gy = gy.to(x.dtype)
        dx = torch.empty_like(x)
        sums = torch.empty(2, C, dtype=torch.float64, device=x.device)
        scratch = torch.empty(5, C, dtype=torch.float32, device=x.device)
        dgamma, dbeta = torch.empty_like(gamma), torch.empty_like(gamma)
        check(lib().sf_batchnorm_eval_bwd(T(x), T(gy), x.numel() // C, gamma.shape[0], gamma.data_ptr(), ctx.eps, rm.data_ptr(), rv.data_ptr(),
                                          sums.data_ptr(), scratch.data_ptr(), T(dx), dgamma.data_ptr(), dbeta.data_ptr(), SF_F32, stream_ptr()),
              "sf_batchnorm_eval_bwd")
        return dx, dgamma, dbeta, None, None, None


def batchnorm(x: Tensor, bn: torch.nn.BatchNorm2d, groups: int, training: bool, stats: Optional["ConvStats"] = None) -> Tensor:
    """``bn(x)`` on NHWC ``x``; in training mode with ``groups`` separate batches (and running-stat updates in order).
    ``stats``: the producing convolution's ``ConvStats`` (training mode) - saves the statistics pass over ``x``."""
    if training:
        momentum = bn.momentum
        if bn.track_running_stats and bn.num_batches_tracked is not None:
            if momentum is None:  # torch's cumulative moving average: factor 1/num_batches_tracked, group after group
                momentum = -(float(bn.num_batches_tracked) + 1.0)  # (host read: only for momentum=None modules)
            bn.num_batches_tracked += groups
        return _BatchNormTrainFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, groups, bn.eps,
                                       momentum if momentum is not None else 0.0, stats)
    if bn.running_mean is None:  # track_running_stats=False: torch normalises with batch statistics in eval mode too
        return _BatchNormTrainFn.apply(x, bn.weight, bn.bias, None, None, groups, bn.eps, 0.0, stats)
    return _BatchNormEvalFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps)


# ----------------------------------------------------------------------------------------------
# pointwise linear map (1x1 conv) and axial attention core
# ----------------------------------------------------------------------------------------------
class _LinearFn(torch.autograd.Function):
    """``y[..., :N] = x @ W.T + b`` with ``W [N, Kp]`` (zero-padded columns); output lanes ``out_lanes``."""

    @staticmethod
    def forward(ctx, x: Tensor, W: Tensor, bias: Optional[Tensor], out_lanes: int, lowp: bool = False):
        W = W.contiguous()
        ctx.save_for_backward(x, W)
        ctx.has_bias, ctx.lowp, ctx.bias = bias is not None, lowp, bias
        return K.linear_fwd(x, W, bias, out_lanes, lowp)

    @staticmethod
    def backward(ctx, gy: Tensor):
        x, W = ctx.saved_tensors
        gy = gy.contiguous()
        N, Kp = W.shape
        dx = None
        if ctx.needs_input_grad[0]:
            # dx = gy[..., :N] @ W : the same kernel with the transposed weight (tiny host-side transpose)
            if gy.shape[-1] == N:
                Wt = W.t().contiguous()
            else:
                Wt = torch.zeros(Kp, gy.shape[-1], dtype=torch.float32, device=W.device)
                Wt[:, :N] = W.t()
            dx = K.linear_fwd(gy, Wt, None, Kp, ctx.lowp)
        if not (ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])):
            return dx, None, None, None, None
        if Kp <= 256:  # (the split-K kernel writes where it is told: the parameters' own gradient slices if an optimizer registered them)
            (dW, dW_ret), (db, db_ret) = grad_out(W), (grad_out(ctx.bias) if ctx.has_bias else (None, None))
            K.linear_bwd_weight(gy, x, N, ctx.has_bias, out=(dW, db))
            return dx, dW_ret, db_ret, None, None
        dW, db = K.linear_bwd_weight_any(gy, x, N, ctx.has_bias)
        return dx, dW, db, None, None


def linear(x: Tensor, weight: Tensor, bias: Optional[Tensor], out_lanes: Optional[int] = None, lowp: bool = False) -> Tensor:
    """Pointwise linear map; ``weight [N, K]`` with K <= x lanes (padded with zero columns here, autograd-tracked).  ``lowp``: forward and input
    gradient with the operands rounded to the 16-bit compute mode's type (1x1 convolutions under the reference's autocast; ignored in f32 mode)."""
    Kp = x.shape[-1]
    if weight.shape[1] != Kp:
        weight = torch.nn.functional.pad(weight, (0, Kp - weight.shape[1]))
    return _LinearFn.apply(x, weight, bias, out_lanes or cpad(weight.shape[0]), lowp)


class _AttnCoreFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv: Tensor, hid: int, heads: int):
        ctx.save_for_backward(qkv)
        ctx.meta = (hid, heads)
        return K.attention_core_fwd(qkv, hid, heads)

    @staticmethod
    def backward(ctx, g: Tensor):
        (qkv,) = ctx.saved_tensors
        return K.attention_core_bwd(qkv, g.contiguous(), *ctx.meta), None, None


def attention_core(qkv: Tensor, hid: int, heads: int) -> Tensor:
    return _AttnCoreFn.apply(qkv, hid, heads)


class _AxialLayerFn(torch.autograd.Function):
    """One axial-attention layer (lucidrains ``AxialAttention(dim, heads, num_dimensions=2, sum_axial_out=True)`` as MetNet uses it) as ONE
    autograd node: q | k | v projections of both axes in one GEMM, the attention core, the two output projections in one GEMM, summed.

    Round 6: the same launches were five nested ``Function``s (parameter blocks, linear, core, linear + a torch add); at 96 maps of
    16 x 16 the layer's forward + backward is ~230 us of device time, and the host needed 460 us to enqueue it (tools/probe_axial_host.py:
    autograd-node and Python overhead per launch, not the kernels).  Here the host makes the library calls back to back; results are those of
    the nested form bit for bit (``tests/test_metnet_gpu.py::test_axial_layer_single_node_equals_nested``).  Lanes must be unpadded (hid % 16 == 0)."""

    @staticmethod
    def forward(ctx, x: Tensor, q0: Tensor, kv0: Tensor, q1: Tensor, kv1: Tensor, o0: Tensor, b0: Tensor, o1: Tensor, b1: Tensor, hid: int, heads: int):
        dev = x.device
        # one buffer: w_in [6h, h] rows [q0 ; k0 | v0 ; q1 ; k1 | v1], w_out [h, 2h] columns [o0 | o1] and - for the backward pass's input-gradient GEMMs -
        # their transposes w_in_t [h, 6h], w_out_t [2h, h], all written by ONE sf_copy_blocks launch
        keep = any(ctx.needs_input_grad)
        wbuf = torch.empty((16 if keep else 8) * hid * hid, dtype=torch.float32, device=dev)
        w_in, w_out = wbuf[: 6 * hid * hid].view(6 * hid, hid), wbuf[6 * hid * hid: 8 * hid * hid].view(hid, 2 * hid)
        blocks = [(q0, 0, 0, hid, w_in, 0, 0, hid, hid, hid), (kv0, 0, 0, hid, w_in, hid, 0, hid, 2 * hid, hid),
                  (q1, 0, 0, hid, w_in, 3 * hid, 0, hid, hid, hid), (kv1, 0, 0, hid, w_in, 4 * hid, 0, hid, 2 * hid, hid),
                  (o0, 0, 0, hid, w_out, 0, 0, 2 * hid, hid, hid), (o1, 0, 0, hid, w_out, 0, hid, 2 * hid, hid, hid)]
        if keep:
            w_in_t, w_out_t = wbuf[8 * hid * hid: 14 * hid * hid].view(hid, 6 * hid), wbuf[14 * hid * hid:].view(2 * hid, hid)
            blocks += [(q0, 0, 0, hid, w_in_t, 0, 0, 6 * hid, hid, hid, True), (kv0, 0, 0, hid, w_in_t, 0, hid, 6 * hid, 2 * hid, hid, True),
                       (q1, 0, 0, hid, w_in_t, 0, 3 * hid, 6 * hid, hid, hid, True), (kv1, 0, 0, hid, w_in_t, 0, 4 * hid, 6 * hid, 2 * hid, hid, True),
                       (o0, 0, 0, hid, w_out_t, 0, 0, hid, hid, hid, True), (o1, 0, 0, hid, w_out_t, hid, 0, hid, hid, hid, True)]
        _copy_blocks(blocks)
        bsum = b0 + b1
        qkv = K.linear_fwd(x, w_in, None, 6 * hid)
        att = K.attention_core_fwd(qkv, hid, heads)   # [n, h, w, 2 * hid] = [axis 0 | axis 1]
        y = K.linear_fwd(att, w_out, bsum, hid)
        ctx.meta = (hid, heads)
        ctx.params = (q0, kv0, q1, kv1, o0, b0, o1, b1)   # (for the gradient sink: identity of the parameters)
        if keep:
            ctx.save_for_backward(x, qkv, att, w_in_t, w_out_t)
        return y

    @staticmethod
    def backward(ctx, gy: Tensor):
        x, qkv, att, w_in_t, w_out_t = ctx.saved_tensors
        hid, heads = ctx.meta
        q0, kv0, q1, kv1, o0, b0, o1, b1 = ctx.params
        gy = gy.contiguous()
        dev = gy.device
        need = ctx.needs_input_grad
        # output projections: d att = gy @ w_out, d w_out = gy^T att, d bias = column sums of gy (both axes' biases receive it)
        datt = K.linear_fwd(gy, w_out_t, None, 2 * hid)
        dw_out = torch.empty(hid, 2 * hid, dtype=torch.float32, device=dev)
        (db0, db0_ret), (db1, db1_ret) = grad_out(b0, needed=need[6]), grad_out(b1, needed=need[8])
        K.linear_bwd_weight(gy, att, hid, True, out=(dw_out, db0))
        dqkv = K.attention_core_bwd(qkv, datt, hid, heads)
        dx = K.linear_fwd(dqkv, w_in_t, None, hid) if need[0] else None
        dw_in = torch.empty(6 * hid, hid, dtype=torch.float32, device=dev)
        K.linear_bwd_weight(dqkv, x, 6 * hid, False, out=(dw_in, None))
        outs = [grad_out(p, needed=need[i]) for p, i in ((q0, 1), (kv0, 2), (q1, 3), (kv1, 4), (o0, 5), (o1, 7))]
        (dq0, _), (dkv0, _), (dq1, _), (dkv1, _), (do0, _), (do1, _) = outs
        _copy_blocks([(dw_in, 0, 0, hid, dq0, 0, 0, hid, hid, hid), (dw_in, hid, 0, hid, dkv0, 0, 0, hid, 2 * hid, hid),
                      (dw_in, 3 * hid, 0, hid, dq1, 0, 0, hid, hid, hid), (dw_in, 4 * hid, 0, hid, dkv1, 0, 0, hid, 2 * hid, hid),
                      (dw_out, 0, 0, 2 * hid, do0, 0, 0, hid, hid, hid), (dw_out, 0, hid, 2 * hid, do1, 0, 0, hid, hid, hid),
                      (db0, 0, 0, hid, db1, 0, 0, hid, 1, hid)])   # (the second bias gradient = the first)
        r = [o[1] for o in outs]
        return dx, r[0], r[1], r[2], r[3], r[4], db0_ret, r[5], db1_ret, None, None


def axial_layer(x: Tensor, a0, a1, hid: int, heads: int) -> Tensor:
    """``a0`` / ``a1``: the two axes' ``SelfAttention`` parameter containers (``to_q``, ``to_kv``, ``to_out``)."""
    return _AxialLayerFn.apply(x, a0.to_q.weight, a0.to_kv.weight, a1.to_q.weight, a1.to_kv.weight, a0.to_out.weight, a0.to_out.bias,
                               a1.to_out.weight, a1.to_out.bias, hid, heads)


# ----------------------------------------------------------------------------------------------
# ConvGRU over a whole sequence (one autograd node per layer)
# ----------------------------------------------------------------------------------------------
class GRUEngine:
    """Maps + packed caches of one ConvGRUCell(input_dim, hidden_dim, 3x3).

    The cell's three convolutions are regrouped (the caller concatenates the reference parameters,
    autograd-tracked): ``Wx [3*hid, cin]`` = [conv_zr[:, :cin] ; conv_h1] applied to ALL timesteps in
    one convolution, and ``Wh [3*hid, hid]`` = [conv_zr[:, cin:] ; conv_h2] applied per step.
    """

    def __init__(self, cin: int, hid: int) -> None:
        self.cin, self.hid = cin, hid
        self.cinp, self.hidp = cpad(cin), cpad(hid)
        gm3 = K.gate_major(hid, 3)
        self.x_fwd = K.custom_map(gm3, K._padded(cin))                 # x -> gx [3*hidp]
        self.x_bwd = K.custom_map(K._padded(cin), gm3)                 # dgx -> dx
        self.x_wgrad = K.GemmMap(gm3, K._padded(cin), 0, 3 * self.hidp)
        self.h_fwd = K.gru_rec_map(hid)                                # h -> [z_h|r_h|h2], fused epilogue
        self.h_bwd = K.custom_map(K._padded(hid), gm3)                 # dgh -> dh_prev
        self.h_wgrad = K.GemmMap(gm3, K._padded(hid), 0, 3 * self.hidp)
        self._key, self._packed = None, {}

    def packed(self, Wx: Tensor, bx: Tensor, Wh: Tensor, bh: Tensor, src_key=None):
        """``src_key``: identity + version of the SOURCE parameters the regrouped tensors were concatenated from (the cats
        are fresh tensors every forward - version 0, and the caching allocator hands the same addresses back - so they
        cannot key the cache themselves).  Without it nothing is cached."""
        key = None if src_key is None else (src_key, generation())
        if key is None or key != self._key:
            self._key = key
            self._packed = {
                "x_fwd": K.pack_weights(Wx, bx, self.x_fwd, False),
                "x_bwd": K.pack_weights(Wx, None, self.x_bwd, True)[0],
                "h_fwd": K.pack_weights(Wh, bh, self.h_fwd, False),
                "h_bwd": K.pack_weights(Wh, None, self.h_bwd, True)[0],
            }
        return self._packed


class _ConvGRUSeqFn(torch.autograd.Function):
    """x ``[T*n,H,W,Cinp]`` (time-major) -> (all hidden states ``[T*n,H,W,hidp]``, last state ``[n,H,W,hidp]``)."""

    @staticmethod
    def forward(ctx, eng: GRUEngine, x: Tensor, Tn: int, Wx: Tensor, bx: Tensor, Wh: Tensor, bh: Tensor, src_key=None):
        N, H, W, _ = x.shape
        n = N // Tn
        hidp, dev = eng.hidp, x.device
        pk = eng.packed(Wx, bx, Wh, bh, src_key)
        keep = any(ctx.needs_input_grad)
        from ._hip import gate_storage_dtype
        persistent = K.convgru_seq_supported(H, W, hidp) and not os.environ.get("SF_GRU_PER_STEP")
        # the x-part of all steps in one convolution; with the persistent kernel in "bf16a" mode it is stored as bf16 (what
        # autocast leaves behind conv_zr / conv_h1) and prefetched by the sequence kernel a whole K loop ahead
        gx = torch.empty(N, H, W, 3 * hidp, dtype=gate_storage_dtype() if persistent else torch.float32, device=dev)
        K.conv3x3(T(x), NULL, N, H, W, pk["x_fwd"][0], pk["x_fwd"][1], eng.x_fwd, T(gx))
        hs = torch.empty(Tn, n, H, W, hidp, dtype=torch.float32, device=dev)
        # saved gates: backward-only data, bf16 in "bf16a" mode (as the ConvLSTM's)
        gates = torch.empty(Tn, n, H, W, 4 * hidp, dtype=gate_storage_dtype(), device=dev) if keep else None
        if persistent:  # all steps in ONE launch, state resident on chip
            K.convgru_seq_fwd(gx, None, Tn, n, H, W, pk["h_fwd"][0], pk["h_fwd"][1], hidp, hs, gates)
        else:
            gxs = gx.view(Tn, n, H, W, 3 * hidp)
            for t in range(Tn):
                K.convgru_step_fwd(T(gxs[t]), hs[t - 1] if t else None, n, H, W, pk["h_fwd"][0], pk["h_fwd"][1], hidp, hs[t],
                                   gates[t] if keep else None)
        ctx.eng, ctx.Tn = eng, Tn
        ctx.pk = pk  # the packed images this forward used (the engine's cache may be rebuilt before the backward runs)
        ctx.set_materialize_grads(False)
        if keep:
            ctx.save_for_backward(x, hs, gates, Wx, Wh)
        return hs.view(N, H, W, hidp), hs[Tn - 1]

    @staticmethod
    def backward(ctx, g_seq: Optional[Tensor], g_last: Optional[Tensor]):
        eng: GRUEngine = ctx.eng
        Tn = ctx.Tn
        x, hs, gates, Wx, Wh = ctx.saved_tensors
        _, n, H, W, hidp = hs.shape
        N, dev = Tn * n, x.device
        pk = ctx.pk
        g_seq = g_seq.contiguous().view(Tn, n, H, W, hidp) if g_seq is not None else None
        g_last = g_last.contiguous() if g_last is not None else None
        # gradients wrt the two convolutions' outputs: only ever read as bf16 MFMA operands (input / weight gradient
        # convolutions), so "bf16a" stores them as bf16 like the gates they were computed from
        gdt = gates.dtype
        dgx = torch.empty(Tn, n, H, W, 3 * hidp, dtype=gdt, device=dev)
        dgh = torch.empty(Tn, n, H, W, 3 * hidp, dtype=gdt, device=dev)
        if K.convgru_seq_bwd_supported(H, W, hidp, gates) and not os.environ.get("SF_GRU_PER_STEP"):
            # the whole time loop in ONE launch, carried gradient in registers (bit-identical dgx / dgh)
            K.convgru_seq_bwd(g_seq, g_last, gates, hs, Tn, n, H, W, pk["h_bwd"], hidp, dgx, dgh)
            steps = ()
        else:
            steps = range(Tn - 1, -1, -1)
            direct = torch.empty(n, H, W, hidp, dtype=torch.float32, device=dev)  # dh * z of the step above
            carry = torch.empty(n, H, W, hidp, dtype=torch.float32, device=dev)   # conv^T(dgh) of the step above
        zeros = None
        have_carry = False
        # "f32e" mode, per-step route: scale words raised by the gate kernel - [0] max |dgx| over the whole sequence, [1 + t] max |dgh[t]|
        words = K.scale_words(1 + Tn, dev) if steps and dgx.dtype == torch.float32 else None
        for t in steps:
            src: List[sfTensor] = []
            if g_seq is not None:
                src.append(T(g_seq[t]))
            if t == Tn - 1 and g_last is not None:
                src.append(T(g_last))
            if have_carry:
                src += [T(direct), T(carry)]
            if not src:
                zeros = zeros if zeros is not None else torch.zeros(n, H, W, hidp, dtype=torch.float32, device=dev)
                src = [T(zeros)]
            if len(src) > 3:  # g_seq + direct + carry (+ g_last only at the last step where there is no carry) <= 3
                raise RuntimeError("unexpected number of gradient sources")
            K.convgru_bwd_gates(src, gates[t], hs[t - 1] if t else None, hidp, dgx[t], dgh[t], direct if t else None,
                                words[0:1] if words is not None else None, words[1 + t: 2 + t] if words is not None else None)
            if t:
                src_h = K.grad_operand_with(dgh[t], words[1 + t: 2 + t]) if words is not None else K.grad_operand(dgh[t])
                K.conv3x3(src_h, NULL, n, H, W, pk["h_bwd"], None, eng.h_bwd, T(carry))
                have_carry = True
        dx = None
        if ctx.needs_input_grad[1]:
            dx = torch.empty_like(x)  # stored like x (fp32: the encoder's last pooling returns fp32)
        gx_word = (lambda t_: K.grad_operand_with(t_, words[0:1])) if words is not None else K.grad_operand
        if ctx.needs_input_grad[1]:
            dgxT = gx_word(dgx.view(N, H, W, 3 * hidp))
            K.conv3x3(dgxT, NULL, N, H, W, pk["x_bwd"], None, eng.x_bwd, T(dx))
        else:
            dgxT = gx_word(dgx.view(N, H, W, 3 * hidp))
        dWx, dbx = torch.empty_like(Wx), torch.empty(Wx.shape[0], dtype=torch.float32, device=dev)
        K.conv3x3_bwd_weight(T(x), NULL, dgxT, N, H, W, eng.x_wgrad, dWx, dbx, False)
        dWh, dbh = torch.empty_like(Wh), torch.empty(Wh.shape[0], dtype=torch.float32, device=dev)
        if Tn > 1:
            gh_all = K.grad_operand_with(dgh[1:], words[2:].amax().reshape(1)) if words is not None and Tn > 1 else K.grad_operand(dgh[1:])
            K.conv3x3_bwd_weight(T(hs[: Tn - 1]), NULL, gh_all, (Tn - 1) * n, H, W, eng.h_wgrad, dWh, dbh, False)
            # bias of the h-part also acts at t = 0 (zero state, bias only): add that step's column sums (tiny torch op)
            dbh = (dbh + dgh[0].float().sum(dim=(0, 1, 2))[K_bias_index(eng, dev)]) * K_bias_mask(eng, dev)
        else:
            dWh.zero_()
            dbh.copy_(dgh[0].float().sum(dim=(0, 1, 2))[K_bias_index(eng, dev)] * K_bias_mask(eng, dev))
        return None, dx, None, dWx, dbx, dWh, dbh, None


_BIAS_CACHE = {}


def _bias_tables(eng: GRUEngine, dev):
    key = (eng.hid, str(dev))
    if key not in _BIAS_CACHE:
        lanes = K.gate_major(eng.hid, 3)  # padded lane -> row
        idx = torch.zeros(3 * eng.hid, dtype=torch.long)
        for lane, row in enumerate(lanes):
            if row >= 0:
                idx[row] = lane
        mask = torch.zeros(3 * eng.hid, dtype=torch.float32)
        mask[2 * eng.hid:] = 1.0  # the step kernel adds the bias to the h2 map only (z/r biases ride on the x-part)
        _BIAS_CACHE[key] = (idx.to(dev), mask.to(dev))
    return _BIAS_CACHE[key]


def K_bias_index(eng: GRUEngine, dev):
    return _bias_tables(eng, dev)[0]


def K_bias_mask(eng: GRUEngine, dev):
    return _bias_tables(eng, dev)[1]


def convgru_sequence(eng: GRUEngine, x: Tensor, Tn: int, Wx: Tensor, bx: Tensor, Wh: Tensor, bh: Tensor, src_key=None) -> Tuple[Tensor, Tensor]:
    return _ConvGRUSeqFn.apply(eng, x, Tn, Wx, bx, Wh, bh, src_key)


# ----------------------------------------------------------------------------------------------
# loss / dropout
# ----------------------------------------------------------------------------------------------
class _MSEFn(torch.autograd.Function):
    """(mean squared error, per-frame means) with the gradient produced in the same pass."""

    @staticmethod
    def forward(ctx, pred: Tensor, target: Tensor, frames: int, inner: int):
        pred, target = pred.contiguous(), target.contiguous()
        n = pred.numel()
        grad = torch.empty_like(pred) if ctx.needs_input_grad[0] else None
        sums = torch.empty(1 + frames, dtype=torch.float64, device=pred.device)
        out = torch.empty(1 + frames, dtype=torch.float32, device=pred.device)
        check(lib().sf_mse_loss(pred.data_ptr(), target.data_ptr(), n, inner, frames, grad.data_ptr() if grad is not None else None,
                                sums.data_ptr(), out.data_ptr(), stream_ptr()), "sf_mse_loss")
        ctx.save_for_backward(grad if grad is not None else pred.new_empty(0))
        loss, per = out[0], out[1:]
        ctx.mark_non_differentiable(per)  # the very tensor object that is returned (a second out[1:] would be another view)
        return loss, per

    @staticmethod
    def backward(ctx, g_loss: Tensor, _g_frames):
        (grad,) = ctx.saved_tensors
        return grad * g_loss, None, None, None


def mse_loss_with_frames(pred: Tensor, target: Tensor, frame_dim: int = 1) -> Tuple[Tensor, Tensor]:
    """``(F.mse_loss(pred, target), per-frame losses)`` for contiguous tensors whose ``frame_dim`` indexes the forecast frame."""
    require_device(pred, "pred")
    if pred.shape != target.shape:
        raise RuntimeError(f"mse: shape mismatch {tuple(pred.shape)} vs {tuple(target.shape)}")
    frames = pred.shape[frame_dim]
    inner = 1
    for d in pred.shape[frame_dim + 1:]:
        inner *= d
    return _MSEFn.apply(pred.float(), target.float(), frames, inner)


class _Dropout2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, p1: float, p2: float, period: int, seed1: int, seed2: int):
        ctx.meta = (p1, p2, period, seed1, seed2)
        y = torch.empty_like(x)
        check(lib().sf_dropout2(x.data_ptr(), x.numel(), p1, p2, period, seed1, seed2, y.data_ptr(), stream_ptr()), "sf_dropout2")
        return y

    @staticmethod
    def backward(ctx, g: Tensor):
        p1, p2, period, seed1, seed2 = ctx.meta
        g = g.contiguous()
        gx = torch.empty_like(g)
        check(lib().sf_dropout2(g.data_ptr(), g.numel(), p1, p2, period, seed1, seed2, gx.data_ptr(), stream_ptr()), "sf_dropout2")
        return gx, None, None, None, None, None


def dropout2(x: Tensor, p1: float, p2: float, period: int) -> Tensor:
    """Elementwise dropout ``p1`` fused with a dropout ``p2`` whose mask repeats every ``period`` elements (shared over time)."""
    if p1 <= 0 and p2 <= 0:
        return x
    return _Dropout2Fn.apply(x.contiguous(), float(p1), float(p2), int(period), *_draw_seeds())


# ----------------------------------------------------------------------------------------------
# CloudGAN side network: general convolution, LeakyReLU, GAN / L1 losses
# ----------------------------------------------------------------------------------------------
class _Conv2dFn(torch.autograd.Function):
    """``nn.Conv2d(k, stride, padding)`` (+ fused ``LeakyReLU(slope)``) on NHWC fp32 ``x [N,H,W,Cp]``; weight in OIHW (sf_conv2d_*)."""

    @staticmethod
    def forward(ctx, x: Tensor, weight: Tensor, bias: Optional[Tensor], stride: int, pad: int, slope: float):
        N, H, W, _ = x.shape
        cout, cin, kh, kw = weight.shape
        oh, ow = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
        w = weight.contiguous()
        y = torch.empty(N, oh, ow, cpad(cout), dtype=torch.float32, device=x.device)
        check(lib().sf_conv2d_fwd(T(x), N, H, W, w.data_ptr(), bias.data_ptr() if bias is not None else None, cin, cout, kh, kw, stride, pad, slope,
                                  T(y), SF_F32, stream_ptr()), "sf_conv2d_fwd")
        ctx.meta = (stride, pad, slope, bias is not None)
        ctx.bias = bias
        ctx.save_for_backward(x, w, y if slope != 1.0 else x.new_empty(0))
        return y

    @staticmethod
    def backward(ctx, gy: Tensor):
        x, w, y = ctx.saved_tensors
        stride, pad, slope, has_bias = ctx.meta
        N, H, W, _ = x.shape
        cout, cin, kh, kw = w.shape
        gy = gy.contiguous()
        if slope != 1.0:
            g2 = torch.empty_like(gy)
            check(lib().sf_leaky_relu(gy.data_ptr(), y.data_ptr(), gy.numel(), slope, g2.data_ptr(), stream_ptr()), "sf_leaky_relu")
            gy = g2
        oh, ow = gy.shape[1], gy.shape[2]
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            check(lib().sf_conv2d_bwd_data(T(gy), N, H, W, w.data_ptr(), cin, cout, kh, kw, stride, pad, T(dx), SF_F32, stream_ptr()), "sf_conv2d_bwd_data")
        (dw, dw_ret), (db, db_ret) = grad_out(w), (grad_out(ctx.bias) if has_bias else (None, None))
        nbytes = lib().sf_conv2d_bwd_weight_workspace_bytes(N, oh, ow, cin, cout, kh, kw)
        ws = torch.empty(nbytes // 4 + 1, dtype=torch.float32, device=x.device)
        check(lib().sf_conv2d_bwd_weight(T(x), T(gy), N, H, W, cin, cout, kh, kw, stride, pad, dw.data_ptr(), db.data_ptr() if db is not None else None, 0,
                                         ws.data_ptr(), nbytes, SF_F32, stream_ptr()), "sf_conv2d_bwd_weight")
        return dx, dw_ret, db_ret, None, None, None


def conv2d(x: Tensor, weight: Tensor, bias: Optional[Tensor], stride: int = 1, padding: int = 0, leaky_slope: float = 1.0) -> Tensor:
    require_device(x, "x")
    return _Conv2dFn.apply(x.contiguous(), weight, bias, int(stride), int(padding), float(leaky_slope))


class _LeakyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, slope: float):
        y = torch.empty_like(x)
        check(lib().sf_leaky_relu(x.data_ptr(), None, x.numel(), slope, y.data_ptr(), stream_ptr()), "sf_leaky_relu")
        ctx.slope = slope
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, gy: Tensor):
        (y,) = ctx.saved_tensors
        gy = gy.contiguous()
        gx = torch.empty_like(gy)
        check(lib().sf_leaky_relu(gy.data_ptr(), y.data_ptr(), gy.numel(), ctx.slope, gx.data_ptr(), stream_ptr()), "sf_leaky_relu")
        return gx, None


def leaky_relu(x: Tensor, slope: float = 0.2) -> Tensor:
    return _LeakyFn.apply(x.contiguous(), float(slope))


class _PairLossFn(torch.autograd.Function):
    """(mean loss, per-group means) of L1 (``target`` given) or BCE-with-logits against constant labels, gradient in the same pass."""

    @staticmethod
    def forward(ctx, pred: Tensor, target: Optional[Tensor], labels: Tuple[float, float], groups: int, c: int):
        rows = pred.numel() // pred.shape[-1]
        grad = torch.empty_like(pred) if ctx.needs_input_grad[0] else None
        sums = torch.empty(1 + groups, dtype=torch.float64, device=pred.device)
        out = torch.empty(1 + groups, dtype=torch.float32, device=pred.device)
        g = T(grad) if grad is not None else NULL
        if target is not None:
            check(lib().sf_l1_loss(T(pred), T(target), rows, groups, c, g, sums.data_ptr(), out.data_ptr(), stream_ptr()), "sf_l1_loss")
        else:
            mode = int(labels[2]) if len(labels) > 2 else 1
            check(lib().sf_gan_loss(mode, T(pred), labels[0], labels[1], rows, groups, c, g, sums.data_ptr(), out.data_ptr(), stream_ptr()), "sf_gan_loss")
        ctx.save_for_backward(grad if grad is not None else pred.new_empty(0))
        loss, per = out[0], out[1:]
        ctx.mark_non_differentiable(per)
        return loss, per

    @staticmethod
    def backward(ctx, g_loss: Tensor, _g_groups):
        (grad,) = ctx.saved_tensors
        return grad * g_loss, None, None, None, None


def l1_loss_groups(pred: Tensor, target: Tensor, groups: int, c: int):
    """``(nn.L1Loss()(pred[..., :c], target[..., :c]), per-group means)`` on NHWC tensors whose leading rows split into ``groups``."""
    return _PairLossFn.apply(pred.contiguous(), target.contiguous(), (0.0, 0.0), groups, c)


GAN_MODES = {"vanilla": 1, "lsgan": 2, "wgangp": 3}


def bce_logits_groups(logits: Tensor, label_even: float, label_odd: float, groups: int, c: int = 1, mode: str = "vanilla"):
    """GANLoss of the first ``c`` lanes against a constant label per group parity: ``vanilla`` = ``nn.BCEWithLogitsLoss()``, ``lsgan`` =
    ``nn.MSELoss()``, ``wgangp`` = ``-mean`` (real label) / ``+mean`` (reference gan/discriminators.py:70-136; ``sf_gan_loss``)."""
    return _PairLossFn.apply(logits.contiguous(), None, (float(label_even), float(label_odd), GAN_MODES[mode]), groups, c)


# ----------------------------------------------------------------------------------------------
# ST-LSTM cell with memory decoupling: the two pointwise stages
# ----------------------------------------------------------------------------------------------
class _STLSTMGatesFn(torch.autograd.Function):
    """(gx, gh, gm, c, m) -> (c', m', mem = [c' | m'], delta_c, delta_m, pre_o); ``sf_stlstm_gates_fwd/bwd``."""

    @staticmethod
    def forward(ctx, gx: Tensor, gh: Tensor, gm: Tensor, c: Tensor, m: Tensor, hidp: int, forget_bias: float):
        gx, gh, gm, c, m = (t.contiguous() for t in (gx, gh, gm, c, m))
        shp, dev = c.shape[:-1], c.device
        pixels = c.numel() // hidp
        new = lambda lanes: torch.empty(*shp, lanes, dtype=torch.float32, device=dev)
        c_new, m_new, mem, dc, dm, po = new(hidp), new(hidp), new(2 * hidp), new(hidp), new(hidp), new(hidp)
        keep = any(ctx.needs_input_grad)
        gates = new(6 * hidp) if keep else None
        check(lib().sf_stlstm_gates_fwd(T(gx), T(gh), T(gm), T(c), T(m), pixels, hidp, forget_bias, T(c_new), T(m_new), T(mem), T(dc), T(dm), T(po),
                                        T(gates) if keep else NULL, SF_F32, stream_ptr()), "sf_stlstm_gates_fwd")
        ctx.hidp = hidp
        if keep:
            ctx.save_for_backward(gates, c, m)
        ctx.set_materialize_grads(False)
        return c_new, m_new, mem, dc, dm, po

    @staticmethod
    def backward(ctx, d_cn, d_mn, d_mem, d_dc, d_dm, d_po):
        gates, c, m = ctx.saved_tensors
        hidp = ctx.hidp
        shp, dev = c.shape[:-1], c.device
        pixels = c.numel() // hidp
        new = lambda lanes: torch.empty(*shp, lanes, dtype=torch.float32, device=dev)
        dgx, dgh, dgm, dc, dm = new(7 * hidp), new(4 * hidp), new(3 * hidp), new(hidp), new(hidp)
        g = [T(t.contiguous()) if t is not None else NULL for t in (d_cn, d_mn, d_mem, d_dc, d_dm, d_po)]
        check(lib().sf_stlstm_gates_bwd(*g, T(gates), T(c), T(m), pixels, hidp, T(dgx), T(dgh), T(dgm), T(dc), T(dm), SF_F32, stream_ptr()),
              "sf_stlstm_gates_bwd")
        return dgx, dgh, dgm, dc, dm, None, None


class _STLSTMOutFn(torch.autograd.Function):
    """``h' = sigmoid(pre_o + conv_o) * tanh(last)``; ``sf_stlstm_out_fwd/bwd``."""

    @staticmethod
    def forward(ctx, pre_o: Tensor, conv_o: Tensor, last: Tensor, hidp: int):
        pre_o, conv_o, last = pre_o.contiguous(), conv_o.contiguous(), last.contiguous()
        pixels = pre_o.numel() // hidp
        h_new = torch.empty_like(pre_o)
        keep = any(ctx.needs_input_grad)
        saved = torch.empty(*pre_o.shape[:-1], 2 * hidp, dtype=torch.float32, device=pre_o.device) if keep else None
        check(lib().sf_stlstm_out_fwd(T(pre_o), T(conv_o), T(last), pixels, hidp, T(h_new), T(saved) if keep else NULL, SF_F32, stream_ptr()),
              "sf_stlstm_out_fwd")
        ctx.meta = (hidp, conv_o.shape[-1], last.shape[-1])
        if keep:
            ctx.save_for_backward(saved)
        return h_new

    @staticmethod
    def backward(ctx, dh: Tensor):
        (saved,) = ctx.saved_tensors
        hidp, co_lanes, last_lanes = ctx.meta
        dh = dh.contiguous()
        pixels = dh.numel() // hidp
        d_a, d_last = torch.empty_like(dh), torch.empty_like(dh)
        check(lib().sf_stlstm_out_bwd(T(dh), T(saved), pixels, hidp, T(d_a), T(d_last), SF_F32, stream_ptr()), "sf_stlstm_out_bwd")
        pad = lambda t, lanes: t if lanes == hidp else torch.nn.functional.pad(t, (0, lanes - hidp))
        return d_a, pad(d_a, co_lanes), pad(d_last, last_lanes), None


def stlstm_gates(gx: Tensor, gh: Tensor, gm: Tensor, c: Tensor, m: Tensor, hidp: int, forget_bias: float = 1.0):
    return _STLSTMGatesFn.apply(gx, gh, gm, c, m, hidp, forget_bias)


def stlstm_out(pre_o: Tensor, conv_o: Tensor, last: Tensor, hidp: int) -> Tensor:
    return _STLSTMOutFn.apply(pre_o, conv_o, last, hidp)

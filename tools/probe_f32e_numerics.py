"""Split-operand "f32e" convolutions - the NUMERICS half of the go / no-go, on the CPU, zero GPU minutes.  Outcome: bf16 x 3 / x 4 NO-GO, fp16 x 3 with a
scaled low part GO (profiles/r06_f32e_numerics.txt) - which is what SF_F32E implements.

gfx950 has no TF32; its bf16 MFMA pipe is 16x the exact-fp32 MFMA pipe.  Writing every fp32 operand as a sum of bf16 parts
(x = hi + lo [+ lo2], hi = bf16(x), lo = bf16(x - hi), lo2 = bf16(x - hi - lo)) and multiplying the parts on the bf16 pipe with fp32
accumulation gives
    x3 : hi*hi + hi*lo + lo*hi                         (per-product error ~ |lo*lo| ~ 2^-18 .. 2^-16)
    x4 : x3 + lo*lo                                    (error = the dropped third parts ~ 2^-17)
    x6 : three-way split, all products down to 2^-24   (hi*hi + hi*lo + lo*hi + lo*lo + hi*lo2 + lo2*hi)
Products of two bf16 values are exact in fp32, so a CPU fp32 convolution of the PARTS is what the MFMA computes up to summation order.
This probe swaps every convolution of the oracle (forward, input gradient, weight gradient - the three GEMMs the kernels run) for the split
form and runs the reference goldens through the UNCHANGED fp32 gates of tests/conftest.py (rtol 1e-4 / atol 1e-5, gradients relative to
max(1, |ref|max), relative L2 <= 1e-3).

    python tools/probe_f32e_numerics.py            # -> profiles/r06_f32e_numerics.txt
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as TF

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import ATOL, GOLDEN, RTOL  # noqa: E402

from oracle import convlstm as OC  # noqa: E402  (checker-side tool, not product)
from oracle import metnet as OM  # noqa: E402

bf = lambda t: t.to(torch.bfloat16).to(torch.float32)


hf = lambda t: t.to(torch.float16).to(torch.float32)


def parts(t, n, kind="bf16"):
    """t -> n fp32 tensors, each exactly representable in the operand type, whose sum approximates t (bf16: to 2^(-9 n)).
    kind "f16": fp16 parts, the k-th part taken from the residual scaled by 2^(11 k) (what a kernel with one accumulator rescale per part level
    would do; the value returned is the UNSCALED part, products and sums of scaled parts being exact powers of two apart);
    kind "f16t": the same behind a per-tensor power-of-two scale that puts |t|max at 2^14 (a dynamic scale a kernel would have to be told)."""
    out, r = [], t
    if kind == "bf16":
        for _ in range(n):
            p = bf(r)
            out.append(p)
            r = r - p
        return out
    s = 1.0
    if kind == "f16t":
        m = float(t.abs().max())
        s = 2.0 ** (14 - (np.frexp(m)[1] if m > 0 else 0))
    r = t * s
    for k in range(n):
        p = hf(r * 2.0 ** (11 * k)) * 2.0 ** (-11 * k)
        out.append(p / s)
        r = r - p
    return out


# which (a-part, b-part) products each mode keeps
MODES = {
    "bf16": (1, [(0, 0)]),
    "x3": (2, [(0, 0), (0, 1), (1, 0)]),
    "x4": (2, [(0, 0), (0, 1), (1, 0), (1, 1)]),
    "x6": (3, [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)]),
    "h3": (2, [(0, 0), (0, 1), (1, 0)], "f16"),      # fp16 parts, scaled low part, no tensor scale
    "h3t": (2, [(0, 0), (0, 1), (1, 0)], "f16t"),    # fp16 parts behind a per-tensor power-of-two scale
}
MODE = ["x3"]


def split_gemm(fn, a, b):
    n, pairs, *kind = MODES[MODE[0]]
    A, B = parts(a, n, *kind), parts(b, n, *kind)
    # small products first: closer to what one fp32 accumulator chain fed [lo terms ... hi terms] would do; order is immaterial at this level
    out = None
    for i, j in sorted(pairs, key=lambda p: -(p[0] + p[1])):
        y = fn(A[i], B[j])
        out = y if out is None else out + y
    return out


class SplitConv(torch.autograd.Function):
    """conv2d(x, w, padding) whose three GEMMs (forward, input gradient, weight gradient) multiply bf16 parts with fp32 accumulation."""

    @staticmethod
    def forward(ctx, x, w, pad):
        ctx.save_for_backward(x, w)
        ctx.pad = pad
        return split_gemm(lambda a, b: TF.conv2d(a, b, padding=pad), x, w)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        pad = ctx.pad
        dx = split_gemm(lambda a, b: torch.nn.grad.conv2d_input(x.shape, b, a, padding=pad), g, w)
        dw = split_gemm(lambda a, b: torch.nn.grad.conv2d_weight(a, w.shape, b, padding=pad), x, g)
        return dx, dw, None


REAL_CONV2D, REAL_CONV3D = TF.conv2d, TF.conv3d


def conv2d_split(x, w, b=None, stride=1, padding=0, *a, **k):
    assert stride == 1 and not a and not k
    pad = padding if isinstance(padding, tuple) else (padding, padding)
    if MODE[0] == "f32":
        return REAL_CONV2D(x, w, b, padding=pad)
    y = SplitConv.apply(x, w, pad)
    return y if b is None else y + b.view(1, -1, 1, 1)


def conv3d_split(x, w, b=None, stride=1, padding=0):
    """The ConvLSTM head: Conv3d(hid -> out, (1,3,3)) = a 2-D convolution per frame (conv_lstm.py:164-169)."""
    B, C, T, H, W = x.shape
    y = conv2d_split(x.permute(0, 2, 1, 3, 4).reshape(B * T, C, H, W), w[:, :, 0], b, padding=(padding[1], padding[2]))
    return y.reshape(B, T, -1, H, W).permute(0, 2, 1, 3, 4)


class _F:
    """torch.nn.functional with the convolutions swapped (the oracle modules call F.conv2d / F.conv3d)."""

    def __getattr__(self, n):
        return {"conv2d": conv2d_split, "conv3d": conv3d_split}.get(n, getattr(TF, n))


OC.F = _F()
OM.F = _F()


def gate(a, e, grad=False):
    """The unchanged fp32 gate: returns (pass, worst err / bound, rel L2)."""
    a, e = a.detach().float(), e.detach().float()
    atol = ATOL * max(1.0, float(e.abs().max())) if grad else ATOL
    bound = atol + RTOL * e.abs()
    err = (a - e).abs()
    worst = float((err / bound).max())
    rl2 = float((a - e).norm() / (e.norm() + 1e-30))
    ok = worst <= 1.0 and (rl2 <= 10 * RTOL or float(e.norm()) <= 100 * atol * e.numel() ** 0.5)
    return ok, worst, rl2


def load(name):
    return {k: torch.from_numpy(v) if v.ndim else v for k, v in np.load(os.path.join(GOLDEN, name)).items()}


def run_cell(case):
    G = load(f"convlstm_cell_{case}.npz")
    x, h, c, w, b = (G[k].clone().requires_grad_() for k in ("x", "h", "c", "weight", "bias"))
    h1, c1 = OC.convlstm_cell(x, h, c, w, b)
    ((h1 * G["gh"]).sum() + (c1 * G["gc"]).sum()).backward()
    res = {"h'": gate(h1, G["h_out"]), "c'": gate(c1, G["c_out"])}
    for n, t, k in (("dx", x, "dx"), ("dh", h, "dh"), ("dc", c, "dc"), ("dW", w, "dweight"), ("db", b, "dbias")):
        res[n] = gate(t.grad, G[k], grad=True)
    return res


def run_train(case):
    """The Lightning step's own loss (MSE mean over ~10^5..10^7 elements: output gradients ~1e-6 and below), split run against the fp32 run:
    what range-limited operand types meet in real training, which the goldens' O(1) cotangents do not show."""
    G = load(f"convlstm_model_{case}.npz")
    fs = int(G["forecast_steps"])
    grads = {}
    saved = MODE[0]
    for mode in ("f32", saved):
        MODE[0] = mode
        P = {k[len("param."):]: v.clone().requires_grad_() for k, v in G.items() if k.startswith("param.")}
        loss, _ = OC.training_loss(G["x"], G["y"], fs, P)
        loss.backward()
        grads[mode] = {k: p.grad for k, p in P.items()}
    MODE[0] = saved
    res = {}
    for k, e in grads["f32"].items():
        a = grads[saved][k]
        rl2 = float((a - e).norm() / e.norm())
        res[k] = (rl2 <= 10 * RTOL, rl2 / (10 * RTOL), rl2)
    return res


def run_model(case):
    G = load(f"convlstm_model_{case}.npz")
    P = {k[len("param."):]: v.clone().requires_grad_() for k, v in G.items() if k.startswith("param.")}
    x = G["x"].clone().requires_grad_()
    fs = int(G["forecast_steps"])
    pred = OC.convlstm_forward(x, fs, P)
    (pred * G["cot"]).sum().backward()
    res = {"pred": gate(pred, G["pred"]), "dx": gate(x.grad, G["dx"], grad=True)}
    for k, v in G.items():
        if k.startswith("grad."):
            res[k[5:]] = gate(P[k[5:]].grad, v, grad=True)
    return res


_ROUTES = {"rec": [], "pos": 0, "replay": False}
_REAL_POOL = OM.max_pool2


def _pool(t, routing=None):
    """First (fp32) run: record which window element the argmax took; split run: follow the same routing (a near-tie that flips is the
    pooling's discontinuity, not the convolution's error - the GPU tests inject the routing the same way, tests/parity_util.py)."""
    if _ROUTES["replay"]:
        r = _ROUTES["rec"][_ROUTES["pos"]]
        _ROUTES["pos"] += 1
        return _REAL_POOL(t, r)
    n, c, h, w = t.shape
    win = t.detach().view(n, c, h // 2, 2, w // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(n, c, h // 2, w // 2, 4)
    r = TF.one_hot(win.argmax(-1), 4).bool().view(n, c, h // 2, w // 2, 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(n, c, h, w)
    _ROUTES["rec"].append(r)
    return _REAL_POOL(t, r)


OM.max_pool2 = _pool


def run_metnet(B=1, hid=32):
    """MetNet (oracle restatement, unpinned) at a reduced but structurally complete size: split run against the fp32 run of the same oracle,
    train-mode BatchNorm, no dropout, pooling routed identically."""
    L = 3
    P0 = OM.init_params(input_channels=12, sat_channels=12, output_channels=12, hidden_dim=hid, forecast_steps=L, seed=7)
    g = torch.Generator().manual_seed(1234)
    imgs = torch.randn(B, 4, 12, 128, 128, generator=g)
    cot = torch.randn(B, L, 12, 8, 8, generator=g)
    outs = {}
    saved = MODE[0]
    for mode in ("f32", saved):
        MODE[0] = mode
        _ROUTES["replay"] = mode != "f32"
        _ROUTES["pos"] = 0
        if mode == "f32":
            _ROUTES["rec"].clear()
        P = {k: v.clone().requires_grad_() for k, v in P0.items()}
        out = OM.metnet_forward(imgs, P, sat_channels=12, input_size=32, forecast_steps=L)
        (out * cot).sum().backward()
        outs[mode] = (out.detach(), {k: p.grad for k, p in P.items() if p.grad is not None})
    MODE[0] = saved
    (o0, g0), (o1, g1) = outs["f32"], outs[saved]
    res = {"out": gate(o1, o0)}
    for k in g0:
        res[k] = gate(g1[k], g0[k], grad=True)
    return res


def report(title, res, lines):
    bad = [k for k, (ok, _, _) in res.items() if not ok]
    worst = max(res.items(), key=lambda kv: kv[1][1])
    wl2 = max(res.items(), key=lambda kv: kv[1][2])
    lines.append(f"  {title:36s} {'PASS' if not bad else 'FAIL'}  worst err/bound {worst[1][1]:8.3f} ({worst[0]})   worst rel L2 {wl2[1][2]:.2e} ({wl2[0]})"
                 + (f"   failing {len(bad)}/{len(res)}: {bad[:4]}" if bad else ""))
    return not bad


def main():
    torch.set_num_threads(8)
    lines = [__doc__.split("\n\n")[0], "",
             f"gates: rtol {RTOL} / atol {ATOL} (gradients: atol x max(1, |ref|max)), rel L2 <= {10 * RTOL}; err/bound <= 1 passes", ""]
    verdict = {}
    want = sys.argv[1:] or ["bf16", "x3", "x4", "x6", "h3", "h3t"]
    for mode in want:
        MODE[0] = mode
        lines.append(f"mode {mode}: {len(MODES[mode][1])} {'fp16' if len(MODES[mode]) > 2 else 'bf16'} products per fp32 product")
        ok = True
        for case in ("a", "b", "odd", "hot"):
            ok &= report(f"cell golden {case}", run_cell(case), lines)
        for case in ("cfg1_h8", "cfg1_h32_hot", "rect_h16_o12", "t1_f1", "cfg1_h64_hot"):
            ok &= report(f"model golden {case}", run_model(case), lines)
        ok &= report("train-loss gradients cfg1_h32_hot", run_train("cfg1_h32_hot"), lines)
        ok &= report("MetNet oracle (vs its own fp32 run)", run_metnet(), lines)
        verdict[mode] = ok
        lines.append("")
        print("\n".join(lines[-13:]), flush=True)
    lines.append("verdict: " + ", ".join(f"{m}: {'GO' if v else 'NO-GO'}" for m, v in verdict.items()))
    out = os.path.join(ROOT, "profiles", "r06_f32e_numerics.txt")
    if not sys.argv[1:]:
        open(out, "w").write("\n".join(lines) + "\n")
    print(lines[-1])


if __name__ == "__main__":
    main()

"""Winograd F(2x2, 3x3) in bf16 - the NUMERICS half of the feasibility question, on the CPU:
what does rounding the TRANSFORMED operands to bf16 (what a bf16-MFMA Winograd kernel must do) cost against the direct bf16 convolution the kernels run
today (operands rounded once, fp32 accumulation)?  Shape of MetNet's conv3 / conv4: 256 -> 256 channels, 32x32 images, activations ~ N(0, 1) behind a
BatchNorm, weights as initialised (kaiming-uniform) or trained-like N(0, 0.03).  Reference: float64 convolution of the UNROUNDED fp32 operands.
    python tools/probe_winograd_numerics.py"""
import torch
import torch.nn.functional as TF

torch.manual_seed(0)
bf = lambda t: t.to(torch.bfloat16).to(torch.float32)
# F(2x2, 3x3) matrices (Lavin & Gray)
Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=torch.float32)
At = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def winograd(x, w, round_ops):
    """x [N,C,H,W] (H, W even), w [O,C,3,3] -> conv2d(x, w, padding=1) via F(2x2,3x3); round_ops: round U and V to bf16 before the products."""
    N, C, H, W = x.shape
    O = w.shape[0]
    xp = TF.pad(x, (1, 1, 1, 1))
    # 4x4 patches with stride 2: [N, C, H/2, W/2, 4, 4]
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)
    V = torch.einsum("ij,nchwjk,lk->nchwil", Bt, d, Bt)          # B^T d B
    U = torch.einsum("ij,ocjk,lk->ocil", G, w, G)                # G g G^T
    if round_ops:
        V, U = bf(V), bf(U)
    M = torch.einsum("nchwil,ocil->nohwil", V.double(), U.double()).float() if False else torch.einsum("nchwil,ocil->nohwil", V, U)   # fp32 accumulate
    Y = torch.einsum("ij,nohwjk,lk->nohwil", At, M, At)          # A^T M A: [N,O,H/2,W/2,2,2]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(N, O, H, W)


def rel(a, r):
    return float((a.double() - r).norm() / r.norm())


for wname, wstd in (("kaiming-uniform init (bound 1/sqrt(9*256))", None), ("N(0, 0.03)", 0.03)):
    N, C, O, H = 4, 256, 256, 32
    x = torch.randn(N, C, H, H)
    w = (torch.rand(O, C, 3, 3) * 2 - 1) / (9 * C) ** 0.5 if wstd is None else torch.randn(O, C, 3, 3) * wstd
    ref = TF.conv2d(x.double(), w.double(), padding=1)
    direct = TF.conv2d(bf(x), bf(w), padding=1)                  # what the kernels compute (fp32 accumulation of bf16 products)
    wino_exact = winograd(x, w, False)                           # fp32 transforms, fp32 products: the algorithm's own conditioning
    wino_bf16 = winograd(bf(x), bf(w), True)                     # stored bf16 operands, transformed operands rounded to bf16 again
    print(f"weights {wname}:")
    print(f"   direct bf16 operands            rel L2 {rel(direct, ref):.3e}")
    print(f"   Winograd, fp32 throughout       rel L2 {rel(wino_exact, ref):.3e}")
    print(f"   Winograd, bf16 U and V          rel L2 {rel(wino_bf16, ref):.3e}   = {rel(wino_bf16, ref) / rel(direct, ref):.2f} x the direct kernel's error")
    # the input gradient is the same convolution with flipped, transposed weights on the output gradient: same figures

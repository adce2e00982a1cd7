"""Does torch's own zeroing (Tensor.zero_, fill_(0), torch.zeros) inside a captured graph survive replays?  (The library's hipMemsetAsync resets did not.)"""
import torch
dev = torch.device("cuda:0")
def check(name, n, op):
    buf = torch.full((n,), 5.0, device=dev)
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        op(buf)
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        op(buf)
    bad = []
    for r in range(4):
        buf.fill_(5.0)
        g.replay(); torch.cuda.synchronize()
        nz = int((buf != 0).sum())
        if nz: bad.append((r, nz, float(buf.abs().max())))
    print(f"{name:12s} n={n:>10d}: {'OK' if not bad else bad}")
for n in (16, 64, 1024, 65536, 1 << 20, 1 << 24):
    check("zero_", n, lambda b: b.zero_())
    check("fill_(0)", n, lambda b: b.fill_(0.0))
    check("mul_(0)", n, lambda b: b.mul_(0.0))

"""Probe: sf_flash_attention_fwd (and _bwd when present) against a float64 evaluation on bf16-rounded operands, and timing at the DGMR shape."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, satflow_amd, bench
from satflow_amd import _hip
L = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "satflow_amd", "lib", "libsatflow_hip.so"))
L.sf_flash_attention_fwd.restype = C.c_int
L.sf_last_error_string.restype = C.c_char_p
dev = torch.device("cuda:0")
vp = lambda t: C.c_void_p(t.data_ptr() if t is not None else 0)

def fwd(q, k, v, scale=1.0, dt=1):
    b, n, dqk = q.shape; dv = v.shape[2]
    out = torch.empty(b, n, dv, device=dev); lse = torch.empty(b, n, device=dev)
    rc = L.sf_flash_attention_fwd(vp(q), q.stride(1), vp(k), k.stride(1), vp(v), v.stride(1), b, n, dqk, dv, C.c_float(scale), vp(out), out.stride(1), vp(lse), dt,
                                  C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, L.sf_last_error_string()
    return out, lse

torch.manual_seed(0)
for (b, n, dqk, dv) in ((2, 256, 32, 256), (3, 384, 16, 128), (1, 128, 32, 32), (2, 1024, 32, 64)):
    q = torch.randn(b, n, dqk, device=dev) * 1.5; k = torch.randn(b, n, dqk, device=dev) * 1.5; v = torch.randn(b, n, dv, device=dev)
    out, lse = fwd(q, k, v)
    r = lambda t: t.to(torch.bfloat16).double()
    S = r(q) @ r(k).transpose(1, 2)
    P = torch.softmax(S, -1)
    ref = P @ r(v)
    err = float((out.double() - ref).norm() / ref.norm())
    lerr = float((lse.double() - torch.logsumexp(S, -1)).abs().max())
    print(f"fwd b={b} n={n} dqk={dqk} dv={dv}: rel L2 {err:.2e}, lse max abs err {lerr:.2e}")
L.sf_flash_attention_bwd.restype = C.c_int

def bwd(q, k, v, out, lse, dout, scale=1.0, dt=1):
    b, n, dqk = q.shape; dv = v.shape[2]
    dq, dk, dvg = torch.full_like(q, float("nan")), torch.full_like(k, float("nan")), torch.full_like(v, float("nan"))
    delta = torch.empty(b, n, device=dev)
    rc = L.sf_flash_attention_bwd(vp(q), q.stride(1), vp(k), k.stride(1), vp(v), v.stride(1), vp(out), out.stride(1), vp(lse), vp(dout), dout.stride(1), b, n, dqk, dv,
                                  C.c_float(scale), vp(dq), dq.stride(1), vp(dk), dk.stride(1), vp(dvg), dvg.stride(1), vp(delta), dt,
                                  C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, L.sf_last_error_string()
    return dq, dk, dvg

for (b, n, dqk, dv, scale) in ((2, 256, 32, 256, 1.0), (3, 384, 16, 128, 0.5), (1, 128, 32, 32, 1.0), (2, 1024, 32, 64, 0.25)):
    q = torch.randn(b, n, dqk, device=dev); k = torch.randn(b, n, dqk, device=dev); v = torch.randn(b, n, dv, device=dev)
    dout = torch.randn(b, n, dv, device=dev)
    out, lse = fwd(q, k, v, scale)
    dq, dk, dvg = bwd(q, k, v, out, lse, dout, scale)
    qd, kd, vd = q.double().requires_grad_(), k.double().requires_grad_(), v.double().requires_grad_()
    ref = torch.softmax(scale * qd @ kd.transpose(1, 2), -1) @ vd
    ref.backward(dout.double())
    rel = lambda a, r: float((a.double() - r).norm() / r.norm())
    print(f"b={b} n={n} dqk={dqk} dv={dv} scale={scale}: out {rel(out, ref):.2e}  dq {rel(dq, qd.grad):.2e}  dk {rel(dk, kd.grad):.2e}  dv {rel(dvg, vd.grad):.2e}")
b, n, dqk, dv = 16, 4096, 32, 256
q = torch.randn(b, n, dqk, device=dev); k = torch.randn(b, n, dqk, device=dev); v = torch.randn(b, n, dv, device=dev)
t = bench.event_time(lambda: fwd(q, k, v), iters=10)
fl = 2.0 * b * n * n * (dqk + dv)
print(f"fwd 16 x 4096 x 4096 x ({dqk} + {dv}): {t*1e3:.3f} ms = {fl/t/1e12:.0f} TF/s")
out, lse = fwd(q, k, v); dout = torch.randn_like(out)
t = bench.event_time(lambda: bwd(q, k, v, out, lse, dout), iters=10)
print(f"bwd: {t*1e3:.3f} ms")

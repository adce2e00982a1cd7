"""Stage-by-stage comparison of the spatial discriminator at BASELINE configs[4] size (chn 64, 8 frames of 256x256) against oracle/dgmr.py in fp32 mode.
python tools/debug/dbg_disc_fullsize.py [chn] [size] [frames]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as TF
from oracle import dgmr as OD
from satflow_amd import functional as F
from satflow_amd import functional_gan as FG
from satflow_amd.models.layers.Discriminator import SpatialDiscriminator

chn = int(sys.argv[1]) if len(sys.argv) > 1 else 64
size = int(sys.argv[2]) if len(sys.argv) > 2 else 256
T = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dev = torch.device("cuda:0")
torch.manual_seed(0)
D = SpatialDiscriminator(chn=chn, n_class=4, in_channels=12).to(dev).train()
P = {k: v.detach().cpu().clone() for k, v in D.state_dict().items()}
x = torch.rand(1, T, 12, size, size) * 2 - 1
cls = torch.tensor([1])
frames = F.nchw_to_nhwc(x.reshape(T, 12, size, size).to(dev))
if os.environ.get("DBG_GEN"):   # the frames bench.DGMRWorkload's generator produces, and ITS discriminator (parameters are views of FlatAdam's flat buffer)
    import bench
    wl = bench.DGMRWorkload(dev, 1, 0, chn=chn, size=size, frames=T)
    D = wl.Ds
    P = {k: v.detach().cpu().clone() for k, v in D.state_dict().items()}
    wl.z.copy_(torch.randn(1, wl.in_dim).to(dev))
    cls = wl.cls.cpu()
    with torch.no_grad():
        frames = wl.G.run(wl.z, wl.cls)
    print("pad lanes max", float(frames[..., 12:].abs().max()), "frames max", float(frames.abs().max()), flush=True)
    x = frames.view(T, 1, size, size, frames.shape[-1])[..., :12].permute(1, 0, 4, 2, 3).contiguous().cpu()


def cmp(name, hip, ref_nchw):
    C = ref_nchw.shape[1]
    h = hip[..., :C].permute(0, 3, 1, 2).cpu()
    e = float((h - ref_nchw).norm() / (ref_nchw.norm() + 1e-30))
    print(f"{name:28s} rel L2 {e:.3e}   |ref| max {float(ref_nchw.abs().max()):.3e}  shape {tuple(ref_nchw.shape)}", flush=True)


ns = {}
with torch.no_grad():
    xr = x.view(T, 12, size, size)
    o = D.pre_conv[0].run(frames); r = OD._sn_conv(xr, P, "pre_conv.0.", ns, padding=1); cmp("pre_conv.0", o, r)
    o = FG.relu(o); r = TF.relu(r)
    o = D.pre_conv[2].run(o); r = OD._sn_conv(r, P, "pre_conv.2.", ns, padding=1); cmp("pre_conv.2", o, r)
    sk = D.pre_skip.run(FG.avg_pool2(frames)); rs = OD._sn_conv(TF.avg_pool2d(xr, 2), P, "pre_skip.", ns); cmp("pre_skip", sk, rs)
    o = FG.avg_pool2(o, sk); r = TF.avg_pool2d(r, 2) + rs; cmp("pool + skip", o, r)
    o = D.conv1.run(o); r = OD.gblock(r, P, "conv1.", ns); cmp("conv1 (GBlock)", o, r)
    o = D.attn.run(o); r = OD.self_attention(r, P, "attn."); cmp("attn", o, r)
    for i, blk in enumerate(D.conv2):
        o = blk.run(o); r = OD.gblock(r, P, f"conv2.{i}.", ns); cmp(f"conv2.{i}", o, r)
    from satflow_amd.models.layers.Discriminator import _head
    s = _head(D, o, cls.to(dev), T, time_major=True)
    rr = OD._disc_head(r, cls, T, P, ns)
    print("scores hip", s.cpu().tolist()); print("scores ref", rr.tolist())

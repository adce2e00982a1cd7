"""Shipped-yaml-like MetNet configuration (hidden 32, 24 lead times, 16 input channels) through one bf16a training step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, satflow_amd
from satflow_amd.models import LitMetNet
from satflow_amd.optim import FlatAdam
dev = torch.device("cuda:0")
for mode in ("bf16a", "bf16", "f32"):
    satflow_amd.set_compute_dtype(mode)
    torch.manual_seed(0)
    m = LitMetNet(input_channels=16, sat_channels=12, input_size=64, output_channels=1, hidden_dim=32, forecast_steps=24, num_layers=1, num_att_layers=2).to(dev)
    opt = FlatAdam(m.parameters(), lr=1e-3)
    x = torch.randn(2, 12, 16, 256, 256, device=dev); y = torch.randn(2, 24, 1, 16, 16, device=dev)
    for it in range(2):
        opt.zero_grad()
        loss = m.training_step((x, y), 0)
        loss.backward()
        opt.step()
    torch.cuda.synchronize()
    print(mode, "loss", float(loss), "finite grads", bool(torch.isfinite(opt.flat_g).all()), "|g|max", float(opt.flat_g.abs().max()))

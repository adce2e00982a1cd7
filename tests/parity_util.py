"""Helpers shared by the GPU parity tests: max-pool routing read back from the HIP kernels (tie-aware comparison) and
the published-error record."""
import json
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def publish(record: dict, name: str = "r06_parity_observed.jsonl") -> None:
    """Append one observed-error record ("publish the observed figure") under gpurun_out/ and print it."""
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", name), "a") as f:
        f.write(json.dumps(record) + "\n")
    print("PARITY", json.dumps(record))


def gpu_pool_routing(net, B: int, Tn: int):
    """Which element of every 2x2 window did the HIP max-poolings route to?  Read back from the kernels themselves.

    Max-pooling's backward is linear in the output gradient and sends each window's value to exactly one input element,
    so running the backward kernels on an all-ones (first pooling: one-hot per lead time) gradient returns the routing
    mask.  ``net.image_encoder.module.capture`` must hold the tensors of the forward pass ("base", "y4").
    Returns ``{("p1", l): bool[B*T,160,S,S], ("p2", l): bool[B*T,256,S/2,S/2]}`` in the oracle's frame order (b*T + t).
    """
    from satflow_amd import kernels as K
    from satflow_amd._hip import SF_F32, T, check, lib, stream_ptr

    enc = net.image_encoder.module
    base, y4 = enc.capture["base"], enc.capture["y4"]
    assert base.dtype == torch.float32 and y4.dtype == torch.float32, "routing read-back is for the fp32 parity mode"
    L = net.forecast_steps
    Fr, S, _, C = base.shape
    assert Fr == Tn * B
    w1 = enc.module[0].weight.detach().contiguous()
    O, I = w1.shape[0], w1.shape[1]
    cimg = net.image_channels
    dev = base.device
    routing = {}

    def to_oracle_order(mask_nhwc, c):  # [T*B, h, w, Cp] (frame t*B + b) -> bool [B*T, c, h, w] (frame b*T + t)
        m = mask_nhwc[..., :c].permute(0, 3, 1, 2)
        m = m.reshape(Tn, B, *m.shape[1:]).transpose(0, 1).reshape(B * Tn, *m.shape[1:])
        return (m > 0.5).cpu()

    ws = torch.empty(lib().sf_leadtime_pool_workspace_floats(L, C), dtype=torch.float32, device=dev)
    dw1 = torch.zeros_like(w1)
    for l in range(L):
        g = torch.zeros(L * Fr, S // 2, S // 2, C, dtype=torch.float32, device=dev)
        g[l * Fr:(l + 1) * Fr] = 1.0
        dbase = torch.empty_like(base)
        check(lib().sf_leadtime_pool_bwd(T(base), T(g), Fr, S, S, w1.data_ptr(), O, I, cimg, L, ws.data_ptr(), T(dbase), dw1.data_ptr(),
                                         SF_F32, stream_ptr()), "sf_leadtime_pool_bwd")
        m = to_oracle_order(dbase, O)
        assert int(m.sum()) == B * Tn * O * (S // 2) ** 2, "every window routes to exactly one element"
        routing[("p1", l)] = m
    n, h, w, c = y4.shape
    ones = torch.ones(n, h // 2, w // 2, c, dtype=torch.float32, device=dev)
    gx = K.maxpool2_bwd(y4, ones, (L, Tn))  # y4 images are [lead][time][batch]
    gx = gx.view(L, Fr, h, w, c)
    for l in range(L):
        m = to_oracle_order(gx[l], enc.output_channels)
        assert int(m.sum()) == B * Tn * enc.output_channels * (h // 2) * (w // 2)
        routing[("p2", l)] = m
    return routing

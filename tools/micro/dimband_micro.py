#!/usr/bin/env python3
"""tb_dimband_update at the cfg5 shape (8192 windows, forecast_len 48, hidden 64, 512 variables) against the two-kernel path.
usage: python tools/micro/dimband_micro.py [B Tf K V]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from timeband_amd import ops  # noqa: E402

B, Tf, K, V = (int(v) for v in sys.argv[1:5]) if len(sys.argv) >= 5 else (8192, 48, 64, 512)
dev = torch.device("cuda", 0)
g = torch.Generator(device="cpu").manual_seed(1)
hid = (torch.rand(B, Tf, K, generator=g) * 2 - 1).to(dev)
W = ((torch.rand(V, K, generator=g) * 2 - 1) / K ** 0.5).to(dev)
b = torch.zeros(V, device=dev)
dmin = torch.zeros(V, dtype=torch.float64, device=dev)
dmax = torch.ones(V, dtype=torch.float64, device=dev) * 3
hist = torch.zeros(Tf - 1, Tf, V, device=dev)
hist2 = torch.zeros_like(hist)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def two():
    pred = ops.linear_act(hid.view(B * Tf, K), W, b, ops.ACT_TANH).view(B, Tf, V)
    return ops.band_update(pred, hist, Tf - 1, hist2, 1.0, dmin, dmax)


print(f"linear + band: {timeit(two):8.1f} us")
for dbg, what in ((0, "fused"), (2, "fused, no sort / stats"), (6, "... and no denorm"), (14, "... and no tanh")):
    if dbg and os.environ.get("TB_DIMBAND_DBG") != "1":
        break
    t = timeit(lambda: ops.dimband_update(hid, W, b, ops.ACT_TANH + (dbg << 8), hist, Tf - 1, 1.0, dmin, dmax))
    print(f"{what:28s} {t:8.1f} us")

#!/usr/bin/env python3
"""Two identical LSTM forwards on two streams, for a kernel trace (do they run at the same time?)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from timeband_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
B, T, In, H = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "512,128,32,256").split(","))
g = torch.Generator(device="cpu").manual_seed(1)
k = 1.0 / H ** 0.5
mk = lambda *s: ((torch.rand(*s, generator=g) * 2 - 1) * k).to(dev)
w = [mk(4 * H, In), mk(4 * H, H), mk(4 * H), mk(4 * H)]
xs = [(torch.rand(B, T, In, generator=g) * 2 - 1).to(dev) for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
for it in range(4):
    torch.cuda.synchronize()
    for s, x in zip(streams, xs):
        with torch.cuda.stream(s), torch.no_grad():
            y = ops.lstm_layer(x, *w)
torch.cuda.synchronize()
ops.check_errors()

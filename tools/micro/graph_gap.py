#!/usr/bin/env python3
"""Gap between DEPENDENT small kernels: 400 in-place adds on one stream, launched one by one vs replayed
from a captured graph.  Tells what graph capture of the iteration could save (its ~200 kernel boundaries)."""
import time

import torch

dev = torch.device("cuda", 0)
x = torch.zeros(1 << 16, device=dev)
N = 400


def chain():
    for _ in range(N):
        x.add_(1.0)


for _ in range(3):
    chain()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    chain()
torch.cuda.synchronize()
t_stream = (time.perf_counter() - t0) / 10
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    chain()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        chain()
torch.cuda.synchronize()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    g.replay()
torch.cuda.synchronize()
t_graph = (time.perf_counter() - t0) / 10
print(f"{N} dependent kernels: stream {1e6 * t_stream / N:.2f} us per kernel, graph {1e6 * t_graph / N:.2f} us per kernel")

#!/usr/bin/env python3
"""How long does the host take to ENQUEUE one training iteration (no synchronisation), next to the
GPU time of the iteration?  If the two are close the step is host-bound.   usage: python tools/host_time.py [cfg1|cfg2]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda", 0)
cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]
ds, trainset, trainer = bench.build(cfg, dev)
B = cfg["B"]


def step(i):
    b = trainset.gather((i % (len(trainset) // B)) * B, B)
    trainer.iteration(b["encoded"], b["decoded"], training=True)


for i in range(5):
    step(i)
torch.cuda.synchronize()
K = 20
t0 = time.perf_counter()
for i in range(K):
    step(5 + i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3 * (t1 - t0) / K:.2f} ms/iteration, total {1e3 * (t2 - t0) / K:.2f} ms/iteration "
      f"(the GPU still had {1e3 * (t2 - t1):.1f} ms of work queued at the end)")

#!/usr/bin/env python3
"""Which stock torch (aten) ops launch device kernels inside one training iteration, and from where.
usage: python tools/aten_ops.py [cfg]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402

cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]
dev = torch.device("cuda", 0)
ds, trainset, trainer = bench.build(cfg, dev)
B = cfg["B"]
for i in range(3):
    b = trainset.gather(i * B, B)
    trainer.iteration(b["encoded"], b["decoded"], training=True)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    b = trainset.gather(3 * B, B)
    trainer.iteration(b["encoded"], b["decoded"], training=True)
    torch.cuda.synchronize()
for ev in prof.key_averages(group_by_stack_n=12):
    t = getattr(ev, "self_device_time_total", 0) or getattr(ev, "self_cuda_time_total", 0)
    if ev.key.startswith("aten::") and t > 0:
        st = [s.split("/")[-1] for s in (ev.stack or []) if "timeband_amd" in s or "bench.py" in s]
        print(f"{ev.key:24s} n={ev.count:3d} {t:8.1f}us <- {st[:3]}")

#!/usr/bin/env python3
"""Where the main stream's time goes inside one training iteration (events at the phase boundaries of
TIMEBANDTrainer.iteration).  usage: [TB_OVERLAP=..] python tools/phase_times.py [cfg] [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402

cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda", 0)
ds, trainset, trainer = bench.build(cfg, dev)
B = cfg["B"]
tot = {}
for i in range(steps + 3):
    trainer._marks = [] if i >= 3 else None
    b = trainset.gather((i % 8) * B, B)
    trainer.iteration(b["encoded"], b["decoded"], training=True)
    torch.cuda.synchronize()
    if trainer._marks:
        m = trainer._marks
        for (n0, e0), (n1, e1) in zip(m[:-1], m[1:]):
            tot[n1] = tot.get(n1, 0.0) + e0.elapsed_time(e1)
print("mask", trainer.overlap_mask, {k: round(v / steps, 3) for k, v in tot.items()}, "sum", round(sum(tot.values()) / steps, 3))

#!/usr/bin/env python3
"""cProfile of the host side of the training iteration (enqueue only): where do the Python microseconds go?
usage: python tools/host_profile.py [cfg1|cfg2]"""
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda", 0)
name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
cfg = bench.CONFIGS[name]
ds, trainset, trainer = bench.build(cfg, dev)
B = cfg["B"]


def step(i):
    b = trainset.gather((i % (len(trainset) // B)) * B, B)
    trainer.iteration(b["encoded"], b["decoded"], training=True)


for i in range(5):
    step(i)
torch.cuda.synchronize()
K = 30
pr = cProfile.Profile()
pr.enable()
for i in range(K):
    step(5 + i)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime")
print(f"{name}: per iteration, top functions by own time (ms)")
rows = []
for (fn, line, func), (cc, nc, tt, ct, callers) in st.stats.items():
    rows.append((tt / K * 1e3, ct / K * 1e3, nc / K, f"{os.path.basename(fn)}:{line}:{func}"))
rows.sort(reverse=True)
for tt, ct, nc, nm in rows[:28]:
    print(f"  own {tt:7.3f}  cum {ct:7.3f}  calls {nc:6.1f}  {nm}")
print("total own ms / iteration:", sum(r[0] for r in rows))

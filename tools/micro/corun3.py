#!/usr/bin/env python3
"""Upper bound for pipelining the generator's three LSTM layers in time: the three layer passes (hidden 64, 128,
256 at batch 512) run on three streams at once on INDEPENDENT inputs, against each alone and their serial sum.
usage: python tools/micro/corun3.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from timeband_amd import ops  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from corun import Layer, run  # noqa: E402,F401

if __name__ == "__main__":
    ss = [torch.cuda.Stream() for _ in range(3)]
    for kind in ("fwd", "bwd"):
        for B in (512,):
            ls = [Layer(B, 36, 64, 1), Layer(B, 64, 128, 2), Layer(B, 128, 256, 3)]
            jobs = [(s, None, l.fwd) if kind == "fwd" else (s, l.fwd, l.bwd) for s, l in zip(ss, ls)]
            alone = [run([j])[0] for j in jobs]
            tog = run(jobs)
            two = run(jobs[1:])
            print(f"{kind} B={B}: alone {[round(a, 3) for a in alone]} serial {sum(alone):.3f} | three together "
                  f"{[round(a, 3) for a in tog]} max {max(tog):.3f} | H128+H256 together {[round(a, 3) for a in two]}", flush=True)
    ops.check_errors()

#!/usr/bin/env python3
"""How much do two LSTM layer passes slow each other down when they run on two streams at once?
Prints the wall ms of each alone and of both together.   usage: python tools/micro/corun.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from timeband_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
T = 128


class Layer:
    def __init__(self, B, In, H, seed):
        g = torch.Generator(device="cpu").manual_seed(seed)
        k = 1.0 / H ** 0.5
        mk = lambda *s: ((torch.rand(*s, generator=g) * 2 - 1) * k).to(dev).requires_grad_()
        self.w = [mk(4 * H, In), mk(4 * H, H), mk(4 * H), mk(4 * H)]
        self.x = (torch.rand(B, T, In, generator=g) * 2 - 1).to(dev).requires_grad_()
        self.dy = (torch.rand(B, T, H, generator=g) * 2 - 1).to(dev)
        self.name = f"B{B}/H{H}"

    def fwd(self):
        self.y = ops.lstm_layer(self.x, *self.w)

    def bwd(self):
        self.y.backward(self.dy)
        self.x.grad = None


def run(jobs, reps=6):
    """jobs: list of (stream, prepare, timed); returns mean ms per job."""
    tot = [0.0] * len(jobs)
    for it in range(2 + reps):
        for s, prep, _ in jobs:
            with torch.cuda.stream(s):
                if prep:
                    prep()
        torch.cuda.synchronize()
        ev = []
        for s, _, timed in jobs:
            with torch.cuda.stream(s):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                timed()
                b.record()
                ev.append((a, b))
        torch.cuda.synchronize()
        if it >= 2:
            for i, (a, b) in enumerate(ev):
                tot[i] += a.elapsed_time(b)
    return [t / reps for t in tot]


if __name__ == "__main__":
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    pairs = [
        (("fwd", 512, 32, 256), ("fwd", 512, 32, 256)),
        (("fwd", 512, 32, 256), ("fwd", 512, 64, 128)),
        (("bwd", 512, 32, 256), ("fwd", 512, 32, 256)),
        (("bwd", 1024, 32, 256), ("fwd", 512, 32, 256)),
        (("bwd", 1024, 32, 256), ("fwd", 512, 64, 128)),
        (("bwd", 512, 32, 256), ("bwd", 512, 128, 256)),
        (("fwd", 1024, 32, 256), ("fwd", 512, 36, 64)),
    ]
    for pa, pb in pairs:
        la, lb = Layer(pa[1], pa[2], pa[3], 1), Layer(pb[1], pb[2], pb[3], 2)

        def job(kind, layer, s):
            return (s, None, layer.fwd) if kind == "fwd" else (s, layer.fwd, layer.bwd)

        ja, jb = job(pa[0], la, s1), job(pb[0], lb, s2)
        (ta,) = run([ja])
        (tb,) = run([jb])
        ca, cb = run([ja, jb])
        print(f"{pa[0]} {la.name:10s} alone {ta:.3f}  | {pb[0]} {lb.name:10s} alone {tb:.3f}  | together "
              f"{ca:.3f} / {cb:.3f}   (serial {ta + tb:.3f}, max {max(ca, cb):.3f})", flush=True)
    ops.check_errors()

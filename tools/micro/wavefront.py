#!/usr/bin/env python3
"""TIMING emulation of a chunked wavefront over the generator's three LSTM layers (hidden 64, 128, 256): the window is
cut into chunks of Tc steps, layer l works on chunk c on its own stream as soon as layer l-1 has finished chunk c.  The
chunks here start from a zero state (no state is carried, so the numbers are wrong and only the clock is read): what a
wavefront with carried state could gain at most, launches and handshakes included.
usage: python tools/micro/wavefront.py [B ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from timeband_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
T = 128
DIMS = [(36, 64), (64, 128), (128, 256)]
SLEEP = int(2e7)


def weights(In, H, seed, grad):
    g = torch.Generator(device="cpu").manual_seed(seed)
    k = 1.0 / H ** 0.5
    mk = lambda *s: ((torch.rand(*s, generator=g) * 2 - 1) * k).to(dev).requires_grad_(grad)
    return [mk(4 * H, In), mk(4 * H, H), mk(4 * H), mk(4 * H)]


def timed(fn, reps=6):
    tot = 0.0
    for it in range(2 + reps):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(SLEEP)      # the host enqueues everything while the GPU spins: only the GPU's time is read
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        if it >= 2:
            tot += a.elapsed_time(b)
    return tot / reps


def forward(B, Tc, train, ss):
    """Returns (serial ms, wavefront ms)."""
    ws = [weights(i, h, n, False) for n, (i, h) in enumerate(DIMS)]
    nch = T // Tc
    xs_full = [torch.rand(B, T, i, device=dev) for i, _ in DIMS]
    xs = [[torch.rand(B, Tc, i, device=dev).requires_grad_(train) for _ in range(nch)] for i, _ in DIMS]
    main = torch.cuda.current_stream()

    def serial():
        with torch.set_grad_enabled(train):
            for l in range(3):
                ops.lstm_layer(xs_full[l].requires_grad_(train), *ws[l])

    def wave():
        for s in ss:
            s.wait_stream(main)
        ev = [[torch.cuda.Event() for _ in range(nch)] for _ in range(3)]
        with torch.set_grad_enabled(train):
            for d in range(nch + 2):            # diagonal order: the host enqueues what can start first
                for l in range(3):
                    c = d - l
                    if 0 <= c < nch:
                        with torch.cuda.stream(ss[l]):
                            if l:
                                ss[l].wait_event(ev[l - 1][c])
                            ops.lstm_layer(xs[l][c], *ws[l])
                            ev[l][c].record(ss[l])
        for s in ss:
            main.wait_stream(s)

    return timed(serial), timed(wave)


def backward(B, Tc, ss):
    """dx only (weights without gradient): the chain part of BPTT; dW is one full-length product per layer either way."""
    ws = [weights(i, h, n, False) for n, (i, h) in enumerate(DIMS)]
    nch = T // Tc
    main = torch.cuda.current_stream()
    xf = [torch.rand(B, T, i, device=dev).requires_grad_() for i, _ in DIMS]
    dyf = [torch.rand(B, T, h, device=dev) for _, h in DIMS]
    xc = [[torch.rand(B, Tc, i, device=dev).requires_grad_() for _ in range(nch)] for i, _ in DIMS]
    dyc = [torch.rand(B, Tc, h, device=dev) for _, h in DIMS]
    st = {}

    def prep_serial():
        st["y"] = [ops.lstm_layer(xf[l], *ws[l]) for l in range(3)]

    def serial():
        for l in (2, 1, 0):
            st["y"][l].backward(dyf[l])
            xf[l].grad = None

    def prep_wave():
        st["yc"] = [[ops.lstm_layer(xc[l][c], *ws[l]) for c in range(nch)] for l in range(3)]

    def wave():
        for s in ss:
            s.wait_stream(main)
        ev = [[torch.cuda.Event() for _ in range(nch)] for _ in range(3)]
        for d in range(nch + 2):
            for l in (2, 1, 0):
                c = d - (2 - l)
                if 0 <= c < nch:
                    with torch.cuda.stream(ss[l]):
                        if l < 2:
                            ss[l].wait_event(ev[l + 1][c])
                        st["yc"][l][c].backward(dyc[l])
                        xc[l][c].grad = None
                        ev[l][c].record(ss[l])
        for s in ss:
            main.wait_stream(s)

    def run(prep, fn, reps=6):
        tot = 0.0
        for it in range(2 + reps):
            prep()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda._sleep(SLEEP)
            a.record()
            fn()
            b.record()
            torch.cuda.synchronize()
            if it >= 2:
                tot += a.elapsed_time(b)
        return tot / reps

    return run(prep_serial, serial), run(prep_wave, wave)


if __name__ == "__main__":
    Bs = [int(a) for a in sys.argv[1:]] or [128, 512]
    ss = [torch.cuda.Stream() for _ in range(3)]
    for B in Bs:
        for Tc in (32, 64):
            for train in (False, True):
                s, w = forward(B, Tc, train, ss)
                print(f"fwd  B={B} Tc={Tc} train={int(train)}: serial {s:.3f} ms  wavefront {w:.3f} ms", flush=True)
            s, w = backward(B, Tc, ss)
            print(f"bwd  B={B} Tc={Tc} (dx only): serial {s:.3f} ms  wavefront {w:.3f} ms", flush=True)
    ops.check_errors()

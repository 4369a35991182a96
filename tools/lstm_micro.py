#!/usr/bin/env python3
"""Micro-benchmark of one LSTM layer through the C-ABI: kernel ms from the library's own HIP-event
timers (recurrent kernels only) and wall ms of forward / backward including the GEMMs.
usage: python tools/lstm_micro.py [B,T,In,H ...]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from timeband_amd import _lib, ops  # noqa: E402

cases = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or \
    [(512, 128, 32, 256), (1024, 128, 32, 256), (128, 128, 32, 256), (512, 128, 64, 128), (512, 128, 36, 64)]
lib = _lib.load()
dev = torch.device("cuda", 0)
kids = {"fwd": _lib.K_LSTM_FWD, "bwd": _lib.K_LSTM_BWD}


def read(k):
    n, ms, by = ctypes.c_int(0), ctypes.c_double(0), ctypes.c_double(0)
    lib.tb_prof_read(k, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(by))
    return n.value, ms.value


for (B, T, In, H) in cases:
    g = torch.Generator(device="cpu").manual_seed(1)
    k = 1.0 / H ** 0.5
    mk = lambda *s: ((torch.rand(*s, generator=g) * 2 - 1) * k).to(dev).requires_grad_()
    w_ih, w_hh, b_ih, b_hh = mk(4 * H, In), mk(4 * H, H), mk(4 * H), mk(4 * H)
    x = (torch.rand(B, T, In, generator=g) * 2 - 1).to(dev).requires_grad_()
    dy = (torch.rand(B, T, H, generator=g) * 2 - 1).to(dev)
    reps = 5
    for it in range(2 + reps):
        if it == 2:
            torch.cuda.synchronize()
            for kk in kids.values():
                lib.tb_prof_enable(kk, 1)
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            tf = tb = 0.0
        if it >= 2:
            e0.record()
        y = ops.lstm_layer(x, w_ih, w_hh, b_ih, b_hh)
        if it >= 2:
            e1.record()
        y.backward(dy)
        if it >= 2:
            e2.record()
            torch.cuda.synchronize()
            tf += e0.elapsed_time(e1)
            tb += e1.elapsed_time(e2)
        x.grad = None
    torch.cuda.synchronize()
    ops.check_errors()
    res = {}
    for name, kk in kids.items():
        n, ms = read(kk)
        lib.tb_prof_enable(kk, 0)
        res[name] = ms / max(n, 1)
    print(f"B={B} T={T} in={In} H={H} mode={os.environ.get('TB_LSTM_MODE', '0')}: "
          f"recurrent fwd {res['fwd']:.3f} ms ({res['fwd'] / T * 1e3:.2f} us/step)  bwd {res['bwd']:.3f} ms "
          f"({res['bwd'] / T * 1e3:.2f} us/step)  | layer fwd {tf / reps:.3f} ms  bwd {tb / reps:.3f} ms", flush=True)

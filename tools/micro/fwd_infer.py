#!/usr/bin/env python3
"""Recurrent forward kernel time with and without the BPTT stash (train vs inference), HIP-event timed."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from timeband_amd import _lib, ops
lib = _lib.load(); dev = torch.device("cuda", 0)
def read(k):
    n, ms, by = ctypes.c_int(0), ctypes.c_double(0), ctypes.c_double(0)
    lib.tb_prof_read(k, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(by)); return n.value, ms.value
cases = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(512, 128, 32, 256), (512, 128, 128, 256), (1024, 128, 32, 256), (512, 128, 64, 128), (512, 128, 36, 64)]
for (B, T, In, H) in cases:
    g = torch.Generator(device="cpu").manual_seed(1); k = 1.0 / H ** 0.5
    mk = lambda *s: ((torch.rand(*s, generator=g) * 2 - 1) * k).to(dev)
    w = [mk(4 * H, In), mk(4 * H, H), mk(4 * H), mk(4 * H)]
    x = (torch.rand(B, T, In, generator=g) * 2 - 1).to(dev)
    for mode in ("train", "infer"):
        for it in range(8):
            if it == 3:
                torch.cuda.synchronize(); lib.tb_prof_enable(_lib.K_LSTM_FWD, 1)
            if mode == "train":
                xs = x.clone().requires_grad_(); y = ops.lstm_layer(xs, *w)
            else:
                with torch.no_grad(): y = ops.lstm_layer(x, *w)
        torch.cuda.synchronize(); n, ms = read(_lib.K_LSTM_FWD); lib.tb_prof_enable(_lib.K_LSTM_FWD, 0)
        print(f"B={B} in={In} H={H} {mode}: {ms / n / T * 1e3:.2f} us/step", flush=True)
ops.check_errors()

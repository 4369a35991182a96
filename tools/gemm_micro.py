#!/usr/bin/env python3
"""Micro-benchmark of the fp32 MFMA GEMM through the C-ABI (linear forward = NT, backward = NN dx + TN dW).
usage: python tools/gemm_micro.py [M,K,N ...]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from timeband_amd import _lib, ops  # noqa: E402

cases = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or \
    [(65536, 256, 1024), (65536, 128, 1024), (65536, 32, 1024), (65536, 64, 512), (65536, 36, 256),
     (131072, 128, 128), (65536, 256, 32), (65536, 256, 1)]
lib = _lib.load()
dev = torch.device("cuda", 0)


def prof():
    n, ms, by = ctypes.c_int(0), ctypes.c_double(0), ctypes.c_double(0)
    lib.tb_prof_read(_lib.K_GEMM, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(by))
    return n.value, ms.value


for (M, K, N) in cases:
    x = torch.randn(M, K, device=dev).requires_grad_()
    W = torch.randn(N, K, device=dev).requires_grad_()
    b = torch.randn(N, device=dev).requires_grad_()
    dy = torch.randn(M, N, device=dev)
    reps = 5
    out = {}
    for phase in ("fwd", "bwd"):
        for it in range(2 + reps):
            if it == 2:
                torch.cuda.synchronize()
                lib.tb_prof_enable(_lib.K_GEMM, 1)
            if phase == "fwd":
                with torch.no_grad():
                    ops.linear_act(x, W, b, 0)
            else:
                y = ops.linear_act(x, W, b, 0)
                if it >= 2:
                    torch.cuda.synchronize()
                    lib.tb_prof_enable(_lib.K_GEMM, 1)   # restart: time only the backward GEMMs
                y.backward(dy)
                if it >= 2:
                    torch.cuda.synchronize()
                    n, ms = prof()
                    out.setdefault("bwd_ms", 0.0)
                    out["bwd_ms"] += ms
                    out["bwd_n"] = n
                    lib.tb_prof_enable(_lib.K_GEMM, 0)
                x.grad = W.grad = b.grad = None
        if phase == "fwd":
            torch.cuda.synchronize()
            n, ms = prof()
            lib.tb_prof_enable(_lib.K_GEMM, 0)
            out["fwd_ms"] = ms / max(n, 1)
    fl = 2.0 * M * K * N
    fwd, bwd = out["fwd_ms"], out["bwd_ms"] / reps
    if fwd <= 0 or bwd <= 0:      # one-column products go through the head kernels, not the GEMM
        print(f"M={M} K={K} N={N}: not a GEMM launch (streaming head kernels)", flush=True)
        continue
    print(f"M={M} K={K} N={N}: fwd {fwd:.3f} ms = {fl / fwd / 1e9:.1f} TF | bwd (dx+dW, {out['bwd_n']} launches) "
          f"{bwd:.3f} ms = {2 * fl / bwd / 1e9:.1f} TF", flush=True)

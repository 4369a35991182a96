#!/usr/bin/env python3
"""The LSTM backward's GEMMs at their real shapes through the C-ABI: dW_ih, dW_hh (tb_lstm_wgrad) and dx (linear backward
with M = rows x steps, K_gemm = 4H).  usage: python tools/micro/wgrad_micro.py [B,T,In,H ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from timeband_amd import _lib, ops  # noqa: E402

cases = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(512, 128, 128, 256), (1024, 128, 32, 256), (1024, 128, 128, 256), (1024, 128, 64, 128)]
lib = _lib.load()
dev = torch.device("cuda", 0)
rt = ops.Runtime.get(dev)


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for (B, T, In, H) in cases:
    x = torch.randn(B, T, In, device=dev)
    y = torch.randn(B, T, H, device=dev)
    dg = torch.randn(B, T, 4 * H, device=dev)
    wih = torch.randn(4 * H, In, device=dev)
    dwi, dwh = torch.zeros(4 * H, In, device=dev), torch.zeros(4 * H, H, device=dev)
    dx = torch.empty(B, T, In, device=dev)
    ws = torch.empty(lib.tb_lstm_wgrad_workspace_bytes(In, H) + 256, dtype=torch.uint8, device=dev)
    P = lambda t: 0 if t is None else t.data_ptr()
    st = lambda: torch.cuda.current_stream().cuda_stream
    f_ih = lambda: _lib.call("tb_lstm_wgrad", P(x), P(y), P(dg), P(dwi), 0, B, T, In, H, 1, P(ws), ws.numel(), st())
    f_hh = lambda: _lib.call("tb_lstm_wgrad", P(x), P(y), P(dg), 0, P(dwh), B, T, In, H, 1, P(ws), ws.numel(), st())
    ws2 = torch.empty(lib.tb_linear_workspace_bytes(B * T, In, 4 * H) + 256, dtype=torch.uint8, device=dev)
    f_dx = lambda: _lib.call("tb_linear_bwd", P(dg), P(x), P(wih), P(dx), 0, 0, B * T, In, 4 * H, 0, P(ws2), ws2.numel(), st())
    t_ih, t_hh, t_dx = timed(f_ih), timed(f_hh), timed(f_dx)
    fl = 2.0 * B * T * 4 * H
    print(f"B={B} T={T} in={In} H={H}: dW_ih {t_ih:7.1f} us ({fl * In * 3 / t_ih / 1e6:6.0f} TF bf16) | dW_hh {t_hh:7.1f} us "
          f"({fl * H * 3 / t_hh / 1e6:6.0f} TF) | dx {t_dx:7.1f} us ({fl * In * 3 / t_dx / 1e6:6.0f} TF)", flush=True)

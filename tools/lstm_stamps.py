#!/usr/bin/env python3
"""Where does a timestep of the persistent LSTM kernels go?  Runs one layer forward+backward on the
DIAGNOSTIC build (python -m timeband_amd.build --stamps -> libtimeband_hip_stamps.so: s_memtime stamps
around the segments of the step loop, cluster 0 / member 0) and prints per-step microseconds per
segment and wave.  Read the SHARES, not the length: the stamped build forbids overlaps the real one has.
usage: python tools/lstm_stamps.py [B,T,In,H ...]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["TB_EXT"] = "0"   # the C++ extension is linked against the regular library: go through ctypes here
import torch  # noqa: E402
from timeband_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), "libtimeband_hip_stamps.so")
from timeband_amd import ops  # noqa: E402

cases = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:] if not a.startswith("--")] or [(512, 128, 32, 256), (1024, 128, 32, 256), (512, 128, 36, 64)]
lib = _lib.load()
dev = torch.device("cuda", 0)
FWD = ["top/rearm/pf", "mfma", "sync1", "publish", "poll", "gates", "vm wait", "pull", "lds+y/stash st", "sync3", "same_l2*1000", "xcc_reg"]
BWD = ["top", "gate grads+lds+pf", "sync", "re-arm", "sync", "dgates st", "factors", "-", "reduce-scatter", "lds+mfma", "X st"]
for (B, T, In, H) in cases:
    g = torch.Generator(device="cpu").manual_seed(1)
    k = 1.0 / H ** 0.5
    mk = lambda *s: ((torch.rand(*s, generator=g) * 2 - 1) * k).to(dev).requires_grad_()
    w_ih, w_hh, b_ih, b_hh = mk(4 * H, In), mk(4 * H, H), mk(4 * H), mk(4 * H)
    x = (torch.rand(B, T, In, generator=g) * 2 - 1).to(dev).requires_grad_()
    dy = (torch.rand(B, T, H, generator=g) * 2 - 1).to(dev)
    for it in range(3):
        if "--corun" in sys.argv:      # the same forward on two streams at once
            x2 = x.detach().clone()
            streams = [torch.cuda.Stream(), torch.cuda.Stream()]
            torch.cuda.synchronize()
            for s_, xx in zip(streams, (x, x2)):
                with torch.cuda.stream(s_), torch.no_grad():
                    y = ops.lstm_layer(xx, w_ih, w_hh, b_ih, b_hh)
            torch.cuda.synchronize()
            continue
        if "--infer" in sys.argv:
            with torch.no_grad():
                y = ops.lstm_layer(x, w_ih, w_hh, b_ih, b_hh)
            continue
        y = ops.lstm_layer(x, w_ih, w_hh, b_ih, b_hh)
        y.backward(dy)
    torch.cuda.synchronize()
    ops.check_errors()
    buf = (ctypes.c_ulonglong * 96)()
    lib.tb_debug_stamps.restype = ctypes.c_int
    assert lib.tb_debug_stamps(buf) == 0
    print(f"B={B} T={T} in={In} H={H}  (s_memtime ticks per step)")
    for kname, names, base in (("fwd", FWD, 0), ("bwd", BWD, 48)):
        for w in range(4):
            v = [buf[base + w * 12 + i] / T for i in range(len(names))]
            print(f"  {kname} wave{w}: " + "  ".join(f"{n}={x_:.0f}" for n, x_ in zip(names, v)) + f"  | total {sum(v):.0f}")

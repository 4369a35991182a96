import sys, os
sys.path.insert(0, "/root/repo")
import torch
from timeband_amd import ops
dev = torch.device("cuda", 0)
K = 256
W = torch.randn(1, K, device=dev) * 0.05
b = torch.zeros(1, device=dev)
for M in (65536, 131072):
    x = torch.randn(M, K, device=dev)
    big = torch.empty(512 * 1024 * 1024 // 4, device=dev)
    def t(fn, n=5):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        tot = 0
        for i in range(n):
            fn(0)
            torch.cuda.synchronize()
            a.record(); fn(1); e.record(); torch.cuda.synchronize()
            tot += a.elapsed_time(e)
        return tot / n * 1e3
    with torch.no_grad():
        warm = t(lambda ph: ops.linear_act(x, W, b, ops.ACT_SIGMOID))
        def cold(ph):
            if ph == 0:
                big.zero_()      # flush L2 / MALL
            else:
                ops.linear_act(x, W, b, ops.ACT_SIGMOID)
        c = t(cold)
        def fresh(ph):
            if ph == 0:
                x.mul_(1.0)      # x just written
            else:
                ops.linear_act(x, W, b, ops.ACT_SIGMOID)
        f = t(fresh)
    print(f"M={M}: warm {warm:.1f} us  after 512MB flush {c:.1f} us  x just rewritten {f:.1f} us  ({M*K*4/1e6:.0f} MB)")

# backward: dx + dW + db
for M in (65536, 131072):
    x = torch.randn(M, K, device=dev, requires_grad=True)
    Wp, bp = W.clone().requires_grad_(), b.clone().requires_grad_()
    dy = torch.randn(M, 1, device=dev)
    tot = 0.0
    for i in range(7):
        y = ops.linear_act(x, Wp, bp, ops.ACT_SIGMOID)
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); y.backward(dy); e.record(); torch.cuda.synchronize()
        if i >= 2:
            tot += a.elapsed_time(e)
        x.grad = None
    print(f"M={M}: backward {tot / 5 * 1e3:.1f} us (through autograd)")

"""HBM write / copy / read bandwidth as torch sees it (fill_, copy_, sum) at the sizes of the gate matrix."""
import torch
dev = torch.device("cuda", 0)
for mb in (67, 268, 1024):
    n = mb * 1024 * 1024 // 4
    a = torch.empty(n, device=dev); b = torch.empty(n, device=dev)
    for name, fn, bytes_ in (("fill", lambda: a.fill_(1.0), 4 * n), ("copy", lambda: b.copy_(a), 8 * n), ("sum", lambda: a.sum(), 4 * n)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"{mb:5d} MB {name}: {ms * 1e3:8.1f} us  {bytes_ / ms / 1e9:7.2f} TB/s")

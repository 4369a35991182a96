import os, sys, subprocess
sys.path.insert(0, "/root/repo")
if len(sys.argv) > 1:
    import torch
    from timeband_amd import ops
    dev = torch.device("cuda", 0)
    B, T, In, H = 32, 6, 32, 256
    g = torch.Generator(device="cpu").manual_seed(1)
    k = 1.0 / H ** 0.5
    mk = lambda *s: ((torch.rand(*s, generator=g) * 2 - 1) * k).to(dev).requires_grad_()
    w = [mk(4 * H, In), mk(4 * H, H), mk(4 * H), mk(4 * H)]
    x = (torch.rand(B, T, In, generator=g) * 2 - 1).to(dev).requires_grad_()
    dy = (torch.rand(B, T, H, generator=g) * 2 - 1).to(dev)
    y = ops.lstm_layer(x, *w); y.backward(dy); torch.cuda.synchronize(); ops.check_errors()
    torch.save(x.grad.cpu(), sys.argv[1])
else:
    import torch
    env = dict(os.environ)
    env["TB_DX_FUSE"] = "0"; subprocess.check_call([sys.executable, __file__, "/tmp/dx0.pt"], env=env)
    env["TB_DX_FUSE"] = "1"; subprocess.check_call([sys.executable, __file__, "/tmp/dx1.pt"], env=env)
    a, b = torch.load("/tmp/dx0.pt"), torch.load("/tmp/dx1.pt")
    print("max abs ref", a.abs().max().item(), "max err", (a - b).abs().max().item())
    e = (a - b).abs()
    print("err by t", e.amax(dim=(0, 2)))
    print("err by feature", e.amax(dim=(0, 1)))
    print("err by row", e.amax(dim=(1, 2)))
    print("ratio sample", (b[0, :, :4] / a[0, :, :4]))

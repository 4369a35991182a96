"""CPU baseline port: the Gen-A iteration on stock PyTorch CPU ops (nn.LSTM -> oneDNN, nn.Linear,
MSELoss, SmoothL1Loss, Adam, RMSprop), i.e. the same library calls the reference makes.

TEST / BENCH INFRASTRUCTURE ONLY (see oracle/timeband_oracle.py header): imported by tests/ and by
bench.py's cpu_baseline leg, never by timeband_amd/.  The reference's files cannot travel to the GPU
box, so this port is what is timed there on the host cores; its equality with the reference is pinned
by the same golden vectors (tests/test_oracle_golden.py::test_torch_port_matches_reference).
Follows source/utils/lstm_layer.py:18-52,68-89, source/loss.py:93-141, source/trainer.py:142-194.
"""
import time

import torch
import torch.nn as nn
import torch.nn.functional as F


class PortG(nn.Module):
    def __init__(self, E, V, To, Tf, H, L=1):
        super().__init__()
        self.H, self.To, self.Tf = H, To, Tf
        self.lstm0 = nn.LSTM(E, H // 4, L, batch_first=True)
        self.lstm1 = nn.LSTM(H // 4, H // 2, L, batch_first=True)
        self.lstm2 = nn.LSTM(H // 2, H, L, batch_first=True)
        self.seq_layer = nn.Sequential(nn.Linear(To, Tf))
        self.dim_layer = nn.Sequential(nn.Linear(H, V), nn.Tanh())

    def forward(self, x):
        B = x.size(0)
        r = self.lstm2(self.lstm1(self.lstm0(x)[0])[0])[0]
        s = self.seq_layer(r.contiguous().view(B, self.H, self.To))
        return self.dim_layer(s.contiguous().view(B, self.Tf, self.H))


class PortD(nn.Module):
    def __init__(self, V, H, L=1):
        super().__init__()
        self.H = H
        self.lstm = nn.LSTM(V, H, L, batch_first=True)
        self.linear = nn.Sequential(nn.Linear(H, 1), nn.Sigmoid())

    def forward(self, y):
        B, T = y.shape[:2]
        r = self.lstm(y)[0]
        return self.linear(r.contiguous().view(B * T, self.H)).view(B, T, 1)


def _gp(true, pred):
    return torch.square(1 - torch.sqrt(torch.sum(torch.square(true - pred)))) / true.size(0)


def dis_loss(true, pred, Dy, DGx, critic, gp_w):
    if critic:
        return DGx.mean() - Dy.mean() + gp_w * _gp(true, pred)
    return F.mse_loss(DGx, torch.full_like(DGx, 0.1)) + F.mse_loss(Dy, torch.full_like(Dy, 0.9))


def gen_loss(true, pred, DGx, l1_w, l2_w, gp_w):
    return (F.mse_loss(DGx, torch.full_like(DGx, 0.1)) + l1_w * F.smooth_l1_loss(pred, true)
            + l2_w * F.mse_loss(pred, true) + gp_w * _gp(true, pred))


class PortTrainer:
    def __init__(self, netD, netG, lr=2e-4, w=(50, 50, 50), critic=1):
        self.netD, self.netG, self.lr, self.w, self.critic = netD, netG, lr, w, critic
        self.new_epoch()

    def new_epoch(self):
        self.optD = torch.optim.Adam(self.netD.parameters(), lr=self.lr)
        self.optG = torch.optim.Adam(self.netG.parameters(), lr=self.lr)

    def iteration(self, x, y, xc, yc):
        D, G = self.netD, self.netG
        l1, l2, gp = self.w
        rms = torch.optim.RMSprop(D.parameters(), lr=self.lr)
        for _ in range(self.critic):
            fake = G(xc)
            rms.zero_grad()
            dis_loss(yc, fake, D(yc), D(fake), True, gp).backward()
            rms.step()
        self.optD.zero_grad()
        self.optG.zero_grad()
        fake = G(x)
        errD = dis_loss(y, fake, D(y), D(fake), False, gp)
        errD.backward()
        self.optD.step()
        fake = G(x)
        errG = gen_loss(y, fake, D(fake), l1, l2, gp)
        errG.backward()
        self.optG.step()
        return errD.item(), errG.item(), fake.detach()


def time_cpu_baseline(V, E, To, Tf, H, L, B, iters=3, warmup=1, threads=None, seed=31):
    """windows/s of the port on the host cores: `warmup` + `iters` iterations at the bench shape."""
    if threads:
        torch.set_num_threads(threads)
    torch.manual_seed(seed)
    D, G = PortD(V, H, L), PortG(E, V, To, Tf, H, L)
    tr = PortTrainer(D, G)
    x, y = torch.rand(B, To, E) * 2 - 1, torch.rand(B, Tf, V) * 2 - 1
    xc, yc = torch.rand(Tf, To, E) * 2 - 1, torch.rand(Tf, Tf, V) * 2 - 1
    for _ in range(warmup):
        tr.iteration(x, y, xc, yc)
    t0 = time.perf_counter()
    for _ in range(iters):
        tr.iteration(x, y, xc, yc)
    dt = time.perf_counter() - t0
    return B * iters / dt, dt / iters, torch.get_num_threads()

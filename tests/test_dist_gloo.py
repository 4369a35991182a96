"""Data-parallel design check on CPU (gloo, world_size 2): the sharding + all-reduce scheme of
SURVEY.md §8e / timeband_amd.trainer reproduces the single-process result.

Scheme under test (the same arithmetic timeband_amd.ops performs on the GPU with RCCL):
  * each rank runs G / D on its contiguous shard of the global batch;
  * the loss sums (sum SmoothL1, sum d^2, ...) are all-reduced; each rank scales its loss gradients
    by the GLOBAL counts, with the gradient-penalty norm taken over the GLOBAL batch;
  * ONE sum all-reduce per network of the flat gradient gives the full-batch gradient.
The oracle is the compute engine here (the HIP kernels need a GPU); what is verified is the
distribution logic.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _flat(g, keys):
    return np.concatenate([g[k].reshape(-1) for k in keys]).astype(np.float64)


def _worker(rank, world, port, golden, out):
    sys.path.insert(0, ROOT)
    from oracle import timeband_oracle as O
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    z = np.load(golden)
    sub = lambda p: {k[len(p) + 1:]: z[k] for k in z.files if k.startswith(p + ".")}
    V, E, To, Tf, H, L, B = (int(v) for v in z["meta"])
    pD, pG = sub("init.netD"), sub("init.netG")
    x, y = z["it1.x"][:4], z["it1.y"][:4]          # global batch 4 -> 2 per rank
    k = x.shape[0] // world
    xs, ys = x[rank * k:(rank + 1) * k], y[rank * k:(rank + 1) * k]

    def allsum(a):
        t = torch.tensor(np.asarray(a, np.float64))
        dist.all_reduce(t)
        return t.numpy()

    # generator phase on the shard, with global sums / counts
    fake, cG = O.generator_fwd(pG, xs, L)
    DGx, cD = O.discriminator_fwd(pD, fake, L)
    sums = allsum(O.recon_sums(ys, fake))
    nP, nD, Bg = fake.size * world, DGx.size * world, x.shape[0]
    adv, dDGx = O.gan_loss(DGx, False, n_global=nD)
    l1, l2, gp, dpred, _ = O.recon_terms(ys, fake, 50, 50, 50, sums=tuple(sums), n_global=nP, batch_global=Bg)
    loss = allsum([adv])[0] + float(l1) + float(l2) + float(gp)
    gD, gG = {}, {}
    dfake = O.discriminator_bwd(gD, dDGx, cD, need_dx=True)
    O.generator_bwd(gG, (dfake + dpred).astype(np.float32), cG)
    keys = sorted(gG)
    flat = allsum(_flat(gG, keys))                 # ONE all-reduce of the flat gradient buffer
    if rank == 0:
        # single-process reference on the full batch
        fake1, cG1 = O.generator_fwd(pG, x, L)
        DGx1, cD1 = O.discriminator_fwd(pD, fake1, L)
        Lo = O.Losses(50, 50, 50)
        e1, dDG1, dp1 = Lo.gen_loss(y, fake1, DGx1)
        g1 = {}
        df1 = O.discriminator_bwd({}, dDG1, cD1, need_dx=True)
        O.generator_bwd(g1, (df1 + dp1).astype(np.float32), cG1)
        ref = _flat(g1, keys)
        np.save(out, np.array([abs(loss - float(e1)) / abs(float(e1)),
                               np.abs(flat - ref).max() / np.abs(ref).max()]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_dp2_matches_single_process(tmp_path, golden_dir):
    golden = os.path.join(golden_dir, "scenario_uneven.npz")
    out = str(tmp_path / "res.npy")
    mp.spawn(_worker, args=(2, _free_port(), golden, out), nprocs=2, join=True)
    loss_err, grad_err = np.load(out)
    assert loss_err < 1e-5, loss_err
    assert grad_err < 1e-4, grad_err


def _worker_replicated(rank, world, port, out):
    """ops.replicated(): _world() must read 1 inside the context and the real size outside."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from timeband_amd import ops
    from timeband_amd.trainer import TIMEBANDTrainer
    a = ops._world()
    with ops.replicated():
        b = ops._world()
    c = ops._world()
    x = torch.arange(12.0).reshape(6, 2)
    cx, cy, shared = TIMEBANDTrainer._shard_critic(x, x)         # 6 windows / 2 ranks -> 3 each
    ok = shared and torch.equal(cx, x[rank * 3:(rank + 1) * 3])
    cx2, _, shared2 = TIMEBANDTrainer._shard_critic(x[:5], x[:5])  # 5 windows / 2 ranks -> ragged slices 2 + 3
    ok = ok and shared2 and cx2.shape[0] == (2 if rank == 0 else 3)
    cx3, _, shared3 = TIMEBANDTrainer._shard_critic(x[:1], x[:1])  # fewer windows than ranks -> replicated
    ok = ok and (not shared3) and cx3.shape[0] == 1
    if rank == 0:
        np.save(out, np.array([a, b, c, int(ok)]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_replicated_mode_and_critic_sharding(tmp_path):
    out = str(tmp_path / "r.npy")
    mp.spawn(_worker_replicated, args=(2, _free_port(), out), nprocs=2, join=True)
    assert list(np.load(out)) == [2, 1, 2, 1]

"""GPU: the data-parallel product path on the card.

  * world = 1 over RCCL ("nccl" backend, one rank), collectives forced on: TIMEBANDTrainer.iteration and a
    whole train_step pass go through every data-parallel code path -- bucketed gradient all-reduces on the
    communication stream, the early reconstruction-sum all-reduce, loader slices, the band halo / adjust
    carry -- and must leave parameters and scores bit-identical to the plain single-process run.
  * world = 2 on ONE GPU (gloo, device tensors staged through the host: a debugging transport, the box has
    one card): two ranks, each on its slice of every global batch, against one process on the whole batch --
    summed gradients, scores and the generator output of the rank's own windows.
Tolerances: gradients 1e-3 (north_star); what differs at all is the order of fp32 sums over the batch.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu

CFG = dict(V=8, NT=4, To=24, Tf=24, H=128, L=1, B=96, N=3000)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_pass(dev, rank, world, nbatch=3, gb=None, critic=1):
    """Seeded trainer; `nbatch` global batches of gb windows through train_step (this rank's slices);
    returns (score, params, D grads, G grads, logged losses)."""
    import bench
    from timeband_amd.dataset import WindowSet
    np.random.seed(3)
    cfg = dict(CFG)
    gb = cfg["B"] if gb is None else gb
    ds, trainset, trainer = bench.build(cfg, dev)
    trainer.critic = critic
    np.random.seed(3)
    loader = WindowSet(ds, trainset.w0, nbatch * gb).loader(gb, rank, world)
    score = trainer.train_step(0, loader, training=True)
    torch.cuda.synchronize()
    params = [p.detach().clone() for n in (trainer.models.netD, trainer.models.netG) for p in n.parameters()]
    return score, params, trainer.optimD.flat.grad.clone(), trainer.optimG.flat.grad.clone(), trainer.losses.loss(nbatch - 1)


def _run_cfg2(dev, iters=2):
    """BASELINE.json configs[1] at full size (32 variables, window 128, hidden 256, batch 512): the discriminator's
    real + fake pass is then 1024 rows -- the EXCLUSIVE 4-member launches -- beside the bucketed all-reduces."""
    import bench
    from timeband_amd import ops
    np.random.seed(3)
    cfg = bench.CONFIGS["cfg2"]
    ds, trainset, trainer = bench.build(cfg, dev)
    np.random.seed(3)
    for i in range(iters):
        b = trainset.gather(i * cfg["B"], cfg["B"])
        trainer.iteration(b["encoded"], b["decoded"], training=True)
    torch.cuda.synchronize()
    err = ops.Runtime.get(dev).err.tolist()
    assert err[0] == 0, err                      # no inter-workgroup wait timed out
    ops.check_errors(dev)
    return [p.detach().clone() for n in (trainer.models.netD, trainer.models.netG) for p in n.parameters()]


def test_world1_rccl_path_is_bit_identical():
    import torch.distributed as dist
    from timeband_amd import dp
    dev = torch.device("cuda", 0)
    base = _run_pass(dev, 0, 1)
    dp.init("nccl", init_method=f"tcp://127.0.0.1:{_free_port()}", rank=0, world_size=1, device_id=dev)
    assert os.environ["NCCL_MAX_NCHANNELS"]      # the channel cap is in place before RCCL comes up
    try:
        dp.force(True)
        assert dp.active()
        got = _run_pass(dev, 0, 1)
        # the loader only attaches `part` for world > 1: drive the sharded post-processing explicitly as well
        import bench
        from timeband_amd.postprocess import BandAdjust
        ds, trainset, trainer = bench.build(dict(CFG), dev)
        a = BandAdjust(ds, "trainer", 1.0, 0.2, 0.3)
        b = BandAdjust(ds, "trainer", 1.0, 0.2, 0.3)
        g = torch.Generator(device="cpu").manual_seed(1)
        for _ in range(3):
            fake = (torch.rand(CFG["B"], CFG["Tf"], CFG["V"], generator=g) * 2 - 1).to(dev)
            ra = a._step_all(fake)
            rb = b.step(fake, (0, CFG["B"], False))
            for x, y in zip(ra, rb):
                assert torch.equal(x, y)
        # cfg2 at full size: RCCL's all-reduce kernels on the communication stream while exclusive and clustered
        # recurrent launches run (the co-residency argument of DESIGN.md section 8) -- no timeout, same bits
        dp.force(False)
        plain2 = _run_cfg2(dev)
        dp.force(True)
        forced2 = _run_cfg2(dev)
        for x, y in zip(plain2, forced2):
            assert torch.equal(x, y)
    finally:
        dp.force(False)
        dist.destroy_process_group()
    assert base[0] == got[0]
    for x, y in zip(base[1], got[1]):
        assert torch.equal(x, y)
    assert torch.equal(base[2], got[2]) and torch.equal(base[3], got[3])
    assert base[4] == got[4]


def _critic_grad(dev):
    """D's (summed) gradient of one critic step on a seeded draw; lr = 0, so the parameters stay put."""
    import bench
    from timeband_amd import dp
    from timeband_amd.optim import FusedRMSprop
    ds, trainset, trainer = bench.build(dict(CFG), dev)
    np.random.seed(5)
    cx, cy = ds.get_random()
    n = cx.shape[0]
    trainer._critic_opt = FusedRMSprop(trainer.models.netD, 0.0)
    cx, cy, shared = trainer._shard_critic(cx, cy)
    assert shared
    with dp.batch_counts(cx.shape[0], n):
        trainer.critic_step(cx, cy)
    torch.cuda.synchronize()
    return trainer._critic_opt.flat.grad.clone()


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    gC = _critic_grad(dev)
    # critic = 0 below: a fresh RMSprop's first step is 10 lr sign(g), so wherever a critic gradient is ~0 two
    # summation orders step D in opposite directions and everything after differs legitimately
    score, params, gD, gG, losses = _run_pass(dev, rank, world, nbatch=2, gb=2 * CFG["B"], critic=0)
    np.savez(f"{out}.{rank}.npz", score=score, gD=gD.cpu().numpy(), gG=gG.cpu().numpy(), gC=gC.cpu().numpy(),
             losses=np.array([float(v) for v in losses.values()]),
             **{f"p{i}": p.cpu().numpy() for i, p in enumerate(params)})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_on_one_gpu_match_one_process(tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "dp")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    dev = torch.device("cuda", 0)
    gC = _critic_grad(dev)
    score, params, gD, gG, losses = _run_pass(dev, 0, 1, nbatch=2, gb=2 * CFG["B"], critic=0)
    r0, r1 = (np.load(f"{out}.{k}.npz") for k in range(2))
    # both ranks hold the same summed gradients and take the same step
    np.testing.assert_array_equal(r0["gD"], r1["gD"])
    np.testing.assert_array_equal(r0["gG"], r1["gG"])
    for i in range(len(params)):
        np.testing.assert_array_equal(r0[f"p{i}"], r1[f"p{i}"])
    # ... and they are the single-process gradients of the whole batch
    np.testing.assert_array_equal(r0["gC"], r1["gC"])
    for got, want in ((r0["gC"], gC.cpu().numpy()), (r0["gD"], gD.cpu().numpy()), (r0["gG"], gG.cpu().numpy())):
        assert np.abs(got - want).max() <= 1e-3 * np.abs(want).max()
    assert abs(float(r0["score"]) - score) <= 1e-4 * abs(score)
    np.testing.assert_allclose(r0["losses"], np.array([float(v) for v in losses.values()]), atol=2e-3)
    lr = 2e-4
    nbad = ntot = 0
    for i, p in enumerate(params):          # two Adam steps each: at most two steps apart, almost all equal
        d = np.abs(r0[f"p{i}"] - p.cpu().numpy())
        assert d.max() <= 4.2 * lr
        nbad += int((d > 0.1 * lr).sum())
        ntot += d.size
    assert nbad / ntot < 5e-3, (nbad, ntot)

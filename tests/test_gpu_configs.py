"""GPU: the other BASELINE.json configs as parity cases, and size-independent properties at full size.

  * cfg4 (long context, hidden 512, 2-layer) at a size the numpy oracle finishes in seconds: forward,
    one whole Gen-A iteration's gradients and updated parameters against oracle/timeband_oracle.py;
  * cfg5 (wide multivariate cleaning: 512 variables) at a short series: clean() outputs and bands
    against the oracle's clean();
  * more recurrent-kernel shapes (every exchange ring path: clusters of 4, 8 and 32 workgroups, several
    batch tiles per cluster, ragged batch, sequences long enough to cycle the ring many times);
  * at cfg2's FULL size (no oracle there): batch-split invariance of G and D (bit-exact: a row's result
    must not depend on which tile, cluster or launch it lands in), gradient additivity over the batch,
    and the overlapped (side-stream) iteration == the single-stream iteration, bit for bit.
Tolerances: forward 1e-4 relative, gradients 1e-3 (north_star); what is observed is ~1e-6.
"""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import timeband_oracle as O  # noqa: E402  (test infrastructure: the checker)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda", 0)


def T(a, dev, dtype=torch.float32):
    return torch.tensor(np.ascontiguousarray(a), dtype=dtype, device=dev)


def rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def params_np(module):
    return {k: v.detach().cpu().numpy().astype(np.float64) for k, v in module.state_dict().items()}


def make_nets(V, E, To, Tf, H, L, dev, seed=31):
    from timeband_amd.utils.initiate import seeding
    from timeband_amd.utils.lstm_layer import LSTMDiscriminator, LSTMGenerator
    seeding(seed)
    netD = LSTMDiscriminator(V, H, L, dev).to(dev)      # D before G: the reference's RNG order
    netG = LSTMGenerator(E, V, To, Tf, H, L, dev).to(dev)
    return netD, netG


def grads_close(module, want, tol, what):
    gmax = max(np.abs(v).max() for v in want.values())
    for k, p in module.named_parameters():
        err = np.abs(p.grad.detach().cpu().numpy().astype(np.float64) - want[k]).max()
        # per tensor; the floor (1 % of the tolerance times the largest gradient of the network) only keeps tensors
        # whose own gradient is ~0 from being compared against their rounding noise
        bound = tol * np.abs(want[k]).max() + 0.01 * tol * gmax + 2e-7    # 2e-7: fp32 rounding of the reference's own cancelling O(1) sums (WGAN bias gradients)
        assert err <= bound, f"{what}:{k} err {err:.3e} > {bound:.3e}"


# ---------------------------------------------------------------------------------------------
MORE_LSTM_CASES = [
    # (B, T, in, H)
    (20, 40, 36, 512),      # clusters of 32, three puller waves, ring cycled 13 times
    (40, 70, 32, 256),      # clusters of 8, ragged third batch tile
    (100, 33, 64, 128),     # clusters of 4
    (600, 9, 16, 128),      # 38 tiles on 64 clusters: one tile per cluster, ragged last tile
    (1100, 20, 16, 128),    # forward with 2 batch tiles per cluster, ring re-armed for 18 steps
    (2100, 12, 16, 128),    # forward with 4 batch tiles per cluster (BPTT: 4 tiles too)
    (200, 20, 36, 512),     # hidden 512 forward with 2 tiles per cluster (k split over 4 waves, 32-member clusters)
    (1100, 5, 8, 256),      # 2 tiles per cluster AND a second launch
    (9, 130, 11, 16), (50, 64, 7, 32), (130, 48, 36, 64),   # single-workgroup layers, long sequences
    # input projection fused into the recurrent kernel (in <= 64, a multiple of 4): one and two x k-steps, in < 32,
    # several tiles per cluster, the padded 36 -> 64 case, hidden 16 / 32
    (40, 9, 4, 16), (300, 21, 12, 32), (530, 17, 36, 64), (1100, 12, 64, 128), (96, 25, 28, 256), (70, 14, 60, 128),
    # fused hidden-64 layers on partly filled tiles, elementwise part compacted over the idle lanes: 4 valid rows per tile
    # (one and two x k-steps; (530, 17, 36, 64) above as well), 8 valid rows, and whole tiles
    (200, 9, 20, 64), (1500, 7, 36, 64), (1500, 6, 32, 64), (3000, 5, 20, 64),
]


@pytest.mark.parametrize("B,Tn,In,H", MORE_LSTM_CASES)
def test_lstm_more_shapes(dev, B, Tn, In, H):
    from timeband_amd import ops
    rng = np.random.default_rng(B * 7 + Tn * 3 + H)
    k = 1.0 / np.sqrt(H)
    w_ih, w_hh = rng.uniform(-k, k, (4 * H, In)), rng.uniform(-k, k, (4 * H, H))
    b_ih, b_hh = rng.uniform(-k, k, 4 * H), rng.uniform(-k, k, 4 * H)
    x = rng.uniform(-1, 1, (B, Tn, In))
    dy = rng.uniform(-1, 1, (B, Tn, H))
    y_ref, cache = O.lstm_layer_fwd(x, w_ih, w_hh, b_ih, b_hh)
    dx_ref, dwi_ref, dwh_ref, dbi_ref, _ = O.lstm_layer_bwd(dy, cache)
    ts = [T(a, dev).requires_grad_() for a in (x, w_ih, w_hh, b_ih, b_hh)]
    y = ops.lstm_layer(*ts)
    y.backward(T(dy, dev))
    torch.cuda.synchronize()
    ops.check_errors()
    assert rel(y.detach().cpu().numpy(), y_ref) < 1e-4, "forward"
    for got, want, name in ((ts[0].grad, dx_ref, "dx"), (ts[1].grad, dwi_ref, "dw_ih"), (ts[2].grad, dwh_ref, "dw_hh"),
                            (ts[3].grad, dbi_ref, "db_ih"), (ts[4].grad, dbi_ref, "db_hh")):
        assert rel(got.cpu().numpy(), want) < 1e-3, name


def test_two_clustered_launches_side_by_side(dev):
    """Two hidden-512 forward launches with two batch tiles per cluster (the variant with the largest LDS
    footprint, all 256 CUs) on two streams at once: both must be able to become fully resident (co-residency
    rule of lstm.hip's make_plan) and give the bits of a launch that runs alone."""
    from timeband_amd import ops
    rng = np.random.default_rng(12)
    B, Tn, In, H = 200, 24, 36, 512
    k = 1.0 / np.sqrt(H)
    ps = [T(rng.uniform(-k, k, s), dev) for s in ((4 * H, In), (4 * H, H), (4 * H,), (4 * H,))]
    xa, xb = T(rng.uniform(-1, 1, (B, Tn, In)), dev), T(rng.uniform(-1, 1, (B, Tn, In)), dev)
    with torch.no_grad():
        ya, yb = ops.lstm_layer(xa, *ps), ops.lstm_layer(xb, *ps)
        torch.cuda.synchronize()
        s1, s2 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        for _ in range(3):
            with torch.cuda.stream(s1):
                ya2 = ops.lstm_layer(xa, *ps)
            with torch.cuda.stream(s2):
                yb2 = ops.lstm_layer(xb, *ps)
        torch.cuda.synchronize()
    ops.check_errors()
    assert torch.equal(ya, ya2) and torch.equal(yb, yb2)


def test_third_clustered_stream_is_held_back(dev):
    """Three streams with full-chip clustered launches queued at once (what TB_OVERLAP=7 can do): only two such
    launches fit a CU side by side, so ops.Runtime orders the third behind the oldest -- no wait timeout, and
    the bits of launches that run alone."""
    from timeband_amd import ops
    rng = np.random.default_rng(13)
    B, Tn, In, H = 512, 32, 36, 256
    k = 1.0 / np.sqrt(H)
    ps = [T(rng.uniform(-k, k, s), dev) for s in ((4 * H, In), (4 * H, H), (4 * H,), (4 * H,))]
    xs = [T(rng.uniform(-1, 1, (B, Tn, In)), dev) for _ in range(3)]
    with torch.no_grad():
        alone = [ops.lstm_layer(x, *ps) for x in xs]
        torch.cuda.synchronize()
        streams = [torch.cuda.Stream(dev) for _ in range(3)]
        for _ in range(4):
            got = []
            for s, x in zip(streams, xs):
                with torch.cuda.stream(s):
                    got.append(ops.lstm_layer(x, *ps))
        torch.cuda.synchronize()
    ops.check_errors()
    for a, b in zip(alone, got):
        assert torch.equal(a, b)


def test_exclusive_launch_is_ordered_against_other_streams(dev):
    """Hidden 256 with 1024 rows runs on 4-member clusters whose workgroups fill a CU's registers (tb_lstm_exclusive):
    queued beside clustered launches on two other streams, forward and backward, it must be ordered by ops.Runtime --
    no wait timeout, and the bits of passes that run alone."""
    from timeband_amd import _lib, ops
    lib = _lib.load()
    assert lib.tb_lstm_exclusive(1024, 32, 256, 0) == 1 and lib.tb_lstm_exclusive(1024, 32, 256, 1) == 1
    assert lib.tb_lstm_exclusive(512, 32, 256, 0) == 0 and lib.tb_lstm_exclusive(1024, 64, 128, 1) == 0
    rng = np.random.default_rng(14)
    Tn, In, H = 24, 32, 256
    k = 1.0 / np.sqrt(H)
    ps = [T(rng.uniform(-k, k, s), dev).requires_grad_() for s in ((4 * H, In), (4 * H, H), (4 * H,), (4 * H,))]
    xs = [T(rng.uniform(-1, 1, (B, Tn, In)), dev).requires_grad_() for B in (1024, 512, 512)]
    dys = [T(rng.uniform(-1, 1, (x.shape[0], Tn, H)), dev) for x in xs]

    def run(x, dy):
        y = ops.lstm_layer(x, *ps)
        (gx,) = torch.autograd.grad(y, x, dy)
        return y.detach(), gx

    alone = [run(x, dy) for x, dy in zip(xs, dys)]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(dev) for _ in range(3)]
    for _ in range(3):
        got = []
        for s, x, dy in zip(streams, xs, dys):
            with torch.cuda.stream(s):
                got.append(run(x, dy))
    torch.cuda.synchronize()
    ops.check_errors()
    for (ya, ga), (yb, gb) in zip(alone, got):
        assert torch.equal(ya, yb) and torch.equal(ga, gb)


# ---------------------------------------------------------------------------------------------
def test_cfg4_long_context_two_layer_iteration(dev):
    """BASELINE.json configs[3] (window 1024, hidden 512, 2-layer) reduced to window 48 / batch 6: the D and
    G phases of a Gen-A iteration with the fused optimizers against the oracle."""
    from timeband_amd.loss import TIMEBANDLoss
    from timeband_amd.optim import FusedAdam
    V, NT, To, Tf, H, L, B = 5, 4, 48, 48, 512, 2, 6
    E = V + NT
    lr = 2e-4
    netD, netG = make_nets(V, E, To, Tf, H, L, dev)
    rng = np.random.default_rng(4)
    x, y = rng.uniform(-1, 1, (B, To, E)), rng.uniform(-1, 1, (B, Tf, V))
    st = O.TrainState(params_np(netD), params_np(netG), n_layers=L, lr=lr)
    st.new_epoch()
    rec = {}
    # no critic draw here: RMSprop's first step is 10*lr*sign(g), so wherever a critic gradient is ~0 any
    # two fp32 implementations step D in opposite directions and everything after differs legitimately
    # (the critic step has its own fixtures in test_gpu_kernels.py::test_iteration_golden)
    fake_ref = O.train_iteration(st, x, y, None, None, training=True, rec=rec)

    losses = TIMEBANDLoss(dict(l1_weight=50, l2_weight=50, gp_weight=50, device=dev))
    adamD, adamG = FusedAdam(netD, lr), FusedAdam(netG, lr)
    xt, yt = T(x, dev), T(y, dev)
    adamD.zero_grad()
    adamG.zero_grad()
    fake = netG(xt)
    Dy, DGx = netD.pair(yt, fake)
    errD = losses.dis_loss(yt, fake, Dy, DGx)
    errD.backward()
    grads_close(netD, rec["dphase.grad.netD"], 1e-3, "D-phase grad D")
    grads_close(netG, rec["dphase.grad.netG"], 1e-3, "D-phase grad G")
    adamD.step()
    for p in netD.parameters():
        p.requires_grad_(False)
    fake = netG(xt)
    errG = losses.gen_loss(yt, fake, netD(fake))
    errG.backward()
    grads_close(netG, rec["gphase.grad.netG"], 1e-3, "G-phase grad G (accumulated)")
    adamG.step()
    torch.cuda.synchronize()
    from timeband_amd import ops
    ops.check_errors()
    assert rel(fake.detach().cpu().numpy(), fake_ref) < 1e-4
    # parameters after the Adam steps: a step is lr-sized, compare in units of lr
    for net, want in ((netD, st.pD), (netG, st.pG)):
        for k_, v in net.state_dict().items():
            d = np.abs(v.cpu().numpy().astype(np.float64) - want[k_])
            # Adam's first step is lr * g / (|g| + 1e-8), a sign where |g| >> 1e-8 and noise below that:
            # bounded by one step each way; the gradients themselves are compared above
            assert d.max() <= 2.1 * lr, (k_, d.max())


def test_cfg4_full_size_properties(dev):
    """BASELINE.json configs[3] at FULL size (window 1024, hidden 512, 2-layer, batch 64; no oracle at this size):
    the persistent kernels' rings cycle 341 times per launch and the 32-bit descriptors span T*4H*rows = 268 MB.
    Finite outputs in range, no wait timeout, bit-exact batch-split invariance of G and D forward, finite
    gradients of a whole D-phase + G-phase backward, and gradients additive over the batch."""
    from timeband_amd import ops
    V, NT, To, Tf, H, L, B = 32, 4, 1024, 1024, 512, 2, 64
    E = V + NT
    netD, netG = make_nets(V, E, To, Tf, H, L, dev)
    g = torch.Generator(device="cpu").manual_seed(11)
    x = (torch.rand(B, To, E, generator=g) * 2 - 1).to(dev)
    y = (torch.rand(B, Tf, V, generator=g) * 2 - 1).to(dev)
    with torch.no_grad():
        full = netG(x)
        part = torch.cat([netG(x[:20]), netG(x[20:])], 0)
        dfull = netD(y)
        dpart = torch.cat([netD(y[:33]), netD(y[33:])], 0)
    torch.cuda.synchronize()
    ops.check_errors()
    assert torch.equal(full, part) and torch.equal(dfull, dpart)
    assert torch.isfinite(full).all() and full.abs().max() < 1.0 and (dfull > 0).all() and (dfull < 1).all()

    def grads(xs, ys):
        for n in (netD, netG):
            n.zero_grad(set_to_none=True)
        fake = netG(xs)
        Dy, DGx = netD.pair(ys, fake)
        (((Dy - 0.9) ** 2).sum() + ((DGx - 0.1) ** 2).sum() + ((fake - ys) ** 2).sum()).backward()
        torch.cuda.synchronize()
        ops.check_errors()
        return [p.grad.detach().double().clone() for n in (netD, netG) for p in n.parameters()]

    g_all = grads(x, y)
    assert all(bool(torch.isfinite(t).all()) for t in g_all)
    g_a, g_b = grads(x[:32], y[:32]), grads(x[32:], y[32:])
    gmax = max(float(t.abs().max()) for t in g_all)
    for ga, a, b in zip(g_all, g_a, g_b):
        assert float((ga - (a + b)).abs().max()) <= 1e-3 * (float(ga.abs().max()) + 0.01 * gmax)
    for n in (netD, netG):
        n.zero_grad(set_to_none=True)


# ---------------------------------------------------------------------------------------------
def test_cfg5_wide_cleaning_vs_oracle(dev, monkeypatch):
    """BASELINE.json configs[4] (512 variables, detect + impute through the cleaner path) on a 260-row
    series: device-resident series -> window gather -> G forward -> band -> adjust, against the oracle."""
    from timeband_amd import synthetic
    from timeband_amd.cleaner import TIMEBANDCleaner
    from timeband_amd.dataset import TIMEBANDDataset
    from timeband_amd.model import TIMEBANDModel
    from timeband_amd.utils.logger import NullLogger
    V, To, Tf, H, L, B, N = 512, 12, 12, 64, 1, 50, 260
    values = synthetic.make_series(N, V, seed=5)
    tfeat = synthetic.time_features(N + Tf)
    core = dict(logger=NullLogger(), device=dev, batch_size=B, observed_len=To, forecast_len=Tf, directory="",
                data_name="synthetic")
    ds = TIMEBANDDataset.from_arrays(values, tfeat, {**core, "stride": 1, "split_rate": 0.9, "min_valid_scale": 2,
                                                     "sliding_step": 0})
    netD, netG = make_nets(V, V + tfeat.shape[1], To, Tf, H, L, dev)
    holder = TIMEBANDModel.__new__(TIMEBANDModel)
    holder.netG, holder.netD = netG, None
    cfg = {**core, "band_width": 1.0, "missing_gamma": 0.2, "anomaly_gamma": 0.3}
    full = ds.prepare_dataset(split=False)
    out_df, bands_df = TIMEBANDCleaner(cfg, ds, holder).clean(full.loader(B))
    series = O.SeriesData(values, tfeat, To, Tf, 0.9)
    outs, lower, upper = O.clean(params_np(netG), series, B, n_layers=L, variant="cleaner", band_width=1.0, a=0.2, b=0.3)
    got = out_df.to_numpy()
    assert got.shape == outs.shape == (N + Tf, V)
    scale = np.abs(outs).max()
    assert np.abs(got - outs).max() < 2e-4 * scale
    bands = bands_df.to_numpy()
    want_b = np.concatenate([lower, upper], axis=1)
    assert bands.shape == want_b.shape      # same column order as the reference: lower block then upper block
    assert np.abs(bands - want_b).max() < 5e-4 * scale
    # the run above formed the generator's output inside the band kernel (tb_dimband_update) for every batch of at least
    # forecast_len - 1 windows; with the output layer and the band as two kernels the results must be the same bits
    monkeypatch.setenv("TB_FUSE_DIMBAND", "0")
    out2, bands2 = TIMEBANDCleaner(cfg, ds, holder).clean(full.loader(B))
    np.testing.assert_array_equal(out2.to_numpy(), got)
    np.testing.assert_array_equal(bands2.to_numpy(), bands)
    # ... and so must the sweep without its pipeline (generator forward of batch i + 1 on a side stream under the band /
    # adjust of batch i: postprocess.WindowSweep) -- everything on one stream, batch after batch
    monkeypatch.delenv("TB_FUSE_DIMBAND")
    monkeypatch.setenv("TB_CLEAN_PIPELINE", "0")
    out3, bands3 = TIMEBANDCleaner(cfg, ds, holder).clean(full.loader(B))
    np.testing.assert_array_equal(out3.to_numpy(), got)
    np.testing.assert_array_equal(bands3.to_numpy(), bands)


# ---------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def cfg2_nets(dev):
    V, E, To, Tf, H, L = 32, 36, 128, 128, 256, 1
    netD, netG = make_nets(V, E, To, Tf, H, L, dev)
    g = torch.Generator(device="cpu").manual_seed(7)
    x = (torch.rand(512, To, E, generator=g) * 2 - 1).to(dev)
    y = (torch.rand(512, Tf, V, generator=g) * 2 - 1).to(dev)
    return netD, netG, x, y


def test_full_size_batch_split_invariance(dev, cfg2_nets):
    """cfg2 size: G(x)[a:b] == G(x[a:b]) and D likewise, bit for bit -- a window's result does not depend on
    the batch tile, cluster, launch or exchange path it is computed in."""
    netD, netG, x, y = cfg2_nets
    from timeband_amd import ops
    with torch.no_grad():
        full = netG(x)
        part = torch.cat([netG(x[:100]), netG(x[100:356]), netG(x[356:])], 0)
        dfull = netD(y)
        dpart = torch.cat([netD(y[:37]), netD(y[37:])], 0)
        Dy, DGx = netD.pair(y, full)
    torch.cuda.synchronize()
    ops.check_errors()
    assert torch.equal(full, part)
    assert torch.equal(dfull, dpart)
    assert torch.equal(Dy, dfull)
    assert torch.isfinite(full).all() and full.abs().max() < 1.0 and (DGx > 0).all() and (DGx < 1).all()


def test_full_size_gradients_add_over_the_batch(dev, cfg2_nets):
    """cfg2 size: parameter gradients of a batch-summed loss are the sum of the two half-batch gradients."""
    netD, netG, x, y = cfg2_nets

    def grads(xs, ys):
        for n in (netD, netG):
            n.zero_grad(set_to_none=True)
        fake = netG(xs)
        (netD(fake).sum() + ((fake - ys) ** 2).sum()).backward()
        return [p.grad.detach().double().clone() for n in (netD, netG) for p in n.parameters()]

    g_all = grads(x, y)
    g_a, g_b = grads(x[:256], y[:256]), grads(x[256:], y[256:])
    gmax = max(float(g.abs().max()) for g in g_all)
    for ga, a, b in zip(g_all, g_a, g_b):
        assert float((ga - (a + b)).abs().max()) <= 1e-3 * (float(ga.abs().max()) + 0.1 * gmax)
    for n in (netD, netG):
        n.zero_grad(set_to_none=True)


@pytest.mark.parametrize("hidden", [128, 256])
def test_overlapped_iteration_equals_single_stream(dev, hidden):
    """TIMEBANDTrainer.iteration with the generator work on side streams (default, and each part of it
    alone: TB_OVERLAP bit mask), with and without the two generator forward passes computed as one 2B-row pass
    (bit 16, LSTMGenerator.twin), and on one stream (TB_OVERLAP=0) leave bit-identical parameters after
    two iterations.  Bit 32 (one generator backward over the 2B rows, the default) sums the parameter gradients
    of the two passes in ONE reduction over 2B rows instead of pass by pass: every mask with it gives the same
    bits as every other, and against the masks without it the discriminator and fake_y stay bit-identical after
    one iteration while G's gradients agree to fp32 association (1e-5 of each tensor's largest entry)."""
    import bench
    cfg = dict(V=8, NT=4, To=24, Tf=24, H=hidden, L=1, B=96, N=3000)

    def run(mask, iters):
        np.random.seed(3)
        ds, trainset, trainer = bench.build(cfg, dev)
        trainer.overlap_mask = mask
        for i in range(iters):
            b = trainset.gather(i * cfg["B"], cfg["B"])
            fake = trainer.iteration(b["encoded"], b["decoded"], training=True)
        torch.cuda.synchronize()
        params = [p.detach().clone() for n in (trainer.models.netD, trainer.models.netG) for p in n.parameters()]
        return params, fake.clone(), [p.grad.detach().clone() for p in trainer.models.netG.parameters()], \
            [p.detach().clone() for p in trainer.models.netD.parameters()]

    plain = (29, 0, 16, 1, 2, 4, 5, 7, 13, 15, 17, 20, 21, 31) if hidden == 128 else (29, 0, 5, 16, 21)
    joint = (49, 48, 61, 63) if hidden == 128 else (49, 48)
    for group in (plain, joint):
        results = [run(mask, 2)[0] for mask in group]
        for other in results[1:]:
            for a, b in zip(results[0], other):
                assert torch.equal(a, b)
    _, fa, ga, da = run(plain[0], 1)
    _, fb, gb, db = run(joint[0], 1)
    assert torch.equal(fa, fb)
    for a, b in zip(da, db):
        assert torch.equal(a, b)
    for a, b in zip(ga, gb):
        assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max()) + 1e-12


@pytest.mark.parametrize("mask", [49, 29])
def test_cpp_extension_path_equals_python_path(dev, mask, monkeypatch):
    """The C++ extension's autograd Functions (csrc/ext.cpp, the default) and the Python ones of ops.py (TB_EXT=0) drive the
    same kernels with the same arguments: bit-identical parameters after two iterations."""
    import bench
    from timeband_amd import ops
    cfg = dict(V=8, NT=4, To=24, Tf=24, H=128, L=1, B=96, N=3000)

    def run():
        np.random.seed(3)
        ds, trainset, trainer = bench.build(cfg, dev)
        trainer.overlap_mask = mask
        for i in range(2):
            b = trainset.gather(i * cfg["B"], cfg["B"])
            trainer.iteration(b["encoded"], b["decoded"], training=True)
        torch.cuda.synchronize()
        ops.check_errors(dev)
        return [p.detach().clone() for n in (trainer.models.netD, trainer.models.netG) for p in n.parameters()]

    assert ops._load_ext() is not None, "the C++ extension is the product's default path"
    with_ext = run()
    monkeypatch.setenv("TB_EXT", "0")
    monkeypatch.setattr(ops, "_EXT", None)
    assert ops._ext() is None
    without = run()
    for a, b in zip(with_ext, without):
        assert torch.equal(a, b)


@pytest.mark.parametrize("E,V,To,Tf,H,L,stride", [(132, 128, 24, 24, 64, 1, 1), (516, 512, 48, 48, 64, 1, 1), (100, 96, 16, 8, 512, 1, 2),
                                                   (72, 68, 12, 12, 64, 2, 1), (11, 7, 48, 48, 64, 1, 1), (36, 32, 20, 20, 256, 1, 3)])
def test_infer_windows_equals_gathered_windows(dev, E, V, To, Tf, H, L, stride):
    """LSTMGenerator.infer_windows (the cleaner's path: overlapping windows read in place from the series, first projection
    once per series row, tb_lstm_fwd_windows) against the generator on the gathered windows.  Where the gathered path
    projects by GEMM too (encoder width > 64) the two do the same arithmetic per row: bit-identical.  Narrow encoders
    are fused into the recurrent kernel on the gathered path (another summation order): 1e-5."""
    from timeband_amd import ops
    from timeband_amd.utils.lstm_layer import LSTMGenerator
    torch.manual_seed(5)
    netG = LSTMGenerator(E, V, To, Tf, H, L).to(dev)
    rows, nwin = 400 * stride, 150
    series = (torch.rand(rows, E) * 2 - 1).to(dev)
    tcols = torch.arange(V, dtype=torch.int32, device=dev)
    first = To + 7
    assert first + (nwin - 1) * stride + Tf <= rows
    enc, _ = ops.window_gather(series, tcols, first, nwin, To, Tf, stride=stride)
    with torch.no_grad():
        want = netG(enc)
    got = netG.infer_windows(series, first, nwin, stride)
    torch.cuda.synchronize()
    ops.check_errors(dev)
    if E > 64:
        assert torch.equal(got, want)
    else:
        assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max())
    with pytest.raises(RuntimeError):
        netG.infer_windows(series, To - 1, 4, stride)          # the first window would start before the series


def test_shared_generator_pass_matches_two_passes(dev):
    """TB_SHARE_GPASS: one generator forward and one backward with the summed upstream gradient against
    the reference's schedule (two identical forwards, two backwards): same fake_y bit for bit, parameters
    equal up to the rounding of that sum (Adam turns it into at most a fraction of one step)."""
    import bench
    cfg = dict(V=8, NT=4, To=24, Tf=24, H=128, L=1, B=96, N=3000)
    res = []
    for share in (False, True):
        np.random.seed(3)
        ds, trainset, trainer = bench.build(cfg, dev)
        trainer.share_gpass = share
        b = trainset.gather(0, cfg["B"])
        fake = trainer.iteration(b["encoded"], b["decoded"], training=True)
        torch.cuda.synchronize()
        res.append((fake.clone(), [p.detach().clone() for p in trainer.models.netG.parameters()],
                    [p.detach().clone() for p in trainer.models.netD.parameters()], trainer.lr))
    (fa, ga, da, lr), (fb, gb, db, _) = res
    assert torch.equal(fa, fb)
    for a, b in zip(da, db):
        assert torch.equal(a, b)
    nbad = ntot = 0
    for a, b in zip(ga, gb):
        d = (a - b).abs()
        assert float(d.max()) <= 2.1 * lr
        nbad += int((d > 0.05 * lr).sum())
        ntot += d.numel()
    assert nbad / ntot < 2e-3, (nbad, ntot)


def test_pair_pass_skips_only_the_real_half_input_gradient(dev):
    """LSTMDiscriminator.pair computes the input gradient of the fake half alone (tb_lstm_bwd_rows): it must equal,
    bit for bit, the fake half of the full input gradient, and the parameter gradients must not change."""
    V, E, To, Tf, H, L = 8, 12, 24, 24, 128, 2
    netD, _ = make_nets(V, E, To, Tf, H, L, dev)
    g = torch.Generator(device="cpu").manual_seed(3)
    real = (torch.rand(40, Tf, V, generator=g) * 2 - 1).to(dev)
    fake = (torch.rand(40, Tf, V, generator=g) * 2 - 1).to(dev)
    w = (torch.rand(80, Tf, 1, generator=g)).to(dev)

    def run(pair):
        netD.zero_grad(set_to_none=True)
        f = fake.clone().requires_grad_()
        if pair:
            Dy, DGx = netD.pair(real, f)
            out = torch.cat([Dy, DGx], 0)
        else:
            out = netD(torch.cat([real, f], 0))
        (out * w).sum().backward()
        return f.grad.clone(), [p.grad.clone() for p in netD.parameters()]

    ga, pa = run(True)
    gb, pb = run(False)
    assert torch.equal(ga, gb)
    for a, b in zip(pa, pb):
        assert torch.equal(a, b)


@pytest.mark.parametrize("critic", [False, True])
def test_pair_loss_gradient_in_one_launch_equals_two(dev, critic):
    """dis_loss on the two halves of the discriminator's real + fake output sends the gradient to their common
    buffer in one launch (ops._pair_base / tb_gan_loss_grad_pair): same loss, same gradients, bit for bit, as the
    two-launch path that copies of the halves take."""
    from timeband_amd import ops
    V, E, To, Tf, H, L = 8, 12, 24, 24, 128, 1
    netD, _ = make_nets(V, E, To, Tf, H, L, dev)
    g = torch.Generator(device="cpu").manual_seed(5)
    real = (torch.rand(40, Tf, V, generator=g) * 2 - 1).to(dev)
    fake = (torch.rand(40, Tf, V, generator=g) * 2 - 1).to(dev)

    def run(paired):
        netD.zero_grad(set_to_none=True)
        f = fake.clone().requires_grad_()
        Dy, DGx = netD.pair(real, f)
        assert ops._pair_base(Dy, DGx) is not None
        if not paired:
            Dy, DGx = Dy.clone(), DGx.clone()
            assert ops._pair_base(Dy, DGx) is None
        total, out = ops.dis_loss(Dy, DGx, real if critic else None, f if critic else None, critic, 10.0 if critic else 0.0, None)
        total.backward()
        return total.detach().clone(), out.clone(), f.grad.clone(), [p.grad.clone() for p in netD.parameters()]

    a, b = run(True), run(False)
    # (the logging vector has four entries for the discriminator's losses; the rest of its eight floats is never written)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1][:4], b[1][:4]) and torch.equal(a[2], b[2])
    for x, y in zip(a[3], b[3]):
        assert torch.equal(x, y)


def test_post_processing_on_its_own_stream_changes_nothing(dev):
    """TIMEBANDTrainer.post_process queues band / adjust / scores of a batch on a side stream, where they run under the next
    iteration (one rank).  Same kernels on the same inputs in the same order: the pass score, the running scores and the
    band history must equal the in-line run bit for bit."""
    import bench
    from timeband_amd.dataset import WindowSet
    cfg = dict(V=8, NT=4, To=24, Tf=24, H=128, L=1, B=96, N=3000)
    res = []
    for overlap in (True, False):
        np.random.seed(3)
        ds, trainset, trainer = bench.build(cfg, dev)
        trainer.post_overlap = overlap
        np.random.seed(3)
        loader = WindowSet(ds, trainset.w0, 4 * cfg["B"]).loader(cfg["B"])
        score = trainer.train_step(0, loader, training=True)
        torch.cuda.synchronize()
        res.append((score, trainer.metric.score(3), trainer.post.hist[trainer.post.cur].clone(), trainer.post.last_row.clone(),
                    [p.detach().clone() for p in trainer.models.netG.parameters()]))
        assert (trainer._post_stream is not None) == overlap
    (sa, ma, ha, la, pa), (sb, mb, hb, lb, pb) = res
    assert sa == sb and ma == mb
    assert torch.equal(ha, hb) and torch.equal(la, lb)
    for x, y in zip(pa, pb):
        assert torch.equal(x, y)


@pytest.mark.parametrize("depth", [0, 1, 2, 4])
def test_window_sweep_depths_equal(dev, depth, monkeypatch):
    """postprocess.WindowSweep with 0 (no pipeline), 1, 2 and 4 side streams: identical cleaned series and bands (the default,
    3, is what test_cfg5_wide_cleaning_vs_oracle runs against the oracle)."""
    from timeband_amd import synthetic
    from timeband_amd.cleaner import TIMEBANDCleaner
    from timeband_amd.dataset import TIMEBANDDataset
    from timeband_amd.model import TIMEBANDModel
    from timeband_amd.utils.logger import NullLogger
    V, To, Tf, H, L, B, N = 40, 12, 12, 64, 1, 30, 400
    values = synthetic.make_series(N, V, seed=9)
    tfeat = synthetic.time_features(N + Tf)
    core = dict(logger=NullLogger(), device=dev, batch_size=B, observed_len=To, forecast_len=Tf, directory="",
                data_name="synthetic")
    ds = TIMEBANDDataset.from_arrays(values, tfeat, {**core, "stride": 1, "split_rate": 0.9, "min_valid_scale": 2,
                                                     "sliding_step": 0})
    torch.manual_seed(11)
    netD, netG = make_nets(V, V + tfeat.shape[1], To, Tf, H, L, dev)
    holder = TIMEBANDModel.__new__(TIMEBANDModel)
    holder.netG, holder.netD = netG, None
    cfg = {**core, "band_width": 1.0, "missing_gamma": 0.2, "anomaly_gamma": 0.3}
    full = ds.prepare_dataset(split=False)
    want = [t.clone() for t in TIMEBANDCleaner(cfg, ds, holder).clean_arrays(full.loader(B))]
    monkeypatch.setenv("TB_CLEAN_PIPELINE", str(depth))
    got = TIMEBANDCleaner(cfg, ds, holder).clean_arrays(full.loader(B))
    torch.cuda.synchronize()
    for a, b in zip(got, want):
        assert torch.equal(torch.nan_to_num(a, nan=1e30), torch.nan_to_num(b, nan=1e30))

#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

This is the only place in the repository that touches /root/reference: it imports the
reference's Gen-A modules (source/utils/lstm_layer.py, source/loss.py, source/trainer.py,
source/cleaner.py, source/predictor.py, source/dataset.py, source/metric.py,
source/model.py, source/utils/initiate.py) as a black box, feeds them seeded inputs
and writes ONLY inputs and outputs (numpy arrays) to *.npz.  No reference source text is
copied anywhere.  Run it in the build container (the GPU box has no /root/reference):

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python tests/golden/make_golden.py

The reference has no tests and no fixtures of its own, so these files
are what pins the oracle (oracle/timeband_oracle.py) and, through it, the HIP path.
torch 2.10.0 CPU (oneDNN LSTM), numpy 2.2.6, pandas 2.3.3.
"""
import os
import sys
import shutil
import tempfile

os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("TIMEBAND_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
sys.path.insert(0, REPO)

import numpy as np  # noqa: E402
import torch  # noqa: E402
from torch.optim import Adam, RMSprop  # noqa: E402
from torch.utils.data import DataLoader  # noqa: E402

from source.utils.lstm_layer import LSTMGenerator, LSTMDiscriminator  # noqa: E402
from source.loss import TIMEBANDLoss  # noqa: E402
from source.metric import TIMEBANDMetric  # noqa: E402
from source.model import TIMEBANDModel  # noqa: E402
from source.dataset import TIMEBANDDataset  # noqa: E402
from source.trainer import TIMEBANDTrainer  # noqa: E402
from source.cleaner import TIMEBANDCleaner  # noqa: E402
from source.predictor import TIMEBANDPredictor  # noqa: E402
from source.dashboard import TIMEBANDDashboard  # noqa: E402
from source.utils.initiate import seeding  # noqa: E402

from timeband_amd import synthetic  # noqa: E402

torch.set_num_threads(1)  # one thread: reproducible reduction order


class StubLogger:
    """3-method stand-in for the reference's structlog Logger (structlog is absent)."""

    def info(self, *a, **k):
        pass

    def warn(self, *a, **k):
        pass

    def debug(self, *a, **k):
        pass


def sd_numpy(module, prefix):
    return {f"{prefix}.{k}": v.detach().cpu().numpy().copy() for k, v in module.state_dict().items()}


def grads_numpy(module, prefix):
    return {f"{prefix}.{k}": (p.grad.detach().cpu().numpy().copy() if p.grad is not None
                              else np.zeros(tuple(p.shape), np.float32))
            for k, p in module.named_parameters()}


LOSS_W = dict(l1_weight=50, l2_weight=50, gp_weight=50)


def make_losses():
    return TIMEBANDLoss({**LOSS_W, "device": torch.device("cpu")})


# --------------------------------------------------------------------------------------
# G1-G5: module-level scenarios (init, forward, losses, grads, optimizer steps)
# --------------------------------------------------------------------------------------
SCENARIOS = {
    # cfg1 of BASELINE.json (7 variables + 4 time features, window 48, hidden 64)
    "cfg1": dict(V=7, E=11, To=48, Tf=48, H=64, L=1, B=8),
    # observed_len != forecast_len, odd batch: exercises the reinterpreting views
    "uneven": dict(V=3, E=5, To=10, Tf=3, H=64, L=1, B=5),
    # 2 stacked layers per nn.LSTM (cfg4 style)
    "twolayer": dict(V=4, E=6, To=12, Tf=12, H=64, L=2, B=4),
}


def run_scenario(name, V, E, To, Tf, H, L, B, lr=2e-4):
    dev = torch.device("cpu")
    seeding(31)
    netD = LSTMDiscriminator(V, H, L, dev)           # D first (model.py:56-57)
    netG = LSTMGenerator(E, V, To, Tf, H, L, dev)
    out = {}
    out.update(sd_numpy(netD, "init.netD"))
    out.update(sd_numpy(netG, "init.netG"))

    x = torch.rand(B, To, E) * 2 - 1
    y = torch.rand(B, Tf, V) * 2 - 1
    xc = torch.rand(Tf, To, E) * 2 - 1                # critic batch = Tf windows (dataset.py:249-252)
    yc = torch.rand(Tf, Tf, V) * 2 - 1
    out.update(x=x.numpy().copy(), y=y.numpy().copy(), xc=xc.numpy().copy(), yc=yc.numpy().copy())

    # ---- G2 forward
    with torch.no_grad():
        Gx = netG(x)
        out["fwd.Gx"] = Gx.numpy().copy()
        out["fwd.Dy"] = netD(y).numpy().copy()
        out["fwd.DGx"] = netD(Gx).numpy().copy()

    # ---- G3 losses on the forward outputs
    losses = make_losses()
    with torch.no_grad():
        Gx_t, Dy_t, DGx_t = (torch.tensor(out[k]) for k in ("fwd.Gx", "fwd.Dy", "fwd.DGx"))
        out["loss.dis"] = np.float32(losses.dis_loss(y, Gx_t, Dy_t, DGx_t).item())
        out["loss.dis_critic"] = np.float32(losses.dis_loss(y, Gx_t, Dy_t, DGx_t, critic=True).item())
        out["loss.gen"] = np.float32(losses.gen_loss(y, Gx_t, DGx_t).item())
        out["loss.gan_fake"] = np.float32(losses.GANloss(DGx_t, False).item())
        out["loss.gan_real"] = np.float32(losses.GANloss(Dy_t, True).item())
        out["loss.l1"] = np.float32(losses.l1loss(y, Gx_t).item())
        out["loss.l2"] = np.float32(losses.l2loss(y, Gx_t).item())
        out["loss.gp"] = np.float32(losses.grad_penalty(y, Gx_t).item())

    # ---- G4/G5: one iteration exactly as trainer.py:142-194 (critic=1)
    losses = make_losses()
    optimD_adam = Adam(netD.parameters(), lr=lr)     # trainer.py:95-98 (once per epoch)
    optimG_adam = Adam(netG.parameters(), lr=lr)

    def one_iteration(tag, xb, yb, xcb, ycb, full=True, final=False):
        def keep(d):
            if full:
                out.update(d)
        # critic (trainer.py:142-158)
        optimD = RMSprop(netD.parameters(), lr=lr)
        fake = netG(xcb)
        optimD.zero_grad()
        Dy = netD(ycb)
        DGx = netD(fake)
        errD = losses.dis_loss(ycb, fake, Dy, DGx, critic=True)
        errD.backward()
        out[f"{tag}.critic.errD"] = np.float32(errD.item())
        keep(grads_numpy(netD, f"{tag}.critic.grad.netD"))
        optimD.step()
        keep(sd_numpy(netD, f"{tag}.critic.after.netD"))
        # trainer.py:160-161
        optimD_adam.zero_grad()
        optimG_adam.zero_grad()
        # D phase (trainer.py:173-180)
        fake = netG(xb)
        Dy = netD(yb)
        DGx = netD(fake)
        errD = losses.dis_loss(yb, fake, Dy, DGx)
        errD.backward()
        out[f"{tag}.dphase.errD"] = np.float32(errD.item())
        keep(grads_numpy(netD, f"{tag}.dphase.grad.netD"))
        keep(grads_numpy(netG, f"{tag}.dphase.grad.netG"))
        optimD_adam.step()
        if full or final:
            out.update(sd_numpy(netD, f"{tag}.dphase.after.netD"))
        # G phase (trainer.py:188-194)
        fake = netG(xb)
        DGx = netD(fake)
        errG = losses.gen_loss(yb, fake, DGx)
        errG.backward()
        out[f"{tag}.gphase.errG"] = np.float32(errG.item())
        keep(grads_numpy(netG, f"{tag}.gphase.grad.netG"))   # accumulated on D-phase grads
        optimG_adam.step()
        if full or final:
            out.update(sd_numpy(netG, f"{tag}.gphase.after.netG"))
        out[f"{tag}.fake_y"] = fake.detach().numpy().copy()

    one_iteration("it0", x, y, xc, yc)
    if name != "cfg1":  # keep the cfg1 file near 1 MB: further iterations on the small ones
        # two more iterations on fresh inputs: checks Adam state carry + per-batch fresh RMSprop;
        # only scalars are kept for it1, scalars + final parameters for it2
        for it in (1, 2):
            xb = torch.rand(B, To, E) * 2 - 1
            yb = torch.rand(B, Tf, V) * 2 - 1
            xcb = torch.rand(Tf, To, E) * 2 - 1
            ycb = torch.rand(Tf, Tf, V) * 2 - 1
            out[f"it{it}.x"], out[f"it{it}.y"] = xb.numpy().copy(), yb.numpy().copy()
            out[f"it{it}.xc"], out[f"it{it}.yc"] = xcb.numpy().copy(), ycb.numpy().copy()
            one_iteration(f"it{it}", xb, yb, xcb, ycb, full=False, final=(it == 2))
    out["meta"] = np.array([V, E, To, Tf, H, L, B], dtype=np.int64)
    out["lr"] = np.float64(lr)
    np.savez_compressed(os.path.join(HERE, f"scenario_{name}.npz"), **out)
    print(f"scenario_{name}.npz: {len(out)} arrays")


# --------------------------------------------------------------------------------------
# Hidden 256 (the clustered kernels' sizes): cfg2's shape and model.cfg's observed 10 / forecast 3.
# Stored compactly so each file stays near 1 MB: the inputs are NOT stored -- the test regenerates them
# with the same seed and torch.rand calls (init is bit-identical by seed) and checks the checksums kept
# here; tensors above 16384 elements are kept as a seeded sample of 4096 entries (`sample_index`).
# --------------------------------------------------------------------------------------
BIG = {
    "h256": dict(V=32, E=36, To=128, Tf=128, H=256, L=1, B=4),        # BASELINE.json configs[1] shape
    "uneven_h256": dict(V=3, E=5, To=10, Tf=3, H=256, L=1, B=5),      # model.cfg: observed 10, forecast 3, hidden 256
}
SAMPLE_ABOVE, SAMPLE_N, SAMPLE_SEED = 16384, 4096, 1234


def sample_index(numel):
    return np.sort(np.random.default_rng(SAMPLE_SEED + numel % 9973).choice(numel, SAMPLE_N, replace=False))


def pack(a):
    a = np.asarray(a)
    return a.copy() if a.size <= SAMPLE_ABOVE else a.reshape(-1)[sample_index(a.size)].copy()


def checksum(a):
    a = np.asarray(a, np.float64).reshape(-1)
    return np.array([a.sum(), (a * a).sum(), (a * np.arange(1, a.size + 1) % 7).sum()])


def big_inputs(V, E, To, Tf, B):
    """The scenario's inputs, drawn right after the networks were built (same calls, same order, in the test)."""
    x = torch.rand(B, To, E) * 2 - 1
    y = torch.rand(B, Tf, V) * 2 - 1
    xc = torch.rand(Tf, To, E) * 2 - 1
    yc = torch.rand(Tf, Tf, V) * 2 - 1
    return x, y, xc, yc


def run_scenario_big(name, V, E, To, Tf, H, L, B, lr=2e-4):
    dev = torch.device("cpu")
    seeding(31)
    netD = LSTMDiscriminator(V, H, L, dev)
    netG = LSTMGenerator(E, V, To, Tf, H, L, dev)
    out = {}
    for pre, net in (("init.netD", netD), ("init.netG", netG)):
        for k, v in net.state_dict().items():
            out[f"{pre}.{k}.checksum"] = checksum(v.numpy())
    x, y, xc, yc = big_inputs(V, E, To, Tf, B)
    for k, v in (("x", x), ("y", y), ("xc", xc), ("yc", yc)):
        out[f"{k}.checksum"] = checksum(v.numpy())
    with torch.no_grad():
        Gx = netG(x)
        out["fwd.Gx"] = Gx.numpy().copy()
        out["fwd.Dy"] = netD(y).numpy().copy()
        out["fwd.DGx"] = netD(Gx).numpy().copy()
        losses = make_losses()
        Dy_t, DGx_t = torch.tensor(out["fwd.Dy"]), torch.tensor(out["fwd.DGx"])
        out["loss.dis"] = np.float32(losses.dis_loss(y, Gx, Dy_t, DGx_t).item())
        out["loss.dis_critic"] = np.float32(losses.dis_loss(y, Gx, Dy_t, DGx_t, critic=True).item())
        out["loss.gen"] = np.float32(losses.gen_loss(y, Gx, DGx_t).item())
        out["loss.gan_fake"] = np.float32(losses.GANloss(DGx_t, False).item())
        out["loss.gan_real"] = np.float32(losses.GANloss(Dy_t, True).item())
        out["loss.l1"] = np.float32(losses.l1loss(y, Gx).item())
        out["loss.l2"] = np.float32(losses.l2loss(y, Gx).item())
        out["loss.gp"] = np.float32(losses.grad_penalty(y, Gx).item())
    # one iteration exactly as trainer.py:142-194 (critic=1)
    losses = make_losses()
    adamD, adamG = Adam(netD.parameters(), lr=lr), Adam(netG.parameters(), lr=lr)

    def keep_grads(net, tag):
        for k, p_ in net.named_parameters():
            out[f"{tag}.{k}"] = pack(p_.grad.detach().numpy())
            out[f"{tag}.{k}.absmax"] = np.float64(p_.grad.detach().abs().max().item())

    def keep_params(net, tag):
        for k, v in net.state_dict().items():
            out[f"{tag}.{k}"] = pack(v.numpy())

    # The WGAN critic's gradients are residues of two cancelling sums (-mean D(real) + mean D(fake), loss.py:107-112;
    # e.g. the LSTM bias: 1.1e-2 - 1.1e-2 -> 4e-4): the scale an fp32 implementation can be held to is the size of
    # the PARTS.  Their largest entries per tensor are recorded for the tolerance (two extra backward passes of the
    # reference's own modules; no parameter is touched).
    with torch.no_grad():
        fake0 = netG(xc)
    parts = []
    for term in (lambda: -netD(yc).mean(), lambda: netD(fake0).mean()):
        netD.zero_grad()
        term().backward()
        parts.append({k: p_.grad.detach().abs().max().item() for k, p_ in netD.named_parameters()})
    netD.zero_grad()
    for k in parts[0]:
        out[f"it0.critic.grad.netD.{k}.partmax"] = np.float64(max(parts[0][k], parts[1][k]))
    optimD = RMSprop(netD.parameters(), lr=lr)
    fake = netG(xc)
    optimD.zero_grad()
    errD = losses.dis_loss(yc, fake, netD(yc), netD(fake), critic=True)
    errD.backward()
    out["it0.critic.errD"] = np.float32(errD.item())
    keep_grads(netD, "it0.critic.grad.netD")
    optimD.step()
    keep_params(netD, "it0.critic.after.netD")
    adamD.zero_grad()
    adamG.zero_grad()
    fake = netG(x)
    errD = losses.dis_loss(y, fake, netD(y), netD(fake))
    errD.backward()
    out["it0.dphase.errD"] = np.float32(errD.item())
    keep_grads(netD, "it0.dphase.grad.netD")
    keep_grads(netG, "it0.dphase.grad.netG")
    adamD.step()
    keep_params(netD, "it0.dphase.after.netD")
    fake = netG(x)
    errG = losses.gen_loss(y, fake, netD(fake))
    errG.backward()
    out["it0.gphase.errG"] = np.float32(errG.item())
    keep_grads(netG, "it0.gphase.grad.netG")
    adamG.step()
    keep_params(netG, "it0.gphase.after.netG")
    out["it0.fake_y"] = fake.detach().numpy().copy()
    out["meta"] = np.array([V, E, To, Tf, H, L, B], dtype=np.int64)
    out["lr"] = np.float64(lr)
    out["sampling"] = np.array([SAMPLE_ABOVE, SAMPLE_N, SAMPLE_SEED], dtype=np.int64)
    path = os.path.join(HERE, f"scenario_{name}.npz")
    np.savez_compressed(path, **out)
    print(f"scenario_{name}.npz: {len(out)} arrays, {os.path.getsize(path) / 1e6:.2f} MB")


# --------------------------------------------------------------------------------------
# G6: band (predicts) and adjust, straight from the three reference classes
# --------------------------------------------------------------------------------------
class _DS:
    pass


def bare(cls, Tf, V, band_width, a, b):
    obj = object.__new__(cls)
    ds = _DS()
    ds.decode_shape = (0, Tf, V)
    obj.dataset = ds
    obj.band_width = band_width
    obj.missing_gamma = a
    obj.anomaly_gamma = b
    return obj


def run_band_adjust():
    rng = np.random.default_rng(7)
    out = {}
    Tf, V = 5, 3
    out["meta"] = np.array([Tf, V], dtype=np.int64)
    out["band_width"] = np.float64(1.5)
    batches = [6, 4, 7, 2]
    band_in = [rng.standard_normal((B, Tf, V)).astype(np.float32) for B in batches]
    for cname, cls in (("trainer", TIMEBANDTrainer), ("cleaner", TIMEBANDCleaner),
                       ("predictor", TIMEBANDPredictor)):
        obj = bare(cls, Tf, V, 1.5, 0.2, 0.3)
        obj.pred_initate()
        for i, B in enumerate(batches):
            pred = torch.tensor(band_in[i].copy())
            out[f"band.pred{i}"] = band_in[i].copy()
            preds, lower, upper = obj.predicts(pred)
            out[f"band.{cname}.preds{i}"] = preds.copy()
            out[f"band.{cname}.lower{i}"] = lower.copy()
            out[f"band.{cname}.upper{i}"] = upper.copy()
            out[f"band.{cname}.state{i}"] = obj.preds.copy()
    # one more band case with Tf=2 (smallest legal forecast_len) and B < Tf-1 style carry
    obj = bare(TIMEBANDCleaner, 2, 2, 1.0, 0.2, 0.3)
    obj.pred_initate()
    for i, B in enumerate([1, 1, 3]):
        pred = torch.tensor(rng.standard_normal((B, 2, 2)).astype(np.float32))
        out[f"band2.pred{i}"] = pred.numpy().copy()
        p, lo, up = obj.predicts(pred)
        out[f"band2.preds{i}"], out[f"band2.lower{i}"], out[f"band2.upper{i}"] = p.copy(), lo.copy(), up.copy()
    # long-horizon band where B < Tf - 1 (carry spans several batches)
    obj = bare(TIMEBANDTrainer, 6, 2, 2.0, 0.2, 0.3)
    obj.pred_initate()
    for i, B in enumerate([2, 3, 2, 4]):
        pred = torch.tensor(rng.standard_normal((B, 6, 2)).astype(np.float32))
        out[f"band6.pred{i}"] = pred.numpy().copy()
        p, lo, up = obj.predicts(pred)
        out[f"band6.preds{i}"], out[f"band6.lower{i}"], out[f"band6.upper{i}"] = p.copy(), lo.copy(), up.copy()

    # adjust: three variants, missing at p=0 and elsewhere, values outside the band
    n = 12
    output0 = rng.standard_normal((n + 1, V))
    preds = rng.standard_normal((n, V))
    lower = preds - 0.4 * np.abs(rng.standard_normal((n, V)))
    upper = preds + 0.4 * np.abs(rng.standard_normal((n, V)))
    masks = (rng.random((n, V)) < 0.3).astype(np.int64)
    masks[0, 0] = 1
    out.update({"adj.output": output0.copy(), "adj.preds": preds, "adj.lower": lower,
                "adj.upper": upper, "adj.masks": masks, "adj.a": np.float64(0.2), "adj.b": np.float64(0.3)})
    for cname, cls in (("trainer", TIMEBANDTrainer), ("cleaner", TIMEBANDCleaner),
                       ("predictor", TIMEBANDPredictor)):
        obj = bare(cls, Tf, V, 1.5, 0.2, 0.3)
        o = output0.copy()
        tgt = obj.adjust(o, preds, masks, lower, upper)
        out[f"adj.{cname}.target"] = np.array(tgt).copy()
    np.savez_compressed(os.path.join(HERE, "band_adjust.npz"), **out)
    print(f"band_adjust.npz: {len(out)} arrays")


# --------------------------------------------------------------------------------------
# metric (reference source/metric.py) on random data
# --------------------------------------------------------------------------------------
def run_metric():
    rng = np.random.default_rng(11)
    out = {}
    m = TIMEBANDMetric({"device": torch.device("cpu")})
    for i in range(3):
        true = rng.standard_normal((9, 4))
        true[rng.random((9, 4)) < 0.1] = 0.0
        pred = true + 0.1 * rng.standard_normal((9, 4))
        mask = (rng.random((9, 4)) < 0.2).astype(np.int64)
        out[f"true{i}"], out[f"pred{i}"], out[f"mask{i}"] = true, pred, mask
        t, p, k = torch.tensor(true), torch.tensor(pred), torch.tensor(mask)
        out[f"nmae{i}"] = np.float64(m.NMAE(t, p, k))
        out[f"rmse{i}"] = np.float64(m.RMSE(t, p, k))
        out[f"nme{i}"] = np.float64(m.NME(t, p, k))
    sc = m.score(2)
    out["score"] = np.array([sc["NME"], sc["NMAE"], sc["RMSE"]])
    np.savez_compressed(os.path.join(HERE, "metric.npz"), **out)
    print("metric.npz")


# --------------------------------------------------------------------------------------
# G7/G8: dataset pipeline, one full Gen-A epoch (train + valid), clean() and predict()
# --------------------------------------------------------------------------------------
def pipeline_config(directory, data_name, V, To, Tf, H, L, B, split_rate=0.8, epochs=1,
                    band_width=1.0, missing_gamma=0.2, anomaly_gamma=0.3):
    logger, dev = StubLogger(), torch.device("cpu")
    core = dict(TAG="golden", data_name=data_name, directory=directory, pretrain=False, workers=0,
                batch_size=B, observed_len=To, forecast_len=Tf, verbosity=9)
    inj = dict(logger=logger, device=dev)
    dataset = dict(reset=True, targets=[f"v{i}" for i in range(V)], drops=[], time_index="Date",
                   time_format="%Y-%m-%d %H:%M:%S",
                   time_encode=dict(year=False, month=True, weekday=True, day=True, hour=True, minute=False),
                   fill_timegap=False, stride=1, split_rate=split_rate, min_valid_scale=2, sliding_step=0)
    models = dict(save_opt=False, load_opt=False, best_score=50.0, hidden_dim=H, layers_num=L,
                  models_path=os.path.join(directory, "models"))
    losses = dict(LOSS_W)
    trainer = dict(models=dict(reload=False, reload_interval=50, save_interval=10), critic=1, epochs=epochs,
                   learning_rate=dict(base=2e-4, decay=0.9, gammaD=1.0, gammaG=1.0),
                   band_width=band_width, missing_gamma=missing_gamma, anomaly_gamma=anomaly_gamma,
                   print_interval=1)
    dashboard = dict(vis_opt=False, vis_column=1, visual_cols=["v0"], scope=120, feats_by_rows=1,
                     xinterval=14, height=10, width=25)
    return dict(core=core, inj=inj, dataset=dataset, models=models, losses=losses, trainer=trainer,
                dashboard=dashboard)


def run_pipeline():
    V, To, Tf, H, L, B, N = 3, 24, 12, 64, 1, 32, 420
    tmp = tempfile.mkdtemp(prefix="tb_golden_")
    try:
        name = "synth"
        synthetic.write_csv(tmp, name, N, V, seed=31)
        os.makedirs(os.path.join(tmp, name), exist_ok=True)
        cfg = pipeline_config(tmp, name, V, To, Tf, H, L, B)
        seeding(31)
        ds = TIMEBANDDataset({**cfg["core"], **cfg["dataset"], **cfg["inj"]})
        out = {}
        out["meta"] = np.array([V, To, Tf, H, L, B, N], dtype=np.int64)
        out["ds.columns"] = np.array(list(ds.data.columns))
        out["ds.normalized"] = ds.data.to_numpy().copy()
        out["ds.observed"] = ds.observed.numpy().copy()
        out["ds.forecast"] = ds.forecast.numpy().copy()
        out["ds.missing"] = np.asarray(ds.missing).copy()
        out["ds.decode_min"] = ds.decode_min.numpy().copy()
        out["ds.decode_max"] = ds.decode_max.numpy().copy()
        out["ds.data_length"] = np.int64(ds.data_length)
        out["ds.times_first_last"] = np.array([ds.times[0], ds.times[-1]])
        import pandas as pd
        orig = pd.read_csv(ds.datapath)
        out["ds.original"] = orig.drop(columns=["Date"]).to_numpy().copy()   # after expand + time features
        trainset, validset = ds.prepare_dataset()
        out["ds.train_len"] = np.int64(len(trainset))
        out["ds.valid_len"] = np.int64(len(validset))
        out["ds.win0.encoded"] = trainset[0]["encoded"].numpy().copy()
        out["ds.win0.decoded"] = trainset[0]["decoded"].numpy().copy()
        out["ds.winlast.encoded"] = validset[len(validset) - 1]["encoded"].numpy().copy()
        out["ds.winlast.decoded"] = validset[len(validset) - 1]["decoded"].numpy().copy()

        models = TIMEBANDModel({**cfg["core"], **cfg["models"], **cfg["inj"]})
        models.initiate(dims=ds.dims)
        out.update(sd_numpy(models.netD, "init.netD"))
        out.update(sd_numpy(models.netG, "init.netG"))
        metric = TIMEBANDMetric({**cfg["core"], **cfg["inj"]})
        losses = TIMEBANDLoss({**cfg["core"], **cfg["losses"], **cfg["inj"]})
        dash = TIMEBANDDashboard({**cfg["core"], **cfg["dashboard"], **cfg["inj"]}, ds)
        tcfg = {**cfg["core"], **cfg["trainer"], **cfg["inj"]}
        trainer = TIMEBANDTrainer(tcfg, ds, models, metric, losses, dash)

        # wrap train_step to snapshot what the epoch produced (the reference keeps it internal)
        calls = []
        orig_step = trainer.train_step

        def wrapped(epoch, loader, training=True):
            score = orig_step(epoch, loader, training=training)
            tag = f"ep{epoch}.{'train' if training else 'valid'}"
            out[f"{tag}.score"] = np.float64(score)
            nb = len(loader)
            ld = trainer.losses.loss(nb - 1)
            out[f"{tag}.losses"] = np.array([ld[k] for k in ("D", "R", "F", "G", "G_", "L1", "L2", "GP")])
            sd = trainer.metric.score(nb - 1)
            out[f"{tag}.scores"] = np.array([sd[k] for k in ("NME", "NMAE", "RMSE")])
            out[f"{tag}.band_state"] = trainer.preds.copy()
            if training:
                out.update(sd_numpy(models.netD, f"{tag}.netD"))
                out.update(sd_numpy(models.netG, f"{tag}.netG"))
            calls.append(tag)
            return score

        trainer.train_step = wrapped
        trainer.train(DataLoader(trainset, batch_size=B), DataLoader(validset, batch_size=B))
        out["calls"] = np.array(calls)

        # clean() and predict() over the whole series with the trained weights
        full = ds.prepare_dataset(split=False)
        cleaner = TIMEBANDCleaner(tcfg, ds, models, metric, losses, dash)
        outputs_df, bands_df = cleaner.clean(DataLoader(full, batch_size=B))
        out["clean.outputs"] = outputs_df.to_numpy().copy()
        out["clean.bands"] = bands_df.to_numpy().copy()
        out["clean.columns"] = np.array(list(bands_df.columns))
        predictor = TIMEBANDPredictor(tcfg, ds, models, metric, losses, dash)
        pred_df = predictor.predict(DataLoader(full, batch_size=B))
        out["predict.outputs"] = pred_df.to_numpy().copy()
        out["full_len"] = np.int64(len(full))
        np.savez_compressed(os.path.join(HERE, "pipeline.npz"), **out)
        print(f"pipeline.npz: {len(out)} arrays; calls={calls}")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    if "h256" in sys.argv[1:]:          # only the hidden-256 scenarios (the other files stay as committed)
        for nm, sc in BIG.items():
            run_scenario_big(nm, **sc)
        sys.exit(0)
    for nm, sc in BIG.items():
        run_scenario_big(nm, **sc)
    for nm, sc in SCENARIOS.items():
        run_scenario(nm, **sc)
    run_band_adjust()
    run_metric()
    run_pipeline()

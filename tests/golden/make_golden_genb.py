#!/usr/bin/env python3
"""Golden vectors for the Gen-B API (tests/golden/genb.npz), made by RUNNING THE REFERENCE.

What the reference can run here: source/losses.py (GANLoss, TimebandLoss) and source/metrics.py
(TIMEBANDMetric) import cleanly; source/core.py does not (it names classes it never imports), so the
network comes from the reference's source/utils/lstm_layer.py -- the same architecture as
models.py:10-117 -- and the Gen-B schedule of core.py:64-130 (persistent RMSprop critic with the LSGAN
loss, D step, then a G step whose gradients start from zero) is driven from this script with the
reference's own loss object and torch's optimizers.  source/models.py imports once `structlog` (not
installed; only the logger's formatting uses it) is stood in for by an empty module object at import
time: TimebandModel.predicts / adjust (models.py:173-215) are then run unbound on a bare instance --
no Logger, no networks -- over three consecutive batches.  Only inputs and outputs are written.

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_golden_genb.py
"""
import os
import sys

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("TIMEBAND_REFERENCE", "/root/reference")
sys.path.insert(0, REF)

import numpy as np  # noqa: E402
import pandas as pd  # noqa: E402
import torch  # noqa: E402
from torch.optim import Adam, RMSprop  # noqa: E402

from source.losses import GANLoss, TimebandLoss  # noqa: E402
from source.metrics import TIMEBANDMetric  # noqa: E402
from source.utils.initiate import seeding  # noqa: E402
from source.utils.lstm_layer import LSTMDiscriminator, LSTMGenerator  # noqa: E402

torch.set_num_threads(1)
dev = torch.device("cpu")
V, E, To, Tf, H, L, B = 5, 9, 10, 5, 64, 1, 6
LR = 2e-4
W = dict(l1_weights=1.5, l2_weights=2.5, gp_weights=0.7)
out = {"meta": np.array([V, E, To, Tf, H, L, B]), "lr": np.float64(LR),
       "weights": np.array([W["l1_weights"], W["l2_weights"], W["gp_weights"]])}


def sd(module, prefix):
    return {f"{prefix}.{k}": v.detach().numpy().copy() for k, v in module.state_dict().items()}


def grads(module, prefix):
    return {f"{prefix}.{k}": (p.grad.detach().numpy().copy() if p.grad is not None
                              else np.zeros(tuple(p.shape), np.float32)) for k, p in module.named_parameters()}


seeding(42)                                   # launcher.py:9
netD = LSTMDiscriminator(V, H, L, dev)        # D before G (models.py:145-158)
netG = LSTMGenerator(E, V, To, Tf, H, L, dev)
out.update(sd(netD, "init.netD"))
out.update(sd(netG, "init.netG"))
rnd = lambda *s: torch.rand(*s) * 2 - 1
data = dict(x0=rnd(B, To, E), y0=rnd(B, Tf, V), x1=rnd(B, To, E), y1=rnd(B, Tf, V),
            xc0=rnd(Tf, To, E), yc0=rnd(Tf, Tf, V), xc1=rnd(Tf, To, E), yc1=rnd(Tf, Tf, V))
out.update({k: v.numpy().copy() for k, v in data.items()})

# ---- GANLoss, all modes (losses.py:36-53)
p = torch.rand(B, Tf, 1)
out["gan.p"] = p.numpy().copy()
for mode in ("vanilla", "lsgan", "wgangp"):
    for real in (True, False):
        q = p.clone().requires_grad_()
        val = GANLoss(mode)(q, real)
        val.backward()
        out[f"gan.{mode}.{int(real)}.val"] = np.float32(val.item())
        out[f"gan.{mode}.{int(real)}.grad"] = q.grad.numpy().copy()

# ---- the Gen-B schedule (core.py:64-130)
losses = TimebandLoss(**W)
optimD, optimG = Adam(netD.parameters(), lr=LR), Adam(netG.parameters(), lr=LR)
optimC = RMSprop(netD.parameters(), lr=LR)
losses.init_loss()
for k in range(2):                             # Core.critic(trainset, 2)
    optimC.zero_grad()
    fake = netG(data[f"xc{k}"])
    errC = losses.dis_loss(netD(data[f"yc{k}"]), netD(fake))
    errC.backward()
    out[f"critic{k}.errC"] = np.float32(errC.item())
    out.update(grads(netD, f"critic{k}.grad.netD"))
    optimC.step()
    out.update(sd(netD, f"critic{k}.after.netD"))
losses.init_loss()                             # train_step, core.py:94
for k in range(2):
    x, y = data[f"x{k}"], data[f"y{k}"]
    optimD.zero_grad()
    fake = netG(x)
    errD = losses.dis_loss(netD(y), netD(fake))
    errD.backward()
    out[f"it{k}.errD"] = np.float32(errD.item())
    out.update(grads(netD, f"it{k}.dphase.grad.netD"))
    optimD.step()
    out.update(sd(netD, f"it{k}.after.netD"))
    optimG.zero_grad()
    fake = netG(x)
    errG = losses.gen_loss(y, fake, netD(fake))
    errG.backward()
    out[f"it{k}.errG"] = np.float32(errG.item())
    out[f"it{k}.fake_y"] = fake.detach().numpy().copy()
    out.update(grads(netG, f"it{k}.gphase.grad.netG"))
    optimG.step()
    out.update(sd(netG, f"it{k}.after.netG"))
log = losses.loss(1)
out["loss.keys"] = np.array(list(log.keys()))
out["loss.values"] = np.array([float(v) for v in log.values()])

# ---- TIMEBANDMetric.scoring (metrics.py:19-27): DataFrames in, pred longer than true
rng = np.random.default_rng(42)
true = rng.normal(size=(Tf, V)) * 3
true[1, 2] = 0.0
mask = (rng.random((Tf, V)) < 0.2).astype(int)
pred = rng.normal(size=(B + Tf - 1, V)) * 3
m = TIMEBANDMetric()
m.init_score()
nmae, rmse, nme = m.scoring(pd.DataFrame(true), pred, pd.DataFrame(mask))
out.update({"metric.true": true, "metric.mask": mask, "metric.pred": pred,
            "metric.out": np.array([float(nmae), float(rmse), float(nme)])})
sc = m.score(0)
out["metric.score"] = np.array([sc["NME"], sc["NMAE"], sc["RMSE"]])

# ---- TimebandModel.predicts / adjust (models.py:160-215): three consecutive batches, history carried
import types  # noqa: E402
from unittest import mock  # noqa: E402

sys.modules.setdefault("structlog", mock.MagicMock(name="structlog"))   # logger.py:3,8-9 touch it at import
from source.models import TimebandModel  # noqa: E402

rng = np.random.default_rng(7)
tm = TimebandModel.__new__(TimebandModel)      # the two methods read these four attributes only
tm.forecast_len, tm.decode_dim = Tf, V
tm.missing_gamma, tm.anomaly_gamma = 0.2, 0.3
tm.pred_initiate()
NB = 3
series = rng.normal(size=(NB * B + Tf, V)) * 2 + 10
miss = (rng.random((NB * B + Tf - 1, V)) < 0.15).astype(np.float64)
out.update({"band.series": series, "band.miss": miss, "band.gammas": np.array([0.2, 0.3])})
carry = series[0].copy()
for k in range(NB):
    pred = torch.tensor(series[1 + k * B:1 + k * B + B, None, :] + rng.normal(size=(B, Tf, V)) * 1.5, dtype=torch.float32)
    pred[k, 1:3, 2] += 9.0                      # horizon outliers: wide std at a few rows
    med, lo, up = tm.predicts(pred)
    n = B + Tf - 1
    output = np.concatenate([carry[None], series[1 + k * B:1 + k * B + n]])
    output[3, 1] += 40.0                        # anomalies on both sides of the band
    output[4, 3] -= 40.0
    out[f"band.{k}.pred"] = pred.numpy().copy()
    out[f"band.{k}.output"] = output.copy()
    masks = miss[k * B:k * B + n]
    target = tm.adjust(output, med, masks, lo, up)
    out.update({f"band.{k}.median": med.copy(), f"band.{k}.lower": lo.copy(), f"band.{k}.upper": up.copy(),
                f"band.{k}.target": np.array(target, copy=True)})
    carry = target[B - 1].copy()                # the row before the next batch's first one

np.savez_compressed(os.path.join(HERE, "genb.npz"), **out)
print("wrote genb.npz with", len(out), "arrays")

"""Gen-B training core (reference source/core.py): `Timeband` with init_optimizer / critic /
train_step / predict / is_best.

Differences from Gen-A that this keeps: the critic uses the LSGAN dis_loss and
ONE RMSprop that persists across calls (core.py:70,72-89); the G phase zeroes G's gradients first,
so the D-phase pass through G does not accumulate into the G step (core.py:121-130); the band is
3*std wide; denormalisation is vectorised.  core.py as shipped names classes it never imports and
hard-codes the csv columns it drops (core.py:40-53): here the data description is keyword
arguments with the reference's values as defaults.
"""
import os

import numpy as np
import pandas as pd
import torch

from .data import TIMEBANDData
from .losses import TimebandLoss
from .metrics import TIMEBANDMetric
from .models import TimebandModel
from .optim import FusedAdam, FusedRMSprop
from .utils.initiate import init_device
from .utils.logger import NullLogger


class Timeband:
    VERSION = "Timeband v2.3.0"

    def __init__(self, datadir: str, filename: str, observed_len: int, forecast_len: int, l1_weights: int,
                 l2_weights: int, gp_weights: int, hidden_dim: int = 256, targets=None,
                 drops=(), time_index=("Date",), time_encode=("year", "month", "weekday", "day"),
                 fill_timegap=False, split_size=1.0, ignore=("KOSPI", "KOSDAQ"), device=None, logger=None):
        self.device = torch.device(device) if device is not None else init_device()
        self.logger = logger if logger is not None else NullLogger()
        self.epochs = 0
        self.best_score = 10.0
        self.idx = observed_len
        self.datadir, self.filename = datadir, filename
        self.observed_len, self.forecast_len = observed_len, forecast_len
        self.Metric = TIMEBANDMetric(device=self.device)
        self.Losses = TimebandLoss(l1_weights=l1_weights, l2_weights=l2_weights, gp_weights=gp_weights)
        if targets is None:     # core.py:40-41: every column but the time index and two index series
            cols = pd.read_csv(os.path.join(datadir, "origin", f"{filename}.csv"), nrows=1).columns
            targets = [c for c in cols if c not in tuple(time_index) + tuple(ignore)]
        self.Data = TIMEBANDData(basedir=datadir, filename=filename, targets=targets, drops=list(drops),
                                 fill_timegap=fill_timegap, time_index=list(time_index),
                                 time_encode=list(time_encode), split_size=split_size, observed_len=observed_len,
                                 forecast_len=forecast_len, device=self.device)
        self.Model = TimebandModel(encode_dim=self.Data.encode_dims, decode_dim=self.Data.decode_dims,
                                   hidden_dim=hidden_dim, observed_len=observed_len, forecast_len=forecast_len,
                                   device=self.device, logger=self.logger)

    def init_optimizer(self, lr_D: float, lr_G: float):
        """core.py:64-70: Adam for D and G, and one RMSprop over D for the critic."""
        self.optimD = FusedAdam(self.Model.netD, lr_D)
        self.optimG = FusedAdam(self.Model.netG, lr_G)
        self.optimC = FusedRMSprop(self.Model.netD, lr_D)

    def critic(self, trainset, critics):
        """core.py:72-89."""
        self.Losses.init_loss()
        for _ in range(critics):
            self.optimC.zero_grad()
            true_x, true_y = self.Data.get_random(trainset)
            true_x, true_y = true_x.to(self.device), true_y.to(self.device)
            # the reference keeps G's graph here; its G gradients are cleared unread by the next
            # optimG.zero_grad() (core.py:122), so G runs without one
            with torch.no_grad():
                fake_y = self.Model.generate(true_x)
            Dy, DGx = self.Model.netD.pair(true_y, fake_y)
            errC = self.Losses.dis_loss(Dy, DGx)
            errC.backward()
            self.optimC.all_reduce_grads()
            self.optimC.step()

    def iteration(self, true_x, true_y, training=False):
        """The D phase and the G phase of one batch (core.py:98-130).  Returns the G-phase fake_y."""
        netD = self.Model.netD
        if training:
            self.optimD.zero_grad()
        with torch.set_grad_enabled(training):
            # errD.backward() of the reference also fills G's gradients, which optimG.zero_grad()
            # (core.py:122) clears before use: the D phase sees fake_y detached
            with torch.no_grad():
                fake_y = self.Model.generate(true_x)
            Dy, DGx = netD.pair(true_y, fake_y)
            errD = self.Losses.dis_loss(Dy, DGx)
        if training:
            errD.backward()
            self.optimD.all_reduce_grads()
            self.optimD.step()
            self.optimG.zero_grad()
        dparams = [p for p in netD.parameters() if p.requires_grad] if training else []
        for p in dparams:       # D's gradients from errG.backward() are cleared unread (core.py:102)
            p.requires_grad_(False)
        try:
            with torch.set_grad_enabled(training):
                fake_y = self.Model.generate(true_x)
                DGx = self.Model.discriminate(fake_y)
                errG = self.Losses.gen_loss(true_y, fake_y, DGx)
            if training:
                errG.backward()
        finally:
            for p in dparams:
                p.requires_grad_(True)
        if training:
            self.optimG.all_reduce_grads()
            self.optimG.step()
        return fake_y.detach()

    def train_step(self, dataloader, training=False):
        """core.py:91-150; returns the mean NMAE."""
        self.Model.pred_initiate()
        self.Metric.init_score()
        self.Losses.init_loss()
        i = -1
        for i, (true_x, true_y) in enumerate(dataloader):
            true_x, true_y = true_x.to(self.device), true_y.to(self.device)
            fake_y = self.iteration(true_x, true_y, training)
            pred_y = self.Data.denormalize(fake_y)
            preds, lower, upper = self.Model.predicts(pred_y)
            batchs, pred_len, _ = pred_y.shape
            reals = self.Data.forecast[self.idx: self.idx + pred_len]
            masks = self.Data.missing_decode[self.idx: self.idx + pred_len]
            self.Metric.scoring(reals, preds, masks)
            self.idx = self.idx + batchs
        if i < 0:
            return float("nan")
        return float(self.Metric.score(i)["NMAE"])

    def predict(self, dataloader):
        """core.py:152-185: outputs / bands DataFrames, each batch's rows overlapping the previous
        batch's last forecast_len-1."""
        self.Model.pred_initiate()
        Tf, V = self.forecast_len, self.Data.decode_dims
        outputs = np.zeros((Tf, V))
        lower_bands, upper_bands = outputs.copy(), outputs.copy()
        with torch.no_grad():
            for true_x, _ in dataloader:
                fake_y = self.Model.generate(true_x.to(self.device))
                preds, lower, upper = self.Model.predicts(self.Data.denormalize(fake_y))
                outputs = np.concatenate([outputs[: 1 - Tf], preds.cpu().numpy()])
                lower_bands = np.concatenate([lower_bands[: 1 - Tf], lower.cpu().numpy()])
                upper_bands = np.concatenate([upper_bands[: 1 - Tf], upper.cpu().numpy()])
        targets = list(self.Data.targets)
        outputs_df = pd.DataFrame(outputs, columns=targets)
        bands_df = pd.concat([pd.DataFrame(lower_bands, columns=[f"{x}_lower" for x in targets]),
                              pd.DataFrame(upper_bands, columns=[f"{x}_upper" for x in targets])], axis=1)
        return outputs_df, bands_df

    def is_best(self, score: float) -> bool:
        if score < self.best_score:
            self.best_score = score
            return True
        return False

"""Device-resident band / adjust state shared by trainer, cleaner and predictor.

Restates the per-batch tail of the reference (source/trainer.py:202-212, cleaner.py:88-106,
predictor.py:86-96) without leaving the GPU: denormalise -> band (median/std over horizons, HIP)
-> adjust (sequential scan, HIP).  The reference's growing `outputs` array is replaced by absolute
row positions: target row p of the batch that starts at window idx is data row To + idx + p.
"""
import torch

from . import ops


class BandAdjust:
    def __init__(self, dataset, variant, band_width, missing_gamma, anomaly_gamma, keep_outputs=False):
        self.ds, self.variant = dataset, variant
        self.w, self.a, self.b = band_width, missing_gamma, anomaly_gamma
        self.keep = keep_outputs
        self.reset()

    def reset(self):
        """pred_initate + idx = 0 (trainer.py:101-102,259-264)."""
        d = self.ds
        Tf, V = d.forecast_len, d.decode_dim
        dev = d.series_dev.device
        self.hist = [torch.zeros(Tf - 1, Tf, V, dtype=torch.float32, device=dev) for _ in range(2)]
        self.cur = 0
        self.nhist = 0
        self.idx = 0
        self.restart_pass()

    def restart_pass(self):
        """`outputs = self.dataset.observed` at the top of every pass (trainer.py:137, cleaner.py:71-73)."""
        d = self.ds
        self.last_row = None
        if self.keep:
            n0 = d.observed_dev.shape[0]
            mk = lambda: torch.cat([d.observed_dev, torch.zeros(d.data_length - n0, d.decode_dim,
                                                                dtype=torch.float64, device=d.observed_dev.device)])
            self.outputs, self.lower_bands, self.upper_bands = mk(), mk(), mk()

    def step(self, fake_y):
        d = self.ds
        B, Tf, V = fake_y.shape
        # the history holds normalised predictions: the band kernel denormalises as it gathers
        preds, lower, upper = ops.band_update(fake_y.detach(), self.hist[self.cur], self.nhist, self.hist[self.cur ^ 1],
                                              self.w, d.dmin_dev, d.dmax_dev)
        self.cur ^= 1
        self.nhist = min(Tf - 1, self.nhist + B)
        n = B + Tf - 1
        if self.idx + n > d.forecast_dev.shape[0]:
            raise RuntimeError("post-processing ran past the end of the series")
        reals = d.forecast_dev[self.idx:self.idx + n]
        masks = d.missing_dev[self.idx:self.idx + n]
        seed = d.observed_dev[-1] if self.last_row is None else self.last_row
        target = ops.adjust_scan(self.variant, seed, reals, preds, masks, lower, upper, self.a, self.b)
        self.last_row = target[-1]
        if self.keep:
            r0 = d.observed_len + self.idx
            self.outputs[r0:r0 + n] = target
            self.lower_bands[r0:r0 + n] = lower
            self.upper_bands[r0:r0 + n] = upper
        self.idx += B
        return reals, preds, masks, lower, upper, target

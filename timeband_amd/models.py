"""Gen-B model API (reference source/models.py) on the gfx950 kernels.

`LSTMGenerator` / `LSTMDiscriminator` take the Gen-B argument order (models.py:23-31, :91) and are
otherwise the Gen-A modules of utils/lstm_layer.py (same sub-module names, so the same state_dict).
`TimebandModel` (models.py:126-215) keeps generate / discriminate / predicts / adjust; the band
(median / std over horizons, fixed 3*std width, models.py:173-194) and the adjust scan
(models.py:196-215) run on the device through tb_band_update / tb_adjust_scan.
"""
import torch

from . import ops
from .utils import lstm_layer as _gen_a
from .utils.initiate import init_device
from .utils.logger import NullLogger


class LSTMGenerator(_gen_a.LSTMGenerator):
    def __init__(self, in_dim, in_seq, out_dim, out_seq, hid_dim=256, n_layers=1, device=None):
        super().__init__(in_dim, out_dim, in_seq, out_seq, hid_dim, n_layers, device)


class LSTMDiscriminator(_gen_a.LSTMDiscriminator):
    def __init__(self, in_seq, in_dim, out_dim, out_seq, hid_dim=256, n_layers=1, device=None):
        super().__init__(in_dim, hid_dim, n_layers, device)
        self.in_seq, self.out_dim, self.out_seq, self.hid_dim = in_seq, out_dim, out_seq, hid_dim


class TimebandBase:
    def __init__(self, device=None, logger=None):
        self.device = torch.device(device) if device is not None else init_device()
        self.logger = logger if logger is not None else NullLogger()


class TimebandModel(TimebandBase):
    BAND_WIDTH = 3.0        # models.py:191-192

    def __init__(self, encode_dim: int, decode_dim: int, hidden_dim: int, observed_len: int, forecast_len: int,
                 device=None, logger=None, missing_gamma=None, anomaly_gamma=None):
        super().__init__(device, logger)
        self.encode_dim, self.decode_dim, self.hidden_dim = encode_dim, decode_dim, hidden_dim
        self.observed_len, self.forecast_len = observed_len, forecast_len
        # the reference reads these in adjust() without ever setting them (models.py:198-199)
        if missing_gamma is not None:
            self.missing_gamma = missing_gamma
        if anomaly_gamma is not None:
            self.anomaly_gamma = anomaly_gamma
        # D before G: the order the reference draws the initial weights in (models.py:145-158)
        self.netD = LSTMDiscriminator(in_dim=decode_dim, in_seq=forecast_len, hid_dim=hidden_dim, out_dim=1,
                                      out_seq=forecast_len, device=self.device).to(self.device)
        self.netG = LSTMGenerator(in_dim=encode_dim, in_seq=observed_len, hid_dim=hidden_dim, out_dim=decode_dim,
                                  out_seq=forecast_len, device=self.device).to(self.device)
        self.pred_initiate()

    def pred_initiate(self):
        """Empty horizon history (models.py:160-165)."""
        Tf, V = self.forecast_len, self.decode_dim
        self._hist = [torch.zeros(max(Tf - 1, 0), Tf, V, dtype=torch.float32, device=self.device) for _ in range(2)]
        self._cur, self._nhist = 0, 0

    def generate(self, x):
        return self.netG(x.to(self.device))

    def discriminate(self, x):
        return self.netD(x.to(self.device))

    def predicts(self, pred):
        """pred: denormalised (B,Tf,V).  Returns preds, lower, upper as float64 device tensors of
        B+Tf-1 rows (the reference returns numpy arrays of the same values, models.py:173-194)."""
        pred = pred.detach().to(self.device, torch.float32)
        B, Tf, _ = pred.shape
        med, lower, upper = ops.band_update(pred, self._hist[self._cur], self._nhist, self._hist[self._cur ^ 1],
                                            self.BAND_WIDTH)
        self._cur ^= 1
        self._nhist = min(Tf - 1, self._nhist + B)
        return med, lower, upper

    def adjust(self, output, preds, masks, lower, upper):
        """output: len+1 rows, row 0 the seed.  With 0/1 masks models.py:196-215 is the trainer's scan
        (trainer.py:238-257): the band masks are taken before the fill but zeroed where it applies."""
        f64 = lambda t: torch.as_tensor(t).to(self.device, torch.float64).contiguous()
        output, preds, lower, upper = f64(output), f64(preds), f64(lower), f64(upper)
        masks = torch.as_tensor(masks).to(self.device, torch.int32).contiguous()
        return ops.adjust_scan("trainer", output[0], output[1:], preds, masks, lower, upper, self.missing_gamma,
                               self.anomaly_gamma)

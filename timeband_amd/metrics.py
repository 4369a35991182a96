"""Gen-B metric API (reference source/metrics.py): no-argument constructor and `scoring`."""
import numpy as np
import torch

from .metric import TIMEBANDMetric as _Metric
from .utils.initiate import init_device


class TIMEBANDMetric(_Metric):
    def __init__(self, device=None) -> None:
        super().__init__(dict(device=torch.device(device) if device is not None else init_device()))

    def scoring(self, true, pred, mask):
        """(nmae, rmse, nme) of one batch, added to the running sums (metrics.py:19-27).  `true` / `mask`
        may be DataFrames; like the reference's index arithmetic, rows of `pred` beyond len(true) are
        not scored."""
        t = np.asarray(getattr(true, "values", true), dtype=np.float64)
        m = np.asarray(getattr(mask, "values", mask))
        p = pred if torch.is_tensor(pred) else torch.as_tensor(np.asarray(pred, dtype=np.float64))
        out = self.update(t, p[: t.shape[0]], m)
        nme, nmae, rmse = out.tolist()
        return nmae, rmse, nme

"""TIMEBANDMetric: NME / NMAE / RMSE on masked, non-zero entries (reference source/metric.py:24-81),
computed by one HIP reduction per batch with the running sums kept on the device."""
import torch

from . import ops


class TIMEBANDMetric:
    def __init__(self, config: dict) -> None:
        self.__dict__ = {**config, **self.__dict__}
        self.esp = 1e-7
        self._running = None
        self._pending = []
        self.init_score()

    def set_config(self, config: dict):
        self.__dict__ = {**config, **self.__dict__}

    def init_score(self):
        self._pending = []
        if self._running is not None:
            self._running.zero_()
        return self.score()

    def _run(self, device):
        if self._running is None or self._running.device != device:
            self._running = torch.zeros(3, dtype=torch.float64, device=device)
        return self._running

    @staticmethod
    def _prep(true, pred, mask, device):
        f = lambda t: torch.as_tensor(t).to(device=device, dtype=torch.float64).contiguous()
        return f(true), f(pred), torch.as_tensor(mask).to(device=device, dtype=torch.int32).contiguous()

    def update(self, true, pred, mask, full=False):
        """All three scores of one batch in one launch, added to the running sums; no host sync.
        Data parallel: each rank holds the rows of its own windows; the scores are ratios of sums over ALL rows of
        the global batch, so only the five additive sums are taken here -- they are summed over the ranks and
        turned into scores when the running values are read (one all-reduce per pass, `_flush`).  `full`: see below."""
        dev = pred.device if torch.is_tensor(pred) and pred.is_cuda else torch.device(self.device)
        t, p, m = self._prep(true, pred, mask, dev)
        if ops.dp.active():
            # full: EVERY rank holds all the rows of this batch (a replicated tail batch, the all-gather fallback of the
            # post-processing) -- they count once, from rank 0
            sums = ops.metric_sums(t, p, m)
            if full and ops.dp.ranks() > 1 and ops.dp.rank() != 0:
                sums = torch.zeros_like(sums)
            self._pending.append(sums)
            self._run(dev)
            return None
        return ops.metrics(t, p, m, self._run(dev))

    def _flush(self):
        if self._pending:
            sums = torch.stack(self._pending)
            self._pending = []
            ops.dp.all_reduce(sums)
            ops.metric_finish(sums, self._running)

    # the reference's per-score entry points (metric.py:30-61): each returns the batch score (host float)
    def _one(self, k, true, pred, mask):
        dev = pred.device if torch.is_tensor(pred) and pred.is_cuda else torch.device(self.device)
        t, p, m = self._prep(true, pred, mask, dev)
        v = ops.metrics(t, p, m, None)[k]
        self._run(dev)[k] += v
        return v.item()

    def NME(self, true, pred, mask):
        return self._one(0, true, pred, mask)

    def NMAE(self, true, pred, mask):
        return self._one(1, true, pred, mask)

    def RMSE(self, true, pred, mask):
        return self._one(2, true, pred, mask)

    def _sums(self):
        self._flush()
        return [0.0, 0.0, 0.0] if self._running is None else self._running.tolist()

    @property
    def nme(self):
        return self._sums()[0]

    @property
    def nmae(self):
        return self._sums()[1]

    @property
    def rmse(self):
        return self._sums()[2]

    def score(self, i: int = 0):
        nme, nmae, rmse = self._sums()
        return {"NME": f"{nme / (i + 1):6.3f}", "NMAE": f"{nmae / (i + 1):7.5f}", "RMSE": f"{rmse / (i + 1):6.3f}"}

"""TIMEBANDLoss on fused HIP reductions (reference: source/loss.py:54-157; Gen-B twin
source/losses.py:56-145 via `TimebandLoss`).

Same methods and return types as the reference: dis_loss / gen_loss return 0-dim tensors that
support .backward(); the eight logging sums are kept, but on the device (no host sync per batch)
and read back only by loss(i).
"""
import torch

from . import ops


class TIMEBANDLoss:
    def __init__(self, config: dict):
        self.set_config(config)
        self._running = self._running_repl = None
        self.init_loss()

    def set_config(self, config: dict):
        self.__dict__ = {**config, **self.__dict__}

    # running sums live in one device vector: [errD_real, errD_fake, err_Gen, errG_l1, errG_l2, errG_GP]
    def _run(self, device):
        if self._running is None or self._running.device != device:
            self._running = torch.zeros(8, dtype=torch.float32, device=device)
            self._running_repl = torch.zeros(8, dtype=torch.float32, device=device)
        # data parallel: a batch that every rank runs in full (dp.replicated(): tail batches, critic draws smaller than
        # the world) contributes its FULL value on every rank -- kept apart from the per-rank shares, which are summed
        # over the ranks when they are read (_sums)
        if ops.dp.is_replicated() and ops.dp.ranks() > 1:
            return self._running_repl
        return self._running

    def init_loss(self):
        if self._running is not None:
            self._running.zero_()
            self._running_repl.zero_()
        return self.loss()

    def gen_loss(self, true, pred, DGx, presum=None):
        """`presum` (data parallel, optional): recon_presum(true, pred) started earlier, so the all-reduce of
        the batch-global gradient-penalty norm runs under the discriminator pass."""
        total, _ = ops.gen_loss(true, pred, DGx, self.l1_weight, self.l2_weight, self.gp_weight,
                                self._run(pred.device), presum)
        return total

    def recon_presum(self, true, pred):
        return ops.recon_presum(true, pred) if ops.dp.active() else None

    def dis_loss(self, true, pred, Dy, DGx, critic: bool = False):
        if critic:
            total, _ = ops.dis_loss(Dy, DGx, true, pred, True, self.gp_weight, None)
        else:
            total, _ = ops.dis_loss(Dy, DGx, None, None, False, 0.0, self._run(Dy.device))
        return total

    def _sums(self):
        if self._running is None:
            return [0.0] * 8
        run = self._running
        if ops.dp.active():
            # data parallel: the adversarial terms (real, fake, G_) were accumulated as this rank's SHARE of the
            # global means -- sum them over the ranks; l1 / l2 / gp come from globally reduced sums already
            run = run.clone()
            ops.dp.all_reduce(run[0:3])
        if ops.dp.ranks() > 1:
            run = run + self._running_repl
        return run.tolist()      # the one host sync, at logging time

    @property
    def errD_real(self):
        return self._sums()[0]

    @property
    def errD_fake(self):
        return self._sums()[1]

    def loss(self, i: int = 0):
        r, f, g_, l1, l2, gp = self._sums()[:6]
        vals = {"D": r + f, "R": r, "F": f, "G": g_ + l1 + l2 + gp, "G_": g_, "L1": l1, "L2": l2, "GP": gp}
        return {k: f"{v / (i + 1):6.3f}" for k, v in vals.items()}


class TimebandLoss(TIMEBANDLoss):
    """Gen-B constructor signature (reference source/losses.py:57); dis_loss(Dy, DGx) has no critic form."""

    def __init__(self, l1_weights, l2_weights, gp_weights):
        super().__init__(dict(l1_weight=l1_weights, l2_weight=l2_weights, gp_weight=gp_weights))

    def dis_loss(self, *args, **kw):
        if len(args) == 2 and not kw:
            Dy, DGx = args
            return super().dis_loss(None, None, Dy, DGx)
        return super().dis_loss(*args, **kw)

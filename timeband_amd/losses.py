"""Gen-B losses (reference source/losses.py): `GANLoss` (:7-53) and `TimebandLoss` (:56-145) on the
fused HIP reductions of csrc/losses.hip.  Gen-B's dis_loss(Dy, DGx) has no critic form: its critic
step (core.py:72-89) uses the same LSGAN loss."""
import torch

from . import _lib, ops
from .loss import TimebandLoss  # noqa: F401  (constructor l1_weights, l2_weights, gp_weights)


class _GanLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, real, wgan):
        ops._need_cuda(p)
        p = ops._f32c(p)
        rt = ops.Runtime.get(p.device)
        sums = ops._loss_sums(p if real else None, None if real else p, None, None, rt)   # all-reduced under DP
        n = ops.dp.count_scale(p.numel())
        if wgan:
            val = -sums[2] / n if real else sums[3] / n
        else:
            val = sums[0 if real else 1] / n
        ctx.args = (real, wgan, n)
        ctx.save_for_backward(p)
        return val.to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        (p,) = ctx.saved_tensors
        real, wgan, n = ctx.args
        dp = torch.empty_like(p)
        up = ops._f32c(g)
        _lib.call("tb_gan_loss_grad", ops._ptr(p), p.numel(), n, 1 if real else 0, 1 if wgan else 0, ops._ptr(up),
                  ops._ptr(dp), ops._stream())
        return dp, None, None


class GANLoss(torch.nn.Module):
    """lsgan / vanilla: mean((p - label)^2), labels 0.9 / 0.1; wgangp: -+mean(p) (losses.py:36-53)."""

    def __init__(self, gan_mode="vanilla", real_label=0.9, fake_label=0.1):
        super().__init__()
        if gan_mode not in ("lsgan", "vanilla", "wgangp"):
            raise NotImplementedError(f"gan mode {gan_mode} not implemented")
        if gan_mode != "wgangp" and (abs(real_label - 0.9) > 1e-12 or abs(fake_label - 0.1) > 1e-12):
            raise NotImplementedError("the fused reduction has the reference's labels 0.9 / 0.1 built in")
        self.gan_mode = gan_mode
        self.register_buffer("real_label", torch.tensor(real_label))
        self.register_buffer("fake_label", torch.tensor(fake_label))

    def get_target_tensor(self, prediction, target_is_real):
        return (self.real_label if target_is_real else self.fake_label).expand_as(prediction)

    def forward(self, prediction, target_is_real):
        return _GanLossFn.apply(prediction, bool(target_is_real), self.gan_mode == "wgangp")

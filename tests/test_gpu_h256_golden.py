"""GPU: the clustered hidden-256 kernels against a RUN OF THE REFERENCE (not only against the oracle): cfg2's shape
(32 variables + 4 time features, window 128, hidden 256) and model.cfg's observed 10 / forecast 3 at hidden 256.
Fixtures: tests/golden/scenario_h256.npz, scenario_uneven_h256.npz (tests/golden/make_golden.py run_scenario_big);
inputs and initial weights are regenerated from the seed and verified by checksum (tests/h256_util.py).
Tolerances (north_star): forward <= 1e-4 relative (max-norm), gradients <= 1e-3 -- max-norm per tensor AND
element-wise with an absolute floor (h256_util.grads_close)."""
import numpy as np
import pytest
import torch

import h256_util as U

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda", 0)


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def grads(net):
    return {k: p.grad.detach().cpu().numpy() for k, p in net.named_parameters()}


def params(net):
    return {k: v.detach().cpu().numpy() for k, v in net.state_dict().items()}


@pytest.mark.parametrize("name", U.NAMES)
def test_forward_h256_vs_reference(dev, name):
    z = U.load(name)
    netD, netG, x, y, xc, yc = U.regenerate(z)
    netD, netG = netD.to(dev), netG.to(dev)
    with torch.no_grad():
        Gx = netG(x.to(dev))
        Dy = netD(y.to(dev))
        DGx = netD(Gx)
    assert rel(Gx.cpu().numpy(), z["fwd.Gx"]) < 1e-4
    assert rel(Dy.cpu().numpy(), z["fwd.Dy"]) < 1e-4
    assert rel(DGx.cpu().numpy(), z["fwd.DGx"]) < 1e-4
    # element-wise too: outputs are O(0.1 .. 1) (tanh / sigmoid), floor 1e-6 absolute
    for got, want in ((Gx, z["fwd.Gx"]), (Dy, z["fwd.Dy"]), (DGx, z["fwd.DGx"])):
        d = np.abs(got.cpu().numpy().astype(np.float64) - want)
        assert (d <= np.maximum(1e-4 * np.abs(want), 1e-6)).all()


@pytest.mark.parametrize("pair", [False, True])
@pytest.mark.parametrize("name", U.NAMES)
def test_iteration_h256_vs_reference(dev, name, pair):
    """trainer.py:142-194 once (critic=1) with the fused optimizers; pair=True runs netD(real), netD(fake) as the
    product's trainer does -- one 2B-row pass (LSTMDiscriminator.pair)."""
    from timeband_amd import ops
    from timeband_amd.loss import TIMEBANDLoss
    from timeband_amd.optim import FusedAdam, FusedRMSprop
    z = U.load(name)
    lr = float(z["lr"])
    netD, netG, x, y, xc, yc = U.regenerate(z)
    netD, netG = netD.to(dev), netG.to(dev)
    x, y, xc, yc = (t.to(dev) for t in (x, y, xc, yc))
    losses = TIMEBANDLoss(dict(l1_weight=50, l2_weight=50, gp_weight=50, device=dev))
    adamD, adamG = FusedAdam(netD, lr), FusedAdam(netG, lr)
    both = (lambda r, f: netD.pair(r, f)) if pair else (lambda r, f: (netD(r), netD(f)))
    optD = FusedRMSprop(netD, lr)
    fake = netG(xc)
    optD.zero_grad()
    errD = losses.dis_loss(yc, fake, *both(yc, fake), critic=True)
    errD.backward()
    assert rel(errD.item(), z["it0.critic.errD"]) < 1e-4
    U.grads_close(grads(netD), z, "it0.critic.grad.netD", 1e-3)
    optD.step()
    U.params_close(params(netD), z, "it0.critic.after.netD", 12 * lr)     # fresh RMSprop: |step| = lr / sqrt(1 - alpha) = 10 lr
    adamD.zero_grad()
    adamG.zero_grad()
    fake = netG(x)
    errD = losses.dis_loss(y, fake, *both(y, fake))
    errD.backward()
    assert rel(errD.item(), z["it0.dphase.errD"]) < 1e-4
    U.grads_close(grads(netD), z, "it0.dphase.grad.netD", 1e-3)
    U.grads_close(grads(netG), z, "it0.dphase.grad.netG", 1e-3)
    adamD.step()
    U.params_close(params(netD), z, "it0.dphase.after.netD", lr)
    fake = netG(x)
    errG = losses.gen_loss(y, fake, netD(fake))
    errG.backward()
    assert rel(errG.item(), z["it0.gphase.errG"]) < 1e-4
    U.grads_close(grads(netG), z, "it0.gphase.grad.netG", 1e-3)
    adamG.step()
    U.params_close(params(netG), z, "it0.gphase.after.netG", lr)
    assert rel(fake.detach().cpu().numpy(), z["it0.fake_y"]) < 1e-4
    ops.check_errors(dev)

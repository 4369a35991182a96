"""Shared by the hidden-256 golden tests (tests/golden/scenario_h256.npz, scenario_uneven_h256.npz; written by
tests/golden/make_golden.py `run_scenario_big` from a run of the reference).  Those files hold checksums of the
initial weights and of the inputs instead of the arrays: both are regenerated here from the seed -- the product's
networks are created in the reference's RNG order (bit-identical init, test_init_matches_reference_rng_order), and the
inputs are the same four torch.rand calls -- and checked against the checksums.  Tensors above 16384 elements were kept
as a seeded sample of 4096 entries."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = ["h256", "uneven_h256"]


def load(name):
    return np.load(os.path.join(GOLDEN, f"scenario_{name}.npz"), allow_pickle=False)


def sample_index(numel, z):
    _, n, seed = (int(v) for v in z["sampling"])
    return np.sort(np.random.default_rng(seed + numel % 9973).choice(numel, n, replace=False))


def pack(a, z):
    a = np.asarray(a)
    return a if a.size <= int(z["sampling"][0]) else a.reshape(-1)[sample_index(a.size, z)]


def checksum(a):
    a = np.asarray(a, np.float64).reshape(-1)
    return np.array([a.sum(), (a * a).sum(), (a * np.arange(1, a.size + 1) % 7).sum()])


def regenerate(z):
    """(netD, netG, x, y, xc, yc) on the CPU, verified against the fixture's checksums."""
    from timeband_amd.utils.initiate import seeding
    from timeband_amd.utils.lstm_layer import LSTMDiscriminator, LSTMGenerator
    V, E, To, Tf, H, L, B = (int(v) for v in z["meta"])
    seeding(31)
    netD = LSTMDiscriminator(V, H, L)          # D first (model.py:56-57)
    netG = LSTMGenerator(E, V, To, Tf, H, L)
    x = torch.rand(B, To, E) * 2 - 1
    y = torch.rand(B, Tf, V) * 2 - 1
    xc = torch.rand(Tf, To, E) * 2 - 1
    yc = torch.rand(Tf, Tf, V) * 2 - 1
    for pre, net in (("init.netD", netD), ("init.netG", netG)):
        for k, v in net.state_dict().items():
            assert np.array_equal(checksum(v.numpy()), z[f"{pre}.{k}.checksum"]), f"{pre}.{k}: init differs from the reference's"
    for k, v in (("x", x), ("y", y), ("xc", xc), ("yc", yc)):
        assert np.array_equal(checksum(v.numpy()), z[f"{k}.checksum"]), f"input {k} differs from the reference run's"
    return netD, netG, x, y, xc, yc


def sub(z, prefix):
    prefix = prefix + "."
    return {k[len(prefix):]: z[k] for k in z.files if k.startswith(prefix) and not k.endswith(".absmax")
            and not k.endswith(".checksum") and not k.endswith(".partmax")}


def grads_close(got, z, tag, tol, floor=2e-7):
    """got: name -> full gradient array.  Max-norm bound per tensor (tol x the tensor's own largest entry -- stored in
    full precision even where only a sample was kept -- plus 1 % of tol x the network's largest gradient, plus
    `floor` for the reference's own fp32 rounding of cancelling O(1) sums) AND an element-wise relative check: every
    kept entry must agree to tol relative or to that same absolute bound, whichever is larger."""
    want = sub(z, tag)
    gmax = max(float(z[f"{tag}.{k}.absmax"]) for k in want)
    for k, w in want.items():
        g = pack(np.asarray(got[k], np.float64), z)
        assert g.shape == w.shape, (tag, k, g.shape, w.shape)
        amax = float(z[f"{tag}.{k}.absmax"])
        if f"{tag}.{k}.partmax" in z.files:
            # the WGAN critic: a residue of two cancelling sums -- tol applies to the size of the parts (recorded by
            # the fixture's script from the reference's own modules)
            amax = max(amax, float(z[f"{tag}.{k}.partmax"]))
        bound = tol * amax + 0.01 * tol * gmax + floor
        err = np.abs(g - w)
        assert err.max() <= bound, f"{tag}:{k} max err {err.max():.3e} > {bound:.3e}"
        elem = err <= np.maximum(tol * np.abs(w), bound)
        assert elem.all(), f"{tag}:{k} {int((~elem).sum())} entries off element-wise"


def params_close(got, z, tag, step):
    """After a first Adam / RMSprop step every element moves by ~lr * sign(g): ill-conditioned where |g| ~ eps, so a
    tiny fraction may differ by up to two steps; the rest agrees to 5 % of a step."""
    want = sub(z, tag)
    nbad = ntot = 0
    for k, w in want.items():
        d = np.abs(pack(np.asarray(got[k], np.float64), z) - w)
        assert d.max() <= 2.1 * step, (tag, k, d.max())
        nbad += int((d > 0.05 * step).sum())
        ntot += d.size
    assert nbad / ntot < 2e-3, (tag, nbad, ntot)

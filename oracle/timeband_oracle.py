"""CPU oracle: a plain-numpy restatement of Timeband's LSTM-GAN hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under timeband_amd/ imports this file; only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may.  It is the checker the HIP
path is compared with, never the thing shipped or measured.

Parity status: PINNED.  Every function below is checked against golden vectors captured
by running the reference's own Gen-A modules in the build container
(tests/golden/make_golden.py -> tests/golden/*.npz; tests/test_oracle_golden.py).  The
reference ships no tests or fixtures of its own, and its arithmetic lives
in a third-party dependency (PyTorch: nn.LSTM, nn.Linear, MSELoss, SmoothL1Loss, Adam,
RMSprop; pinned upstream at torch==1.9.1, executed here on torch 2.10.0 CPU), whose
published semantics are restated here.

Each function cites the reference file:line it follows (paths relative to /root/reference).
All network arithmetic is float32 like the reference; post-processing is float64 like the
reference's numpy code.
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


# --------------------------------------------------------------------------------------
# activations
# --------------------------------------------------------------------------------------
def sigmoid(x):
    x = np.asarray(x)
    out = np.empty_like(x)
    pos = x >= 0
    out[pos] = 1.0 / (1.0 + np.exp(-x[pos]))
    ex = np.exp(x[~pos])
    out[~pos] = ex / (1.0 + ex)
    return out


# --------------------------------------------------------------------------------------
# LSTM (torch.nn.LSTM semantics: gate order i, f, g, o; two biases; batch_first;
# zero initial state).  Call sites: source/utils/lstm_layer.py:31-33,44-46,75,83
# --------------------------------------------------------------------------------------
def lstm_layer_fwd(x, w_ih, w_hh, b_ih, b_hh):
    """x (B,T,in) -> y (B,T,h) and the cache BPTT needs."""
    B, T, _ = x.shape
    h4 = w_ih.shape[0]
    h = h4 // 4
    dt = x.dtype
    proj = (x.reshape(B * T, -1) @ w_ih.T + (b_ih + b_hh)).reshape(B, T, h4)
    y = np.empty((B, T, h), dt)
    gates = np.empty((B, T, h4), dt)     # activated i, f, g, o
    cs = np.empty((B, T, h), dt)
    hp = np.zeros((B, h), dt)
    cp = np.zeros((B, h), dt)
    for t in range(T):
        a = proj[:, t] + hp @ w_hh.T
        i = sigmoid(a[:, :h])
        f = sigmoid(a[:, h:2 * h])
        g = np.tanh(a[:, 2 * h:3 * h])
        o = sigmoid(a[:, 3 * h:])
        c = f * cp + i * g
        hh = o * np.tanh(c)
        gates[:, t, :h], gates[:, t, h:2 * h], gates[:, t, 2 * h:3 * h], gates[:, t, 3 * h:] = i, f, g, o
        cs[:, t] = c
        y[:, t] = hh
        hp, cp = hh, c
    return y, (x, w_ih, w_hh, gates, cs, y)


def lstm_layer_bwd(dy, cache, need_dx=True):
    x, w_ih, w_hh, gates, cs, y = cache
    B, T, h = y.shape
    dt = x.dtype
    dG = np.empty((B, T, 4 * h), dt)     # grads wrt pre-activations
    dh_rec = np.zeros((B, h), dt)
    dc = np.zeros((B, h), dt)
    for t in range(T - 1, -1, -1):
        i, f = gates[:, t, :h], gates[:, t, h:2 * h]
        g, o = gates[:, t, 2 * h:3 * h], gates[:, t, 3 * h:]
        c = cs[:, t]
        cprev = cs[:, t - 1] if t > 0 else np.zeros((B, h), dt)
        tc = np.tanh(c)
        dh = dy[:, t] + dh_rec
        do = dh * tc
        dc = dc + dh * o * (1 - tc * tc)
        di = dc * g
        dg = dc * i
        df = dc * cprev
        dG[:, t, :h] = di * i * (1 - i)
        dG[:, t, h:2 * h] = df * f * (1 - f)
        dG[:, t, 2 * h:3 * h] = dg * (1 - g * g)
        dG[:, t, 3 * h:] = do * o * (1 - o)
        dc = dc * f
        dh_rec = dG[:, t] @ w_hh
    dG2 = dG.reshape(B * T, 4 * h)
    dw_ih = dG2.T @ x.reshape(B * T, -1)
    hprev = np.concatenate([np.zeros((B, 1, h), dt), y[:, :-1]], axis=1).reshape(B * T, h)
    dw_hh = dG2.T @ hprev
    db = dG2.sum(axis=0)
    dx = (dG2 @ w_ih).reshape(x.shape) if need_dx else None
    return dx, dw_ih, dw_hh, db, db.copy()


def lstm_fwd(params, prefix, x, n_layers):
    """nn.LSTM(in, h, n_layers): layer l consumes layer l-1's output."""
    caches = []
    for l in range(n_layers):
        x, c = lstm_layer_fwd(x, params[f"{prefix}.weight_ih_l{l}"], params[f"{prefix}.weight_hh_l{l}"],
                              params[f"{prefix}.bias_ih_l{l}"], params[f"{prefix}.bias_hh_l{l}"])
        caches.append(c)
    return x, caches


def lstm_bwd(grads, prefix, dy, caches, need_dx=True):
    for l in range(len(caches) - 1, -1, -1):
        dy, dwi, dwh, dbi, dbh = lstm_layer_bwd(dy, caches[l], need_dx or l > 0)
        _acc(grads, f"{prefix}.weight_ih_l{l}", dwi)
        _acc(grads, f"{prefix}.weight_hh_l{l}", dwh)
        _acc(grads, f"{prefix}.bias_ih_l{l}", dbi)
        _acc(grads, f"{prefix}.bias_hh_l{l}", dbh)
    return dy


def _acc(grads, key, val):
    grads[key] = grads[key] + val if key in grads else val


# --------------------------------------------------------------------------------------
# Generator / Discriminator.  source/utils/lstm_layer.py:38-52 and :78-89
# --------------------------------------------------------------------------------------
def generator_fwd(p, x, n_layers=1):
    """lstm0 -> lstm1 -> lstm2 -> Linear(To->Tf) on view(B,H,To) -> Linear(H->V)+tanh on view(B,Tf,H).
    The two `.view`s REINTERPRET memory (lstm_layer.py:49-50); numpy reshape of a contiguous
    array is the same thing."""
    B, To, _ = x.shape
    r, c0 = lstm_fwd(p, "lstm0", x, n_layers)
    r, c1 = lstm_fwd(p, "lstm1", r, n_layers)
    r, c2 = lstm_fwd(p, "lstm2", r, n_layers)
    H = r.shape[2]
    Wseq, bseq = p["seq_layer.0.weight"], p["seq_layer.0.bias"]
    Wdim, bdim = p["dim_layer.0.weight"], p["dim_layer.0.bias"]
    Tf = Wseq.shape[0]
    v1 = np.ascontiguousarray(r).reshape(B, H, To)
    s = v1 @ Wseq.T + bseq                                  # (B,H,Tf)
    v2 = np.ascontiguousarray(s).reshape(B, Tf, H)
    out = np.tanh(v2 @ Wdim.T + bdim)                       # (B,Tf,V)
    return out, (c0, c1, c2, v1, v2, out, Wseq, Wdim)


def generator_bwd(grads, dout, cache):
    c0, c1, c2, v1, v2, out, Wseq, Wdim = cache
    B, Tf, V = out.shape
    H, To = v1.shape[1], v1.shape[2]
    dz = dout * (1 - out * out)
    _acc(grads, "dim_layer.0.weight", dz.reshape(-1, V).T @ v2.reshape(-1, H))
    _acc(grads, "dim_layer.0.bias", dz.reshape(-1, V).sum(0))
    ds = (dz @ Wdim).reshape(B, H, Tf)
    _acc(grads, "seq_layer.0.weight", ds.reshape(-1, Tf).T @ v1.reshape(-1, To))
    _acc(grads, "seq_layer.0.bias", ds.reshape(-1, Tf).sum(0))
    dr = (ds @ Wseq).reshape(B, To, H)
    dr = lstm_bwd(grads, "lstm2", dr, c2)
    dr = lstm_bwd(grads, "lstm1", dr, c1)
    lstm_bwd(grads, "lstm0", dr, c0, need_dx=False)


def discriminator_fwd(p, yin, n_layers=1):
    B, T, _ = yin.shape
    r, c = lstm_fwd(p, "lstm", yin, n_layers)
    H = r.shape[2]
    W, b = p["linear.0.weight"], p["linear.0.bias"]
    out = sigmoid(r.reshape(B * T, H) @ W.T + b).reshape(B, T, 1)
    return out, (c, r, out, W)


def discriminator_bwd(grads, dout, cache, need_dx=True):
    c, r, out, W = cache
    B, T, H = r.shape
    dz = (dout * out * (1 - out)).reshape(B * T, 1)
    _acc(grads, "linear.0.weight", dz.T @ r.reshape(B * T, H))
    _acc(grads, "linear.0.bias", dz.sum(0))
    dr = (dz @ W).reshape(B, T, H)
    return lstm_bwd(grads, "lstm", dr, c, need_dx)


# --------------------------------------------------------------------------------------
# Losses with analytic gradients.  source/loss.py
# --------------------------------------------------------------------------------------
REAL_LABEL, FAKE_LABEL = F32(0.9), F32(0.1)     # loss.py:63-64


def gan_loss(p, target_is_real, mode="lsgan", n_global=None):
    """loss.py:34-51.  Returns (loss, dloss/dp).  n_global: element count of the GLOBAL batch when
    `p` is one data-parallel shard (the returned loss is then this shard's share of the mean)."""
    n = F32(p.size if n_global is None else n_global)
    if mode in ("lsgan", "vanilla"):
        t = REAL_LABEL if target_is_real else FAKE_LABEL
        d = p - t
        return F32(np.sum(d * d, dtype=np.float64) / n), (2 * d / n).astype(p.dtype)
    sgn = F32(-1.0) if target_is_real else F32(1.0)
    return F32(sgn * np.sum(p, dtype=np.float64) / n), np.full(p.shape, sgn / n, p.dtype)


def recon_sums(true, pred):
    """(sum SmoothL1(d), sum d^2) of one shard, d = pred - true; what data-parallel ranks all-reduce."""
    d = (pred - true).astype(pred.dtype)
    ad = np.abs(d)
    hub = np.where(ad < 1, F32(0.5) * d * d, ad - F32(0.5))
    return np.sum(hub, dtype=np.float64), np.sum(d.astype(np.float64) ** 2)


def recon_terms(true, pred, l1_w, l2_w, gp_w, sums=None, n_global=None, batch_global=None):
    """loss.py:127-141: l1_w*SmoothL1 + l2_w*MSE + gp_w*(1-||true-pred||_2)^2/B and the
    gradient of their sum wrt pred.  sums / n_global / batch_global: the GLOBAL (all-reduced)
    sums and counts when (true, pred) is one data-parallel shard."""
    d = (pred - true).astype(pred.dtype)
    N = F32(d.size if n_global is None else n_global)
    B = F32(true.shape[0] if batch_global is None else batch_global)
    Sh, S = recon_sums(true, pred) if sums is None else sums
    l1 = F32(l1_w) * F32(Sh / N)
    l2 = F32(l2_w) * F32(S / N)
    n = np.sqrt(S)
    gp = F32(gp_w) * F32((1 - n) ** 2 / B)
    coef_gp = -2.0 * gp_w * (1 - n) / (float(B) * n) if n > 0 else 0.0
    dpred = (F32(l1_w) * np.clip(d, -1, 1) / N + F32(l2_w) * 2 * d / N + F32(coef_gp) * d).astype(pred.dtype)
    dpred_gp_only = (F32(coef_gp) * d).astype(pred.dtype)
    return l1, l2, gp, dpred, dpred_gp_only


class Losses:
    """Running sums + loss/grad evaluation; mirrors TIMEBANDLoss (loss.py:54-157)."""

    def __init__(self, l1_weight=50, l2_weight=50, gp_weight=50):
        self.l1_w, self.l2_w, self.gp_w = l1_weight, l2_weight, gp_weight
        self.init_loss()

    def init_loss(self):
        self.errD_real = self.errD_fake = 0.0
        self.err_Gen = self.errG_l1 = self.errG_l2 = self.errG_GP = 0.0
        return self.loss()

    def dis_loss(self, true, pred, Dy, DGx, critic=False):
        """loss.py:106-119.  Returns loss, dDy, dDGx, dpred (dpred only in critic mode)."""
        if critic:
            lr_, dDy = gan_loss(Dy, True, "wgangp")
            lf_, dDGx = gan_loss(DGx, False, "wgangp")
            _, _, gp, _, dpred = recon_terms(true, pred, 0, 0, self.gp_w)
            return F32(lf_ + lr_ + gp), dDy, dDGx, dpred
        lr_, dDy = gan_loss(Dy, True)
        lf_, dDGx = gan_loss(DGx, False)
        self.errD_real += float(lr_)
        self.errD_fake += float(lf_)
        return F32(lf_ + lr_), dDy, dDGx, None

    def gen_loss(self, true, pred, DGx):
        """loss.py:93-104: the adversarial term uses the FAKE label (loss.py:94)."""
        lg, dDGx = gan_loss(DGx, False)
        l1, l2, gp, dpred, _ = recon_terms(true, pred, self.l1_w, self.l2_w, self.gp_w)
        self.err_Gen += float(lg)
        self.errG_l1 += float(l1)
        self.errG_l2 += float(l2)
        self.errG_GP += float(gp)
        return F32(lg + l1 + l2 + gp), dDGx, dpred

    def loss(self, i=0):
        """loss.py:143-157."""
        errG = self.err_Gen + self.errG_l1 + self.errG_l2 + self.errG_GP
        errD = self.errD_fake + self.errD_real
        vals = dict(D=errD, R=self.errD_real, F=self.errD_fake, G=errG, G_=self.err_Gen,
                    L1=self.errG_l1, L2=self.errG_l2, GP=self.errG_GP)
        return {k: f"{v / (i + 1):6.3f}" for k, v in vals.items()}


# --------------------------------------------------------------------------------------
# Optimizers (torch.optim defaults).  Schedule: source/trainer.py:95-98,143-144
# --------------------------------------------------------------------------------------
class Adam:
    """torch.optim.Adam(lr, betas=(.9,.999), eps=1e-8), no weight decay, no amsgrad."""

    def __init__(self, lr, b1=0.9, b2=0.999, eps=1e-8):
        self.lr, self.b1, self.b2, self.eps = lr, b1, b2, eps
        self.t = 0
        self.m, self.v = {}, {}

    def step(self, params, grads):
        self.t += 1
        bc1 = 1 - self.b1 ** self.t
        bc2 = 1 - self.b2 ** self.t
        step_size = self.lr / bc1
        for k, g in grads.items():
            if k not in self.m:
                self.m[k] = np.zeros_like(params[k])
                self.v[k] = np.zeros_like(params[k])
            m = self.m[k] = (self.m[k] + (g - self.m[k]) * F32(1 - self.b1)).astype(F32)
            v = self.v[k] = (self.v[k] * F32(self.b2) + F32(1 - self.b2) * g * g).astype(F32)
            denom = (np.sqrt(v) / F32(np.sqrt(bc2)) + F32(self.eps)).astype(F32)
            params[k] = (params[k] - F32(step_size) * (m / denom)).astype(F32)


class RMSprop:
    """torch.optim.RMSprop(lr, alpha=.99, eps=1e-8), no momentum, not centered."""

    def __init__(self, lr, alpha=0.99, eps=1e-8):
        self.lr, self.alpha, self.eps = lr, alpha, eps
        self.sq = {}

    def step(self, params, grads):
        for k, g in grads.items():
            if k not in self.sq:
                self.sq[k] = np.zeros_like(params[k])
            sq = self.sq[k] = (self.sq[k] * F32(self.alpha) + F32(1 - self.alpha) * g * g).astype(F32)
            avg = (np.sqrt(sq) + F32(self.eps)).astype(F32)
            params[k] = (params[k] - F32(self.lr) * (g / avg)).astype(F32)


# --------------------------------------------------------------------------------------
# One Gen-A training iteration.  source/trainer.py:142-194
# --------------------------------------------------------------------------------------
class TrainState:
    def __init__(self, pD, pG, n_layers=1, lr=2e-4, gammaD=1.0, gammaG=1.0, losses=None, critic=1):
        self.pD = {k: np.array(v, F32) for k, v in pD.items()}
        self.pG = {k: np.array(v, F32) for k, v in pG.items()}
        self.L = n_layers
        self.lr, self.gammaD, self.gammaG = lr, gammaD, gammaG
        self.losses = losses or Losses()
        self.critic = critic
        self.new_epoch()

    def new_epoch(self):
        """trainer.py:95-98: Adam for D and G is re-created (moments reset) every epoch."""
        self.adamD = Adam(self.lr * self.gammaD)
        self.adamG = Adam(self.lr * self.gammaG)


def critic_step(st, xc, yc, rec=None):
    """trainer.py:142-158: fresh RMSprop per batch; WGAN + GP loss; fake_y NOT detached
    (G's grads from this backward are discarded by the zero_grad at trainer.py:161)."""
    opt = RMSprop(st.lr * st.gammaD)
    for _ in range(st.critic):
        fake, _ = generator_fwd(st.pG, xc, st.L)
        Dy, cDy = discriminator_fwd(st.pD, yc, st.L)
        DGx, cDG = discriminator_fwd(st.pD, fake, st.L)
        errD, dDy, dDGx, _ = st.losses.dis_loss(yc, fake, Dy, DGx, critic=True)
        gD = {}
        discriminator_bwd(gD, dDy, cDy, need_dx=False)
        discriminator_bwd(gD, dDGx, cDG, need_dx=False)
        if rec is not None:
            rec["critic.errD"] = errD
            rec["critic.grad.netD"] = {k: v.copy() for k, v in gD.items()}
        opt.step(st.pD, gD)
    return errD


def train_iteration(st, x, y, xc=None, yc=None, training=True, rec=None):
    """One pass of trainer.py:142-194.  Returns fake_y of the G phase."""
    if training and xc is not None:
        critic_step(st, xc, yc, rec)
    # D phase (trainer.py:173-180): fake_y is not detached -> errD.backward() also fills G grads
    fake, cG = generator_fwd(st.pG, x, st.L)
    Dy, cDy = discriminator_fwd(st.pD, y, st.L)
    DGx, cDG = discriminator_fwd(st.pD, fake, st.L)
    errD, dDy, dDGx, _ = st.losses.dis_loss(y, fake, Dy, DGx)
    gD, gG = {}, {}
    if training:
        discriminator_bwd(gD, dDy, cDy, need_dx=False)
        dfake = discriminator_bwd(gD, dDGx, cDG, need_dx=True)
        generator_bwd(gG, dfake, cG)
        if rec is not None:
            rec["dphase.errD"] = errD
            rec["dphase.grad.netD"] = {k: v.copy() for k, v in gD.items()}
            rec["dphase.grad.netG"] = {k: v.copy() for k, v in gG.items()}
        st.adamD.step(st.pD, gD)
    # G phase (trainer.py:188-194): G grads ACCUMULATE on the D-phase grads (zero_grad is at :160-161)
    fake, cG = generator_fwd(st.pG, x, st.L)
    DGx, cDG = discriminator_fwd(st.pD, fake, st.L)
    errG, dDGx, dpred = st.losses.gen_loss(y, fake, DGx)
    if training:
        scratch = {}
        dfake = discriminator_bwd(scratch, dDGx, cDG, need_dx=True)
        generator_bwd(gG, (dfake + dpred).astype(F32), cG)
        if rec is not None:
            rec["gphase.errG"] = errG
            rec["gphase.grad.netG"] = {k: v.copy() for k, v in gG.items()}
        st.adamG.step(st.pG, gG)
    return fake


# --------------------------------------------------------------------------------------
# Gen-B schedule.  source/core.py:64-130 with source/losses.py:86-105
# --------------------------------------------------------------------------------------
class GenBState:
    """core.py:64-70: Adam(D), Adam(G) and ONE RMSprop over D that persists across critic calls."""

    def __init__(self, pD, pG, n_layers=1, lr_D=2e-4, lr_G=2e-4, losses=None):
        self.pD = {k: np.array(v, F32) for k, v in pD.items()}
        self.pG = {k: np.array(v, F32) for k, v in pG.items()}
        self.L = n_layers
        self.losses = losses or Losses()
        self.adamD, self.adamG, self.rmsC = Adam(lr_D), Adam(lr_G), RMSprop(lr_D)


def genb_critic_step(st, xc, yc, rec=None):
    """core.py:74-89: LSGAN dis_loss (losses.py:99-105), one step of the persistent RMSprop."""
    fake, _ = generator_fwd(st.pG, xc, st.L)
    Dy, cDy = discriminator_fwd(st.pD, yc, st.L)
    DGx, cDG = discriminator_fwd(st.pD, fake, st.L)
    errC, dDy, dDGx, _ = st.losses.dis_loss(None, None, Dy, DGx)
    gD = {}
    discriminator_bwd(gD, dDy, cDy, need_dx=False)
    discriminator_bwd(gD, dDGx, cDG, need_dx=False)
    if rec is not None:
        rec["errC"] = errC
        rec["grad.netD"] = {k: v.copy() for k, v in gD.items()}
    st.rmsC.step(st.pD, gD)
    return errC


def genb_iteration(st, x, y, training=True, rec=None):
    """core.py:98-130: D step, then a G step whose gradients start from zero (optimG.zero_grad at
    :122 -- unlike Gen-A, the D-phase pass through G does not reach the G update)."""
    fake, _ = generator_fwd(st.pG, x, st.L)
    Dy, cDy = discriminator_fwd(st.pD, y, st.L)
    DGx, cDG = discriminator_fwd(st.pD, fake, st.L)
    errD, dDy, dDGx, _ = st.losses.dis_loss(None, None, Dy, DGx)
    if training:
        gD = {}
        discriminator_bwd(gD, dDy, cDy, need_dx=False)
        discriminator_bwd(gD, dDGx, cDG, need_dx=False)
        if rec is not None:
            rec["errD"] = errD
            rec["dphase.grad.netD"] = {k: v.copy() for k, v in gD.items()}
        st.adamD.step(st.pD, gD)
    fake, cG = generator_fwd(st.pG, x, st.L)
    DGx, cDG = discriminator_fwd(st.pD, fake, st.L)
    errG, dDGx, dpred = st.losses.gen_loss(y, fake, DGx)
    if training:
        gG, scratch = {}, {}
        dfake = discriminator_bwd(scratch, dDGx, cDG, need_dx=True)
        generator_bwd(gG, (dfake + dpred).astype(F32), cG)
        if rec is not None:
            rec["errG"] = errG
            rec["gphase.grad.netG"] = {k: v.copy() for k, v in gG.items()}
        st.adamG.step(st.pG, gG)
    return fake


# --------------------------------------------------------------------------------------
# Dataset: normalise / windows / denormalise.  source/dataset.py
# --------------------------------------------------------------------------------------
def interpolate(a):
    """dataset.py:111-115: forward fill, then backward fill, then 0."""
    a = np.array(a, np.float64)
    n = a.shape[0]
    for j in range(a.shape[1]):
        col = a[:, j]
        idx = np.where(~np.isnan(col), np.arange(n), -1)
        np.maximum.accumulate(idx, out=idx)
        good = idx >= 0
        col[good] = col[idx[good]]
        nz = np.where(~np.isnan(col))[0]
        if nz.size:
            col[:nz[0]] = col[nz[0]]
        col[np.isnan(col)] = 0.0
    return a


def minmax_info(data, split_rate):
    """dataset.py:297-321: min/max over the first int(len*split_rate) rows; -inf min -> max;
    min == max -> max + 1."""
    k = int(len(data) * split_rate)
    mn = np.nanmin(data[:k], axis=0).astype(np.float64)
    mx = np.nanmax(data[:k], axis=0).astype(np.float64)
    mn = np.where(mn == -np.inf, mx, mn)
    mx = np.where(mn == mx, mx + 1, mx)
    return mn, mx


def normalize(data, mn, mx):
    """dataset.py:216-227."""
    return 2 * ((data - mn) / (mx - mn)) - 1


def denormalize(pred, dmin, dmax):
    """dataset.py:229-242: fp32 0.5*(x+1), times fp64 delta, plus fp64 min, stored back into
    the fp32 tensor (so the result is rounded to fp32)."""
    delta = (np.asarray(dmax, np.float64) - np.asarray(dmin, np.float64))
    half = (F32(0.5) * (np.asarray(pred, F32) + F32(1))).astype(F32)
    return (half.astype(np.float64) * delta + np.asarray(dmin, np.float64)).astype(F32)


def window_starts(data_len, To, Tf, stride=1):
    """dataset.py:199-214: i in range(To, stop+1, stride), stop = data_len - Tf; window w reads
    rows [i-To, i) of all columns and rows [i, i+Tf) of the target columns."""
    stop = data_len - Tf
    return np.arange(To, stop + 1, stride)


def windowing(norm, target_cols, To, Tf, stride=1, idx=None):
    starts = window_starts(len(norm), To, Tf, stride)
    if idx is not None:
        starts = starts[idx]
    enc = np.stack([norm[i - To:i] for i in starts]).astype(F32)
    dec = np.stack([norm[i:i + Tf][:, target_cols] for i in starts]).astype(F32)
    return enc, dec


def split_index(data_len, To, Tf, split_rate, min_valid_scale):
    """dataset.py:181-185."""
    valid_minlen = int(min_valid_scale * Tf)
    valid_idx = min(int(data_len * split_rate), data_len - valid_minlen)
    return valid_idx - Tf - To


class SeriesData:
    """What TIMEBANDDataset.init_dataset + load_dataset produce (dataset.py:68-158), from raw
    arrays instead of a csv: `values` (N,V) with NaN = missing, `tfeat` (N+Tf, n_time) the time
    features of the expanded index (dataset.py:262-295)."""

    def __init__(self, values, tfeat, To, Tf, split_rate, target_cols=None):
        N, V = values.shape
        expanded = np.concatenate([values, np.full((Tf, V), np.nan)], axis=0)       # expand_data
        frame = np.concatenate([tfeat, expanded], axis=1)                          # parse_timeinfo order
        nt = tfeat.shape[1]
        self.target_cols = (np.arange(V) if target_cols is None else np.asarray(target_cols)) + nt
        self.original = frame
        self.missing_all = np.isnan(frame).astype(np.int64)
        data = interpolate(frame)
        self.observed = data[:To + Tf - 1][:, self.target_cols]
        self.forecast = data[To:][:, self.target_cols]
        self.missing = self.missing_all[To:][:, self.target_cols]
        self.mn, self.mx = minmax_info(data, split_rate)
        self.norm = normalize(data, self.mn, self.mx)
        self.decode_min, self.decode_max = self.mn[self.target_cols], self.mx[self.target_cols]
        self.data_length = len(data)
        self.To, self.Tf = To, Tf


# --------------------------------------------------------------------------------------
# Band (predicts) and adjust.  source/trainer.py:238-287, cleaner.py:130-179, predictor.py:106-155
# --------------------------------------------------------------------------------------
class Band:
    def __init__(self, Tf, V, band_width):
        """pred_initate (trainer.py:259-264)."""
        self.Tf, self.V, self.w = Tf, V, band_width
        self.preds = np.full((Tf - 1, Tf, V), np.nan)

    def predicts(self, pred):
        """trainer.py:266-287."""
        B, Tf, V = pred.shape
        self.preds = np.concatenate([self.preds[1 - Tf:] if Tf > 1 else self.preds[:0],
                                     np.full((B, Tf, V), np.nan)])
        for f in range(Tf):
            self.preds[f:B + f, f] = pred[:, f]
        with np.errstate(all="ignore"):
            med = np.nanmedian(self.preds, axis=1)
            std = np.nanstd(self.preds, axis=1)
        for f in range(Tf - 1, 0, -1):
            gamma = (Tf - f) / (Tf - 1)
            std[-f] += std[-f - 1] * gamma
        return med, med - self.w * std, med + self.w * std


def adjust(variant, output, preds, masks, lower, upper, a, b):
    """variant in {"trainer","cleaner","predictor","genb"}: trainer.py:238-257, cleaner.py:130-149,
    predictor.py:106-125, models.py:196-215.  `output` has len(preds)+1 rows and is updated in place."""
    for p in range(preds.shape[0]):
        value = output[p + 1]
        m = masks[p]
        if variant == "trainer":
            value = (1 - m) * value + m * (a * preds[p] + (1 - a) * output[p])
            lm = (value < lower[p]) * (1 - m)
            um = (value > upper[p]) * (1 - m)
            value = (1 - lm) * value + lm * (b * preds[p] + (1 - b) * value)
            value = (1 - um) * value + um * (b * preds[p] + (1 - b) * value)
        elif variant == "genb":
            lm = (value < lower[p]) * (1 - m)
            um = (value > upper[p]) * (1 - m)
            value = (1 - m) * value + m * (a * preds[p] + (1 - a) * output[p])
            value = (1 - lm) * value + lm * (b * preds[p] + (1 - b) * value)
            value = (1 - um) * value + um * (b * preds[p] + (1 - b) * value)
        elif variant == "cleaner":
            lm = value < lower[p]
            um = value > upper[p]
            value = (1 - m) * value + m * (a * preds[p] + (1 - a) * output[p])
            value = (1 - lm) * value + lm * (b * preds[p] + (1 - b) * value)
            value = (1 - um) * value + um * (b * preds[p] + (1 - b) * value)
        else:
            lm = value < lower[p]
            um = value > upper[p]
            value = (1 - lm) * value + lm * (b * preds[p] + (1 - b) * value)
            value = (1 - um) * value + um * (b * preds[p] + (1 - b) * value)
            value = (1 - m) * value + m * (a * preds[p] + (1 - a) * output[p])
        output[p + 1] = value
    return output[1:]


# --------------------------------------------------------------------------------------
# Metrics.  source/metric.py:30-73
# --------------------------------------------------------------------------------------
def metrics(true, pred, mask, esp=1e-7):
    """Returns (nme, nmae, rmse) of one batch."""
    sel = mask == 0
    t, p = true[sel], pred[sel]
    rmse = np.sqrt(np.mean((t - p) ** 2))
    nz = t != 0
    t2, p2 = t[nz], p[nz]
    ne = (t2 - p2) / (esp + t2)
    return np.mean(ne), np.mean(np.abs(ne)), rmse


# --------------------------------------------------------------------------------------
# Post-processing loop shared by train_step / clean / predict
# --------------------------------------------------------------------------------------
class PostProcess:
    """The per-batch tail of trainer.py:202-212 / cleaner.py:88-106 / predictor.py:86-96."""

    def __init__(self, series: SeriesData, variant, band_width, a, b):
        self.s, self.variant, self.a, self.b = series, variant, a, b
        self.band = Band(series.Tf, len(series.target_cols), band_width)
        self.idx = 0
        self.restart()

    def restart(self):
        self.outputs = self.s.observed.copy()
        self.lower_bands = self.s.observed.copy()
        self.upper_bands = self.s.observed.copy()

    def step(self, fake_y):
        s, Tf = self.s, self.s.Tf
        pred_y = denormalize(fake_y, s.decode_min, s.decode_max)
        preds, lower, upper = self.band.predicts(pred_y)
        n = preds.shape[0]
        reals = s.forecast[self.idx:self.idx + n]
        masks = s.missing[self.idx:self.idx + n]
        output = np.concatenate([self.outputs[-1:], reals])
        target = adjust(self.variant, output, preds, masks, lower, upper, self.a, self.b)
        cut = 1 - Tf
        head = (lambda z: z[:cut]) if cut != 0 else (lambda z: z[:0])
        self.lower_bands = np.concatenate([head(self.lower_bands), lower])
        self.upper_bands = np.concatenate([head(self.upper_bands), upper])
        self.outputs = np.concatenate([head(self.outputs), target])
        self.idx += fake_y.shape[0]
        return reals, preds, masks, lower, upper, target


def clean(pG, series: SeriesData, batch_size, n_layers=1, variant="cleaner", band_width=1.0,
          a=0.2, b=0.3):
    """cleaner.py:55-128 (variant="cleaner") / predictor.py:56-104 (variant="predictor")."""
    enc, _ = windowing(series.norm, series.target_cols, series.To, series.Tf)
    pp = PostProcess(series, variant, band_width, a, b)
    for s0 in range(0, len(enc), batch_size):
        fake, _ = generator_fwd(pG, enc[s0:s0 + batch_size], n_layers)
        pp.step(fake[:, :series.Tf])
    return pp.outputs, pp.lower_bands, pp.upper_bands


# --------------------------------------------------------------------------------------
# A whole Gen-A epoch pass (train or valid).  source/trainer.py:125-236, dataset.py:244-257
# --------------------------------------------------------------------------------------
def train_epoch_pass(st: TrainState, series: SeriesData, pp: PostProcess, enc, dec, batch_size,
                     training, rng, train_len):
    """Runs train_step over windows (enc, dec); returns (nmae_mean, per-pass metric sums).
    `rng` is a numpy RandomState standing for the global np.random the reference draws the
    critic index from (dataset.py:250); `train_len` = len(trainset)."""
    st.losses.init_loss()
    pp.restart()                      # trainer.py:137: outputs = dataset.observed
    Tf = series.Tf
    trainset_enc, trainset_dec = series._train_windows
    sums = np.zeros(3)
    nb = 0
    for s0 in range(0, len(enc), batch_size):
        x, y = enc[s0:s0 + batch_size], dec[s0:s0 + batch_size]
        xc = yc = None
        if training:
            idx = rng.randint(train_len - Tf)
            xc, yc = trainset_enc[idx:idx + Tf], trainset_dec[idx:idx + Tf]
        fake = train_iteration(st, x, y, xc, yc, training=training)
        reals, preds, masks, lower, upper, target = pp.step(fake)
        sums += np.array(metrics(reals, preds, masks))
        nb += 1
    return sums[1] / nb, sums / nb, nb

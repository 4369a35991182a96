"""Fused Adam / RMSprop over ONE flat buffer per network (reference schedule:
source/trainer.py:95-98 -- Adam re-created every epoch; :143-144 -- RMSprop re-created every batch).

`flatten(module)` re-points every parameter (and its .grad) at a slice of one contiguous fp32
buffer, so an optimizer step is one kernel launch and a data-parallel gradient exchange is one
all-reduce with no packing.  Stock torch.optim optimizers still work on the same parameters.
"""
import contextlib
import os

import torch

from . import dp, ops


class FlatParams:
    def __init__(self, module):
        params = [p for p in module.parameters()]
        if not params:
            raise ValueError("module has no parameters")
        dev = params[0].device
        if dev.type != "cuda":
            raise RuntimeError("flatten the module after moving it to the GPU")
        n = sum(p.numel() for p in params)
        self.data = torch.empty(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        self.params = params
        off = 0
        for p in params:
            k = p.numel()
            self.data[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.data[off:off + k].view(p.shape)
            p.grad = self.grad[off:off + k].view(p.shape)
            p._tb_grad_slot = p.grad        # the backward kernels add into it directly (ops._slots)
            p._tb_flat = self
            off += k
        self.buckets = self._make_buckets(module)

    def _make_buckets(self, module):
        """Contiguous [lo, hi) ranges of the flat buffer in the order backward completes them: the
        top-level sub-modules in REVERSE registration order (generator: dim_layer, seq_layer, lstm2 | lstm1
        | lstm0), neighbours merged until a bucket holds at least TB_DP_BUCKET floats (default 64 Ki: a
        smaller message costs a collective's latency for nothing)."""
        min_floats = int(os.environ.get("TB_DP_BUCKET", 65536))
        groups, off = [], 0           # [name, lo, hi, params] in registration order
        for name, p in module.named_parameters():
            top = name.split(".")[0]
            if not groups or groups[-1][0] != top:
                groups.append([top, off, off, []])
            groups[-1][2] = off + p.numel()
            groups[-1][3].append(p)
            off += p.numel()
        buckets, cur = [], None
        for top, lo, hi, ps in reversed(groups):
            if cur is None:
                cur = dict(lo=lo, hi=hi, params=list(ps), names=[top])
            else:
                cur["lo"] = lo
                cur["params"] += ps
                cur["names"].append(top)
            if cur["hi"] - cur["lo"] >= min_floats:
                buckets.append(cur)
                cur = None
        if cur is not None:
            buckets.append(cur)
        for i, b in enumerate(buckets):
            b["nparams"] = len(b["params"])
            for p in b["params"]:
                p._tb_bucket = i
        return buckets

    @contextlib.contextmanager
    def alt(self):
        """Backward passes enqueued inside add their parameter gradients into a SECOND buffer (`grad2`, zero between
        uses) instead of `grad`: two backward passes of one network can then run on two streams at once.  merge_alt()
        adds it in: (0 + g1) + (0 + g2) -- the same two-term sum, bit for bit, as accumulating g2 onto g1."""
        if getattr(self, "grad2", None) is None:
            self.grad2 = torch.zeros_like(self.grad)
        off = 0
        for p in self.params:
            k = p.numel()
            p._tb_grad_slot = self.grad2[off:off + k].view(p.shape)
            off += k
        try:
            yield
        finally:
            for p in self.params:
                p._tb_grad_slot = p.grad

    def merge_alt(self):
        if getattr(self, "grad2", None) is not None:
            ops.grad_merge(self.grad, self.grad2)

    def check(self):
        """Parameters / grads must still alias the flat buffers (e.g. nobody set .grad = None)."""
        off = 0
        for p in self.params:
            k = p.numel()
            if p.data.data_ptr() != self.data.data_ptr() + 4 * off:
                raise RuntimeError("a parameter no longer aliases the flat buffer (module.to()/load after flatten?)")
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * off:
                p.grad = self.grad[off:off + k].view(p.shape)
                p._tb_grad_slot = p.grad
            off += k


def flatten(module) -> FlatParams:
    fp = getattr(module, "_tb_flat", None)
    if fp is None:
        fp = FlatParams(module)
        object.__setattr__(module, "_tb_flat", fp)
    else:
        fp.check()
    return fp


class _FlatOptimizer:
    def __init__(self, module, lr):
        self.flat = flatten(module)
        self.lr = lr
        self._reducer = None

    def _join(self):
        ops.Runtime.get(self.flat.grad.device).join_wgrad()     # weight-gradient products on the side stream

    def zero_grad(self, set_to_none: bool = False):
        self.flat.check()
        self._join()
        ops.zero_(self.flat.grad)

    # Data parallel: sum all-reduce of the gradient buffer (the loss kernels already divide by GLOBAL counts).
    # begin_reduce() BEFORE the last backward pass that adds into this network's gradients: its buckets are
    # then reduced on the communication stream as BPTT completes them; all_reduce_grads() (before step())
    # reduces what has not been and waits for all of it.
    def begin_reduce(self, visits=1):
        if dp.active():
            self._reducer = dp.GradReducer(self.flat)
            self._reducer.arm(visits)

    def all_reduce_grads(self):
        self._join()
        if not dp.active():
            self._reducer = None
            return
        if self._reducer is None:
            self._reducer = dp.GradReducer(self.flat)
            self._reducer.arm()
        t0 = dp.comm_bracket(self.flat.grad.device)
        self._reducer.finish()
        dp.comm_bracket_end(t0, self.flat.grad.device)
        self._reducer = None


class FusedAdam(_FlatOptimizer):
    """torch.optim.Adam defaults (betas .9/.999, eps 1e-8)."""

    def __init__(self, module, lr, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(module, lr)
        self.betas, self.eps = betas, eps
        self.m = torch.zeros_like(self.flat.data)
        self.v = torch.zeros_like(self.flat.data)
        self.t = 0

    def step(self):
        self._join()
        self.t += 1
        ops.adam_step(self.flat.data, self.flat.grad, self.m, self.v, self.lr, self.t, self.betas[0], self.betas[1],
                      self.eps)


class FusedRMSprop(_FlatOptimizer):
    """torch.optim.RMSprop defaults (alpha .99, eps 1e-8, no momentum)."""

    def __init__(self, module, lr, alpha=0.99, eps=1e-8):
        super().__init__(module, lr)
        self.alpha, self.eps = alpha, eps
        self.sq = None          # allocated only if the optimizer outlives one step

    def step(self):
        self._join()
        if self.sq is None:
            self.sq = torch.empty_like(self.flat.data)
            ops.rmsprop_step(self.flat.data, self.flat.grad, self.sq, self.lr, self.alpha, self.eps, fresh=True)
        else:
            ops.rmsprop_step(self.flat.data, self.flat.grad, self.sq, self.lr, self.alpha, self.eps, fresh=False)

"""torch.autograd bindings over the C-ABI (include/timeband_hip.h).

PyTorch is plumbing here: device memory (caching allocator), the current HIP stream, autograd
bookkeeping and torch.distributed.  Every arithmetic op of the hot path is a HIP kernel of
libtimeband_hip.so; there is no eager/CPU fallback -- a CPU tensor raises.
"""
from __future__ import annotations

import torch

from . import _lib, dp

ACT_NONE, ACT_TANH, ACT_SIGMOID = 0, 1, 2

# The C++ extension (csrc/ext.cpp -> _tbext.so, built by timeband_amd.build.build_ext): torch::autograd::Functions for the
# LSTM layer, Linear (+ activation) and the head over the same C-ABI -- one call into C++ per op instead of ~20 ctypes /
# allocator / stream look-ups from the autograd.Functions below, which stay as the reference implementation
# (TB_EXT=0 selects them; tests compare the two bit for bit).
_EXT = None


def _load_ext():
    global _EXT
    import os
    if os.environ.get("TB_EXT", "1") == "0":
        return None
    if _EXT is None:
        _lib.load()            # libtimeband_hip.so first (it has to bind to torch's HIP runtime)
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_tbext.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: build it with `python -m timeband_amd.build` "
                               "(or set TB_EXT=0 for the Python autograd path over the same kernels)")
        from . import _tbext
        _tbext.set_grads_hook(grads_enqueued)
        _EXT = _tbext
    return _EXT


def _ext():
    """The extension module, or None when the Python path is selected (TB_EXT=0, or the weight-gradient side stream)."""
    if Runtime.wgrad_overlap:
        return None
    return _EXT if _EXT is not None else _load_ext()


def _slot(p):
    return getattr(p, "_tb_grad_slot", None) if (p is not None and p.requires_grad) else None


def _ptr(t):
    return 0 if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(device_index=None):
    """Raw handle of the current stream (66 look-ups per iteration: the C call is ~10x cheaper than building a
    torch.cuda.Stream object each time)."""
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device() if device_index is None else device_index)
    return torch.cuda.current_stream().cuda_stream


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("timeband_amd ops run on the GPU only (got a CPU tensor); "
                               "there is no CPU fallback in the product path")


def _f32c(t):
    if t.dtype != torch.float32:
        raise RuntimeError(f"expected float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


class Runtime:
    """Per-device scratch: one growable workspace, the sticky error words, lazily initialised."""
    _by_device = {}
    wgrad_overlap = False      # set by the trainer: LSTM weight-gradient GEMMs on the side stream below

    def __init__(self, device):
        self.device = device
        with torch.cuda.device(device):
            _lib.call("tb_init", device.index if device.index is not None else torch.cuda.current_device())
        self.err = torch.zeros(4, dtype=torch.int32, device=device)
        self._ws = {}       # one workspace per stream: launches on different streams may run concurrently
        self.index = device.index if device.index is not None else torch.cuda.current_device()
        ext = _load_ext()
        if ext is not None:
            ext.set_runtime(self.index, self.err)    # the extension shares the error words and owns the co-residency guard

    @classmethod
    def get(cls, device):
        device = torch.device(device)
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        rt = cls._by_device.get(device)
        if rt is None:
            rt = cls._by_device[device] = Runtime(device)
        return rt

    def workspace(self, nbytes):
        key = _stream(self.device.index)
        ws = self._ws.get(key)
        if ws is None or ws.numel() < nbytes:
            ws = self._ws[key] = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=self.device)
        return ws

    # Weight-gradient stream.  The dW_ih / dW_hh products of an LSTM layer's backward read the 268-537 MB
    # dgates matrix but feed nothing in the backward chain (only dx does): with `wgrad_overlap` on they are
    # queued on ONE side stream per device -- one, so that the passes that add into the same gradient slots
    # keep their order -- beside the latency-bound BPTT kernels of the layers below.  Whoever reads or
    # overwrites parameter gradients afterwards (optimizer step, all-reduce, zero_grad) calls join_wgrad().
    def wgrad_stream(self):
        if getattr(self, "_wg", None) is None:
            self._wg = torch.cuda.Stream(self.device)
            self._wg_dirty = False
        return self._wg

    def join_wgrad(self):
        """The current stream waits for every weight-gradient product queued so far."""
        if getattr(self, "_wg", None) is not None and self._wg_dirty:
            torch.cuda.current_stream(self.device).wait_stream(self._wg)

    # Co-residency guard.  A clustered recurrent launch (hidden > 64) makes progress only with ALL its workgroups
    # resident, and every such variant is sized so that TWO launches fit side by side on a CU (256 of 512
    # registers, 80 of 160 KiB LDS; csrc/lstm.hip make_plan).  A third one in flight -- three streams, e.g.
    # TB_OVERLAP=7 -- can leave two launches each half resident, spinning until the 2 s timeout.  So before a
    # clustered launch on stream S: of the OTHER streams with a clustered launch not yet known to be finished,
    # all but the most recent one are waited for by S (oldest first).  With the default streams every such wait
    # is on work S depends on anyway.
    # An EXCLUSIVE launch (tb_lstm_exclusive: workgroups of ~390 registers, nothing clustered fits beside them) waits
    # for every other stream's clustered launch, and every later clustered launch elsewhere waits for it.
    def clustered_begin(self, exclusive=False):
        if _EXT is not None:          # one guard state for both paths: the extension's
            return _EXT.clustered_begin(self.index, bool(exclusive))
        cl = self.__dict__.setdefault("_clustered", {})
        if not cl:
            return
        cur = torch.cuda.current_stream(self.device)
        live = []
        for key, (seq, ev, excl) in list(cl.items()):
            if key == cur.cuda_stream:
                continue
            if ev.query():
                del cl[key]
            elif excl or exclusive:
                cur.wait_event(ev)
            else:
                live.append((seq, ev))
        live.sort(key=lambda q: q[0])
        for seq, ev in live[:-1]:
            cur.wait_event(ev)

    def clustered_end(self, exclusive=False):
        if _EXT is not None:
            return _EXT.clustered_end(self.index, bool(exclusive))
        cl = self.__dict__.setdefault("_clustered", {})
        cur = torch.cuda.current_stream(self.device)
        self._cl_seq = getattr(self, "_cl_seq", 0) + 1
        old = cl.get(cur.cuda_stream)
        ev = old[1] if old is not None else torch.cuda.Event()
        ev.record(cur)
        cl[cur.cuda_stream] = (self._cl_seq, ev, bool(exclusive))

    def check(self):
        """Host sync: raise if a persistent kernel reported an inter-workgroup wait timeout."""
        e = self.err.tolist()
        if e[0] != 0:
            self.err.zero_()
            raise RuntimeError(f"timeband_amd: persistent LSTM kernel reported error words {e}")


def check_errors(device=None):
    for dev, rt in list(Runtime._by_device.items()):
        if device is None or torch.device(device) == dev:
            rt.check()


# ----------------------------------------------------------------------------------------------
# Parameter gradients straight into a gradient buffer.  optim.FlatParams gives every parameter a
# `_tb_grad_slot` (its slice of the network's flat gradient buffer, which is also its .grad); when
# all the parameters of an op have one, the backward kernels ADD their result into the slots
# (`accumulate` of the C-ABI) and autograd gets None for them -- the same sums as autograd's own
# `grad += new` without a temporary and an add kernel per parameter and backward pass.
# ----------------------------------------------------------------------------------------------
def _slots(need, *params):
    """Slots of the parameters that need a gradient, or None if any of those has none."""
    out = []
    for n, p in zip(need, params):
        s = getattr(p, "_tb_grad_slot", None) if (n and p is not None) else None
        if n and p is not None and (s is None or s.shape != p.shape or not s.is_contiguous()):
            return None
        out.append(s)
    return out if any(s is not None for s in out) else None


# Data parallel: armed dp.GradReducer objects hear which parameters' gradients have just been ENQUEUED on
# the current stream (only the slot path adds in place; gradients returned to autograd land later, and the
# reducer's finish() covers them).
_REDUCERS = []


def reducers_changed():
    """dp.GradReducer.arm / finish: the extension's backward passes call grads_enqueued only while a reducer listens."""
    if _EXT is not None:
        _EXT.arm_grads_hook(bool(_REDUCERS))


def grads_enqueued(slots, params):
    if _REDUCERS and slots is not None:
        done = [p for s_, p in zip(slots, params) if s_ is not None]
        for r in list(_REDUCERS):
            r.notify(done)


# ----------------------------------------------------------------------------------------------
# LSTM layer
# ----------------------------------------------------------------------------------------------
def lstm_fwd_raw(x, w_ih, w_hh, b_ih, b_hh, train, split=0):
    """One LSTM layer forward, no autograd: (y (B,T,H), gates (B,T,4H), cst (B,T,H) or None).  split > 0: the batch is
    launched in pieces of `split` rows (rows are independent: the same bits whichever way the batch is cut -- it keeps a
    hidden-256 layer of more than 512 rows on the 8-member cluster layout, which may share the chip with another
    clustered launch, instead of the exclusive 4-member one)."""
    ext = _ext()
    if ext is not None:
        Runtime.get(x.device)
        return tuple(ext.lstm_fwd_raw(x, w_ih, w_hh, b_ih, b_hh, bool(train), int(split)))
    x, w_ih, w_hh, b_ih, b_hh = (_f32c(t) for t in (x, w_ih, w_hh, b_ih, b_hh))
    B, T, In = x.shape
    H = w_hh.shape[1]
    if w_ih.shape != (4 * H, In) or w_hh.shape != (4 * H, H):
        raise RuntimeError("lstm: weight shapes do not match the input")
    lib = _lib.load()
    if not lib.tb_lstm_supported(H):
        raise RuntimeError(f"lstm: hidden size {H} is not supported by the HIP kernels "
                           "(supported: 16, 32, 64, 128, 256, 512)")
    rt = Runtime.get(x.device)
    y = torch.empty(B, T, H, dtype=torch.float32, device=x.device)
    gates = torch.empty(B, T, 4 * H, dtype=torch.float32, device=x.device)
    cst = torch.empty(B, T, H, dtype=torch.float32, device=x.device) if train else None
    step = B if split <= 0 else min(int(split), B)
    for b0 in range(0, B, step):
        nb = min(step, B - b0)
        nws = lib.tb_lstm_workspace_bytes(nb, T, In, H, 0)
        ws = rt.workspace(nws)
        excl = H > 64 and bool(lib.tb_lstm_exclusive(nb, In, H, 0))
        if H > 64:
            rt.clustered_begin(excl)
        _lib.call("tb_lstm_fwd", x.data_ptr() + 4 * b0 * T * In, _ptr(w_ih), _ptr(w_hh), _ptr(b_ih), _ptr(b_hh),
                  y.data_ptr() + 4 * b0 * T * H, gates.data_ptr() + 16 * b0 * T * H,
                  0 if cst is None else cst.data_ptr() + 4 * b0 * T * H, nb, T, In, H, int(train), _ptr(ws), ws.numel(),
                  _ptr(rt.err), _stream())
        if H > 64:
            rt.clustered_end(excl)
    return y, gates, cst


class _LSTMLayerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh, train, dx_from=0, pre=None, split=0):
        # pre = (y, gates, cst): this layer's forward was already computed (LSTMGenerator.twin: one pass over the
        # rows of two forward passes) -- only the graph node is made here, around those buffers
        _need_cuda(x, w_ih, w_hh, b_ih, b_hh)
        ctx.params = (w_ih, w_hh, b_ih, b_hh)
        ctx.dx_from = int(dx_from)
        x, w_ih, w_hh = (_f32c(t) for t in (x, w_ih, w_hh))
        B, T, In = x.shape
        H = w_hh.shape[1]
        if pre is None:
            y, gates, cst = lstm_fwd_raw(x, w_ih, w_hh, b_ih, b_hh, train, split)
        else:
            y, gates, cst = pre
            if y.shape != (B, T, H) or gates.shape != (B, T, 4 * H) or not (y.is_contiguous() and gates.is_contiguous()):
                raise RuntimeError("lstm: precomputed buffers do not match the input")
            y = y.view_as(y)        # a tensor of its own for autograd (same memory)
        if train:
            ctx.save_for_backward(x, w_ih, w_hh, y, gates, cst)
            ctx.dims = (B, T, In, H)
            ctx.used = False
        return y

    @staticmethod
    def backward(ctx, dy):
        if ctx.used:
            raise RuntimeError("lstm: backward called twice (the gate stash is consumed in place)")
        ctx.used = True
        x, w_ih, w_hh, y, gates, cst = ctx.saved_tensors
        B, T, In, H = ctx.dims
        dy = _f32c(dy)
        dev = x.device
        rt = Runtime.get(dev)
        lib = _lib.load()
        need = ctx.needs_input_grad
        dx = torch.empty_like(x) if need[0] else None
        slots = _slots(need[1:5], *ctx.params)
        if slots is not None:
            dw_ih, dw_hh, db_ih, db_hh = slots
        else:
            dw_ih = torch.empty_like(w_ih) if need[1] else None
            dw_hh = torch.empty_like(w_hh) if need[2] else None
            db_ih = torch.empty(4 * H, dtype=torch.float32, device=dev) if need[3] else None
            db_hh = torch.empty(4 * H, dtype=torch.float32, device=dev) if need[4] else None
        nws = lib.tb_lstm_workspace_bytes(B, T, In, H, 1)
        ws = rt.workspace(nws)
        side = slots is not None and Runtime.wgrad_overlap and (dw_ih is not None or dw_hh is not None)
        excl = H > 64 and bool(lib.tb_lstm_exclusive(B, In, H, 1))
        if H > 64:
            rt.clustered_begin(excl)
        # dx_from: the input gradient of the batch rows before it is never read (the real half of the discriminator's
        # real + fake pass) and is not computed; those rows of dx stay as allocated
        _lib.call("tb_lstm_bwd_rows", _ptr(dy), _ptr(x), _ptr(w_ih), _ptr(w_hh), _ptr(y), _ptr(gates), _ptr(cst),
                  _ptr(dx), _ptr(None if side else dw_ih), _ptr(None if side else dw_hh), _ptr(db_ih), _ptr(db_hh),
                  B, T, In, H, 1 if slots is not None else 0, _ptr(ws), ws.numel(), _ptr(rt.err), _stream(),
                  min(max(ctx.dx_from, 0), B))
        if H > 64:
            rt.clustered_end(excl)
        if side:
            cur = torch.cuda.current_stream(dev)
            wg = rt.wgrad_stream()
            wg.wait_stream(cur)
            with torch.cuda.stream(wg):
                ws2 = rt.workspace(lib.tb_lstm_wgrad_workspace_bytes(In, H))
                _lib.call("tb_lstm_wgrad", _ptr(x), _ptr(y), _ptr(gates), _ptr(dw_ih), _ptr(dw_hh), B, T, In, H, 1,
                          _ptr(ws2), ws2.numel(), _stream())
                for t in (x, y, gates):
                    t.record_stream(wg)
                rt._wg_dirty = True
                grads_enqueued(slots, ctx.params)      # an armed reducer records its event on THIS stream
            return dx, None, None, None, None, None, None, None, None
        if slots is not None:
            grads_enqueued(slots, ctx.params)
            return dx, None, None, None, None, None, None, None, None
        return dx, dw_ih, dw_hh, db_ih, db_hh, None, None, None, None


def lstm_layer(x, w_ih, w_hh, b_ih, b_hh, dx_from=0, pre=None, split=0):
    """One nn.LSTM layer (batch_first, zero initial state) -> (B,T,H).  dx_from: the caller promises that nobody reads
    the input gradient of the batch rows [0, dx_from) -- it is then not computed (those rows of x.grad are garbage).
    pre: see _LSTMLayerFn.forward."""
    ext = _ext()
    if ext is not None:
        if not x.is_cuda:
            _need_cuda(x)
        Runtime.get(x.device)
        py, pg, pc = pre if pre is not None else (None, None, None)
        return ext.lstm_layer(x, w_ih, w_hh, b_ih, b_hh, int(dx_from), int(split), _slot(w_ih), _slot(w_hh), _slot(b_ih),
                              _slot(b_hh), py, pg, pc)
    # grad mode is off inside Function.forward, so the stash decision is taken here
    train = bool(torch.is_grad_enabled() and any(t.requires_grad for t in (x, w_ih, w_hh, b_ih, b_hh)))
    return _LSTMLayerFn.apply(x, w_ih, w_hh, b_ih, b_hh, train, dx_from, pre, split)


import os as _os
_TWIN_SPLIT = _os.environ.get("TB_TWIN_SPLIT", "1") != "0"     # experiment knob: 0 = hidden-256 twin layers as ONE (exclusive) launch


def lstm_fwd_windows(pseries, first_row, stride, nwin, T, w_hh):
    """Recurrent forward (inference) of `nwin` windows of T steps whose pre-activations are rows first_row + b * stride + t
    of the per-series-row projection `pseries` (rows, 4H): see tb_lstm_fwd_windows.  -> y (nwin, T, H)."""
    _need_cuda(pseries, w_hh)
    pseries, w_hh = _f32c(pseries), _f32c(w_hh.detach())
    H = w_hh.shape[1]
    if pseries.dim() != 2 or pseries.shape[1] != 4 * H:
        raise RuntimeError("lstm_fwd_windows: pseries must be (rows, 4H)")
    lib = _lib.load()
    rt = Runtime.get(pseries.device)
    y = torch.empty(nwin, T, H, dtype=torch.float32, device=pseries.device)
    ws = rt.workspace(lib.tb_lstm_workspace_bytes(nwin, T, 0, H, 0))
    excl = H > 64 and bool(lib.tb_lstm_exclusive(nwin, 0, H, 0))
    if H > 64:
        rt.clustered_begin(excl)
    _lib.call("tb_lstm_fwd_windows", _ptr(pseries), pseries.shape[0], int(first_row), int(stride), _ptr(w_hh), _ptr(y),
              nwin, T, H, _ptr(ws), ws.numel(), _ptr(rt.err), _stream())
    if H > 64:
        rt.clustered_end(excl)
    return y


def _lstm_params(lstm_module, l):
    return (getattr(lstm_module, f"weight_ih_l{l}"), getattr(lstm_module, f"weight_hh_l{l}"),
            getattr(lstm_module, f"bias_ih_l{l}"), getattr(lstm_module, f"bias_hh_l{l}"))


def lstm_stack(x, lstm_module, dx_from=0, pre=None, split=0):
    """Run x through the layers of a torch.nn.LSTM parameter container with the HIP kernels (dx_from: see lstm_layer;
    it applies to the first layer, whose input is x).  pre: one (y, gates, cst) per layer, see _LSTMLayerFn.forward.
    split: forward launches of hidden >= 256 layers in pieces of that many rows (lstm_fwd_raw)."""
    for l in range(lstm_module.num_layers):
        x = lstm_layer(x, *_lstm_params(lstm_module, l), dx_from if l == 0 else 0, None if pre is None else pre[l],
                       split if (lstm_module.hidden_size >= 256 and _TWIN_SPLIT) else 0)
    return x


def lstm_stack_raw(x, lstm_module, split=0):
    """lstm_stack without autograd, keeping the BPTT stash: (output, [(y, gates, cst) per layer])."""
    out = []
    for l in range(lstm_module.num_layers):
        y, gates, cst = lstm_fwd_raw(x, *(p.detach() for p in _lstm_params(lstm_module, l)), True,
                                     split if (lstm_module.hidden_size >= 256 and _TWIN_SPLIT) else 0)
        out.append((y, gates, cst))
        x = y
    return x, out


# ----------------------------------------------------------------------------------------------
# Linear (+ activation)
# ----------------------------------------------------------------------------------------------
def linear_act_raw(x, W, b, act):
    """act(x @ W.T + b) for a 2-D x, no autograd (GEMM tiles; any output width)."""
    ext = _ext()
    if ext is not None:
        return ext.linear_act_raw(x, W, b, int(act))
    x, W = _f32c(x), _f32c(W)
    b = _f32c(b) if b is not None else None
    M, K = x.shape
    N = W.shape[0]
    if W.shape[1] != K:
        raise RuntimeError("linear: weight shape does not match the input")
    y = torch.empty(M, N, dtype=torch.float32, device=x.device)
    _lib.call("tb_linear_act_fwd", _ptr(x), _ptr(W), _ptr(b), _ptr(y), M, K, N, act, _stream())
    return y


class _LinearActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W, b, act, pre=None):
        # pre: the output, already computed (LSTMGenerator.twin); only the graph node is made here
        _need_cuda(x, W, b)
        ctx.params = (W, b)
        x, W = _f32c(x), _f32c(W)
        M, K = x.shape
        N = W.shape[0]
        if pre is None:
            y = linear_act_raw(x, W, b, act)
        else:
            if pre.shape != (M, N) or not pre.is_contiguous():
                raise RuntimeError("linear: precomputed output does not match the input")
            y = pre.view_as(pre)
        ctx.save_for_backward(x, W, y if act else None)
        ctx.act = act
        ctx.has_b = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W, y = ctx.saved_tensors
        M, K = x.shape
        N = W.shape[0]
        dy = _f32c(dy)
        dev = x.device
        rt = Runtime.get(dev)
        lib = _lib.load()
        if ctx.act:
            dz = torch.empty_like(dy)
            _lib.call("tb_act_bwd", _ptr(dy), _ptr(y), _ptr(dz), dy.numel(), ctx.act, _stream())
        else:
            dz = dy
        need = ctx.needs_input_grad
        dx = torch.empty_like(x) if need[0] else None
        slots = _slots((need[1], need[2] and ctx.has_b), *ctx.params)
        if slots is not None:
            dW, db = slots
        else:
            dW = torch.empty_like(W) if need[1] else None
            db = torch.empty(N, dtype=torch.float32, device=dev) if (need[2] and ctx.has_b) else None
        ws = rt.workspace(lib.tb_linear_workspace_bytes(M, K, N))
        _lib.call("tb_linear_bwd", _ptr(dz), _ptr(x), _ptr(W), _ptr(dx), _ptr(dW), _ptr(db), M, K, N,
                  1 if slots is not None else 0, _ptr(ws), ws.numel(), _stream())
        if slots is not None:
            grads_enqueued(slots, ctx.params)
            return dx, None, None, None, None
        return dx, dW, db, None, None


class _HeadFn(torch.autograd.Function):
    """act(x @ w.T + b) for a single output column (the discriminator head): streaming kernels."""

    @staticmethod
    def forward(ctx, x, W, b, act):
        _need_cuda(x, W, b)
        ctx.params = (W, b)
        x, W = _f32c(x), _f32c(W)
        b = _f32c(b) if b is not None else None
        M, K = x.shape
        y = torch.empty(M, 1, dtype=torch.float32, device=x.device)
        _lib.call("tb_head_fwd", _ptr(x), _ptr(W), _ptr(b), _ptr(y), M, K, act, _stream())
        ctx.save_for_backward(x, W, y)
        ctx.act = act
        ctx.has_b = b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W, y = ctx.saved_tensors
        M, K = x.shape
        dy = _f32c(dy)
        dev = x.device
        rt = Runtime.get(dev)
        need = ctx.needs_input_grad
        dx = torch.empty_like(x) if need[0] else None
        slots = _slots((need[1], need[2] and ctx.has_b), *ctx.params)
        if slots is not None:
            dW, db = slots
        else:
            dW = torch.empty_like(W) if need[1] else None
            db = torch.empty(1, dtype=torch.float32, device=dev) if (need[2] and ctx.has_b) else None
        ws = rt.workspace(_lib.load().tb_head_workspace_bytes(M, K))
        _lib.call("tb_head_bwd", _ptr(dy), _ptr(y), _ptr(x), _ptr(W), _ptr(dx), _ptr(dW), _ptr(db), M, K, ctx.act,
                  1 if slots is not None else 0, _ptr(ws), ws.numel(), _stream())
        if slots is not None:
            grads_enqueued(slots, ctx.params)
            return dx, None, None, None
        return dx, dW, db, None


def _aligned16(*ts):
    return all(t is None or t.data_ptr() % 16 == 0 for t in ts)


def linear_act(x2d, W, b, act=ACT_NONE, pre=None):
    """act(x2d @ W.T + b) for a 2-D x.  pre: see _LinearActFn.forward."""
    ext = _ext()
    if ext is not None:
        if not x2d.is_cuda:
            _need_cuda(x2d)
        Runtime.get(x2d.device)
        return ext.linear_act(x2d, W, b, int(act), _slot(W), _slot(b), pre)
    if pre is None and W.shape[0] == 1 and _lib.load().tb_head_supported(W.shape[1]) and x2d.is_contiguous() \
            and W.is_contiguous() and _aligned16(x2d, W):
        return _HeadFn.apply(x2d, W, b, act)
    return _LinearActFn.apply(x2d, W, b, act, pre)


# ----------------------------------------------------------------------------------------------
# Losses
# ----------------------------------------------------------------------------------------------
# Data parallel (dp.py): every rank holds a contiguous share of the global batch.  The gradient kernels
# divide by GLOBAL element counts (dp.count_scale), so summed gradients are exact.  Only the reconstruction
# sums behind the gradient penalty -- (1 - ||true - pred||)^2 / B with the norm over the WHOLE batch,
# loss.py:136-141 -- have to be global before backward; `recon_presum` starts their all-reduce as soon as the
# generator's output exists, under the discriminator pass.  The adversarial sums stay rank-local: the
# returned loss value carries this rank's share of those terms (their gradients need no sum), and the
# logging sums are made global when they are read (TIMEBANDLoss.loss).
replicated = dp.replicated


def _world():
    return dp.world()


def _loss_sums(Dy, DGx, tru, pred, rt, reduce=True):
    lib = _lib.load()
    sums = torch.empty(8, dtype=torch.float64, device=rt.device)
    ws = rt.workspace(lib.tb_loss_workspace_bytes())
    _lib.call("tb_loss_sums", _ptr(Dy), 0 if Dy is None else Dy.numel(), _ptr(DGx),
              0 if DGx is None else DGx.numel(), _ptr(tru), _ptr(pred), 0 if pred is None else pred.numel(),
              _ptr(sums), _ptr(ws), ws.numel(), _stream())
    if reduce and dp.active():
        dp.all_reduce(sums)
    return sums


def _loss_sums_finish(Dy, DGx, tru, pred, rt, mode, nD_real, nD_fake, nP, batch, l1_w, l2_w, gp_w, out, running, total):
    """_loss_sums + tb_loss_finish with nothing between them (no all-reduce of the sums): two launches instead of three."""
    lib = _lib.load()
    sums = torch.empty(8, dtype=torch.float64, device=rt.device)
    ws = rt.workspace(lib.tb_loss_workspace_bytes())
    _lib.call("tb_loss_sums_finish", _ptr(Dy), 0 if Dy is None else Dy.numel(), _ptr(DGx),
              0 if DGx is None else DGx.numel(), _ptr(tru), _ptr(pred), 0 if pred is None else pred.numel(),
              _ptr(sums), _ptr(ws), ws.numel(), mode, nD_real, nD_fake, nP, batch, float(l1_w), float(l2_w), float(gp_w),
              _ptr(out), _ptr(running), _ptr(total), _stream())
    return sums


def _pair_base(Dy, DGx):
    """The discriminator's real + fake pass returns its two halves as views of ONE output buffer: that buffer (the
    autograd root both are views of) if Dy and DGx are exactly its first and second half, else None."""
    base = getattr(Dy, "_base", None)
    if base is None or base is not getattr(DGx, "_base", None) or not base.is_contiguous() or base.dtype != torch.float32:
        return None
    n = Dy.numel()
    if DGx.numel() != n or base.numel() != 2 * n or not (Dy.is_contiguous() and DGx.is_contiguous()):
        return None
    if Dy.data_ptr() != base.data_ptr() or DGx.data_ptr() != base.data_ptr() + 4 * n:
        return None
    return base


class ReconPresum:
    """sum huber(pred - true), sum (pred - true)^2 of this rank's windows, all-reduce in flight."""

    def __init__(self, tru, pred):
        _need_cuda(tru, pred)
        tru, pred = _f32c(tru.detach()), _f32c(pred.detach())
        rt = Runtime.get(pred.device)
        self.sums = _loss_sums(None, None, tru, pred, rt, reduce=False)
        self.handle = dp.all_reduce_async(self.sums[4:6]) if dp.active() else None

    def get(self):
        if self.handle is not None:
            self.handle.wait()
            self.handle = None
        return self.sums


def recon_presum(tru, pred):
    return ReconPresum(tru, pred)


class _DisLossFn(torch.autograd.Function):
    """TIMEBANDLoss.dis_loss (loss.py:106-119): returns (total, out[8]); out is not differentiable."""

    @staticmethod
    def forward(ctx, Dy, DGx, tru, pred, critic, gp_w, running, base):
        # base: the buffer Dy and DGx are the two halves of (_pair_base), or None.  With it the gradient goes to that
        # buffer in ONE launch -- instead of two, plus autograd's concatenation of the halves behind them
        _need_cuda(Dy, DGx, tru, pred)
        Dy, DGx = _f32c(Dy), _f32c(DGx)
        rt = Runtime.get(Dy.device)
        out = torch.empty(8, dtype=torch.float32, device=Dy.device)
        nP = dp.count_scale(pred.numel()) if critic else 1.0
        batch = dp.count_scale(tru.shape[0]) if critic else 1.0
        nDy, nDGx = dp.count_scale(Dy.numel()), dp.count_scale(DGx.numel())
        total = torch.empty((), dtype=torch.float32, device=out.device)    # separate storage: autograd outputs must not alias
        if critic:
            tru, pred = _f32c(tru), _f32c(pred)
        # the penalty's gradient needs the global norm; without it (the trainer's critic step runs G without
        # a graph) nothing does, and the critic's loss value -- never logged, loss.py:107-112 -- stays local
        if critic and ctx.needs_input_grad[3] and dp.active():
            sums = _loss_sums(Dy, DGx, tru, pred, rt, reduce=True)
            _lib.call("tb_loss_finish", _ptr(sums), 1, nDy, nDGx, nP, batch, 0.0, 0.0, float(gp_w), _ptr(out), 0,
                      _ptr(total), _stream())
        else:
            sums = _loss_sums_finish(Dy, DGx, tru if critic else None, pred if critic else None, rt, 1 if critic else 0,
                                     nDy, nDGx, nP, batch, 0.0, 0.0, gp_w, out, None if critic else running, total)
        ctx.critic, ctx.gp_w, ctx.counts = critic, gp_w, (nDy, nDGx, nP, batch)
        ctx.paired = base is not None
        ctx.base_shape = None if base is None else tuple(base.shape)
        ctx.save_for_backward(Dy, DGx, tru if critic else None, pred if critic else None, sums)
        ctx.mark_non_differentiable(out)
        ctx.set_materialize_grads(False)     # no zero-fill launch for `out`'s unused gradient
        return total, out

    @staticmethod
    def backward(ctx, gtotal, _gout):
        if gtotal is None:
            return (None,) * 8
        Dy, DGx, tru, pred, sums = ctx.saved_tensors
        nDy, nDGx, nP, batch = ctx.counts
        up = _f32c(gtotal)
        wgan = 1 if ctx.critic else 0
        dbase = None
        if ctx.paired and ctx.needs_input_grad[7]:
            # Dy is the first half of the buffer, DGx the second (checked by _pair_base)
            dbase = torch.empty(ctx.base_shape, dtype=torch.float32, device=Dy.device)
            _lib.call("tb_gan_loss_grad_pair", _ptr(Dy), Dy.numel(), nDy, nDGx, wgan, _ptr(up), _ptr(dbase), _stream())
        dDy = torch.empty_like(Dy) if ctx.needs_input_grad[0] and not ctx.paired else None
        dDGx = torch.empty_like(DGx) if ctx.needs_input_grad[1] and not ctx.paired else None
        if dDy is not None:
            _lib.call("tb_gan_loss_grad", _ptr(Dy), Dy.numel(), nDy, 1, wgan, _ptr(up), _ptr(dDy), _stream())
        if dDGx is not None:
            _lib.call("tb_gan_loss_grad", _ptr(DGx), DGx.numel(), nDGx, 0, wgan, _ptr(up), _ptr(dDGx), _stream())
        dpred = None
        if ctx.critic and ctx.needs_input_grad[3]:
            dpred = torch.empty_like(pred)
            _lib.call("tb_recon_loss_grad", _ptr(tru), _ptr(pred), pred.numel(), nP, batch, _ptr(sums), 0.0, 0.0,
                      float(ctx.gp_w), _ptr(up), _ptr(dpred), 0, _stream())
        return dDy, dDGx, None, dpred, None, None, None, dbase


class _GenLossFn(torch.autograd.Function):
    """TIMEBANDLoss.gen_loss (loss.py:93-104): adversarial term with the FAKE label + l1 + l2 + gp."""

    @staticmethod
    def forward(ctx, tru, pred, DGx, l1_w, l2_w, gp_w, running, presum):
        _need_cuda(tru, pred, DGx)
        tru, pred, DGx = _f32c(tru), _f32c(pred), _f32c(DGx)
        rt = Runtime.get(pred.device)
        if presum is not None:
            sums = _loss_sums(None, DGx, None, None, rt, reduce=False)
            sums[4:6].copy_(presum.get()[4:6])      # the global reconstruction sums, reduced under the D pass
        elif dp.active():
            sums = _loss_sums(None, DGx, tru, pred, rt, reduce=False)
            dp.all_reduce(sums[4:6])
        else:
            sums = None
        out = torch.empty(8, dtype=torch.float32, device=pred.device)
        nDGx, nP, batch = dp.count_scale(DGx.numel()), dp.count_scale(pred.numel()), dp.count_scale(tru.shape[0])
        total = torch.empty((), dtype=torch.float32, device=out.device)    # separate storage: autograd outputs must not alias
        if sums is None:
            sums = _loss_sums_finish(None, DGx, tru, pred, rt, 2, 1.0, nDGx, nP, batch, l1_w, l2_w, gp_w, out, running, total)
        else:
            _lib.call("tb_loss_finish", _ptr(sums), 2, 1.0, nDGx, nP, batch, float(l1_w), float(l2_w), float(gp_w),
                      _ptr(out), _ptr(running), _ptr(total), _stream())
        ctx.w = (l1_w, l2_w, gp_w)
        ctx.counts = (nDGx, nP, batch)
        ctx.save_for_backward(tru, pred, DGx, sums)
        ctx.mark_non_differentiable(out)
        ctx.set_materialize_grads(False)     # no zero-fill launch for `out`'s unused gradient
        return total, out

    @staticmethod
    def backward(ctx, gtotal, _gout):
        if gtotal is None:
            return (None,) * 8
        tru, pred, DGx, sums = ctx.saved_tensors
        l1_w, l2_w, gp_w = ctx.w
        nDGx, nP, batch = ctx.counts
        up = _f32c(gtotal)
        dpred = dDGx = None
        if ctx.needs_input_grad[1]:
            dpred = torch.empty_like(pred)
            _lib.call("tb_recon_loss_grad", _ptr(tru), _ptr(pred), pred.numel(), nP, batch, _ptr(sums), float(l1_w),
                      float(l2_w), float(gp_w), _ptr(up), _ptr(dpred), 0, _stream())
        if ctx.needs_input_grad[2]:
            dDGx = torch.empty_like(DGx)
            _lib.call("tb_gan_loss_grad", _ptr(DGx), DGx.numel(), nDGx, 0, 0, _ptr(up), _ptr(dDGx), _stream())
        return None, dpred, dDGx, None, None, None, None, None


def dis_loss(Dy, DGx, tru=None, pred=None, critic=False, gp_w=0.0, running=None):
    base = _pair_base(Dy, DGx) if (Dy.requires_grad and DGx.requires_grad and torch.is_grad_enabled()) else None
    if base is not None:
        # the halves enter as data, the gradient leaves through the buffer they are views of
        return _DisLossFn.apply(Dy.detach(), DGx.detach(), tru, pred, critic, gp_w, running, base)
    return _DisLossFn.apply(Dy, DGx, tru, pred, critic, gp_w, running, None)


def gen_loss(tru, pred, DGx, l1_w, l2_w, gp_w, running=None, presum=None):
    return _GenLossFn.apply(tru, pred, DGx, l1_w, l2_w, gp_w, running, presum)


# ----------------------------------------------------------------------------------------------
# data side
# ----------------------------------------------------------------------------------------------
def window_gather(series, target_cols, first, nwin, To, Tf, stride=1, starts=None):
    """enc (nwin,To,E), dec (nwin,Tf,V) from the device-resident normalised series (len,E)."""
    _need_cuda(series, target_cols)
    if not series.is_contiguous() or series.dtype != torch.float32 or target_cols.dtype != torch.int32:
        raise RuntimeError("window_gather: series must be contiguous float32, target_cols int32")
    E = series.shape[1]
    V = target_cols.numel()
    enc = torch.empty(nwin, To, E, dtype=torch.float32, device=series.device)
    dec = torch.empty(nwin, Tf, V, dtype=torch.float32, device=series.device)
    if starts is not None:
        lo, hi = int(starts.min()), int(starts.max())
        if lo - To < 0 or hi + Tf > series.shape[0]:
            raise RuntimeError("window_gather: a window leaves the series")
    _lib.call("tb_window_gather", _ptr(series), series.shape[0], E, _ptr(starts), int(first), int(stride), nwin,
              To, Tf, _ptr(target_cols), V, _ptr(enc), _ptr(dec), _stream())
    return enc, dec


def interpolate(data, tpos=None, V=0, row_off=0):
    """ffill / bfill / 0 of a float64 (rows, E) device frame IN PLACE; returns the int32 missing mask of the target
    columns for rows >= row_off when `tpos` (E, int32: target position or -1) is given."""
    _need_cuda(data, tpos)
    if data.dtype != torch.float64 or not data.is_contiguous():
        raise RuntimeError("interpolate: data must be contiguous float64")
    rows, E = data.shape
    missing = None
    if tpos is not None:
        if tpos.dtype != torch.int32 or tpos.numel() != E:
            raise RuntimeError("interpolate: tpos must be int32 of length E")
        missing = torch.empty(max(0, rows - row_off), V, dtype=torch.int32, device=data.device)
    rt = Runtime.get(data.device)
    ws = rt.workspace(_lib.load().tb_interpolate_workspace_bytes(rows, E))
    _lib.call("tb_interpolate", _ptr(data), rows, E, _ptr(tpos), int(V), int(row_off), _ptr(missing), _ptr(ws),
              ws.numel(), _stream())
    return missing


def minmax(data, mn=None, mx=None, finish=True):
    """Column min / max of a float64 (rows, E) device frame; mn / mx given: fold this chunk into them."""
    _need_cuda(data)
    data = _c(data, torch.float64)
    rows, E = data.shape
    init = mn is None
    if init:
        mn = torch.empty(E, dtype=torch.float64, device=data.device)
        mx = torch.empty(E, dtype=torch.float64, device=data.device)
    rt = Runtime.get(data.device)
    ws = rt.workspace(_lib.load().tb_minmax_workspace_bytes(E))
    _lib.call("tb_minmax", _ptr(data), rows, E, _ptr(mn), _ptr(mx), int(init), int(finish), _ptr(ws), ws.numel(), _stream())
    return mn, mx


def normalize(data, mn, mx, out32=None, want64=False):
    """2*((x-min)/(max-min))-1 of a float64 (rows, E) device frame -> float32 series (and the float64 frame)."""
    _need_cuda(data, mn, mx)
    data, mn, mx = _c(data, torch.float64), _c(mn, torch.float64), _c(mx, torch.float64)
    rows, E = data.shape
    if out32 is None:
        out32 = torch.empty(rows, E, dtype=torch.float32, device=data.device)
    elif out32.shape != (rows, E) or out32.dtype != torch.float32 or not out32.is_contiguous():
        raise RuntimeError("normalize: out32 must be contiguous float32 (rows, E)")
    out64 = torch.empty(rows, E, dtype=torch.float64, device=data.device) if want64 else None
    _lib.call("tb_normalize", _ptr(data), rows, E, _ptr(mn), _ptr(mx), _ptr(out32), _ptr(out64), _stream())
    return (out32, out64) if want64 else out32


def take_columns(data, cols, out=None):
    _need_cuda(data, cols)
    data = _c(data, torch.float64)
    if cols.dtype != torch.int32:
        raise RuntimeError("take_columns: cols must be int32")
    rows, E = data.shape
    V = cols.numel()
    if out is None:
        out = torch.empty(rows, V, dtype=torch.float64, device=data.device)
    _lib.call("tb_take_columns", _ptr(data), rows, E, _ptr(cols), V, _ptr(out), _stream())
    return out


def denormalize(x, dmin, dmax):
    _need_cuda(x, dmin, dmax)
    x, dmin, dmax = _f32c(x), _c(dmin, torch.float64), _c(dmax, torch.float64)
    out = torch.empty_like(x)
    _lib.call("tb_denormalize", _ptr(x), x.numel(), x.shape[-1], _ptr(dmin), _ptr(dmax), _ptr(out), _stream())
    return out


def band_update(pden, hist_in, nhist, hist_out, band_width, dmin=None, dmax=None, rows_out=None, smear=True):
    """dmin / dmax given: `pden` and the history hold NORMALISED predictions, denormalised as they are read.
    rows_out / smear: tb_band_update_part (a data-parallel rank's own rows); hist_out may then be None."""
    _need_cuda(pden, hist_in, hist_out)
    if dmin is not None:
        dmin, dmax = _c(dmin, torch.float64), _c(dmax, torch.float64)
    pden = _f32c(pden)
    B, Tf, V = pden.shape
    for h in (hist_in, hist_out):
        if h is not None and (h.shape != (Tf - 1, Tf, V) or not h.is_contiguous() or h.dtype != torch.float32):
            raise RuntimeError("band_update: history buffers must be contiguous float32 (Tf-1,Tf,V)")
    R = B + Tf - 1 if rows_out is None else int(rows_out)
    mk = lambda: torch.empty(R, V, dtype=torch.float64, device=pden.device)
    med, sd, lower, upper = mk(), mk(), mk(), mk()
    _lib.call("tb_band_update_part", _ptr(pden), B, Tf, V, _ptr(hist_in), int(nhist), _ptr(hist_out), _ptr(dmin),
              _ptr(dmax), float(band_width), _ptr(med), _ptr(sd), _ptr(lower), _ptr(upper), R, 1 if smear else 0,
              _stream())
    return med, lower, upper


def dimband_supported(Tf, K):
    """Shapes tb_dimband_update takes (the others go through linear_act + band_update)."""
    return 2 <= Tf <= 48 and K in (32, 64)


def dimband_update(hid, weight, bias, act, hist_in, nhist, band_width, dmin=None, dmax=None, rows_out=None, smear=True):
    """band_update(act(hid . weight^T + bias), ...) in one kernel, the predictions never written to memory: hid (B, Tf, K) the
    generator's last hidden buffer, weight (V, K).  The band history is NOT updated here (see BandAdjust.step_hidden).
    Same results bit for bit as the two-kernel path."""
    _need_cuda(hid, weight, hist_in)
    if dmin is not None:
        dmin, dmax = _c(dmin, torch.float64), _c(dmax, torch.float64)
    hid, weight = _f32c(hid), _f32c(weight)
    bias = None if bias is None else _f32c(bias)
    B, Tf, K = hid.shape
    V = weight.shape[0]
    if weight.shape[1] != K or not dimband_supported(Tf, K):
        raise RuntimeError(f"dimband_update: unsupported shape hid {tuple(hid.shape)} weight {tuple(weight.shape)}")
    if hist_in is not None and (hist_in.shape != (Tf - 1, Tf, V) or not hist_in.is_contiguous() or hist_in.dtype != torch.float32):
        raise RuntimeError("dimband_update: the history must be contiguous float32 (Tf-1,Tf,V)")
    R = B + Tf - 1 if rows_out is None else int(rows_out)
    mk = lambda: torch.empty(R, V, dtype=torch.float64, device=hid.device)
    med, sd, lower, upper = mk(), mk(), mk(), mk()
    _lib.call("tb_dimband_update", _ptr(hid), B, Tf, K, _ptr(weight), _ptr(bias), int(act), V, _ptr(hist_in), int(nhist),
              _ptr(dmin), _ptr(dmax), float(band_width), _ptr(med), _ptr(sd), _ptr(lower), _ptr(upper), R, 1 if smear else 0,
              _stream())
    return med, lower, upper


VARIANTS = {"trainer": 0, "cleaner": 1, "predictor": 2}


def _c(t, dtype):
    if t.dtype != dtype:
        raise RuntimeError(f"expected {dtype}, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def adjust_scan(variant, seed, reals, preds, masks, lower, upper, a, b):
    _need_cuda(seed, reals, preds, masks, lower, upper)
    f64 = torch.float64
    seed, reals, preds, lower, upper = (_c(t, f64) for t in (seed, reals, preds, lower, upper))
    masks = _c(masks, torch.int32)
    n, V = preds.shape
    if reals.shape != (n, V) or masks.shape != (n, V) or lower.shape != (n, V) or upper.shape != (n, V) \
            or seed.numel() != V:
        raise RuntimeError("adjust_scan: shape mismatch")
    target = torch.empty(n, V, dtype=torch.float64, device=preds.device)
    _lib.call("tb_adjust_scan", VARIANTS[variant], _ptr(seed), _ptr(reals), _ptr(preds), _ptr(masks), _ptr(lower),
              _ptr(upper), n, V, float(a), float(b), _ptr(target), _stream())
    return target


def metrics(reals, preds, masks, running=None):
    _need_cuda(reals, preds, masks)
    reals, preds, masks = _c(reals, torch.float64), _c(preds, torch.float64), _c(masks, torch.int32)
    if reals.shape != preds.shape or masks.shape != preds.shape:
        raise RuntimeError("metrics: shape mismatch")
    out = torch.empty(3, dtype=torch.float64, device=preds.device)
    _lib.call("tb_metrics", _ptr(reals), _ptr(preds), _ptr(masks), preds.numel(), _ptr(out), _ptr(running), _stream())
    return out


def metric_sums(reals, preds, masks):
    """The five additive sums behind NME / NMAE / RMSE of one batch's rows (data parallel: summed over ranks)."""
    _need_cuda(reals, preds, masks)
    reals, preds, masks = _c(reals, torch.float64), _c(preds, torch.float64), _c(masks, torch.int32)
    if reals.shape != preds.shape or masks.shape != preds.shape:
        raise RuntimeError("metrics: shape mismatch")
    out = torch.empty(5, dtype=torch.float64, device=preds.device)
    _lib.call("tb_metric_sums", _ptr(reals), _ptr(preds), _ptr(masks), preds.numel(), _ptr(out), _stream())
    return out


def metric_finish(sums, running=None):
    """Scores (k,3) of k batches from their sums (k,5); running[3] += each of them, in order."""
    _need_cuda(sums)
    sums = _c(sums, torch.float64)
    k = sums.shape[0]
    out = torch.empty(k, 3, dtype=torch.float64, device=sums.device)
    _lib.call("tb_metric_finish", _ptr(sums), k, _ptr(out), _ptr(running), _stream())
    return out


def zero_(t):
    """t <- 0 on the current stream (contiguous tensor)."""
    if not t.is_contiguous():
        raise RuntimeError("zero_: contiguous tensors only")
    _lib.call("tb_zero", _ptr(t), t.numel() * t.element_size(), _stream())
    return t


def grad_merge(g, g2):
    """g += g2; g2 <- 0 (flat float32 buffers)."""
    _lib.call("tb_grad_merge", _ptr(g), _ptr(g2), g.numel(), _stream())


def concat2(a, b, b_add=None):
    """torch.cat([a, b], 0) for contiguous tensors of the same trailing shape, as two device copies; with `b_add` (float32, b's
    shape) the second part is b + b_add, formed in the one launch that writes the result."""
    _need_cuda(a, b)
    if a.shape[1:] != b.shape[1:] or a.dtype != b.dtype or not (a.is_contiguous() and b.is_contiguous()):
        raise RuntimeError("concat2: contiguous tensors of the same dtype and trailing shape")
    out = torch.empty((a.shape[0] + b.shape[0],) + tuple(a.shape[1:]), dtype=a.dtype, device=a.device)
    if b_add is not None:
        if b_add.shape != b.shape or a.dtype != torch.float32 or b_add.dtype != torch.float32 or not b_add.is_contiguous():
            raise RuntimeError("concat2: b_add must be a contiguous float32 tensor of b's shape")
        _lib.call("tb_concat2_sum", _ptr(a), a.numel(), _ptr(b), _ptr(b_add), b.numel(), _ptr(out), _stream())
        return out
    _lib.call("tb_concat2", _ptr(a), a.numel() * a.element_size(), _ptr(b), b.numel() * b.element_size(), _ptr(out),
              _stream())
    return out


class _Concat2Fn(torch.autograd.Function):
    """concat2 with the gradient handed back as the two halves (views: no kernel)."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.na = a.shape[0]
        return concat2(a.detach(), b.detach())

    @staticmethod
    def backward(ctx, g):
        return (g[:ctx.na] if ctx.needs_input_grad[0] else None), (g[ctx.na:] if ctx.needs_input_grad[1] else None)


def cat2(a, b):
    if a.is_contiguous() and b.is_contiguous() and a.dtype == b.dtype and a.shape[1:] == b.shape[1:]:
        return _Concat2Fn.apply(a, b)
    return torch.cat([a, b], 0)


def adam_step(p, g, m, v, lr, step, b1=0.9, b2=0.999, eps=1e-8):
    _lib.call("tb_adam_step", _ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), float(lr), b1, b2, eps, int(step),
              _stream())


def rmsprop_step(p, g, sq, lr, alpha=0.99, eps=1e-8, fresh=False):
    _lib.call("tb_rmsprop_step", _ptr(p), _ptr(g), _ptr(sq), p.numel(), float(lr), alpha, eps, int(fresh), _stream())

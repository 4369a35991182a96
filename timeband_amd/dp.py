"""Batch-level data parallelism of the training iteration: one process per GPU over torch.distributed
(backend "nccl" = RCCL over xGMI on MI355X).

The reference is single-device (source/utils/initiate.py:21-28 picks cuda:0 or cpu; no collective
anywhere), so nothing here has a counterpart there; the scheme is SURVEY.md §8(e):

  * rank r owns a CONTIGUOUS slice of every global batch of windows (`shard`): contiguity keeps the
    band / adjust post-processing a halo exchange with the neighbour instead of a gather;
  * parameters are replicated; gradients are summed with all-reduces over slices of each network's
    flat gradient buffer (`GradReducer`: buckets in the order BPTT completes them, issued on a side
    stream as soon as a bucket's last backward kernel is enqueued);
  * the loss-gradient kernels divide by GLOBAL counts (`count_scale`), so plain sums are exact;
  * the reconstruction sums behind the batch-global gradient-penalty norm are all-reduced
    asynchronously under the discriminator pass (`all_reduce_async`);
  * band: the predictions of a rank's last forecast_len-1 windows go to the right neighbour
    (`ring_shift`); adjust: a one-row carry travels down the ranks (`send_row` / `recv_row`).

Everything is expressed over three primitives (sum all-reduce, ring shift, point-to-point row) so that the
same host logic runs over RCCL on GPUs and over gloo in the CPU tests.  With gloo and device tensors the
data is staged through the host (a debugging path: two ranks on one GPU).
"""
from __future__ import annotations

import contextlib
import os

import torch
import torch.distributed as dist

_REPLICATED = False
_SCALE = None          # (n_local, n_global) of the batch being processed, or None = equal shards
_FORCE = os.environ.get("TB_DP_FORCE", "0") == "1"   # take the collective paths even with one rank (tests)


def init(backend="nccl", **kw):
    """torch.distributed.init_process_group with this path's RCCL settings (call it instead, BEFORE any collective).
    NCCL_MAX_NCHANNELS (kept if the environment already sets it): RCCL runs one workgroup per channel, and its
    all-reduces run on the communication stream BESIDE the clustered recurrent kernels, which need all their workgroups
    resident.  The messages here are 1.2 MB (D) and 2.2 MB (G) of gradients in buckets of
    >= 256 KB plus 16-byte loss sums -- latency-bound on xGMI, not bandwidth-bound -- so 8 channels (one per XCD; at
    least one ring per xGMI link of the 7) lose nothing and bound RCCL's footprint to 8 of the 256 CUs."""
    os.environ.setdefault("NCCL_MAX_NCHANNELS", "8")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if not dist.is_initialized():
        dist.init_process_group(backend, **kw)
    return dist.get_rank(), dist.get_world_size()


def initialized() -> bool:
    return dist.is_available() and dist.is_initialized()


def world() -> int:
    """Ranks that share the batch being processed (1 inside `replicated()`)."""
    if _REPLICATED or not initialized():
        return 1
    return dist.get_world_size()


def ranks() -> int:
    """Processes in the job, whatever the batch being processed is (1 if torch.distributed is not initialised; the
    forced single-rank test mode counts as data parallel through active())."""
    return dist.get_world_size() if initialized() else 1


def is_replicated() -> bool:
    return _REPLICATED


def rank() -> int:
    return dist.get_rank() if initialized() else 0


def active() -> bool:
    """True when collectives have to be issued for the batch being processed."""
    if _REPLICATED or not initialized():
        return False
    return dist.get_world_size() > 1 or _FORCE


def force(on: bool = True):
    global _FORCE
    _FORCE = bool(on)


@contextlib.contextmanager
def replicated():
    """Every rank runs the SAME batch (e.g. a critic batch that does not divide by the world size):
    losses use local counts and nothing is all-reduced -- the kernels are deterministic, so all ranks
    take bit-identical steps."""
    global _REPLICATED
    old, _REPLICATED = _REPLICATED, True
    try:
        yield
    finally:
        _REPLICATED = old


@contextlib.contextmanager
def batch_counts(n_local: int, n_global: int):
    """Window counts of the batch being processed: the loss kernels scale by GLOBAL element counts, which is
    local count x n_global / n_local (exact: every window has the same number of elements)."""
    global _SCALE
    old, _SCALE = _SCALE, (int(n_local), int(n_global))
    try:
        yield
    finally:
        _SCALE = old


def count_scale(numel_local: int) -> float:
    """Global element count for a tensor with `numel_local` elements on this rank."""
    if _REPLICATED or not initialized():
        return float(numel_local)
    if _SCALE is not None:
        n_local, n_global = _SCALE
        return float(numel_local // max(n_local, 1) * n_global) if n_local else 0.0
    return float(numel_local * dist.get_world_size())


def shard(n: int, r: int | None = None, w: int | None = None):
    """[lo, hi) of rank r's contiguous slice of n consecutive windows (ragged tails: sizes differ by <= 1)."""
    r = rank() if r is None else r
    w = (dist.get_world_size() if initialized() else 1) if w is None else w
    return n * r // w, n * (r + 1) // w


# ---------------------------------------------------------------------------------------------------
# primitives
# ---------------------------------------------------------------------------------------------------
_comm_streams = {}


def _comm_stream(device):
    s = _comm_streams.get(device)
    if s is None:
        s = _comm_streams[device] = torch.cuda.Stream(device)
    return s


def _staged(t) -> bool:
    return t.is_cuda and dist.get_backend() != "nccl"


class _Handle:
    """An all-reduce in flight.  wait(): the CURRENT stream (the host too, for host-staged backends)
    continues only after the sum has landed in the tensor."""

    def __init__(self, work=None, tensor=None, host=None):
        self.work, self.tensor, self.host = work, tensor, host

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
        if self.host is not None:
            self.tensor.copy_(self.host)
            self.host = None


def all_reduce_async(t, after=None) -> _Handle:
    """Sum all-reduce of `t` in place, issued on the communication side stream so that the stream that
    produced `t` keeps running.  The collective starts after `after` (a torch.cuda.Event) or, by default,
    after the work enqueued so far on the current stream."""
    if not initialized():
        return _Handle()
    if t.is_cuda and not _staged(t):
        cur = torch.cuda.current_stream(t.device)
        if after is None:
            after = torch.cuda.Event()
            after.record(cur)
        cs = _comm_stream(t.device)
        with torch.cuda.stream(cs):
            cs.wait_event(after)
            work = dist.all_reduce(t, async_op=True)      # RCCL orders itself after the stream current here
        return _Handle(work, t)
    if t.is_cuda:                                          # gloo + device tensor: through the host
        if after is not None:
            after.synchronize()
        else:
            torch.cuda.current_stream(t.device).synchronize()
        host = t.detach().cpu()
        dist.all_reduce(host)
        return _Handle(None, t, host)
    return _Handle(dist.all_reduce(t, async_op=True), t)


def all_reduce(t):
    all_reduce_async(t).wait()
    return t


def ring_shift(send, recv):
    """send -> rank+1, recv <- rank-1 (wrapping); both tensors of the same shape.  One matched pair per
    rank, so it cannot deadlock whatever the backend's buffering."""
    w = dist.get_world_size()
    r = dist.get_rank()
    if w == 1:
        recv.copy_(send)
        return recv
    if _staged(send):
        torch.cuda.current_stream(send.device).synchronize()
        hs, hr = send.detach().cpu(), torch.empty(recv.shape, dtype=recv.dtype)
        reqs = dist.batch_isend_irecv([dist.P2POp(dist.isend, hs, (r + 1) % w), dist.P2POp(dist.irecv, hr, (r - 1) % w)])
        for q in reqs:
            q.wait()
        recv.copy_(hr)
        return recv
    reqs = dist.batch_isend_irecv([dist.P2POp(dist.isend, send, (r + 1) % w), dist.P2POp(dist.irecv, recv, (r - 1) % w)])
    for q in reqs:
        q.wait()
    return recv


def send_row(t, dst):
    if _staged(t):
        torch.cuda.current_stream(t.device).synchronize()
        dist.send(t.detach().cpu(), dst)
    else:
        dist.send(t, dst)


def recv_row(t, src):
    if _staged(t):
        h = torch.empty(t.shape, dtype=t.dtype)
        dist.recv(h, src)
        t.copy_(h)
    else:
        dist.recv(t, src)
    return t


def broadcast(t, src=0):
    """In place, from rank `src`."""
    if _staged(t):
        torch.cuda.current_stream(t.device).synchronize()
        h = t.detach().cpu()
        dist.broadcast(h, src)
        t.copy_(h)
    else:
        dist.broadcast(t, src)
    return t


def all_gather_cat(t):
    """Concatenation over ranks along dim 0 (shards may differ in length)."""
    w = dist.get_world_size()
    n = torch.tensor([t.shape[0]], dtype=torch.int64)
    sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(w)]
    if dist.get_backend() == "nccl":
        n = n.to(t.device)
        sizes = [s.to(t.device) for s in sizes]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    m = max(sizes)
    src = t.detach()
    staged = _staged(src)
    if staged:
        torch.cuda.current_stream(src.device).synchronize()
        src = src.cpu()
    pad = torch.zeros((m,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    pad[: src.shape[0]] = src
    parts = [torch.empty_like(pad) for _ in range(w)]
    dist.all_gather(parts, pad)
    out = torch.cat([p[:k] for p, k in zip(parts, sizes)], 0)
    return out.to(t.device) if staged else out


# ---------------------------------------------------------------------------------------------------
# bucketed gradient all-reduce
# ---------------------------------------------------------------------------------------------------
# Exposed communication: how long the stream that needs the reduced gradients had to WAIT for them (bench.py
# --gpus N reports it).  With timing on, _FlatOptimizer.all_reduce_grads() brackets its wait with two events.
_COMM_EVENTS = None


def comm_timing(on: bool):
    global _COMM_EVENTS
    _COMM_EVENTS = [] if on else None


def comm_bracket(device):
    """An event on the current stream now (None if timing is off); pair it with comm_bracket_end()."""
    if _COMM_EVENTS is None or device.type != "cuda":
        return None
    ev = torch.cuda.Event(enable_timing=True)
    ev.record(torch.cuda.current_stream(device))
    return ev


def comm_bracket_end(begin, device):
    if begin is not None and _COMM_EVENTS is not None:
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream(device))
        _COMM_EVENTS.append((begin, ev))


def comm_wait_ms() -> float:
    """Sum of the bracketed waits so far (synchronise the device first); clears the list."""
    global _COMM_EVENTS
    if not _COMM_EVENTS:
        return 0.0
    total = sum(b.elapsed_time(e) for b, e in _COMM_EVENTS)
    _COMM_EVENTS = []
    return total


class GradReducer:
    """Sum all-reduce of one network's flat gradient buffer in buckets, overlapped with BPTT.

    `flat.buckets` lists contiguous [lo, hi) ranges of the buffer in the order backward completes them
    (generator: dim_layer + seq_layer + lstm2, then lstm1, then lstm0 -- SURVEY.md §5).  While armed, the
    backward ops report the parameters whose gradients they have just enqueued (ops.grads_enqueued); when
    a bucket's last parameter has reported, an event is recorded on that stream and the bucket's
    all-reduce is issued on the communication stream behind it -- the earlier layers' BPTT kernels keep
    running on theirs.  finish() issues whatever has not fired (ops that bypass the flat slots) and makes
    the current stream wait for every bucket."""

    def __init__(self, flat):
        self.flat = flat
        self.pending = None
        self.handles = []
        self.fired = None

    def arm(self, visits=1):
        """visits: how often the armed backward pass reaches each parameter (1 unless a module is applied more than
        once in the graph, e.g. LSTMDiscriminator.pair()'s two-pass fallback for unequal shapes)."""
        from . import ops
        self.pending = [b["nparams"] * int(visits) for b in self.flat.buckets]
        self.fired = [False] * len(self.flat.buckets)
        self.handles = []
        ops._REDUCERS.append(self)
        ops.reducers_changed()

    def notify(self, params):
        if self.pending is None:
            return
        for p in params:
            if p is None or getattr(p, "_tb_flat", None) is not self.flat:
                continue
            b = p._tb_bucket
            if self.fired[b]:
                # the bucket's all-reduce is already in flight on the communication stream and these kernels add into
                # the slice it reads and writes: a race and a wrong sum that nothing downstream could repair
                raise RuntimeError("dp.GradReducer: a parameter's gradient was produced again after its bucket's "
                                   "all-reduce had been issued (a module applied more than once in the armed backward "
                                   "pass): arm the reducer with visits=<times each parameter is reached>")
            self.pending[b] -= 1
            if self.pending[b] == 0 and not self.fired[b]:
                self._fire(b)

    def _fire(self, b):
        bk = self.flat.buckets[b]
        self.fired[b] = True
        self.handles.append(all_reduce_async(self.flat.grad[bk["lo"]:bk["hi"]]))

    def finish(self):
        from . import ops
        if self.pending is None:
            return
        if self in ops._REDUCERS:
            ops._REDUCERS.remove(self)
            ops.reducers_changed()
        for b in range(len(self.flat.buckets)):
            if not self.fired[b]:
                self._fire(b)
        for h in self.handles:
            h.wait()
        self.pending, self.handles = None, []

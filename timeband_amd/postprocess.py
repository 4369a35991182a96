"""Device-resident band / adjust state shared by trainer, cleaner and predictor.

Restates the per-batch tail of the reference (source/trainer.py:202-212, cleaner.py:88-106,
predictor.py:86-96) without leaving the GPU: denormalise -> band (median/std over horizons, HIP)
-> adjust (sequential scan, HIP).  The reference's growing `outputs` array is replaced by absolute
row positions: target row p of the batch that starts at window idx is data row To + idx + p.

Data parallel (no counterpart in the reference).  Rank r holds the predictions of a
contiguous slice [lo, lo+b) of the global batch, so it owns the band / adjust rows [idx+lo, idx+lo+b) --
the last rank also the forecast_len-1 overhang rows behind the batch.  A row's band needs the predictions
of the forecast_len-1 windows before it: every rank sends its last forecast_len-1 windows' predictions to its
right neighbour (one ring shift; what wraps round to rank 0 is the history for the NEXT global batch).
The adjust scan is a chain along the rows: its one-row carry travels down the ranks, and from the last
rank back to rank 0 as the next batch's seed row (the reference seeds each batch with the LAST overhang
row of the previous one, trainer.py:209).  Slices shorter than forecast_len-1 windows fall back to an
all-gather of the predictions and the single-rank arithmetic on every rank.
"""
import os

import torch

from . import dp, ops


class BandAdjust:
    def __init__(self, dataset, variant, band_width, missing_gamma, anomaly_gamma, keep_outputs=False, out_dtype=None):
        """keep_outputs: assemble the whole cleaned series and its bands on the device (clean / predict).
        out_dtype: their storage type -- float64 like the reference's arrays by default; float32 when three
        float64 copies of the series would not leave room in HBM (512 variables x 10M rows: 123 GB)."""
        self.ds, self.variant = dataset, variant
        self.w, self.a, self.b = band_width, missing_gamma, anomaly_gamma
        self.keep = keep_outputs
        self.fuse_last_layer = os.environ.get("TB_FUSE_DIMBAND", "1") != "0"   # step_windows: output layer inside the band kernel
        if out_dtype is None:
            dev = dataset.series_dev.device
            big = dev.type == "cuda" and 3 * dataset.data_length * dataset.decode_dim * 8 > \
                0.3 * torch.cuda.get_device_properties(dev).total_memory
            out_dtype = torch.float32 if big else torch.float64
        self.out_dtype = out_dtype
        self.reset()

    def reset(self):
        """pred_initate + idx = 0 (trainer.py:101-102,259-264)."""
        d = self.ds
        Tf, V = d.forecast_len, d.decode_dim
        dev = d.series_dev.device
        self.hist = [torch.zeros(Tf - 1, Tf, V, dtype=torch.float32, device=dev) for _ in range(2)]
        self.cur = 0
        self.nhist = 0
        self.idx = 0
        self._rank0_state = False      # data parallel: after a sharded step only rank 0 holds hist / last_row
        self.full_rows = True          # the last step's rows: all of the batch's (True) or this rank's own (sharded)
        self.restart_pass()

    def restart_pass(self):
        """`outputs = self.dataset.observed` at the top of every pass (trainer.py:137, cleaner.py:71-73)."""
        d = self.ds
        self.last_row = None
        if self.keep:
            n0 = d.observed_dev.shape[0]
            mk = lambda: torch.cat([d.observed_dev.to(self.out_dtype),
                                    torch.zeros(d.data_length - n0, d.decode_dim, dtype=self.out_dtype,
                                                device=d.observed_dev.device)])
            self.outputs, self.lower_bands, self.upper_bands = mk(), mk(), mk()

    def step(self, fake_y, part=None):
        """part (data parallel): (lo, n, replicated) -- fake_y holds windows [lo, lo + len(fake_y)) of the global
        batch of n windows that starts at self.idx (replicated: every rank holds all n)."""
        if part is not None and (dp.ranks() > 1 or dp.active()):
            if not part[2]:
                lo, n = int(part[0]), int(part[1])
                Tf = fake_y.shape[1]
                W = dp.ranks()
                if min(dp.shard(n, r, W)[1] - dp.shard(n, r, W)[0] for r in range(W)) >= Tf - 1:
                    return self._step_shard(fake_y, lo, n)
                fake_y = dp.all_gather_cat(fake_y.detach().contiguous())     # short slices: the whole batch everywhere
            # a step that every rank runs in full (the all-gather fallback above, or a replicated tail batch with
            # fewer windows than ranks): everybody needs the state that sharded steps left on rank 0 only
            self._sync_state()
        return self._step_all(fake_y)

    def step_windows(self, netG, series, first, count, stride=1, part=None):
        """One cleaning / prediction batch from window POSITIONS (the windows whose forecasts start at series rows first,
        first + stride, ...): generator forward over the windows in place, then step().  Where the shapes allow it
        (ops.dimband_supported, single rank, whole forecasts used) the generator's output layer runs INSIDE the band
        kernel and the (count, Tf, V) predictions are never written: same results bit for bit."""
        Tf = self.ds.forecast_len
        fuse = (self.fuse_last_layer and part is None and netG.outseq_len == Tf and count >= Tf - 1 and series.is_cuda
                and ops.dimband_supported(Tf, netG.hidden_dim))
        if not fuse:
            return self.step(netG.infer_windows(series, first, count, stride)[:, :Tf], part)
        return self.step_hidden(netG, netG.infer_windows(series, first, count, stride, hidden=True))

    def sweep(self, netG, series, stride=1):
        """A pipelined pass over window positions (WindowSweep): the generator forward of batch i + 1 runs on a side stream
        under the band / adjust of batch i."""
        return WindowSweep(self, netG, series, stride)

    def step_hidden(self, netG, hid):
        """_step_all for predictions that exist only as the generator's last hidden buffer `hid` (B, Tf, hidden)."""
        d = self.ds
        self.full_rows = True
        B, Tf, K = hid.shape
        dim = netG.dim_layer[0]
        preds, lower, upper = ops.dimband_update(hid, dim.weight.detach(), dim.bias.detach(), ops.ACT_TANH, self.hist[self.cur],
                                                 self.nhist, self.w, d.dmin_dev, d.dmax_dev)
        # the next batch's history: the last Tf-1 windows' predictions, from the same layer on just those windows
        self.hist[self.cur ^ 1] = netG.last_layer(hid[B - (Tf - 1):]).contiguous()
        self.cur ^= 1
        self.nhist = Tf - 1
        return self._adjust_all(B, Tf, preds, lower, upper)

    def _sync_state(self):
        """Before a step that every rank runs in full: the band history and the adjust seed row live on rank 0
        after sharded steps (the ring shift and the carry end there) -- hand them to everybody."""
        if not self._rank0_state:
            return
        d = self.ds
        dp.broadcast(self.hist[self.cur], 0)
        self.nhist = d.forecast_len - 1
        # `last_row is None` means "seed = the last observed row" (trainer.py:137,209; restart_pass clears it on every
        # rank at the top of a pass while the band history stays on rank 0): rank 0 sends that row itself, so what
        # the others receive is always a defined seed
        row = d.observed_dev[-1] if self.last_row is None else self.last_row
        row = row.to(torch.float64).clone().contiguous()
        self.last_row = dp.broadcast(row, 0)
        self._rank0_state = False

    def _step_all(self, fake_y):
        d = self.ds
        self.full_rows = True
        B, Tf, V = fake_y.shape
        # the history holds normalised predictions: the band kernel denormalises as it gathers
        preds, lower, upper = ops.band_update(fake_y.detach(), self.hist[self.cur], self.nhist, self.hist[self.cur ^ 1],
                                              self.w, d.dmin_dev, d.dmax_dev)
        self.cur ^= 1
        self.nhist = min(Tf - 1, self.nhist + B)
        return self._adjust_all(B, Tf, preds, lower, upper)

    def _adjust_all(self, B, Tf, preds, lower, upper):
        d = self.ds
        n = B + Tf - 1
        if self.idx + n > d.forecast_dev.shape[0]:
            raise RuntimeError("post-processing ran past the end of the series")
        reals = d.forecast_dev[self.idx:self.idx + n]
        masks = d.missing_dev[self.idx:self.idx + n]
        seed = d.observed_dev[-1] if self.last_row is None else self.last_row
        target = ops.adjust_scan(self.variant, seed, reals, preds, masks, lower, upper, self.a, self.b)
        self.last_row = target[-1]
        self._keep(self.idx, target, lower, upper)
        self.idx += B
        return reals, preds, masks, lower, upper, target

    def _keep(self, first, target, lower, upper):
        if self.keep:
            r0 = self.ds.observed_len + first
            n = target.shape[0]
            self.outputs[r0:r0 + n] = target
            self.lower_bands[r0:r0 + n] = lower
            self.upper_bands[r0:r0 + n] = upper

    def _step_shard(self, fake_y, lo, n):
        """This rank's rows of the global batch [idx, idx + n): halo in from the left, carry down the ranks."""
        d = self.ds
        b, Tf, V = fake_y.shape
        W, r = dp.world(), dp.rank()
        last = r == W - 1
        fake_y = fake_y.detach().contiguous()
        # band halo: my last Tf-1 windows -> right neighbour; rank 0 keeps what wraps round for the next batch
        halo = torch.empty(Tf - 1, Tf, V, dtype=torch.float32, device=fake_y.device)
        dp.ring_shift(fake_y[b - (Tf - 1):], halo)
        if r == 0:
            hist_in, nhist = self.hist[self.cur], self.nhist
            self.hist[self.cur ^ 1] = halo
            self.cur ^= 1
            self.nhist = Tf - 1
        else:
            hist_in, nhist = halo, Tf - 1
        rows = b + Tf - 1 if last else b
        first = self.idx + lo
        if first + rows > d.forecast_dev.shape[0]:
            raise RuntimeError("post-processing ran past the end of the series")
        preds, lower, upper = ops.band_update(fake_y, hist_in, nhist, None, self.w, d.dmin_dev, d.dmax_dev,
                                              rows_out=rows, smear=last)
        reals = d.forecast_dev[first:first + rows]
        masks = d.missing_dev[first:first + rows]
        # adjust carry: the row before my first one
        if r == 0:
            seed = d.observed_dev[-1] if self.last_row is None else self.last_row
        else:
            seed = dp.recv_row(torch.empty(V, dtype=torch.float64, device=fake_y.device), r - 1)
        target = ops.adjust_scan(self.variant, seed, reals, preds, masks, lower, upper, self.a, self.b)
        if W == 1:
            self.last_row = target[-1]
        else:
            dp.send_row(target[-1].contiguous(), (r + 1) % W)
            if r == 0:  # the last rank's last overhang row seeds the next global batch (trainer.py:209)
                self.last_row = dp.recv_row(torch.empty(V, dtype=torch.float64, device=fake_y.device), W - 1)
        self._keep(first, target, lower, upper)
        self.idx += n
        self._rank0_state = W > 1
        self.full_rows = False
        return reals, preds, masks, lower, upper, target


class WindowSweep:
    """The cleaning / prediction sweep as a two-stage pipeline over its batches.  The batches of a sweep are independent
    through the generator -- it reads the series and its own parameters, nothing a previous batch wrote -- while the band
    history and the adjust carry chain them on the post-processing side (source/cleaner.py:88-106).  So the generator
    forward of batch i + 1 is queued on a side stream BEFORE the band / adjust of batch i on the caller's stream:
    the latency-bound recurrent kernels of one batch then share the chip with the vector-bound band kernel of the other.
    Same kernels, same arguments, same order per batch: the results do not change.

        sweep = post.sweep(netG, series, stride)
        for first, count in positions: sweep.push(first, count)
        sweep.flush()

    TB_CLEAN_PIPELINE side streams (default 3: two forwards and one band / adjust in flight -- the recurrent kernels are
    latency-bound at one workgroup per CU, two of them fill each other's gaps; 0 = no pipeline); a side stream's next
    forward waits for the post-processing of its previous batch, so the look-ahead costs that many batches'
    intermediates, not the sweep's."""

    def __init__(self, post, netG, series, stride=1):
        self.post, self.netG, self.series, self.stride = post, netG, series, stride
        depth = int(os.environ.get("TB_CLEAN_PIPELINE", "3"))     # side streams (0: no pipeline, everything on the caller's stream)
        self.pipelined = series.is_cuda and depth > 0
        self.depth = max(1, depth)
        self._side = None
        self._pending = []                    # forwards queued and not yet post-processed: (tensor, fused?, ready event, side stream)
        self._done = [None] * self.depth      # post-processing event of the last batch of each side stream
        self._n = 0

    def _fuse(self, count):
        p, g = self.post, self.netG
        Tf = p.ds.forecast_len
        return (p.fuse_last_layer and g.outseq_len == Tf and count >= Tf - 1 and self.series.is_cuda
                and ops.dimband_supported(Tf, g.hidden_dim))

    def _forward(self, first, count):
        fuse = self._fuse(count)
        out = self.netG.infer_windows(self.series, first, count, self.stride, hidden=fuse)
        return (out if fuse else out[:, :self.post.ds.forecast_len]), fuse

    def _post(self, out, fuse):
        return self.post.step_hidden(self.netG, out) if fuse else self.post.step(out)

    def push(self, first, count):
        if not self.pipelined:
            return self._post(*self._forward(first, count))
        dev = self.series.device
        main = torch.cuda.current_stream(dev)
        if self._side is None:
            self._side = tuple(torch.cuda.Stream(dev) for _ in range(self.depth))
            for s_ in self._side:
                s_.wait_stream(main)           # whatever prepared the series / the parameters on the caller's stream
        k = self._n % self.depth
        side = self._side[k]
        if self._done[k] is not None:
            side.wait_event(self._done[k])     # look-ahead bound: this stream's previous batch has been post-processed
        with torch.cuda.stream(side):
            out, fuse = self._forward(first, count)
            ready = torch.cuda.Event()
            ready.record(side)
        out.record_stream(main)
        self._pending.append((out, fuse, ready, k))
        self._n += 1
        # post-process what is old enough: `depth - 1` forwards stay queued in front of the band / adjust chain
        res = None
        while len(self._pending) >= self.depth:
            res = self._drain()
        return res

    def _drain(self):
        out, fuse, ready, k = self._pending.pop(0)
        main = torch.cuda.current_stream(self.series.device)
        main.wait_event(ready)
        res = self._post(out, fuse)
        ev = torch.cuda.Event()
        ev.record(main)
        self._done[k] = ev
        return res

    def flush(self):
        """Post-process the batches still queued (the sweep's tail)."""
        res = None
        while self._pending:
            res = self._drain()
        return res

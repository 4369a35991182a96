"""TIMEBANDTrainer on the HIP hot path (reference source/trainer.py:18-305).

Same constructor, config keys, `train` / `train_step` / `model_update` behaviour and step
semantics as the reference's Gen-A trainer, including its quirks:
  * critic loop with a FRESH RMSprop per batch (trainer.py:142-158); the reference's non-detached
    fake_y there only produces G gradients that are zeroed unread, so they are not computed;
  * `zero_grad` of D and G once per iteration, before the D phase (:160-161), so the generator
    step uses D-phase + G-phase gradients;
  * Adam re-created every epoch (:95-98);
  * generator adversarial term against the FAKE label (loss.py:94).
What changes is where the work runs: windows are gathered on the device, losses / optimizers /
band / adjust / scores are HIP kernels, and nothing is copied to the host inside the batch loop
(the reference syncs on `.cpu()` and `.numpy()` every batch, trainer.py:202,223-234).
Data parallel (dp.py): one process per GPU, every rank holds the same parameters and steps on its
contiguous slice of each global batch (`trainset.loader(batch_size, rank, world)`); each network's flat
gradient buffer is summed in buckets, in the order BPTT completes them, on a communication stream
(RCCL over xGMI) while the earlier layers' BPTT still runs; the two sums behind the batch-global
gradient-penalty norm are all-reduced under the discriminator pass; band / adjust exchange a halo with the
neighbouring rank (postprocess.py).
"""
import os

import torch

from . import dp, ops
from .loss import TIMEBANDLoss
from .metric import TIMEBANDMetric
from .model import TIMEBANDModel
from .optim import FusedAdam, FusedRMSprop
from .postprocess import BandAdjust


class TIMEBANDTrainer:
    def __init__(self, config: dict, dataset, models: TIMEBANDModel, metric: TIMEBANDMetric,
                 losses: TIMEBANDLoss, dashboard=None) -> None:
        self.logger = config["logger"]
        self.dataset, self.models, self.metric, self.losses, self.dashboard = dataset, models, metric, losses, dashboard
        self.set_config(config)
        self.forecast_len = self.dataset.forecast_len
        self.post = BandAdjust(dataset, "trainer", self.band_width, self.missing_gamma, self.anomaly_gamma)
        self.optimD = self.optimG = None
        # The D-phase generator forward depends only on G's parameters, which do not change until the very
        # end of the iteration: it runs on a side stream under the critic step; the graph is cut at fake_y
        # so the generator's backward passes run on side streams too (latency-bound persistent kernels
        # leave most of the chip idle).  TB_OVERLAP is a bit mask (see iteration()); 0 = one stream.
        # Default 5: putting the G-phase forward under the D-phase backward as well (bit 2) measured
        # 1.4 % slower -- the two stretch each other by more than the forward takes.
        # Bit 16 (on by default, any number of streams): the two generator forward passes of an iteration -- the same
        # input through the same parameters, trainer.py:173,188 -- as ONE pass over 2B rows (LSTMGenerator.twin).
        # Bit 8 with it (default since the twin pass: 6.84 -> 6.75 ms/step): the generator phase is short now, so the
        # D-phase generator backward is still running when the G-phase one could start.
        # Bit 32 (default; replaces 4 and 8): the generator's two backward passes as ONE pass over the 2B rows of the twin
        # forward, behind the generator phase's discriminator pass -- the recurrent kernels cost the same for 2B rows
        # as for B (three fewer latency-bound launches, no co-running in the generator phase): 6.73 -> 6.21 ms/step.
        # The two passes' parameter gradients are then summed in one reduction over 2B rows instead of one after the
        # other: the same sum up to fp32 association (TB_OVERLAP=29 keeps them apart, bit-identical to one stream).
        ov = os.environ.get("TB_OVERLAP", "49")
        self.overlap_mask = int(ov) if ov.isdigit() else 49
        self.overlap = (self.overlap_mask & 15) != 0
        # TB_SHARE_GPASS=1 (off by default; not the reference's schedule): G's parameters are the same in
        # the D phase and the G phase, so netG(true_x) is the same tensor both times and backward is linear
        # in the upstream gradient -- one generator forward and ONE backward with the summed gradient give
        # the same update up to fp32 rounding of that sum.
        self.share_gpass = os.environ.get("TB_SHARE_GPASS", "0") == "1"
        # TB_WGRAD_OVERLAP=1 (default off: measured zero-sum, 9.10 vs 9.06 ms -- the GEMMs and the BPTT kernels of the
        # layers below slow each other by what the overlap gains): the LSTM weight-gradient GEMMs of every backward pass go to a side
        # stream (ops.Runtime.wgrad_stream); the optimizers join it before they read the gradients
        self.wgrad_overlap = os.environ.get("TB_WGRAD_OVERLAP", "0") == "1"
        self._side = None
        # band / adjust / scores of a batch on a stream of their own, under the NEXT iteration (one rank; TB_POST_OVERLAP=0: off)
        self.post_overlap = os.environ.get("TB_POST_OVERLAP", "1") != "0"
        self._post_stream = None
        self._ones = {}
        self._marks = None      # tools/phase_times.py: list of (name, event on the main stream) per iteration

    def _one(self, loss):
        """d(loss)/d(loss) = 1 from a cached device scalar: backward() without autograd's ones-fill launch."""
        one = self._ones.get(loss.device)
        if one is None:
            one = self._ones[loss.device] = torch.ones((), dtype=loss.dtype, device=loss.device)
        return one

    def _mark(self, name, dev):
        if self._marks is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream(dev))
            self._marks.append((name, ev))

    def set_config(self, config: dict = None) -> None:
        self.__dict__ = {**config, **self.__dict__}
        lr = config["learning_rate"]
        self.lr, self.lr_decay = lr["base"], lr["decay"]
        self.lr_gammaG, self.lr_gammaD = lr["gammaG"], lr["gammaD"]
        m = config["models"]
        self.reload_option, self.reload_interval, self.save_interval = m["reload"], m["reload_interval"], m["save_interval"]
        self.reload_counts = 0

    # ------------------------------------------------------------------ epoch level
    def model_update(self, epoch: int, score: float) -> None:
        """Checkpoint / reload policy of trainer.py:65-85."""
        if score < self.models.best_score:
            self.reload_counts = -1
            self.models.best_score = score
            self.models.save(f"{score:.3f}", best=True)
        if (epoch + 1) % self.save_interval == 0:
            self.models.save()
        if self.reload_option:
            self.reload_counts += 1
            if self.reload_counts >= self.reload_interval:
                self.reload_counts = 0
                self.logger.info(f" - Learning rate decay {self.lr} to {self.lr * self.lr_decay}")
                self.lr *= self.lr_decay
                self.models.load("BEST")

    def new_epoch(self):
        """Fresh Adam for D and G, idx = 0, empty band state (trainer.py:95-102)."""
        self.optimD = FusedAdam(self.models.netD, self.lr * self.lr_gammaD)
        self.optimG = FusedAdam(self.models.netG, self.lr * self.lr_gammaG)
        self.idx = 0
        self.pred_initate()

    def pred_initate(self):
        self.post.reset()

    def train(self, trainset, validset) -> None:
        self.logger.info("Train the model")
        valid_score, epoch = float("inf"), -1
        for epoch in range(self.epochs):
            self.new_epoch()
            train_score = self.train_step(epoch, trainset, training=True)
            valid_score = self.train_step(epoch, validset, training=False)
            if (epoch + 1) % self.print_interval == 0:
                self.logger.info(f"[Ep {epoch + 1}] T ({train_score:.4f}) V ({valid_score:.4f})", level=1)
            if train_score - valid_score < train_score * 0.25:
                self.model_update(epoch, valid_score)
        self.model_update(epoch, valid_score)

    # ------------------------------------------------------------------ one iteration of the hot path
    def critic_step(self, true_x, true_y):
        """trainer.py:143-158 for one draw."""
        netD, netG = self.models.netD, self.models.netG
        # The reference keeps fake_y's graph here (trainer.py:147-152), so errD.backward() also fills
        # G's gradients -- which optimG.zero_grad() (:161) discards before anything reads them.  Dead
        # work with no effect on any result: G runs without a graph (and without the BPTT stash).
        with torch.no_grad():
            fake_y = netG(true_x)
        self._critic_opt.zero_grad()
        Dy, DGx = netD.pair(true_y, fake_y)
        errD = self.losses.dis_loss(true_y, fake_y, Dy, DGx, critic=True)
        self._critic_opt.begin_reduce()
        errD.backward(self._one(errD))
        self._critic_opt.all_reduce_grads()
        self._critic_opt.step()

    @staticmethod
    def _shard_critic(cx, cy):
        """Data parallel: every rank draws the same critic batch (same numpy seed).  If it has at least one
        window per rank each rank keeps its contiguous slice (sharded, gradients all-reduced); otherwise all
        ranks run the whole batch redundantly (replicated, nothing reduced)."""
        if not dp.active():
            return cx, cy, True
        n = cx.shape[0]
        lo, hi = dp.shard(n)
        if n < dp.world():
            return cx, cy, False
        return cx[lo:hi].contiguous(), cy[lo:hi].contiguous(), True

    def iteration(self, true_x, true_y, training=True, critic_batches=None, n_global=None):
        """trainer.py:142-194.  `critic_batches`: iterable of (x, y) critic draws (default: `critic`
        draws from dataset.get_random()).  Data parallel: true_x / true_y are this rank's slice of a global
        batch of `n_global` windows (default: equal slices on every rank).  Returns the G-phase fake_y."""
        netD, netG = self.models.netD, self.models.netG
        overlap = (self.overlap_mask & 15) != 0 and training and true_x.is_cuda
        ops.Runtime.wgrad_overlap = bool(self.wgrad_overlap and training and true_x.is_cuda)
        # which parts overlap (TB_OVERLAP bit mask, default 5): 1 = D-phase generator forward under the
        # critic step, 2 = G-phase generator forward under the D-phase backward, 4 = graph cut at fake_y
        # (the generator's share of errD.backward() runs on the stream of its forward -- the side stream
        # with bit 1 -- under the G phase's forward and discriminator pass)
        # 8 (needs 4) = the generator's two backward passes at once: the G-phase one adds into a second gradient buffer
        # (optim.FlatParams.alt), merged before the step -- the same two-term sum bit for bit
        ov1, ov2, ov4, ov8 = ((overlap and bool(self.overlap_mask & b)) for b in (1, 2, 4, 8))
        ov8 = ov8 and ov4
        share = self.share_gpass and training
        twin = training and bool(self.overlap_mask & 16) and not share and true_x.is_cuda
        # bit 32 (needs 16): ONE generator backward pass over the 2B rows of the twin forward, behind the generator phase's
        # discriminator pass, instead of one per phase (LSTMGenerator.twin_joint)
        joint = twin and bool(self.overlap_mask & 32)
        if share or twin:
            ov2 = False
        if joint:
            ov4 = ov8 = False
        fake_y2 = out2 = None
        if overlap:
            main = torch.cuda.current_stream(true_x.device)
            if self._side is None:
                self._side = (torch.cuda.Stream(true_x.device), torch.cuda.Stream(true_x.device))
                # G's parameter gradients are produced on the side streams and accumulated into leaves
                # created on the main one: intended here
                quiet = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
                if quiet is not None:
                    quiet(False)
            s1, s2 = self._side
        self._mark("start", true_x.device)
        nloc = true_x.shape[0]
        counts = lambda: dp.batch_counts(nloc, nloc * dp.world() if n_global is None else n_global)
        if ov1:
            s1.wait_stream(main)
            with torch.cuda.stream(s1):          # D-phase generator forward, under the critic step
                if joint:
                    out2, fake_y, fake_y2 = netG.twin_joint(true_x)
                    out2.record_stream(main)     # fake_y / fake_y2 are views of its storage
                elif twin:                       # ... and the G-phase one with it
                    fake_y, fake_y2 = netG.twin(true_x, s2 if ov8 else None)
                    fake_y2.record_stream(main)
                else:
                    fake_y = netG(true_x)
            fake_y.record_stream(main)
        if training:
            self._critic_opt = FusedRMSprop(netD, self.lr * self.lr_gammaD)
            if critic_batches is None:
                critic_batches = (self.dataset.get_random() for _ in range(self.critic))
            for cx, cy in critic_batches:
                ncr = cx.shape[0]
                cx, cy, shared = self._shard_critic(cx, cy)
                if shared:
                    with dp.batch_counts(cx.shape[0], ncr):
                        self.critic_step(cx, cy)
                else:
                    with dp.replicated():
                        self.critic_step(cx, cy)
        self._mark("critic", true_x.device)
        self.optimD.zero_grad()
        self.optimG.zero_grad()
        # discriminator phase: fake_y keeps its graph, errD.backward() reaches G too
        with torch.set_grad_enabled(training), counts():
            if ov1:
                main.wait_stream(s1)
            elif joint:
                out2, fake_y, fake_y2 = netG.twin_joint(true_x)
            elif twin:
                fake_y, fake_y2 = netG.twin(true_x, s2 if ov8 else None)
            else:
                fake_y = netG(true_x)
            leaf_a = fake_y
            if ov4 or share:
                # cut the graph at fake_y: D's backward runs on this stream, G's (below) on its forward's
                fake_g, fake_y = fake_y, fake_y.detach().requires_grad_()
            Dy, DGx = netD.pair(true_y, fake_y)
            errD = self.losses.dis_loss(true_y, fake_y, Dy, DGx)
        if ov2:
            s2.wait_stream(main)
            with torch.cuda.stream(s2):          # G-phase generator forward, under the D-phase backward
                fake_y2 = netG(true_x)
            fake_y2.record_stream(main)
        dphase = None
        if training:
            self.optimD.begin_reduce()           # D's gradients are complete after this backward pass
            errD.backward(self._one(errD))
            if share:
                dfake_dphase = fake_y.grad
            elif ov4:        # the generator's share of errD.backward(), under the G-phase forward and discriminator pass
                # Enqueued BEHIND the G-phase generator forward (below): the host needs a few hundred microseconds to queue a
                # whole generator backward, and the main stream -- the critical chain -- would sit idle meanwhile.
                dphase = (fake_g, fake_y.grad, torch.cuda.Event())
                dphase[2].record(main)
        # generator phase.  errG.backward() of the reference also accumulates into D's parameter
        # gradients (trainer.py:183-186); the next zero_grad (:146 / :160) clears them unread, so D's
        # weight gradients are not computed here -- only the path through D back to fake_y.
        dparams = [p for p in netD.parameters() if p.requires_grad] if training else []
        for p in dparams:
            p.requires_grad_(False)
        try:
            with torch.set_grad_enabled(training), counts():
                if share:
                    fake_y = fake_g.detach().requires_grad_()
                elif ov2:
                    main.wait_stream(s2)
                    fake_y = fake_y2
                elif twin:
                    fake_y = fake_y2             # computed with the D-phase pass (same input, same parameters)
                else:
                    fake_y = netG(true_x)
                if dphase is not None:
                    s1.wait_event(dphase[2])
                    # fake_y.grad was allocated on the main stream and its last reference goes two lines down: without
                    # this the block returns to the main stream's pool while s1's backward kernels still read it
                    dphase[1].record_stream(s1)
                    with torch.cuda.stream(s1):
                        dphase[0].backward(dphase[1])
                    dphase = None
                # data parallel: the all-reduce of the gradient-penalty norm's sums starts here and runs under
                # the discriminator pass below
                presum = self.losses.recon_presum(true_y, fake_y) if training and dp.active() else None
                if training:
                    # Adam(D) after the D-phase backward (trainer.py:179-180), placed behind the generator forward
                    # -- which does not read D -- so that D's gradient all-reduce hides under it
                    self.optimD.all_reduce_grads()
                    self.optimD.step()
                    self._mark("dphase", true_x.device)
                if ov4 and not share:
                    fake_g, fake_y = fake_y, fake_y.detach().requires_grad_()
                # joint backward: D reads fake_y through a second leaf on the same values, so that its input gradient and the
                # loss terms' gradient arrive in two buffers (no accumulation kernel); concat2 below sums them on the way
                fake_d = fake_y.detach().requires_grad_() if (joint and training) else fake_y
                DGx = netD(fake_d)
                errG = self.losses.gen_loss(true_y, fake_y, DGx, presum=presum)
            if training:
                if not (share or ov4 or joint):
                    self.optimG.begin_reduce()   # errG.backward() runs through D into G: G's last backward pass
                errG.backward(self._one(errG))
                self._mark("gphase_D", true_x.device)
                if joint:
                    # both generator backward passes (trainer.py:179 through fake_y's graph, :193) as one over 2B rows
                    self.optimG.begin_reduce()
                    g2 = ops.concat2(leaf_a.grad, fake_y.grad, fake_d.grad)
                    if ov1:
                        g2.record_stream(s1)
                        s1.wait_stream(main)
                        with torch.cuda.stream(s1):
                            out2.backward(g2)
                        main.wait_stream(s1)
                    else:
                        out2.backward(g2)
                elif share:
                    self.optimG.begin_reduce()
                    fake_g.backward(dfake_dphase + fake_y.grad)
                    if overlap:
                        main.wait_stream(s1)
                        main.wait_stream(s2)
                elif ov4 and ov8:   # G-phase generator backward BESIDE the D-phase one, into the second gradient buffer
                    s2.wait_stream(main)
                    with torch.cuda.stream(s2), self.optimG.flat.alt():
                        fake_g.backward(fake_y.grad)
                    main.wait_stream(s1)
                    main.wait_stream(s2)
                    self.optimG.flat.merge_alt()
                elif ov4:    # G-phase generator backward after the D-phase one (both add into G's gradients)
                    s2.wait_stream(main)
                    s2.wait_stream(s1)
                    self.optimG.begin_reduce()
                    with torch.cuda.stream(s2):
                        fake_g.backward(fake_y.grad)
                    main.wait_stream(s2)
                elif overlap:
                    main.wait_stream(s1)
                    main.wait_stream(s2)
        finally:
            for p in dparams:
                p.requires_grad_(True)
        if training:
            self.optimG.all_reduce_grads()
            self.optimG.step()
        self._mark("end", true_x.device)
        ops.Runtime.wgrad_overlap = False
        return fake_y.detach()

    def post_process(self, fake_y, part=None, batch_rows=None):
        """The per-batch tail of trainer.py:202-236: band, adjust, scores -- all on the device.  They read the iteration's
        fake_y and state of their own and nothing feeds back into the networks, so on one rank they are queued on a side
        stream and run under the next iteration (the recurrent kernels leave the vector and memory pipes mostly idle); the
        values are the same.  join_post() before anything reads the scores or resets the state."""
        vis = self.dashboard is not None and getattr(self.dashboard, "vis_opt", False)
        side = self.post_overlap and part is None and fake_y.is_cuda and not vis
        if side:
            main = torch.cuda.current_stream(fake_y.device)
            if self._post_stream is None:
                self._post_stream = torch.cuda.Stream(fake_y.device)
            self._post_stream.wait_stream(main)
            fake_y.record_stream(self._post_stream)
            with torch.cuda.stream(self._post_stream):
                out = self.post.step(fake_y, part)
                self.metric.update(out[0], out[1], out[2], full=self.post.full_rows)
        else:
            out = self.post.step(fake_y, part)
            self.metric.update(out[0], out[1], out[2], full=self.post.full_rows)
        self.idx = self.post.idx
        if vis:
            reals, preds, masks, lower, upper, target = out
            self.dashboard.vis(batch_rows, reals.cpu().numpy(), preds.cpu().numpy(), lower.cpu().numpy(),
                               upper.cpu().numpy(), target.cpu().numpy())
        return out

    def join_post(self):
        """The current stream waits for the post-processing queued so far."""
        if self._post_stream is not None:
            torch.cuda.current_stream(self._post_stream.device).wait_stream(self._post_stream)

    def train_step(self, epoch: int, dataset, training: bool = True):
        """One pass over `dataset` (any iterable of {"encoded","decoded"} batches); returns the mean
        NMAE like trainer.py:236.  One host sync, at the end.  Data parallel: `dataset` is
        `windowset.loader(batch_size, rank, world)`, whose batches carry this rank's slice and its `part`."""
        if self.optimD is None:
            self.new_epoch()
        self.join_post()            # the previous pass's tail, before its state is replaced
        self.losses.init_loss()
        self.metric.init_score()
        self.post.restart_pass()
        dev = self.device
        i = -1
        for i, data in enumerate(dataset):
            true_x = data["encoded"].to(dev)
            true_y = data["decoded"].to(dev)
            part = data.get("part") if isinstance(data, dict) else None
            if part is not None and part[2]:
                with dp.replicated():        # a tail batch with fewer windows than ranks: the same step everywhere
                    fake_y = self.iteration(true_x, true_y, training)
            else:
                fake_y = self.iteration(true_x, true_y, training, n_global=None if part is None else part[1])
            # (post.full_rows: every rank holds ALL rows of this batch -- then only rank 0's sums count)
            self.post_process(fake_y, part, true_y.shape[0])
        if i < 0:
            return float("nan")
        self.join_post()
        ops.check_errors(dev)
        self.logger.info(desc(training, epoch, self.metric.score(i), self.losses.loss(i)), level=0)
        return self.metric.nmae / (i + 1)

    # kept for API parity with the reference (trainer.py:238-287); they run the same HIP kernels
    def adjust(self, output, preds, masks, lower, upper):
        dev = self.device
        t = lambda a, dt=torch.float64: torch.as_tensor(a).to(device=dev, dtype=dt).contiguous()
        out = t(output)
        tgt = ops.adjust_scan("trainer", out[0].contiguous(), out[1:].contiguous(), t(preds), t(masks, torch.int32),
                              t(lower), t(upper), self.missing_gamma, self.anomaly_gamma)
        return tgt.cpu().numpy()


def desc(training, epoch, score, losses):
    process = "Train" if training else "Valid"
    return (f"[{process} e{epoch + 1:4d}] (NMAE Score {score['NMAE']} NME {score['NME']} / RMSE {score['RMSE']} ) "
            f"D {losses['D']} ( R {losses['R']} F {losses['F']} ) "
            f"G {losses['G']} ( G {losses['G_']} L1 {losses['L1']} L2 {losses['L2']} GP {losses['GP']} )")

"""LSTMGenerator / LSTMDiscriminator on the gfx950 kernels.

Drop-in for the reference's source/utils/lstm_layer.py (Gen-A signatures :18 and :68) and the Gen-B
twins in source/models.py:23-31,91 (see `from_genb`).  The torch.nn.LSTM / nn.Linear sub-modules
are kept ONLY as parameter containers: same construction order (so the same seed gives the same
initial weights), same state_dict keys (`lstm0.weight_ih_l0`, `seq_layer.0.weight`, ...), so a
state_dict taken from a reference network loads.  (Only state_dict checkpoints are accepted: the
reference's model.py:84-95 pickles whole modules, and TIMEBANDModel.load refuses to unpickle.)  Their
forward() is never called: all arithmetic goes through timeband_amd.ops (HIP kernels).
"""
import contextlib

import torch
import torch.nn as nn

from .. import ops


class LSTMGenerator(nn.Module):
    """(B, inseq_len, in_dim) -> (B, outseq_len, out_dim) in (-1, 1).

    lstm0 (H/4) -> lstm1 (H/2) -> lstm2 (H); then Linear(inseq_len -> outseq_len) applied to the
    (B,To,H) buffer REINTERPRETED as (B,H,To), and Linear(H -> out_dim)+Tanh on the result
    reinterpreted as (B,Tf,H) -- memory reinterpretation, not transposes
    (reference lstm_layer.py:49-50)."""

    def __init__(self, in_dim, out_dim, inseq_len, outseq_len, hidden_dim=256, n_layers=1, device=None):
        super().__init__()
        if hidden_dim % 4:
            raise ValueError("hidden_dim must be divisible by 4")
        self.device = device
        self.n_layers = n_layers
        self.hidden_dim = hidden_dim
        self.out_dim = out_dim
        self.inseq_len = inseq_len
        self.outseq_len = outseq_len
        widths = (hidden_dim // 4, hidden_dim // 2, hidden_dim)
        prev = in_dim
        for i, w in enumerate(widths):          # creation order = the reference's RNG order
            setattr(self, f"lstm{i}", nn.LSTM(prev, w, n_layers, batch_first=True))
            prev = w
        self.seq_layer = nn.Sequential(nn.Linear(inseq_len, outseq_len))
        self.dim_layer = nn.Sequential(nn.Linear(hidden_dim, out_dim), nn.Tanh())

    @classmethod
    def from_genb(cls, in_dim, in_seq, out_dim, out_seq, hid_dim=256, n_layers=1, device=None):
        """Gen-B argument order (reference source/models.py:23-31)."""
        return cls(in_dim, out_dim, in_seq, out_seq, hid_dim, n_layers, device)

    def forward(self, input, split=0):
        B = input.size(0)
        r = input
        for i in range(3):
            r = ops.lstm_stack(r, getattr(self, f"lstm{i}"), split=split)
        seq, dim = self.seq_layer[0], self.dim_layer[0]
        s = ops.linear_act(r.view(B * self.hidden_dim, self.inseq_len), seq.weight, seq.bias, ops.ACT_NONE)
        out = ops.linear_act(s.view(B * self.outseq_len, self.hidden_dim), dim.weight, dim.bias, ops.ACT_TANH)
        return out.view(B, self.outseq_len, self.out_dim)


    def infer_windows(self, series, first, nwin, stride=1, hidden=False):
        """self(windows) without a graph for the `nwin` windows whose encoder rows are series rows
        [first + w * stride - inseq_len, first + w * stride) -- the cleaning / prediction pass (cleaner.py:67-68 over the
        windows of dataset.py:207-212).  The windows overlap, so the first layer's input projection is computed once per
        SERIES row and the recurrent kernel reads it in place (ops.lstm_fwd_windows): the encoder windows are never
        gathered and the projection costs inseq_len / stride times fewer flops.  Bit-identical to self(gathered windows).
        hidden=True: stop in front of the last layer and return its input, the (nwin, outseq_len, hidden_dim) buffer -- the
        caller forms the output inside the band kernel (BandAdjust.step_hidden)."""
        To = self.inseq_len
        l0 = self.lstm0
        r0 = first - To
        nrows = (nwin - 1) * stride + To
        if r0 < 0 or r0 + nrows > series.shape[0]:
            raise RuntimeError("infer_windows: a window leaves the series")
        with torch.no_grad():
            # b_ih + b_hh, summed first like the projection GEMM's two-bias epilogue; cached until a parameter changes
            key = (l0.bias_ih_l0._version, l0.bias_hh_l0._version, l0.bias_ih_l0.data_ptr())
            if getattr(self, "_bias0", (None, None))[0] != key:
                self._bias0 = (key, (l0.bias_ih_l0 + l0.bias_hh_l0).detach())
            bias = self._bias0[1]
            pseries = ops.linear_act_raw(series[r0:r0 + nrows], l0.weight_ih_l0.detach(), bias, ops.ACT_NONE)
            r = ops.lstm_fwd_windows(pseries, 0, stride, nwin, To, l0.weight_hh_l0)
            for l in range(1, l0.num_layers):
                r = ops.lstm_layer(r, *ops._lstm_params(l0, l))
            for i in (1, 2):
                r = ops.lstm_stack(r, getattr(self, f"lstm{i}"))
            seq, dim = self.seq_layer[0], self.dim_layer[0]
            s = ops.linear_act(r.view(nwin * self.hidden_dim, To), seq.weight, seq.bias, ops.ACT_NONE)
            if hidden:
                return s.view(nwin, self.outseq_len, self.hidden_dim)
            out = ops.linear_act(s.view(nwin * self.outseq_len, self.hidden_dim), dim.weight, dim.bias, ops.ACT_TANH)
        return out.view(nwin, self.outseq_len, self.out_dim)

    def last_layer(self, hid):
        """The output layer on a (n, outseq_len, hidden_dim) buffer of infer_windows(hidden=True) (no graph)."""
        dim = self.dim_layer[0]
        n = hid.shape[0]
        with torch.no_grad():
            out = ops.linear_act(hid.reshape(n * self.outseq_len, self.hidden_dim), dim.weight, dim.bias, ops.ACT_TANH)
        return out.view(n, self.outseq_len, self.out_dim)

    def twin_joint(self, input):
        """twin() with ONE autograd graph over the 2B rows: returns (out2, fake_a, fake_b) -- out2 (2B, Tf, V) carries
        the graph; fake_a / fake_b are its two halves as detached leaves (requires_grad) for the discriminator phase and
        the generator phase.  The caller runs the two generator backward passes of the reference (trainer.py:179,193) as
        one: out2.backward(cat(fake_a.grad, fake_b.grad)) -- the parameter gradients are then summed over the 2B rows
        in one pass instead of accumulated pass by pass (the same sum up to fp32 association)."""
        B = input.size(0)
        x = input.contiguous()
        out2 = self(ops.concat2(x, x), split=B)
        return out2, out2[:B].detach().requires_grad_(), out2[B:].detach().requires_grad_()

    def twin(self, input, second_stream=None):
        """(self(input), self(input)): the two generator forward passes of one training iteration -- the reference calls
        netG(true_x) in the discriminator phase and again in the generator phase with unchanged parameters
        (trainer.py:173,188) -- computed as ONE pass over 2B rows (the input twice), like LSTMDiscriminator.pair.
        Both passes are executed in full, row by row the same arithmetic as two calls (a row's result does not depend
        on the batch it sits in: bit-identical outputs), but the persistent recurrent kernels are bound by their
        per-timestep latency, not by rows, so 2B rows cost little more than B.  Each returned tensor carries an autograd
        graph of its own over its half of the BPTT stash: the two backward passes run separately, as in the reference.
        second_stream: the stream the second pass's backward kernels are to run on (autograd runs a backward node on the
        stream that was current when its forward node was made; default: the current one)."""
        B = input.size(0)
        seq, dim = self.seq_layer[0], self.dim_layer[0]
        with torch.no_grad():
            x2 = ops.concat2(input.contiguous(), input.contiguous())
            r, stash = x2, []
            for i in range(3):
                r, st = ops.lstm_stack_raw(r, getattr(self, f"lstm{i}"), split=B)
                stash.append(st)
            s2 = ops.linear_act_raw(r.view(2 * B * self.hidden_dim, self.inseq_len), seq.weight.detach(), seq.bias.detach(),
                                    ops.ACT_NONE)
            o2 = ops.linear_act_raw(s2.view(2 * B * self.outseq_len, self.hidden_dim), dim.weight.detach(),
                                    dim.bias.detach(), ops.ACT_TANH)
        outs = []
        for h in (0, 1):
            rows = slice(h * B, (h + 1) * B)
            ctxm = torch.cuda.stream(second_stream) if (h == 1 and second_stream is not None) else contextlib.nullcontext()
            with ctxm:
                outs.append(self._twin_graph(input, stash, s2, o2, rows, h))
        return outs[0], outs[1]

    def _twin_graph(self, input, stash, s2, o2, rows, h):
        """The autograd nodes of one of twin()'s two passes, around its rows of the jointly computed buffers."""
        B = input.size(0)
        seq, dim = self.seq_layer[0], self.dim_layer[0]
        r = input
        for i in range(3):
            r = ops.lstm_stack(r, getattr(self, f"lstm{i}"), pre=[tuple(t[rows] for t in st) for st in stash[i]])
        n = B * self.hidden_dim
        s = ops.linear_act(r.view(n, self.inseq_len), seq.weight, seq.bias, ops.ACT_NONE, pre=s2[h * n:(h + 1) * n])
        n = B * self.outseq_len
        out = ops.linear_act(s.view(n, self.hidden_dim), dim.weight, dim.bias, ops.ACT_TANH, pre=o2[h * n:(h + 1) * n])
        return out.view(B, self.outseq_len, self.out_dim)


class LSTMDiscriminator(nn.Module):
    """(B, T, in_dim) -> per-timestep score (B, T, 1) in (0, 1)."""

    def __init__(self, in_dim, hidden_dim=256, n_layers=1, device=None):
        super().__init__()
        self.device = device
        self.n_layers = n_layers
        self.hidden_dim = hidden_dim
        self.in_dim = in_dim
        self.lstm = nn.LSTM(in_dim, hidden_dim, n_layers, batch_first=True)
        self.linear = nn.Sequential(nn.Linear(hidden_dim, 1), nn.Sigmoid())

    @classmethod
    def from_genb(cls, in_seq, in_dim, out_dim, out_seq, hid_dim=256, n_layers=1, device=None):
        """Gen-B argument order (reference source/models.py:91)."""
        return cls(in_dim, hid_dim, n_layers, device)

    def forward(self, input, dx_from=0):
        B, T = input.size(0), input.size(1)
        r = ops.lstm_stack(input, self.lstm, dx_from)
        lin = self.linear[0]
        out = ops.linear_act(r.view(B * T, self.hidden_dim), lin.weight, lin.bias, ops.ACT_SIGMOID)
        return out.view(B, T, 1)

    def pair(self, real, fake):
        """(netD(real), netD(fake)) of trainer.py:149-150,174-175 as ONE pass over the 2B rows: the rows
        of a batch are independent through D, so the results are the same, but the persistent
        recurrent kernels are latency-bound per timestep and run 2B rows in little more than the time
        of B (and the weight-gradient GEMMs run once over both halves)."""
        if real.shape != fake.shape:
            return self(real), self(fake)
        # the real half is data: nobody reads its input gradient, so the first layer does not compute it
        out = self(ops.cat2(real, fake), dx_from=0 if real.requires_grad else real.size(0))
        Dy, DGx = out.view(2, real.size(0), real.size(1), 1).unbind(0)
        return Dy, DGx

"""LSTMGenerator / LSTMDiscriminator on the gfx950 kernels.

Drop-in for the reference's source/utils/lstm_layer.py (Gen-A signatures :18 and :68) and the Gen-B
twins in source/models.py:23-31,91 (see `from_genb`).  The torch.nn.LSTM / nn.Linear sub-modules
are kept ONLY as parameter containers: same construction order (so the same seed gives the same
initial weights), same state_dict keys (`lstm0.weight_ih_l0`, `seq_layer.0.weight`, ...), so a
state_dict taken from a reference network loads.  (Only state_dict checkpoints are accepted: the
reference's model.py:84-95 pickles whole modules, and TIMEBANDModel.load refuses to unpickle.)  Their
forward() is never called: all arithmetic goes through timeband_amd.ops (HIP kernels).
"""
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

    def forward(self, input):
        B = input.size(0)
        r = input
        for i in range(3):
            r = ops.lstm_stack(r, getattr(self, f"lstm{i}"))
        seq, dim = self.seq_layer[0], self.dim_layer[0]
        s = ops.linear_act(r.view(B * self.hidden_dim, self.inseq_len), seq.weight, seq.bias, ops.ACT_NONE)
        out = ops.linear_act(s.view(B * self.outseq_len, self.hidden_dim), dim.weight, dim.bias, ops.ACT_TANH)
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
        out = self(torch.cat([real, fake], 0), dx_from=0 if real.requires_grad else real.size(0))
        Dy, DGx = out.view(2, real.size(0), real.size(1), 1).unbind(0)
        return Dy, DGx

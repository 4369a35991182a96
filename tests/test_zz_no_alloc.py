"""GPU: the entry point that allocates nothing (tb_lstm_fwd_static) against tb_lstm_fwd, through the C-ABI.

Two calls in a row on the same GPU (they share the library's static scratch, which keeps the first call's
contents), the second with a smaller input; every output -- y, the activated gates and the cell stash --
must equal tb_lstm_fwd's bit for bit (same kernels, only the exchange area differs).  Collected last on
purpose (file name): the round's end runs pytest -x.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(entry, x, ps, H, train, dev, rt):
    from timeband_amd import _lib
    lib = _lib.load()
    B, T, In = x.shape
    y = torch.full((B, T, H), float("nan"), device=dev)
    gates = torch.full((B, T, 4 * H), float("nan"), device=dev)
    cst = torch.full((B, T, H), float("nan"), device=dev)
    st = torch.cuda.current_stream().cuda_stream
    ptr = lambda t: t.data_ptr()
    if entry == "static":
        _lib.call("tb_lstm_fwd_static", ptr(x), *(ptr(p) for p in ps), ptr(y), ptr(gates), ptr(cst), B, T, In, H,
                  int(train), ptr(rt.err), st)
    else:
        ws = rt.workspace(lib.tb_lstm_workspace_bytes(B, T, In, H, 0))
        _lib.call("tb_lstm_fwd", ptr(x), *(ptr(p) for p in ps), ptr(y), ptr(gates), ptr(cst), B, T, In, H, int(train),
                  ptr(ws), ws.numel(), ptr(rt.err), st)
    return y, gates, cst


@pytest.mark.parametrize("H,In,first,second", [(256, 32, (70, 14), (24, 6)), (512, 36, (200, 9), (40, 5)),
                                                (128, 16, (1100, 7), (33, 12)), (64, 11, (130, 20), (17, 8))])
def test_static_entry_matches_workspace_entry(H, In, first, second):
    from timeband_amd import _lib, ops
    dev = torch.device("cuda", 0)
    rt = ops.Runtime.get(dev)
    lib = _lib.load()
    assert lib.tb_lstm_static_scratch_bytes() == 8 << 20
    g = torch.Generator(device="cpu").manual_seed(H + In)
    k = 1.0 / np.sqrt(H)
    ps = [((torch.rand(s, generator=g) * 2 - 1) * k).to(dev) for s in ((4 * H, In), (4 * H, H), (4 * H,), (4 * H,))]
    for (B, T) in (first, second):          # two calls in a row, the second smaller
        x = (torch.rand(B, T, In, generator=g) * 2 - 1).to(dev)
        for train in (1, 0):
            got = _run("static", x, ps, H, train, dev, rt)
            want = _run("workspace", x, ps, H, train, dev, rt)
            torch.cuda.synchronize()
            ops.check_errors()
            n = 3 if train else 1
            for a, b, name in list(zip(got, want, ("y", "gates", "cstash")))[:n]:
                assert not torch.isnan(a).any(), name
                assert torch.equal(a, b), (name, B, T, train)


def test_static_entry_rejects_what_does_not_fit_and_bad_sizes():
    from timeband_amd import _lib
    lib = _lib.load()
    # unsupported hidden size: a clear error, nothing launched
    rc = lib.tb_lstm_fwd_static(0, 0, 0, 0, 0, 0, 0, 0, 4, 4, 4, 48, 0, 0, 0)
    assert rc != 0 and b"not supported" in lib.tb_last_error()

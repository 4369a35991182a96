"""CPU tests: the C-ABI library loads and exports every symbol include/timeband_hip.h declares,
the ctypes table matches the header, and the host-side logic that needs no GPU."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "timeband_hip.h")


def header_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tb_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    names = header_functions()
    for must in ("tb_lstm_fwd", "tb_lstm_bwd", "tb_linear_act_fwd", "tb_linear_bwd", "tb_loss_sums",
                 "tb_adam_step", "tb_rmsprop_step", "tb_window_gather", "tb_band_update", "tb_adjust_scan"):
        assert must in names
    text = open(HEADER).read()
    # every block cites the reference file it replaces
    for cite in ("source/utils/lstm_layer.py", "source/loss.py", "source/trainer.py", "source/dataset.py",
                 "source/cleaner.py", "source/predictor.py", "source/metric.py"):
        assert cite in text, cite


def test_library_exports_every_declared_symbol():
    from timeband_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build the library first: python -m timeband_amd.build"
    lib = _lib.load()                      # binds every entry of SIGNATURES (AttributeError if missing)
    declared = header_functions()
    assert sorted(_lib.SIGNATURES) == declared, set(declared) ^ set(_lib.SIGNATURES)
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), name
    assert lib.tb_version() >= 1           # host-only call, no GPU needed
    assert lib.tb_lstm_supported(256) == 1 and lib.tb_lstm_supported(100) == 0
    assert lib.tb_loss_workspace_bytes() > 0


def test_cpp_extension_loads_and_exposes_its_ops():
    """timeband_amd/_tbext.so (csrc/ext.cpp: the PyTorch-ROCm C++ extension over the C-ABI) imports without a GPU and
    carries the autograd entry points ops.py routes through; a CPU tensor is refused, not computed."""
    import torch
    from timeband_amd import ops
    ext = ops._load_ext()
    assert ext is not None, "build the extension first: python -m timeband_amd.build"
    for name in ("lstm_layer", "lstm_fwd_raw", "linear_act", "linear_act_raw", "set_runtime", "clustered_begin",
                 "clustered_end", "set_grads_hook", "arm_grads_hook"):
        assert hasattr(ext, name), name
    with pytest.raises(RuntimeError, match="GPU only"):
        ext.linear_act_raw(torch.zeros(4, 8), torch.zeros(3, 8), None, 0)


def test_product_path_has_no_oracle_import():
    """Nothing under timeband_amd/ may import, call or link anything under oracle/."""
    pkg = os.path.join(ROOT, "timeband_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "timeband_oracle" not in src and "torch_cpu_port" not in src, f


def test_cpu_tensor_is_rejected_loudly():
    import torch
    from timeband_amd import ops
    x = torch.zeros(2, 3, 4)
    w = torch.zeros(64, 4)
    with pytest.raises(RuntimeError, match="GPU only"):
        ops.lstm_layer(x, w, torch.zeros(64, 16), torch.zeros(64), torch.zeros(64))
    with pytest.raises(RuntimeError, match="GPU only"):
        ops.linear_act(torch.zeros(4, 4), torch.zeros(2, 4), None)


def test_window_math_matches_oracle():
    from oracle import timeband_oracle as O
    from timeband_amd.dataset import TIMEBANDDataset
    ds = TIMEBANDDataset.__new__(TIMEBANDDataset)
    for n, To, Tf, stride in ((172, 24, 12, 1), (100, 10, 3, 1), (100, 10, 3, 2), (30, 20, 11, 1)):
        ds.observed_len, ds.forecast_len, ds.stride, ds.data_length = To, Tf, stride, n
        assert ds.n_windows() == len(O.window_starts(n, To, Tf, stride))


def test_time_feature_order():
    from timeband_amd import synthetic
    tf = synthetic.time_features(30, enabled=("month", "weekday", "day", "hour"))
    assert tf.shape == (30, 4)
    assert list(tf[:3, 0]) == [0.0, 1.0, 2.0]       # hour first
    assert tf[0, 1] == 4.0 and tf[0, 2] == 1.0 and tf[0, 3] == 1.0   # 2021-01-01: Friday, day 1, month 1


def test_init_matches_reference_rng_order(golden_dir):
    """Same seed -> bit-identical initial weights as the reference (D before G, model.py:56-57)."""
    import torch  # noqa: F401
    from timeband_amd.utils.lstm_layer import LSTMGenerator, LSTMDiscriminator
    from timeband_amd.utils.initiate import seeding
    for name in ("cfg1", "twolayer"):
        z = np.load(os.path.join(golden_dir, f"scenario_{name}.npz"))
        V, E, To, Tf, H, L, B = (int(v) for v in z["meta"])
        seeding(31)
        netD = LSTMDiscriminator(V, H, L)
        netG = LSTMGenerator(E, V, To, Tf, H, L)
        for k, v in netD.state_dict().items():
            assert np.array_equal(v.numpy(), z[f"init.netD.{k}"]), k
        for k, v in netG.state_dict().items():
            assert np.array_equal(v.numpy(), z[f"init.netG.{k}"]), k


def test_genb_data_front_end_on_cpu(tmp_path):
    """timeband_amd.data.TIMEBANDData (reference source/data.py): directory layout, artefacts, time features
    prepended in the order given, min-max normalisation from the saved minmax.csv, and windows equal to
    the reference's slicing loop (data.py:139-143), here restated literally."""
    import numpy as np
    import pandas as pd
    import torch
    from timeband_amd import synthetic
    from timeband_amd.data import TIMEBANDData
    V, To, Tf, N = 3, 6, 4, 120
    base = str(tmp_path)
    synthetic.write_csv(os.path.join(base, "origin"), "s", N, V, seed=3, missing_rate=0.05)
    d = TIMEBANDData(basedir=base, filename="s", targets=[f"v{i}" for i in range(V)], drops=[], fill_timegap=False,
                     time_index=["Date"], time_encode=["weekday", "hour"], split_size=0.75, observed_len=To,
                     forecast_len=Tf, device="cpu")
    assert (d.encode_dims, d.decode_dims) == (V + 2, V)
    frame = d.init_dataset()
    for f in (d.target_path, d.minmax_path, d.missing_path):
        assert os.path.exists(f)
    assert list(d.observed.columns[:2]) == ["hour", "weekday"]        # each feature is prepended
    miss = pd.read_csv(d.missing_path)
    assert int(miss[[f"v{i}" for i in range(V)]].to_numpy().sum()) > 0
    trainset, validset = d.prepare_trainset(frame)
    # the reference's loop over the same normalised frame
    raw = pd.read_csv(d.target_path, parse_dates=["Date"]).set_index("Date").ffill().bfill().replace(np.nan, 0)
    full = d.parse_timeinfo(raw)
    mm = pd.read_csv(d.minmax_path, index_col=["Feats"])
    full = 2 * ((full - mm["min"]) / (mm["max"] - mm["min"])) - 1
    x, y = full.to_numpy(), full[[f"v{i}" for i in range(V)]].to_numpy()
    obs, fut = [], []
    for i in range(To, len(full) - Tf + 1):
        obs.append(x[i - To:i])
        fut.append(y[i:i + Tf])
    split = min(int(len(full) * 0.75), len(full) - To - Tf)
    assert len(trainset) == split and len(trainset) + len(validset) == len(obs)
    np.testing.assert_allclose(np.concatenate([trainset.encoded, validset.encoded]), np.array(obs), rtol=0, atol=0)
    np.testing.assert_allclose(np.concatenate([trainset.decoded, validset.decoded]), np.array(fut), rtol=0, atol=0)
    e0, d0 = trainset[0]
    assert e0.dtype == torch.float32 and e0.shape == (To, V + 2) and d0.shape == (Tf, V)
    xs, ys = d.get_random(trainset)
    assert xs.shape == (Tf, To, V + 2) and ys.shape == (Tf, Tf, V)
    assert torch.allclose(d.delta, d.decode_max - d.decode_min)
    sub_frame = pd.read_csv(d.target_path, parse_dates=["Date"]).iloc[-To - Tf:]
    predset = d.prepare_predset(sub_frame)
    assert len(predset) == Tf and predset.shape() == (Tf, To, V + 2)


def test_whole_module_checkpoint_is_refused_with_guidance(tmp_path):
    """The reference pickles whole modules (model.py:84-95); TIMEBANDModel.load never unpickles: such a file
    ends in the documented RuntimeError, not in an UnpicklingError or the silent fallback to load("")."""
    import torch
    from timeband_amd.model import TIMEBANDModel
    from timeband_amd.utils.logger import NullLogger
    m = TIMEBANDModel({"logger": NullLogger(), "models_path": str(tmp_path), "load_opt": True})
    m.netD, m.netG = torch.nn.Linear(2, 2), torch.nn.Linear(2, 2)
    torch.save(m.netD, m.get_path("netD"))
    torch.save(m.netG, m.get_path("netG"))
    with pytest.raises(RuntimeError, match="not a state_dict checkpoint"):
        m.load()
    torch.save(m.netD.state_dict(), m.get_path("netD"))
    torch.save(m.netG.state_dict(), m.get_path("netG"))
    m.load()

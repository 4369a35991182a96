"""GPU: the config-driven host API end to end (csv -> TIMEBANDDataset -> TIMEBANDTrainer.train ->
TIMEBANDCleaner.clean / TIMEBANDPredictor.predict) against what the reference itself produced on
the same csv and seed (tests/golden/pipeline.npz, made by tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def sub(z, prefix):
    prefix += "."
    return {k[len(prefix):]: z[k] for k in z.files if k.startswith(prefix)}


@pytest.fixture(scope="module")
def world(tmp_path_factory):
    from timeband_amd import synthetic
    from timeband_amd.dataset import TIMEBANDDataset
    from timeband_amd.loss import TIMEBANDLoss
    from timeband_amd.metric import TIMEBANDMetric
    from timeband_amd.model import TIMEBANDModel
    from timeband_amd.utils.initiate import seeding
    from timeband_amd.utils.logger import NullLogger
    z = np.load(os.path.join(GOLDEN, "pipeline.npz"))
    V, To, Tf, H, L, B, N = (int(v) for v in z["meta"])
    tmp = str(tmp_path_factory.mktemp("tb"))
    synthetic.write_csv(tmp, "synth", N, V, seed=31)
    dev = torch.device("cuda", 0)
    inj = dict(logger=NullLogger(), device=dev)
    core = dict(TAG="t", data_name="synth", directory=tmp, pretrain=False, workers=0, batch_size=B,
                observed_len=To, forecast_len=Tf, verbosity=9)
    dcfg = dict(reset=True, targets=[f"v{i}" for i in range(V)], drops=[], time_index="Date",
                time_format="%Y-%m-%d %H:%M:%S",
                time_encode=dict(year=False, month=True, weekday=True, day=True, hour=True, minute=False),
                fill_timegap=False, stride=1, split_rate=0.8, min_valid_scale=2, sliding_step=0)
    seeding(31)
    ds = TIMEBANDDataset({**core, **dcfg, **inj})
    models = TIMEBANDModel({**core, **inj, "save_opt": True, "load_opt": True, "best_score": 50.0, "hidden_dim": H,
                            "layers_num": L, "models_path": os.path.join(tmp, "models")})
    models.initiate(ds.dims)
    tcfg = {**core, **inj, "models": dict(reload=False, reload_interval=50, save_interval=10), "critic": 1,
            "epochs": 1, "learning_rate": dict(base=2e-4, decay=0.9, gammaD=1.0, gammaG=1.0), "band_width": 1.0,
            "missing_gamma": 0.2, "anomaly_gamma": 0.3, "print_interval": 1}
    losses = TIMEBANDLoss({**core, **inj, "l1_weight": 50, "l2_weight": 50, "gp_weight": 50})
    metric = TIMEBANDMetric({**core, **inj})
    return dict(z=z, ds=ds, models=models, tcfg=tcfg, losses=losses, metric=metric, B=B, tmp=tmp,
                dims=(V, To, Tf, H, L, B, N))


def test_dataset_matches_reference(world):
    z, ds = world["z"], world["ds"]
    V, To, Tf, H, L, B, N = world["dims"]
    assert ds.data_length == int(z["ds.data_length"])
    assert list(ds.data.columns) == list(z["ds.columns"])
    np.testing.assert_allclose(ds.data.to_numpy(), z["ds.normalized"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(ds.observed.numpy(), z["ds.observed"], rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(ds.forecast.numpy(), z["ds.forecast"], rtol=1e-12, atol=1e-15)
    np.testing.assert_array_equal(np.asarray(ds.missing), z["ds.missing"])
    np.testing.assert_allclose(ds.decode_min.numpy(), z["ds.decode_min"], rtol=1e-12)
    np.testing.assert_allclose(ds.decode_max.numpy(), z["ds.decode_max"], rtol=1e-12)
    assert [ds.times[0], ds.times[-1]] == list(z["ds.times_first_last"])
    trainset, validset = ds.prepare_dataset()
    assert len(trainset) == int(z["ds.train_len"]) and len(validset) == int(z["ds.valid_len"])
    first, last = trainset[0], validset[len(validset) - 1]
    np.testing.assert_allclose(first["encoded"].cpu().numpy(), z["ds.win0.encoded"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(first["decoded"].cpu().numpy(), z["ds.win0.decoded"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(last["encoded"].cpu().numpy(), z["ds.winlast.encoded"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(last["decoded"].cpu().numpy(), z["ds.winlast.decoded"], rtol=1e-6, atol=1e-7)
    for f in ("original_data.csv", "missing_label.csv", "anomaly_label.csv", "minmax_info.csv", "normalized.csv"):
        assert os.path.exists(os.path.join(ds.basepath, f)), f
    assert len(ds.prepare_dataset(split=False)) == int(z["full_len"])


def test_device_minmax_normalize_match_reference(world):
    """tb_minmax / tb_normalize / tb_take_columns on the uploaded frame against the reference's own frame
    (ds.normalized of pipeline.npz: bit-exact in float64, the float32 series = its float32 cast), the
    reference's fix-up rules on degenerate columns, chunk folding, and the streamed from_arrays front end."""
    from oracle import timeband_oracle as O
    from timeband_amd import ops, synthetic
    from timeband_amd.dataset import TIMEBANDDataset
    from timeband_amd.utils.logger import NullLogger
    z, ds = world["z"], world["ds"]
    V, To, Tf, H, L, B, N = world["dims"]
    dev = torch.device("cuda", 0)
    assert np.array_equal(ds.series_dev.cpu().numpy(), z["ds.normalized"].astype(np.float32))
    np.testing.assert_array_equal(ds.forecast_dev.cpu().numpy(), ds.forecast.numpy())
    # degenerate columns: constant -> max + 1; all -inf -> min = max (dataset.py:311-312); NaN skipped
    rng = np.random.default_rng(2)
    a = rng.normal(size=(1000, 70))
    a[:, 3] = 2.5
    a[:, 7] = -np.inf
    a[5, 9] = np.nan
    t = torch.from_numpy(a).to(dev)
    mn, mx = ops.minmax(t)
    wmn, wmx = O.minmax_info(a, 1.0)
    np.testing.assert_array_equal(mn.cpu().numpy(), wmn)
    np.testing.assert_array_equal(mx.cpu().numpy(), wmx)
    # folded over row chunks == in one go
    m1, x1 = ops.minmax(t[:300], finish=False)
    m1, x1 = ops.minmax(t[300:777], m1, x1, finish=False)
    m1, x1 = ops.minmax(t[777:], m1, x1, finish=True)
    assert torch.equal(m1, mn) and torch.equal(x1, mx)
    # fill kernel: gaps longer than a row block (1024), a leading gap, an empty column; the missing mask
    b = rng.normal(size=(5000, 6))
    b[rng.random((5000, 6)) < 0.3] = np.nan
    b[:1500, 1] = np.nan
    b[2000:4300, 2] = np.nan
    b[:, 5] = np.nan
    tb = torch.from_numpy(b.copy()).to(dev)
    tpos = torch.tensor([-1, 0, -1, 1, -1, 2], dtype=torch.int32, device=dev)
    miss = ops.interpolate(tb, tpos, 3, 100)
    np.testing.assert_array_equal(tb.cpu().numpy(), O.interpolate(b))
    np.testing.assert_array_equal(miss.cpu().numpy(), np.isnan(b)[100:][:, [1, 3, 5]].astype(np.int32))
    # from_arrays with tiny chunks == oracle SeriesData
    values = synthetic.make_series(300, 5, seed=9)
    values[:40, 2] = np.nan
    tfeat = synthetic.time_features(300 + Tf)
    old = TIMEBANDDataset.CHUNK_ROWS
    TIMEBANDDataset.CHUNK_ROWS = 64
    try:
        d2 = TIMEBANDDataset.from_arrays(values, tfeat, dict(logger=NullLogger(), device=dev, batch_size=8, observed_len=To,
                                                             forecast_len=Tf, stride=1, split_rate=0.8,
                                                             min_valid_scale=2, sliding_step=0, directory="", data_name="x"))
    finally:
        TIMEBANDDataset.CHUNK_ROWS = old
    s = O.SeriesData(values, tfeat, To, Tf, 0.8)
    assert np.array_equal(d2.series_dev.cpu().numpy(), s.norm.astype(np.float32))
    np.testing.assert_array_equal(d2.forecast_dev.cpu().numpy(), s.forecast)
    np.testing.assert_array_equal(d2.observed_dev.cpu().numpy(), s.observed)
    np.testing.assert_array_equal(d2.missing_dev.cpu().numpy(), s.missing)
    np.testing.assert_array_equal(d2.decode_min.numpy(), s.decode_min)
    np.testing.assert_array_equal(d2.decode_max.numpy(), s.decode_max)
    # prepare_dataset(_from): windows of the sub-frame data[_from:] (reference dataset.py:160-166)
    sub = d2.prepare_dataset(_from=17, split=False)
    enc, dec = O.windowing(s.norm[17:], s.target_cols, To, Tf)
    assert len(sub) == len(enc)
    got = sub.gather(3, 4)
    assert np.array_equal(got["encoded"].cpu().numpy(), enc[3:7]) and np.array_equal(got["decoded"].cpu().numpy(), dec[3:7])


def test_initial_weights_match_reference(world):
    z, m = world["z"], world["models"]
    for tag, net in (("init.netD", m.netD), ("init.netG", m.netG)):
        for k, v in net.state_dict().items():
            assert np.array_equal(v.cpu().numpy(), z[f"{tag}.{k}"]), (tag, k)


def test_clean_and_predict_with_reference_weights(world):
    """Inference + band + adjust with the weights the REFERENCE trained: isolates the cleaning path."""
    from timeband_amd.cleaner import TIMEBANDCleaner
    from timeband_amd.model import TIMEBANDModel
    from timeband_amd.predictor import TIMEBANDPredictor
    from timeband_amd.utils.lstm_layer import LSTMGenerator
    z, ds, B = world["z"], world["ds"], world["B"]
    V, To, Tf, H, L, _, N = world["dims"]
    dev = torch.device("cuda", 0)
    holder = TIMEBANDModel.__new__(TIMEBANDModel)
    holder.netG = LSTMGenerator(ds.encode_dim, V, To, Tf, H, L, dev).to(dev)
    holder.netG.load_state_dict({k: torch.tensor(v) for k, v in sub(z, "ep0.train.netG").items()})
    holder.netD = None
    full = ds.prepare_dataset(split=False)
    out_df, bands_df = TIMEBANDCleaner(world["tcfg"], ds, holder).clean(full.loader(B))
    assert out_df.shape == (N + Tf, V) and list(bands_df.columns) == list(z["clean.columns"])
    np.testing.assert_allclose(out_df.to_numpy(), z["clean.outputs"], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(bands_df.to_numpy(), z["clean.bands"], rtol=2e-4, atol=2e-4)
    assert out_df.index.name == "Date" and out_df.index[0] == ds.times[0]
    pred_df = TIMEBANDPredictor(world["tcfg"], ds, holder).predict(full.loader(B))
    np.testing.assert_allclose(pred_df.to_numpy(), z["predict.outputs"], rtol=2e-4, atol=2e-5)


def test_train_epoch_matches_reference(world):
    """trainer.train(): one Gen-A epoch (train pass with critic/D/G steps, then valid pass) against the
    reference's parameters, printed loss means and scores."""
    from timeband_amd.trainer import TIMEBANDTrainer
    from timeband_amd.utils.initiate import seeding
    z, ds, m, B = world["z"], world["ds"], world["models"], world["B"]
    np.random.seed(31)      # seeding(31) state at the first get_random() of the reference run
    trainset, validset = ds.prepare_dataset()
    trainer = TIMEBANDTrainer(world["tcfg"], ds, m, world["metric"], world["losses"], None)
    res = {}
    orig = trainer.train_step

    def wrapped(epoch, loader, training=True):
        score = orig(epoch, loader, training=training)
        tag = "train" if training else "valid"
        n = len(loader)
        res[tag] = dict(score=score, losses=trainer.losses.loss(n - 1), scores=trainer.metric.score(n - 1))
        if training:
            res["pD"] = {k: v.cpu().numpy().copy() for k, v in m.netD.state_dict().items()}
            res["pG"] = {k: v.cpu().numpy().copy() for k, v in m.netG.state_dict().items()}
        return score

    trainer.train_step = wrapped
    trainer.train(trainset.loader(B), validset.loader(B))
    for tag, params in (("netD", res["pD"]), ("netG", res["pG"])):
        want, init = sub(z, f"ep0.train.{tag}"), sub(z, f"init.{tag}")
        nbad = ntot = 0
        for k in want:
            drift = np.abs(want[k] - init[k]).max()
            nbad += int((np.abs(params[k] - want[k]) > 0.02 * drift + 1e-7).sum())
            ntot += want[k].size
        assert nbad / ntot < 2e-3, (tag, nbad, ntot)
    keys = ("D", "R", "F", "G", "G_", "L1", "L2", "GP")
    for phase in ("train", "valid"):
        want = [float(v) for v in z[f"ep0.{phase}.losses"]]
        got = [float(res[phase]["losses"][k]) for k in keys]
        np.testing.assert_allclose(got, want, rtol=2e-3, atol=2e-3)
        w = [float(v) for v in z[f"ep0.{phase}.scores"]]
        g = [float(res[phase]["scores"][k]) for k in ("NME", "NMAE", "RMSE")]
        np.testing.assert_allclose(g, w, rtol=5e-3, atol=2e-3)
        np.testing.assert_allclose(res[phase]["score"], float(z[f"ep0.{phase}.score"]), rtol=5e-3)
    # checkpoint round trip in the reference's file naming (state_dict, weights_only loadable)
    m.save("0.123", best=True)
    for f in ("netD_BEST.pth", "netG_BEST.pth", "netD_0.123.pth"):
        assert os.path.exists(os.path.join(m.models_path, f)), f
    before = m.netG.state_dict()["lstm2.weight_hh_l0"].clone()
    with torch.no_grad():
        m.netG.lstm2.weight_hh_l0.add_(1.0)
    m.load("BEST")
    assert torch.equal(m.netG.state_dict()["lstm2.weight_hh_l0"], before)

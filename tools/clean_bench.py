#!/usr/bin/env python3
"""Throughput of the cleaning / inference path (BASELINE.json configs[4] shape: 512 variables + 4 time
features, window 48, hidden 64, 8192 windows per launch) on one MI355X: device-resident series -> window
gather -> generator forward -> denormalise -> band (median / std over horizons) -> adjust scan, all on
the device; the series length is a parameter (default 400 000 rows; the 10M-row case is the same loop
25x longer).  Prints one JSON line.   usage: python tools/clean_bench.py [--rows N] [--batch B]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from timeband_amd import ops, synthetic  # noqa: E402
from timeband_amd.cleaner import TIMEBANDCleaner  # noqa: E402
from timeband_amd.dataset import TIMEBANDDataset  # noqa: E402
from timeband_amd.model import TIMEBANDModel  # noqa: E402
from timeband_amd.utils.initiate import seeding  # noqa: E402
from timeband_amd.utils.logger import NullLogger  # noqa: E402
from timeband_amd.utils.lstm_layer import LSTMGenerator  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=400000)
ap.add_argument("--batch", type=int, default=8192)
ap.add_argument("--vars", type=int, default=512)
args = ap.parse_args()
V, To, Tf, H, L, B, N = args.vars, 48, 48, 64, 1, args.batch, args.rows
dev = torch.device("cuda", 0)
seeding(31)
values = synthetic.make_series(N, V, seed=31)
tfeat = synthetic.time_features(N + Tf)
core = dict(logger=NullLogger(), device=dev, batch_size=B, observed_len=To, forecast_len=Tf, directory="",
            data_name="synthetic")
ds = TIMEBANDDataset.from_arrays(values, tfeat, {**core, "stride": 1, "split_rate": 0.9, "min_valid_scale": 2,
                                                 "sliding_step": 0})
del values
holder = TIMEBANDModel.__new__(TIMEBANDModel)
holder.netG = LSTMGenerator(V + tfeat.shape[1], V, To, Tf, H, L, dev).to(dev)
holder.netD = None
cleaner = TIMEBANDCleaner({**core, "band_width": 1.0, "missing_gamma": 0.2, "anomaly_gamma": 0.3}, ds, holder)
full = ds.prepare_dataset(split=False)
nwin = len(full)
for rep in range(2):                      # first pass warms the allocator and the kernels
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out, lo, up = cleaner.clean_arrays(full.loader(B))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
ops.check_errors(dev)
assert bool(torch.isfinite(out).all())
print(json.dumps({"metric": "cleaning windows/sec", "value": round(nwin / dt, 1), "unit": "windows/s",
                  "rows_per_s": round(out.shape[0] / dt, 1), "variables": V, "window": To, "hidden": H,
                  "windows_per_launch": B, "rows": N, "windows": nwin, "seconds": round(dt, 4),
                  "ms_per_launch": round(dt / ((nwin + B - 1) // B) * 1e3, 3)}))

"""Seeded synthetic multivariate hourly series (SURVEY.md §8d "Synthetic inputs").

Used by the golden-vector generator (tests/golden/make_golden.py), the tests and
bench.py.  Nothing here comes from the reference: the reference ships no data and no
generator (its README points at a config.sample.json that is absent).

    x[t, v] = sin(2*pi*t/24*(1+0.1 v)) + 0.5*sin(2*pi*t/168 + v) + 0.1*N(0,1)

with `missing_rate` of the entries set to NaN and `spike_rate` of them displaced by
+-5 sigma.  Timestamps are hourly from 2021-01-01.
"""
from __future__ import annotations

import os

import numpy as np


def make_series(length: int, n_vars: int, seed: int = 31, missing_rate: float = 0.01,
                spike_rate: float = 0.002, dtype=np.float64) -> np.ndarray:
    """(length, n_vars) array, NaN where a value is missing."""
    rng = np.random.default_rng(seed)
    t = np.arange(length, dtype=np.float64)[:, None]
    v = np.arange(n_vars, dtype=np.float64)[None, :]
    x = (np.sin(2.0 * np.pi * t / 24.0 * (1.0 + 0.1 * v))
         + 0.5 * np.sin(2.0 * np.pi * t / 168.0 + v)
         + 0.1 * rng.standard_normal((length, n_vars)))
    if spike_rate > 0:
        sigma = x.std(axis=0, keepdims=True)
        spikes = rng.random((length, n_vars)) < spike_rate
        sign = np.where(rng.random((length, n_vars)) < 0.5, -1.0, 1.0)
        x = np.where(spikes, x + 5.0 * sigma * sign, x)
    if missing_rate > 0:
        miss = rng.random((length, n_vars)) < missing_rate
        x = np.where(miss, np.nan, x)
    return x.astype(dtype)


def hourly_times(length: int, start: str = "2021-01-01"):
    import pandas as pd
    return pd.date_range(start=start, periods=length, freq="h")


def time_features(length: int, start: str = "2021-01-01",
                  enabled=("month", "weekday", "day", "hour")) -> np.ndarray:
    """Time-feature columns in the reference's final column order.

    `parse_timeinfo` (reference source/dataset.py:281-295) prepends year, month, day,
    weekday, hour, minute one after another, so the final left-to-right order is
    minute, hour, weekday, day, month, year (restricted to the enabled ones).
    """
    idx = hourly_times(length, start)
    order = ["minute", "hour", "weekday", "day", "month", "year"]
    cols = []
    for name in order:
        if name in enabled:
            cols.append(np.asarray(getattr(idx, name), dtype=np.float64))
    if not cols:
        return np.zeros((length, 0))
    return np.stack(cols, axis=1)


def write_csv(directory: str, data_name: str, length: int, n_vars: int, seed: int = 31,
              time_index: str = "Date", **kw) -> str:
    """Write `<directory>/<data_name>.csv` in the layout the reference's dataset reads
    (reference source/dataset.py:79-80): a parseable time column plus value columns."""
    import pandas as pd
    x = make_series(length, n_vars, seed, **kw)
    df = pd.DataFrame(x, columns=[f"v{i}" for i in range(n_vars)])
    df.insert(0, time_index, hourly_times(length))
    os.makedirs(directory, exist_ok=True)
    path = os.path.join(directory, f"{data_name}.csv")
    df.to_csv(path, index=False)
    return path


class LazySeries:
    """The same kind of signal as make_series, generated per ROW SLICE on a device with torch: for series too
    long to build on the host (512 variables x 10M rows = 41 GB of float64).  Input generation only -- nothing
    of the product path runs here.  Deterministic per (seed, slice start)."""

    def __init__(self, length, n_vars, seed=31, device="cuda", missing_rate=0.01, spike_rate=0.002):
        self.shape = (int(length), int(n_vars))
        self.seed, self.device, self.missing_rate, self.spike_rate = seed, device, missing_rate, spike_rate

    def __getitem__(self, sl):
        import torch
        r0, r1, step = sl.indices(self.shape[0])
        assert step == 1
        g = torch.Generator(device=self.device).manual_seed(self.seed * 1000003 + r0)
        t = torch.arange(r0, r1, dtype=torch.float64, device=self.device)[:, None]
        v = torch.arange(self.shape[1], dtype=torch.float64, device=self.device)[None, :]
        n = (r1 - r0, self.shape[1])
        rnd = lambda: torch.rand(n, generator=g, device=self.device)
        x = (torch.sin(2.0 * np.pi * t / 24.0 * (1.0 + 0.1 * v)) + 0.5 * torch.sin(2.0 * np.pi * t / 168.0 + v)
             + 0.1 * torch.randn(n, generator=g, device=self.device, dtype=torch.float64))
        if self.spike_rate > 0:
            sign = torch.where(rnd() < 0.5, -1.0, 1.0)
            x = torch.where(rnd() < self.spike_rate, x + 5.0 * 0.8 * sign, x)
        if self.missing_rate > 0:
            x = torch.where(rnd() < self.missing_rate, float("nan"), x)
        return x


class LazyTimeFeatures:
    """hour, weekday, day, month columns of an hourly index with 30-day months (pandas cannot index 10M hours: its
    timestamps end in 2262), per row slice, on the host."""

    def __init__(self, length):
        self.shape = (int(length), 4)

    def __getitem__(self, sl):
        r0, r1, step = sl.indices(self.shape[0])
        t = np.arange(r0, r1, dtype=np.int64)
        d = t // 24
        return np.stack([t % 24, (d + 4) % 7, d % 30 + 1, (d // 30) % 12 + 1], axis=1).astype(np.float64)

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

"""TIMEBANDDataset with device-resident series and on-device window gather.

Drop-in for the reference's source/dataset.py (config-driven constructor, the same attributes:
observed / forecast / missing / decode_min / decode_max / times / data_length / dims /
encode_shape / decode_shape, the same on-disk artefacts original_data.csv, missing_label.csv,
anomaly_label.csv, minmax_info.csv, normalized.csv) with one structural change: windows are never
materialised.  The reference builds an (N,T,E) float64 array of overlapping windows
(dataset.py:199-214) -- impossible at 10M rows x 516 columns -- whereas here the normalised series
lives once in HBM as (len,E) float32 and a batch of windows is gathered by a HIP kernel
(ops.window_gather) when it is needed.  csv parsing / time features stay on the host in pandas
(out of scope of the hot path).
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import ops

_TIME_ORDER = ["minute", "hour", "weekday", "day", "month", "year"]   # final left-to-right column order


class WindowSet:
    """A contiguous range of sliding windows [w0, w0+length) over the device-resident series.
    Window w starts (first forecast row) at series row i = To + w*stride (dataset.py:208-210)."""

    def __init__(self, ds, w0, length):
        self.ds, self.w0, self.length = ds, int(w0), int(length)

    def __len__(self):
        return self.length

    def shape(self, target="encode"):
        d = self.ds
        return (self.length, d.observed_len, d.encode_dim) if target == "encode" else \
            (self.length, d.forecast_len, d.decode_dim)

    def gather(self, start, count):
        d = self.ds
        if start < 0 or count < 0 or start + count > self.length:
            raise IndexError("window range out of bounds")
        first = d.observed_len + (self.w0 + start) * d.stride
        enc, dec = ops.window_gather(d.series_dev, d.tcols_dev, first, count, d.observed_len, d.forecast_len,
                                     stride=d.stride)
        return {"encoded": enc, "decoded": dec}

    def __getitem__(self, idx):
        if isinstance(idx, slice):
            start, stop, step = idx.indices(self.length)
            if step != 1:
                raise IndexError("only contiguous slices are supported")
            return self.gather(start, max(0, stop - start))
        if idx < 0:
            idx += self.length
        if idx >= self.length:
            raise IndexError(idx)
        out = self.gather(idx, 1)
        return {k: v[0] for k, v in out.items()}

    def loader(self, batch_size, rank=0, world=1):
        """Sequential (shuffle=False) batches, like the DataLoader the reference's caller builds.
        Data parallel (world > 1): `batch_size` is the GLOBAL batch; rank r gets its contiguous slice of each
        global batch (dp.shard), and every batch carries `part` = (offset of the slice, windows in the global
        batch) for the halo logic of the post-processing."""
        return _Loader(self, batch_size, rank, world)


class _Loader:
    def __init__(self, ws, batch_size, rank=0, world=1):
        self.ws, self.batch_size, self.rank, self.world = ws, int(batch_size), int(rank), int(world)
        if not 0 <= self.rank < self.world:
            raise ValueError("loader: rank outside the world")

    def __len__(self):
        return (len(self.ws) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        from . import dp
        n = len(self.ws)
        for s in range(0, n, self.batch_size):
            k = min(self.batch_size, n - s)
            if self.world == 1:
                yield self.ws.gather(s, k)
                continue
            if k < self.world:          # fewer windows than ranks: every rank runs the whole (tail) batch
                batch = self.ws.gather(s, k)
                batch["part"] = (0, k, True)
            else:
                lo, hi = dp.shard(k, self.rank, self.world)
                batch = self.ws.gather(s + lo, hi - lo)
                batch["part"] = (lo, k, False)
            yield batch


class TIMEBANDDataset:
    def __init__(self, config: dict) -> None:
        self.logger = config["logger"]
        self.set_config(config)
        self.init_dataset()
        self.data = self.load_dataset()

    def set_config(self, config: dict) -> None:
        self.__dict__ = {**config, **self.__dict__}
        self.stride = int(getattr(self, "stride", 1))
        self.basepath = os.path.join(self.directory, self.data_name)
        self.datapath = os.path.join(self.basepath, "original_data.csv")
        self.missing_path = os.path.join(self.basepath, "missing_label.csv")
        self.anomaly_path = os.path.join(self.basepath, "anomaly_label.csv")
        self.minmax_path = os.path.join(self.basepath, "minmax_info.csv")
        self.normalized_path = os.path.join(self.basepath, "normalized.csv")

    # ------------------------------------------------------------------ host preprocessing (pandas)
    def init_dataset(self) -> None:
        """csv -> expanded frame with time features + label files (reference dataset.py:68-109)."""
        import pandas as pd
        os.makedirs(self.basepath, exist_ok=True)
        fresh = getattr(self, "reset", False) or not os.path.exists(self.datapath)
        if fresh:
            raw = pd.read_csv(os.path.join(self.directory, f"{self.data_name}.csv"), parse_dates=[self.time_index])
            raw = raw.drop(columns=list(self.drops))
            if getattr(self, "fill_timegap", False):
                raw = self._fill_timegap(raw)
            # forecast_len empty rows after the last timestamp (dataset.py:262-279)
            t = raw[self.time_index]
            gap = t.iloc[1] - t.iloc[0]
            extra = pd.DataFrame({self.time_index: [t.iloc[-1] + gap * (k + 1) for k in range(self.forecast_len)]})
            frame = pd.concat([raw, extra], axis=0, ignore_index=True)
            frame = frame.set_index(self.time_index).sort_index()
            # time features are prepended one by one in this order (dataset.py:281-295)
            idx = frame.index
            for name in ("year", "month", "day", "weekday", "hour", "minute"):
                if self.time_encode.get(name) is True:
                    frame.insert(0, name, np.asarray(getattr(idx, name)))
            frame.to_csv(self.datapath)
            frame.isna().astype(int).to_csv(self.missing_path)
            pd.DataFrame(np.zeros(frame.shape), columns=frame.columns, index=frame.index).to_csv(self.anomaly_path)
            self.logger.info(f"  Data has been saved to {self.datapath}")

    def _fill_timegap(self, raw):
        """Insert empty rows where timestamps are missing (reference utils/time.py:14-35, which no
        longer runs on pandas >= 2: DataFrame.append was removed)."""
        import pandas as pd
        t = raw[self.time_index]
        gap = t.iloc[1] - t.iloc[0]
        full = pd.date_range(t.iloc[0], t.iloc[-1], freq=gap)
        return raw.set_index(self.time_index).reindex(full).rename_axis(self.time_index).reset_index()

    def load_dataset(self, datapath: str = None):
        """reference dataset.py:117-158, then the device upload."""
        import pandas as pd
        datapath = self.datapath if datapath is None else datapath
        data = pd.read_csv(datapath, parse_dates=[self.time_index]).set_index(self.time_index)
        data = data.ffill().bfill().fillna(0)                       # dataset.py:111-115
        To, Tf = self.observed_len, self.forecast_len
        tgt = data[self.targets]
        self.observed = torch.from_numpy(tgt[: To + Tf - 1].to_numpy(dtype=np.float64).copy())
        self.forecast = torch.from_numpy(tgt[To:].to_numpy(dtype=np.float64).copy())
        self.missing = pd.read_csv(self.missing_path)[self.targets][To:].to_numpy()
        self.anomaly = pd.read_csv(self.anomaly_path)[self.targets][To:].to_numpy()
        mn, mx = self.minmax_info(data)
        norm = 2 * ((data - mn) / (mx - mn)) - 1                    # dataset.py:216-227
        norm.to_csv(self.normalized_path)
        pd.DataFrame({"min": mn, "max": mx}).to_csv(self.minmax_path)
        self.decode_min = torch.tensor(mn[self.targets].to_numpy(dtype=np.float64))
        self.decode_max = torch.tensor(mx[self.targets].to_numpy(dtype=np.float64))
        self.times = data.index.strftime(self.time_format).tolist()
        cols = list(data.columns)
        self._finish(norm.to_numpy(dtype=np.float64), [cols.index(c) for c in self.targets])
        return norm

    def minmax_info(self, data):
        """reference dataset.py:297-321."""
        k = int(len(data) * self.split_rate)
        mn, mx = data[:k].min(), data[:k].max()
        mn[mn == -np.inf] = mx
        mx[mn == mx] = mx + 1
        return mn, mx

    # ------------------------------------------------------------------ device state
    def _finish(self, norm64, target_cols):
        dev = self.device
        if torch.device(dev).type != "cuda":
            raise RuntimeError("TIMEBANDDataset keeps the series in HBM: config['device'] must be a GPU")
        self.data_length = norm64.shape[0]
        self.encode_dim = norm64.shape[1]
        self.decode_dim = len(target_cols)
        self.dims = {"encode": self.encode_dim, "decode": self.decode_dim}
        self.encode_shape = (self.batch_size, self.observed_len, self.encode_dim)
        self.decode_shape = (self.batch_size, self.forecast_len, self.decode_dim)
        # the reference casts each float64 window to float32 (utils/dataset.py:19-20): same values
        self.series_dev = torch.tensor(np.ascontiguousarray(norm64), dtype=torch.float64).to(torch.float32).to(dev).contiguous()
        self.tcols_dev = torch.tensor(target_cols, dtype=torch.int32, device=dev)
        self.forecast_dev = self.forecast.to(dev).contiguous()
        self.observed_dev = self.observed.to(dev).contiguous()
        self.missing_dev = torch.tensor(np.ascontiguousarray(self.missing), dtype=torch.int32, device=dev).contiguous()
        self.dmin_dev, self.dmax_dev = self.decode_min.to(dev).contiguous(), self.decode_max.to(dev).contiguous()

    @classmethod
    def from_arrays(cls, values, tfeat, config: dict, target_cols=None):
        """Build the dataset from raw arrays without csv / pandas: `values` (N,V) with NaN = missing,
        `tfeat` (N+forecast_len, n_time) time features in final column order.  Same arithmetic as
        init_dataset + load_dataset; used by bench.py and the large-size tests."""
        self = cls.__new__(cls)
        self.logger = config["logger"]
        self.__dict__ = {**config, **self.__dict__}
        self.stride = int(getattr(self, "stride", 1))
        To, Tf = self.observed_len, self.forecast_len
        N, V = values.shape
        frame = np.concatenate([tfeat, np.concatenate([values, np.full((Tf, V), np.nan)], 0)], axis=1)
        nt = tfeat.shape[1]
        tc = (np.arange(V) if target_cols is None else np.asarray(target_cols)) + nt
        miss = np.isnan(frame)
        data = _interpolate(frame)
        k = int(len(data) * self.split_rate)
        mn, mx = data[:k].min(0), data[:k].max(0)
        mn = np.where(mn == -np.inf, mx, mn)
        mx = np.where(mn == mx, mx + 1, mx)
        norm = 2 * ((data - mn) / (mx - mn)) - 1
        self.observed = torch.from_numpy(data[: To + Tf - 1][:, tc].copy())
        self.forecast = torch.from_numpy(data[To:][:, tc].copy())
        self.missing = miss[To:][:, tc].astype(np.int64)
        self.anomaly = np.zeros_like(self.missing)
        self.decode_min, self.decode_max = torch.tensor(mn[tc]), torch.tensor(mx[tc])
        self.targets = [f"v{i}" for i in range(len(tc))]
        self.times = None
        self._finish(norm, list(tc))
        self.data = None
        return self

    # ------------------------------------------------------------------ windows
    def n_windows(self, data_len=None):
        data_len = self.data_length if data_len is None else data_len
        stop = data_len - self.forecast_len
        return max(0, (stop - self.observed_len) // self.stride + 1)

    def prepare_dataset(self, _from: int = 0, k_step: int = 0, split=True):
        """reference dataset.py:160-197; returns WindowSet(s) instead of materialised arrays."""
        if _from != 0:
            raise NotImplementedError("_from != 0 is never used by the reference's callers")
        data_len = self.data_length - getattr(self, "sliding_step", 0) + k_step
        nwin = self.n_windows(data_len)
        if not split:
            full = WindowSet(self, 0, nwin)
            self.logger.info(f" - Data shape : {full.shape()}")
            return full
        valid_minlen = int(self.min_valid_scale * self.forecast_len)
        valid_idx = min(int(data_len * self.split_rate), data_len - valid_minlen)
        split_idx = max(0, min(nwin, valid_idx - self.forecast_len - self.observed_len))
        self.trainset = WindowSet(self, 0, split_idx)
        validset = WindowSet(self, split_idx, nwin - split_idx)
        self.logger.info(f"- Train shape: {self.trainset.shape()} Valid shape: {validset.shape()}")
        return self.trainset, validset

    def get_random(self):
        """Critic batch: forecast_len consecutive training windows from a random start
        (reference dataset.py:244-257; draws from the global numpy RandomState like the reference)."""
        idx = np.random.randint(self.trainset.length - self.forecast_len)
        b = self.trainset.gather(idx, self.forecast_len)
        return b["encoded"], b["decoded"]

    def denormalize(self, data):
        """reference dataset.py:229-242 (fp32 result); out of place."""
        if not hasattr(self, "decode_max") or not hasattr(self, "decode_min"):
            raise Exception("Try to denormalize, but the input was not normalized")
        return ops.denormalize(data, self.dmin_dev, self.dmax_dev)


def _interpolate(a):
    """ffill, bfill, then 0 (reference dataset.py:111-115) on a float64 array."""
    a = np.array(a, np.float64)
    n = a.shape[0]
    idx = np.where(~np.isnan(a), np.arange(n)[:, None], -1)
    np.maximum.accumulate(idx, axis=0, out=idx)
    filled = np.take_along_axis(a, np.maximum(idx, 0), axis=0)
    filled[idx < 0] = np.nan
    # backward fill of the leading NaNs
    first = np.argmax(~np.isnan(filled), axis=0)
    for j in range(a.shape[1]):
        col = filled[:, j]
        if np.isnan(col).all():
            col[:] = 0.0
        else:
            col[:first[j]] = col[first[j]]
    return filled

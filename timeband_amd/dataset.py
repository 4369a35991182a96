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

    def __init__(self, ds, w0, length, row0=0):
        """row0: first series row of the sub-frame the windows are cut from (prepare_dataset's `_from`)."""
        self.ds, self.w0, self.length, self.row0 = ds, int(w0), int(length), int(row0)

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
        first = self.row0 + d.observed_len + (self.w0 + start) * d.stride
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
        return self._batches(self.ws.gather)

    def positions(self):
        """The same batches WITHOUT gathering them: {"windows": (first forecast row of the first window, count, stride)}
        (+ "part" under data parallelism) -- for consumers that read the windows in place from the device-resident
        series (LSTMGenerator.infer_windows: the cleaning / prediction pass)."""
        d = self.ws.ds

        def where(start, count):
            if start < 0 or count < 0 or start + count > self.ws.length:
                raise IndexError("window range out of bounds")
            return {"windows": (self.ws.row0 + d.observed_len + (self.ws.w0 + start) * d.stride, count, d.stride)}
        return self._batches(where)

    def _batches(self, make):
        from . import dp
        n = len(self.ws)
        for s in range(0, n, self.batch_size):
            k = min(self.batch_size, n - s)
            if self.world == 1:
                yield make(s, k)
                continue
            if k < self.world:          # fewer windows than ranks: every rank runs the whole (tail) batch
                batch = make(s, k)
                batch["part"] = (0, k, True)
            else:
                lo, hi = dp.shard(k, self.rank, self.world)
                batch = make(s + lo, hi - lo)
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
        """reference dataset.py:117-158.  csv parsing and the ffill / bfill stay in pandas (host I/O); the min-max
        statistics and the normalisation run on the device (tb_minmax, tb_normalize) on the uploaded frame."""
        import pandas as pd
        datapath = self.datapath if datapath is None else datapath
        data = pd.read_csv(datapath, parse_dates=[self.time_index]).set_index(self.time_index)
        data = data.ffill().bfill().fillna(0)                       # dataset.py:111-115
        To = self.observed_len
        self.missing = pd.read_csv(self.missing_path)[self.targets][To:].to_numpy()
        self.anomaly = pd.read_csv(self.anomaly_path)[self.targets][To:].to_numpy()
        self.times = data.index.strftime(self.time_format).tolist()
        cols = list(data.columns)
        self._check_device()
        raw = torch.from_numpy(np.ascontiguousarray(data.to_numpy(dtype=np.float64))).to(self.device)
        norm64 = self._device_state(raw, [cols.index(c) for c in self.targets], want64=True)
        self.missing_dev = torch.tensor(np.ascontiguousarray(self.missing), dtype=torch.int32, device=self.device).contiguous()
        # the reference's artefacts (dataset.py:62-66), written from the device results
        mn, mx = self._mn.cpu().numpy(), self._mx.cpu().numpy()
        norm = pd.DataFrame(norm64.cpu().numpy(), columns=data.columns, index=data.index)
        norm.to_csv(self.normalized_path)
        pd.DataFrame({"min": pd.Series(mn, index=data.columns), "max": pd.Series(mx, index=data.columns)}).to_csv(self.minmax_path)
        return norm

    def minmax_info(self, data):
        """reference dataset.py:297-321 on a (rows, E) float64 device frame -> (min, max) device vectors."""
        k = int(data.shape[0] * self.split_rate)
        return ops.minmax(data[:k])

    # ------------------------------------------------------------------ device state
    def _check_device(self):
        if torch.device(self.device).type != "cuda":
            raise RuntimeError("TIMEBANDDataset keeps the series in HBM: config['device'] must be a GPU")

    def _device_state(self, raw, target_cols, want64=False):
        """From the interpolated float64 frame in HBM: min-max over the first split_rate rows, the float32
        normalised series, the raw target columns (observed / forecast) and the decode range.  The host copies
        the reference exposes (observed, forecast, decode_min / decode_max) are made only for frames small
        enough to be worth a host tensor (the 10M x 512 case keeps everything on the device)."""
        To, Tf = self.observed_len, self.forecast_len
        R, E = raw.shape
        self.data_length, self.encode_dim, self.decode_dim = R, E, len(target_cols)
        self.dims = {"encode": self.encode_dim, "decode": self.decode_dim}
        self.encode_shape = (self.batch_size, To, self.encode_dim)
        self.decode_shape = (self.batch_size, Tf, self.decode_dim)
        self.tcols_dev = torch.tensor(target_cols, dtype=torch.int32, device=raw.device)
        self._mn, self._mx = self.minmax_info(raw)
        res = ops.normalize(raw, self._mn, self._mx, want64=want64)
        self.series_dev, norm64 = res if want64 else (res, None)
        self.forecast_dev = ops.take_columns(raw[To:], self.tcols_dev)
        self.observed_dev = ops.take_columns(raw[: To + Tf - 1], self.tcols_dev)
        idx = self.tcols_dev.long()
        self.dmin_dev, self.dmax_dev = self._mn[idx].contiguous(), self._mx[idx].contiguous()
        self.decode_min, self.decode_max = self.dmin_dev.cpu(), self.dmax_dev.cpu()
        self.observed = self.observed_dev.cpu()
        self.forecast = self.forecast_dev.cpu() if R * self.decode_dim <= self.HOST_COPY_LIMIT else None
        return norm64

    HOST_COPY_LIMIT = 1 << 26       # elements: above it no host mirror of forecast / missing is kept
    CHUNK_ROWS = 1 << 18            # from_arrays: rows interpolated on the host and uploaded at a time

    @classmethod
    def from_arrays(cls, values, tfeat, config: dict, target_cols=None):
        """Build the dataset from raw arrays without csv / pandas: `values` (N,V) with NaN = missing,
        `tfeat` (N+forecast_len, n_time) time features in final column order.  Same arithmetic as
        init_dataset + load_dataset; used by bench.py and the large-size tests.  `values` / `tfeat` only have to
        support row slicing (numpy arrays, or objects that produce host arrays or device tensors per slice): the
        frame goes up in row chunks and the wide case (512 variables x 10M rows) never holds a second host copy."""
        self = cls.__new__(cls)
        self.logger = config["logger"]
        self.__dict__ = {**config, **self.__dict__}
        self.stride = int(getattr(self, "stride", 1))
        self._check_device()
        dev = self.device
        To, Tf = self.observed_len, self.forecast_len
        N, V = values.shape
        nt = tfeat.shape[1]
        R, E = N + Tf, nt + V
        tc = (np.arange(V) if target_cols is None else np.asarray(target_cols)) + nt
        # the raw frame goes up in row chunks (host arrays, or device tensors from a device-side generator);
        # everything after that is device work: fill (tb_interpolate, which also records the missing mask),
        # min-max, normalisation, target columns
        raw = torch.empty(R, E, dtype=torch.float64, device=dev)
        up = lambda a: (a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a))).to(device=dev, dtype=torch.float64)
        for r0 in range(0, R, self.CHUNK_ROWS):
            r1 = min(R, r0 + self.CHUNK_ROWS)
            raw[r0:r1, :nt] = up(tfeat[r0:r1])
            if r0 < N:
                raw[r0:min(r1, N), nt:] = up(values[r0:min(r1, N)])
            if r1 > N:
                raw[max(r0, N):r1, nt:] = float("nan")          # forecast_len empty rows (dataset.py:262-279)
        tpos = torch.full((E,), -1, dtype=torch.int32)
        tpos[torch.as_tensor(tc, dtype=torch.long)] = torch.arange(len(tc), dtype=torch.int32)
        self.missing_dev = ops.interpolate(raw, tpos.to(dev), len(tc), To)
        keep_host = R * len(tc) <= self.HOST_COPY_LIMIT
        self.missing = self.missing_dev.cpu().numpy().astype(np.int64) if keep_host else None
        self.anomaly = np.zeros_like(self.missing) if keep_host else None
        self._device_state(raw, list(tc))
        del raw
        self.targets = [f"v{i}" for i in range(len(tc))]
        self.times = None
        self.data = None
        return self

    # ------------------------------------------------------------------ windows
    def n_windows(self, data_len=None):
        data_len = self.data_length if data_len is None else data_len
        stop = data_len - self.forecast_len
        return max(0, (stop - self.observed_len) // self.stride + 1)

    def prepare_dataset(self, _from: int = 0, k_step: int = 0, split=True):
        """reference dataset.py:160-197; returns WindowSet(s) instead of materialised arrays."""
        _from = int(_from)
        if _from < 0:
            _from += self.data_length
        sliding_len = self.data_length - getattr(self, "sliding_step", 0) + k_step
        data_len = max(0, min(sliding_len, self.data_length) - _from)       # len(self.data[_from:sliding_len])
        nwin = self.n_windows(data_len)
        if not split:
            full = WindowSet(self, 0, nwin, _from)
            self.logger.info(f" - Data shape : {full.shape()}")
            return full
        valid_minlen = int(self.min_valid_scale * self.forecast_len)
        valid_idx = min(int(data_len * self.split_rate), data_len - valid_minlen)
        split_idx = max(0, min(nwin, valid_idx - self.forecast_len - self.observed_len))
        self.trainset = WindowSet(self, 0, split_idx, _from)
        validset = WindowSet(self, split_idx, nwin - split_idx, _from)
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

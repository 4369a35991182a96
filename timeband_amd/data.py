"""Gen-B data front-end (reference source/data.py): `TIMEBANDData` with explicit keyword arguments,
the origin/ target/ meta/ directory layout and the minmax.csv / missing.csv artefacts, and
`MyDataset` whose items are (encoded, decoded) tensors.

Host-side pandas as in the reference (this is set-up, not the hot path); the windows are views of
one contiguous array instead of a Python list of slices.  The reference's debugging stop in
prepare_trainset (data.py:126-127, print + input()) is not reproduced.
"""
import os
from typing import List, Optional, Union

import numpy as np
import pandas as pd
import torch
import torch.utils.data as data

from .utils.initiate import init_device


class MyDataset(data.Dataset):
    def __init__(self, encoded, decoded):
        super().__init__()
        self.encoded, self.decoded = encoded, decoded
        self.length = len(self.encoded)

    def shape(self, target="encode"):
        return self.encoded.shape if target == "encode" else self.decoded.shape

    def __len__(self):
        return self.length

    def __getitem__(self, idx):
        return (torch.tensor(np.asarray(self.encoded[idx]), dtype=torch.float32),
                torch.tensor(np.asarray(self.decoded[idx]), dtype=torch.float32))


class TIMEBANDData:
    def __init__(self, basedir: str, filename: str, targets: List[str], drops: List[str], fill_timegap: bool,
                 time_index: List[str], time_encode: List[str], split_size: Union[int, float], observed_len: int,
                 forecast_len: int, device=None) -> None:
        self.device = torch.device(device) if device is not None else init_device()
        self.basedir, self.filename = basedir, filename
        self.targets, self.drops = list(targets), list(drops)
        self.fill_timegap = fill_timegap
        self.time_index = [time_index] if isinstance(time_index, str) else list(time_index)
        self.time_encode = list(time_encode)
        self.split_size = split_size
        self.observed_len, self.forecast_len = observed_len, forecast_len
        # data.py:62-72
        self.basepath = os.path.join(basedir, "origin")
        self.datapath = os.path.join(basedir, "target")
        self.metapath = os.path.join(basedir, "meta", filename)
        os.makedirs(self.datapath, exist_ok=True)
        os.makedirs(self.metapath, exist_ok=True)
        self.origin_path = os.path.join(self.basepath, f"{filename}.csv")
        self.target_path = os.path.join(self.datapath, f"{filename}.csv")
        self.minmax_path = os.path.join(self.metapath, "minmax.csv")
        self.missing_path = os.path.join(self.metapath, "missing.csv")
        head = pd.read_csv(self.origin_path, parse_dates=self.time_index, index_col=self.time_index, nrows=1)
        head = self.parse_timeinfo(head).drop(self.drops, axis=1)
        self.encode_dims = len(head.columns)
        self.decode_dims = len(self.targets)

    def init_dataset(self, index_s: Optional[int] = 0, index_e: Optional[int] = None) -> pd.DataFrame:
        """data.py:87-114: writes target/<name>.csv, meta/<name>/missing.csv and minmax.csv."""
        frame = pd.read_csv(self.origin_path, parse_dates=self.time_index).drop(self.drops, axis=1)
        frame = frame[index_s:index_e]
        if self.fill_timegap:
            frame = self._fill_timegap(frame)
        frame.to_csv(self.target_path, index=False)
        full = self.parse_timeinfo(frame.set_index(self.time_index))
        self.observed = full
        self.forecast = full[self.targets]
        full.isna().astype(int).to_csv(self.missing_path)
        self.minmax_info(full).to_csv(self.minmax_path, index=False)
        return frame

    def _fill_timegap(self, frame):
        """utils/time.py:14-35 (which needs DataFrame.append, gone from pandas 2)."""
        col = self.time_index[0]
        t = frame[col].reset_index(drop=True)
        full = pd.date_range(t.iloc[0], t.iloc[-1], freq=t.iloc[1] - t.iloc[0])
        return frame.set_index(col).reindex(full).rename_axis(col).reset_index()

    def _windows(self, x, y, stride, stop):
        To, Tf = self.observed_len, self.forecast_len
        starts = np.arange(To, stop, stride)
        xv = np.lib.stride_tricks.sliding_window_view(x, To, axis=0).transpose(0, 2, 1)
        enc = xv[starts - To] if len(starts) else np.empty((0, To, x.shape[1]))
        if y is None:
            return enc, None
        yv = np.lib.stride_tricks.sliding_window_view(y, Tf, axis=0).transpose(0, 2, 1)
        dec = yv[starts] if len(starts) else np.empty((0, Tf, y.shape[1]))
        return enc, dec

    def prepare_trainset(self, frame: pd.DataFrame, stride: int = 1):
        """data.py:116-160."""
        frame = frame.set_index(self.time_index)
        frame = frame.ffill().bfill().replace(np.nan, 0)
        frame = self.normalize(self.parse_timeinfo(frame))
        missing = frame.isna().astype(int)
        self.missing_encode, self.missing_decode = missing, missing[self.targets]
        x = frame.to_numpy(dtype=np.float64)
        y = frame[self.targets].to_numpy(dtype=np.float64)
        encoded, decoded = self._windows(x, y, stride, len(frame) - self.forecast_len + 1)
        self.dimensions = x.shape[-1], y.shape[-1]
        split = len(frame) - self.split_size
        if type(self.split_size) is float:
            split = int(len(frame) * self.split_size)
        split = min(split, len(frame) - self.observed_len - self.forecast_len)
        return MyDataset(encoded[:split], decoded[:split]), MyDataset(encoded[split:], decoded[split:])

    def prepare_predset(self, frame: pd.DataFrame, stride: int = 1):
        """data.py:162-178."""
        frame = self.normalize(self.parse_timeinfo(frame.set_index(self.time_index)))
        missing = frame.isna().astype(int)
        self.missing_encode, self.missing_decode = missing, missing[self.targets]
        encoded, _ = self._windows(frame.to_numpy(dtype=np.float64), None, stride, len(frame))
        return MyDataset(encoded, encoded)

    def minmax_info(self, frame: pd.DataFrame) -> pd.DataFrame:
        """data.py:180-196."""
        split = int(len(frame) * self.split_size)
        return pd.DataFrame({"Feats": frame.columns, "min": frame[:split].min().to_numpy(),
                             "max": frame[:split].max().to_numpy()})

    def normalize(self, frame: pd.DataFrame) -> pd.DataFrame:
        """[-1, 1] by the saved min/max (data.py:198-214)."""
        minmax = pd.read_csv(self.minmax_path, index_col=["Feats"])
        lo, hi = minmax["min"], minmax["max"]
        frame = 2 * ((frame - lo) / (hi - lo)) - 1
        f64 = lambda s: torch.tensor(s[self.targets].to_numpy(dtype=np.float64)).to(self.device)
        self.decode_min, self.decode_max = f64(lo), f64(hi)
        self.delta = self.decode_max - self.decode_min
        return frame

    def denormalize(self, x):
        """data.py:216-225 on the device (tb_denormalize)."""
        if not hasattr(self, "decode_max") or not hasattr(self, "decode_min"):
            raise Exception("Try to denormalize, but the input was not normalized")
        from . import ops
        return ops.denormalize(x.detach().to(self.device), self.decode_min, self.decode_max)

    def parse_timeinfo(self, frame: pd.DataFrame) -> pd.DataFrame:
        """data.py:227-245: each requested feature is prepended, in the order given."""
        for name in self.time_encode:
            if name not in ("year", "month", "weekday", "day", "hour", "minute"):
                raise KeyError(name)
            info = pd.DataFrame({name: np.asarray(getattr(frame.index, name))}, index=frame.index)
            frame = pd.concat([info, frame], axis=1)
        return frame

    def get_random(self, dataset: MyDataset):
        """forecast_len consecutive windows from a random start (data.py:247-257)."""
        idx = np.random.randint(len(dataset) - self.forecast_len)
        return dataset[idx: idx + self.forecast_len]

"""TIMEBANDCleaner / batched detect-and-impute on the HIP path (reference source/cleaner.py:21-179).

clean(dataset) returns (outputs_df, bands_df) like the reference; `clean_arrays` returns the device
tensors without building DataFrames (what the wide 512-variable / 10M-row case uses)."""
import os

import torch

from .postprocess import BandAdjust


class TIMEBANDCleaner:
    variant = "cleaner"

    def __init__(self, config: dict, dataset, models, metric=None, losses=None, dashboard=None) -> None:
        self.logger = config["logger"]
        self.dataset, self.models, self.metric, self.losses, self.dashboard = dataset, models, metric, losses, dashboard
        self.__dict__ = {**config, **self.__dict__}
        self.forecast_len = self.dataset.forecast_len

    def clean_arrays(self, dataset):
        post = BandAdjust(self.dataset, self.variant, self.band_width, self.missing_gamma, self.anomaly_gamma,
                          keep_outputs=True)
        netG = self.models.netG
        dev = self.device
        # the product's own loader hands out window POSITIONS: the generator then reads the overlapping windows in place
        # from the device-resident series (no gathered copy, the first projection once per series row); any other
        # iterable of {"encoded": ...} batches (the reference's DataLoader shape) works as before
        in_place = hasattr(dataset, "positions") and os.environ.get("TB_CLEAN_INPLACE", "1") != "0"
        sweep = None
        with torch.no_grad():
            for data in (dataset.positions() if in_place else dataset):
                part = data.get("part") if isinstance(data, dict) else None
                if "windows" in data:
                    first, count, stride = data["windows"]
                    if part is None:     # one rank: generator forward of the next batch under this batch's band / adjust
                        if sweep is None or sweep.stride != stride:
                            if sweep is not None:
                                sweep.flush()
                            sweep = post.sweep(netG, self.dataset.series_dev, stride)
                        sweep.push(first, count)
                    else:
                        post.step_windows(netG, self.dataset.series_dev, first, count, stride, part)
                else:
                    post.step(netG(data["encoded"].to(dev))[:, : self.forecast_len], part)
            if sweep is not None:
                sweep.flush()
        from . import ops
        ops.check_errors(dev)
        n = self.dataset.observed_len + post.idx + self.forecast_len - 1      # rows written so far
        return post.outputs[:n], post.lower_bands[:n], post.upper_bands[:n]

    def clean(self, dataset):
        import pandas as pd
        self.logger.info("RUN the model")
        out, lo, up = (t.cpu().numpy() for t in self.clean_arrays(dataset))
        ds = self.dataset
        index = ds.times[: out.shape[0]] if ds.times is not None else None
        outputs_df = pd.DataFrame(out, columns=ds.targets, index=index)
        bands_df = pd.concat([pd.DataFrame(lo, columns=[f"{x}_lower" for x in ds.targets], index=index),
                              pd.DataFrame(up, columns=[f"{x}_upper" for x in ds.targets], index=index)], axis=1)
        name = getattr(ds, "time_index", None)
        bands_df.index.name = outputs_df.index.name = name
        return outputs_df, bands_df

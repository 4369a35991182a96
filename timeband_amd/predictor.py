"""TIMEBANDPredictor (reference source/predictor.py:22-155): the cleaner's pass with the
predictor's order of the adjust terms (anomaly pull-back first, then missing fill) and no bands."""
from .cleaner import TIMEBANDCleaner


class TIMEBANDPredictor(TIMEBANDCleaner):
    variant = "predictor"

    def predict(self, dataset):
        import pandas as pd
        self.logger.info("Predicts the forecast data")
        out, _, _ = self.clean_arrays(dataset)
        out = out.cpu().numpy()
        ds = self.dataset
        index = ds.times[: out.shape[0]] if ds.times is not None else None
        return pd.DataFrame(out, columns=ds.targets, index=index)

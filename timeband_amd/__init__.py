"""timeband_amd: MI355X-native hot path of Timeband's LSTM-GAN training and cleaning.

Public surface mirrors the reference's Gen-A modules (source/*.py): TIMEBANDDataset, TIMEBANDModel,
TIMEBANDLoss, TIMEBANDMetric, TIMEBANDTrainer, TIMEBANDCleaner, TIMEBANDPredictor,
utils.lstm_layer.LSTMGenerator / LSTMDiscriminator.  Submodules import torch lazily-by-use; the
HIP library is loaded on first kernel call (timeband_amd._lib.load) and its absence is an error.
"""
__all__ = ["TIMEBANDDataset", "TIMEBANDModel", "TIMEBANDLoss", "TIMEBANDMetric", "TIMEBANDTrainer",
           "TIMEBANDCleaner", "TIMEBANDPredictor"]


def __getattr__(name):
    import importlib
    table = {"TIMEBANDDataset": "dataset", "TIMEBANDModel": "model", "TIMEBANDLoss": "loss",
             "TIMEBANDMetric": "metric", "TIMEBANDTrainer": "trainer", "TIMEBANDCleaner": "cleaner",
             "TIMEBANDPredictor": "predictor"}
    if name in table:
        return getattr(importlib.import_module(f"{__name__}.{table[name]}"), name)
    raise AttributeError(name)

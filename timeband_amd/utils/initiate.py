"""Seeding and device selection (reference: source/utils/initiate.py:7-28).

One process per GPU: the device is LOCAL_RANK (torchrun) instead of the reference's fixed cuda:0.
There is no CPU device for the hot path: init_device raises when no GPU is visible.
"""
import os
import random

import numpy as np
import torch


def seeding(seed: int = 31) -> None:
    """Same three generators the reference seeds, so that parameter init (torch) and the critic's
    window draw (global numpy RandomState, source/dataset.py:250) reproduce the reference run."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)


def init_device() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("timeband_amd needs an MI355X: no GPU is visible and the hot path has no CPU fallback")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    return torch.device("cuda", local)

"""ctypes binding of libtimeband_hip.so (C-ABI: include/timeband_hip.h).

The product path has NO fallback: if the library is missing or a call fails, a RuntimeError is
raised.  Nothing here imports oracle/.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TB_LIB_PATH") or os.path.join(_HERE, "libtimeband_hip.so")   # TB_LIB_PATH: A/B builds (tools)

_lib = None

P = C.c_void_p
I = C.c_int
L = C.c_long
D = C.c_double
F = C.c_float
Z = C.c_size_t

# name -> (restype, argtypes); must list every symbol include/timeband_hip.h declares
SIGNATURES = {
    "tb_version": (I, []),
    "tb_last_error": (C.c_char_p, []),
    "tb_init": (I, [I]),
    "tb_device_cus": (I, []),
    "tb_prof_enable": (I, [I, I]),
    "tb_prof_read": (I, [I, C.POINTER(I), C.POINTER(D), C.POINTER(D)]),
    "tb_lstm_supported": (I, [I]),
    "tb_lstm_exclusive": (I, [I, I, I, I]),
    "tb_lstm_workspace_bytes": (Z, [I, I, I, I, I]),
    "tb_lstm_fwd": (I, [P, P, P, P, P, P, P, P, I, I, I, I, I, P, Z, P, P]),
    "tb_lstm_fwd_windows": (I, [P, L, L, L, P, P, I, I, I, P, Z, P, P]),
    "tb_lstm_wgrad_workspace_bytes": (Z, [I, I]),
    "tb_lstm_wgrad": (I, [P, P, P, P, P, I, I, I, I, I, P, Z, P]),
    "tb_lstm_static_init": (I, []),
    "tb_lstm_static_scratch_bytes": (Z, []),
    "tb_lstm_fwd_static": (I, [P, P, P, P, P, P, P, P, I, I, I, I, I, P, P]),
    "tb_lstm_bwd": (I, [P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, P, Z, P, P]),
    "tb_lstm_bwd_rows": (I, [P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, P, Z, P, P, I]),
    "tb_linear_act_fwd": (I, [P, P, P, P, I, I, I, I, P]),
    "tb_linear_workspace_bytes": (Z, [I, I, I]),
    "tb_linear_bwd": (I, [P, P, P, P, P, P, I, I, I, I, P, Z, P]),
    "tb_act_bwd": (I, [P, P, P, L, I, P]),
    "tb_colsum_workspace_bytes": (Z, [I]),
    "tb_colsum": (I, [P, L, I, P, P, Z, P]),
    "tb_head_supported": (I, [I]),
    "tb_head_workspace_bytes": (Z, [L, I]),
    "tb_head_fwd": (I, [P, P, P, P, L, I, I, P]),
    "tb_head_bwd": (I, [P, P, P, P, P, P, P, L, I, I, I, P, Z, P]),
    "tb_loss_workspace_bytes": (Z, []),
    "tb_loss_sums": (I, [P, L, P, L, P, P, L, P, P, Z, P]),
    "tb_loss_finish": (I, [P, I, D, D, D, D, F, F, F, P, P, P, P]),
    "tb_loss_sums_finish": (I, [P, L, P, L, P, P, L, P, P, Z, I, D, D, D, D, F, F, F, P, P, P, P]),
    "tb_gan_loss_grad_pair": (I, [P, L, D, D, I, P, P, P]),
    "tb_gan_loss_grad": (I, [P, L, D, I, I, P, P, P]),
    "tb_recon_loss_grad": (I, [P, P, L, D, D, P, F, F, F, P, P, I, P]),
    "tb_adam_step": (I, [P, P, P, P, L, D, D, D, D, I, P]),
    "tb_rmsprop_step": (I, [P, P, P, L, D, D, D, I, P]),
    "tb_zero": (I, [P, Z, P]),
    "tb_grad_merge": (I, [P, P, L, P]),
    "tb_concat2": (I, [P, Z, P, Z, P, P]),
    "tb_concat2_sum": (I, [P, L, P, P, L, P, P]),
    "tb_interpolate_workspace_bytes": (Z, [L, I]),
    "tb_interpolate": (I, [P, L, I, P, I, L, P, P, Z, P]),
    "tb_minmax_workspace_bytes": (Z, [I]),
    "tb_minmax": (I, [P, L, I, P, P, I, I, P, Z, P]),
    "tb_normalize": (I, [P, L, I, P, P, P, P, P]),
    "tb_take_columns": (I, [P, L, I, P, I, P, P]),
    "tb_window_gather": (I, [P, L, I, P, L, L, I, I, I, P, I, P, P, P]),
    "tb_denormalize": (I, [P, L, I, P, P, P, P]),
    "tb_band_update": (I, [P, I, I, I, P, I, P, P, P, D, P, P, P, P, P]),
    "tb_band_update_part": (I, [P, I, I, I, P, I, P, P, P, D, P, P, P, P, I, I, P]),
    "tb_dimband_update": (I, [P, I, I, I, P, P, I, I, P, I, P, P, D, P, P, P, P, I, I, P]),
    "tb_metric_sums": (I, [P, P, P, L, P, P]),
    "tb_metric_finish": (I, [P, I, P, P, P]),
    "tb_adjust_scan": (I, [I, P, P, P, P, P, P, I, I, D, D, P, P]),
    "tb_metrics": (I, [P, P, P, L, P, P, P]),
}

K_GEMM, K_LSTM_FWD, K_LSTM_BWD, K_LOSS, K_OPTIM, K_GATHER, K_BAND = range(7)


def load():
    """Load the shared library (once).  Raises RuntimeError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m timeband_amd.build` "
            "(there is no CPU/PyTorch fallback for the hot path)")
    # torch must be imported first: the wheel bundles its own HIP runtime, and our library has to bind
    # to THAT copy (same soname) -- loading /opt/rocm's runtime first leaves two runtimes in the process
    # and torch then sees no device.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error() -> str:
    return load().tb_last_error().decode("utf-8", "replace")


def check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"{what} failed (code {rc}): {last_error()}")


def call(name: str, *args):
    """Call an int-returning entry point and raise on a non-zero return."""
    rc = getattr(load(), name)(*args)
    check(rc, name)

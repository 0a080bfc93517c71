"""Run-time configuration of the HIP path.

feature dtype: element type of the NHWC feature maps between kernels.
  * torch.bfloat16 (default): bf16 storage, bf16 MFMA, fp32 accumulation -- the benchmark configuration (BASELINE.json configs[1]).
  * torch.float32: "parity mode" -- fp32 storage, every MFMA product computed as a bf16 hi/lo split (3 MFMAs), which
    reproduces the reference's fp32 results within the 1e-3 relative tolerance BASELINE.json states.
"""
from __future__ import annotations

import contextlib

import torch

_feature_dtype = torch.bfloat16


def feature_dtype() -> torch.dtype:
    return _feature_dtype


def set_feature_dtype(dtype: torch.dtype) -> None:
    global _feature_dtype
    if dtype not in (torch.bfloat16, torch.float32):
        raise ValueError("feature dtype must be torch.bfloat16 or torch.float32")
    _feature_dtype = dtype


@contextlib.contextmanager
def feature_dtype_as(dtype: torch.dtype):
    old = _feature_dtype
    set_feature_dtype(dtype)
    try:
        yield
    finally:
        set_feature_dtype(old)

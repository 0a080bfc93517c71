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
_conv_operands = "bf16"      # "bf16" | "fp8": MFMA operand type of the eligible convolution launches (feature maps stay bf16)


def feature_dtype() -> torch.dtype:
    return _feature_dtype


import os as _os
_flow_gemm = _os.environ.get("LCGAN_FLOW_GEMM", "1") != "0"     # A/B switch: the flow layer as 1x1 GEMM + scatter (ops.FlowConvFn) or the generic up-conv


def flow_gemm() -> bool:
    return _flow_gemm


def set_flow_gemm(on: bool) -> None:
    global _flow_gemm
    _flow_gemm = bool(on)


_synth_fork = _os.environ.get("LCGAN_SYNTH_FORK", "1") != "0"   # A/B switch: a SynthesisBlock's three consumers of its input as one autograd node (ops.SynthForkFn)


def synth_fork() -> bool:
    return _synth_fork


def set_synth_fork(on: bool) -> None:
    global _synth_fork
    _synth_fork = bool(on)


_batched_passes = _os.environ.get("LCGAN_BATCHED_PASSES", "1") != "0"   # A/B switch: the passes of an iteration that share weights as ONE batch (worker.py)


def batched_passes() -> bool:
    return _batched_passes


def set_batched_passes(on: bool) -> None:
    global _batched_passes
    _batched_passes = bool(on)


_act_masks = _os.environ.get("LCGAN_ACT_MASKS", "1") != "0"   # A/B switch: leaky-ReLU sign masks as a by-product of the discriminator's convolution epilogues (ops.Conv2dFn / ConvPoolFn)


def act_masks() -> bool:
    return _act_masks


_gz_demod = _os.environ.get("LCGAN_GZ_DEMOD", "1") != "0"     # A/B switch: a modulated convolution's activation backward stores d[b,o] * gz, so that its data- and weight-gradient launches run without a per-sample operand scale (ops._modconv_backward)


def gz_demod() -> bool:
    return _gz_demod


def set_gz_demod(on: bool) -> None:
    global _gz_demod
    _gz_demod = bool(on)


def set_act_masks(on: bool) -> None:
    global _act_masks
    _act_masks = bool(on)


def set_feature_dtype(dtype: torch.dtype) -> None:
    global _feature_dtype
    if dtype not in (torch.bfloat16, torch.float32):
        raise ValueError("feature dtype must be torch.bfloat16 or torch.float32")
    _feature_dtype = dtype


def conv_operands() -> str:
    return _conv_operands


def set_conv_operands(kind: str) -> None:
    """"fp8" = BASELINE configs[4]: the stride-1 3x3 / 1x1 forward and data-gradient convolutions on grids >= 16 x 16 run on MX block-scaled
    e4m3 operands (v_mfma_scale_f32_32x32x64_f8f6f4, fp32 accumulate, bf16 feature maps); weight gradients, stride-2 / transposed
    convolutions and low-resolution layers stay bf16.  Precision: ~4 % relative L2 per convolution output (tests/test_kernels_gpu.py)."""
    global _conv_operands
    if kind not in ("bf16", "fp8"):
        raise ValueError("conv operands must be 'bf16' or 'fp8'")
    _conv_operands = kind


@contextlib.contextmanager
def conv_operands_as(kind: str):
    old = _conv_operands
    set_conv_operands(kind)
    try:
        yield
    finally:
        set_conv_operands(old)


@contextlib.contextmanager
def feature_dtype_as(dtype: torch.dtype):
    old = _feature_dtype
    set_feature_dtype(dtype)
    try:
        yield
    finally:
        set_feature_dtype(old)


def default_args(res=32, batch=8, **kw):
    """The argument namespace main.py's parser produces with its defaults (main.py:12-60 of the reference), for callers that build
    a WORKER without the CLI: bench.py, scripts/, tests."""
    import types
    a = types.SimpleNamespace(
        phase="train", img_resolution=res, batch_size=batch, geo_latent_dim=64, app_latent_dim=512, geo_noise_dim=64,
        app_noise_dim=64, max_flow_scale=0.1, geo_projection_dim=256, app_projection_dim=256, tau=0.05, l_adv=1.0, l_aux=0.5,
        l_r1=10.0, l_s=1e-7, g_lr=0.002, d_lr=0.002, beta1=0.0, beta2=0.99, g_ema_decay=0.9999, g_ema_start=0,
        freezeD_start=100000, freezeD_layer=5, dataset_path="synthetic", model_name="", save_dir="model", sample_dir="samples",
        best=False, epoch=1, print_interval=100, save_interval=5000, show_interval=1000, psi=2.0, w_psi=1.0, num_fakes=10,
        ctrl_dim=-1, num_videos=10, img_ch=3)
    for k, v in kw.items():
        setattr(a, k, v)
    return a

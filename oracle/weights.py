"""Deterministic, numpy-seeded weights shared by the golden-vector generator and the tests
(TEST INFRASTRUCTURE).  Weights are never stored in fixtures: both sides rebuild them from
(key, shape, seed) with the rule below, so a fixture only carries inputs and expected outputs.

Rule: keys are visited in sorted order with ONE numpy Generator(PCG64(seed)); every tensor is
standard-normal * scale(key):
  * `.weight.weight` of an lr_mul=0.01 linear      -> 100   (reference init is randn / lr_mul, custom_layers.py:11)
  * other `.weight.weight`, const, basis/diagonal  -> 1
  * style-affine bias (`.linear.bias` of a SynthesisLayer, init 1.0 custom_layers.py:96) -> 1 + 0.1 * randn
  * every other bias (init 0 in the reference)     -> 0.1 * randn for lr_mul=1 layers, 10 * randn for lr_mul=0.01
    layers (so the effective bias b*lr_mul is ~0.1 and the bias path is exercised)
  * avg_latent buffers                             -> 0.1 * randn
"""
from __future__ import annotations

import numpy as np
import torch

from .lcgan_ref import lr_mul_of


def _scale_shift(key: str):
    lm = lr_mul_of(key)
    if key.endswith(".weight.weight"):
        return 1.0 / lm, 0.0
    if key.endswith(".linear.bias") and ("modulated_conv" in key or "flow_layer" in key):
        return 0.1, 1.0
    if key.endswith(".bias"):
        return 0.1 / lm, 0.0
    if key.startswith("avg_latent"):
        return 0.1, 0.0
    return 1.0, 0.0


def seeded_state(shapes: dict, seed: int) -> dict:
    rng = np.random.Generator(np.random.PCG64(seed))
    out = {}
    for key in sorted(shapes):
        scale, shift = _scale_shift(key)
        a = rng.standard_normal(shapes[key], dtype=np.float32) * np.float32(scale) + np.float32(shift)
        out[key] = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
    return out


def seeded_tensor(shape, seed: int, kind: str = "normal") -> torch.Tensor:
    rng = np.random.Generator(np.random.PCG64(seed))
    if kind == "normal":
        a = rng.standard_normal(shape, dtype=np.float32)
    elif kind == "uniform_pm1":  # dataset range, custom_dataset.py:81-86
        a = rng.random(shape, dtype=np.float32) * 2 - 1
    else:
        raise ValueError(kind)
    return torch.from_numpy(a)


def grad_stats(t: torch.Tensor, key: str, nproj: int = 8) -> dict:
    """Global statistics of a gradient tensor that stay meaningful when ONE element differs (a leaky-ReLU mask flip at a
    pre-activation of ~1e-7 is rounding noise, yet moves single gradient entries by percents): L1 and L2 norms and
    `nproj` projections on +-1 vectors drawn from PCG64(crc32(key))."""
    import zlib
    f = t.detach().double().reshape(-1).cpu()
    rng = np.random.Generator(np.random.PCG64(zlib.crc32(key.encode())))
    signs = torch.from_numpy(rng.integers(0, 2, size=(nproj, f.numel()), dtype=np.int8).astype(np.float64) * 2 - 1)
    return {"abssum": float(f.abs().sum()), "l2": float(f.norm()), "proj": (signs @ f).numpy()}

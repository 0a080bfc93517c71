"""Frechet Inception Distance statistics -- counterpart of the reference's eval/fid.py:4-27 and of the feature bookkeeping of
worker.py:381-425.  The InceptionV3 feature extractor itself (eval/inception.py: torchvision + weights downloaded from GitHub) is
NOT part of this package: `WORKER.fid_evaluate(feature_extractor)` takes any callable images [B,3,R,R] in [-1,1] -> features [B,F]
(the reference's `InceptionV3([3], normalize_input=False)(x)[0].view(B, -1)` where that is available)."""
from __future__ import annotations

import numpy as np


def feature_statistics(features: np.ndarray):
    """(mean, covariance) as worker.py:410-417 computes them: np.mean(f, 0), np.cov(f, rowvar=False)"""
    f = np.asarray(features, dtype=np.float64)
    return f.mean(axis=0), np.cov(f, rowvar=False)


def calc_fid(sample_mean, sample_cov, real_mean, real_cov, eps: float = 1e-6) -> float:
    """|mu_s - mu_r|^2 + tr(C_s) + tr(C_r) - 2 tr((C_s C_r)^(1/2)); a singular product is regularised with eps on both diagonals and
    an imaginary part of the matrix root beyond 1e-3 on the diagonal is an error, as in eval/fid.py:5-17."""
    from scipy import linalg
    sample_cov, real_cov = np.atleast_2d(sample_cov), np.atleast_2d(real_cov)
    root = linalg.sqrtm(sample_cov @ real_cov)
    if not np.isfinite(root).all():
        print("product of cov matrices is singular")
        shift = eps * np.eye(sample_cov.shape[0])
        root = linalg.sqrtm((sample_cov + shift) @ (real_cov + shift))
    if np.iscomplexobj(root):
        if not np.allclose(np.diagonal(root).imag, 0, atol=1e-3):
            raise ValueError(f"Imaginary component {np.max(np.abs(root.imag))}")
        root = root.real
    d = np.asarray(sample_mean, dtype=np.float64) - np.asarray(real_mean, dtype=np.float64)
    return float(d @ d + np.trace(sample_cov) + np.trace(real_cov) - 2.0 * np.trace(root))

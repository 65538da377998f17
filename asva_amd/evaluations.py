"""Evaluation statistics of the reference (avgen/evaluations/dists.py): the Frechet distance between two feature sets —
the reduction step of FID / FVD (avgen/evaluations/eval.py:236-260).  The feature EXTRACTORS (InceptionV3, I3D, CLIP,
the AVSync classifier: avgen/evaluations/{fid,fvd,clip,avsync}, avsync/**) are separate pretrained networks outside the
denoising path and are not rebuilt here: pass their features to this function.
Host-side float64 linear algebra on a (n_samples, n_features) matrix; not a device kernel."""
from __future__ import annotations

import numpy as np
import torch


def frechet_distance(x1, x2, eps: float = 1e-6) -> float:
    """d^2 = |mu1 - mu2|^2 + Tr(C1 + C2 - 2 sqrt(C1 C2)) for x1, x2: (n, d) feature tensors (dists.py:62-125:
    unbiased covariances, matrix square root of the product, eps-regularised retry when it is singular, imaginary
    round-off discarded)."""
    from scipy import linalg

    a = x1.detach().cpu().double().numpy() if torch.is_tensor(x1) else np.asarray(x1, dtype=np.float64)
    b = x2.detach().cpu().double().numpy() if torch.is_tensor(x2) else np.asarray(x2, dtype=np.float64)
    mu1, mu2 = np.atleast_1d(a.mean(0)), np.atleast_1d(b.mean(0))
    s1, s2 = np.atleast_2d(np.cov(a, rowvar=False)), np.atleast_2d(np.cov(b, rowvar=False))
    assert mu1.shape == mu2.shape and s1.shape == s2.shape, "feature sets have different dimensions"
    diff = mu1 - mu2
    covmean = linalg.sqrtm(s1.dot(s2))
    if not np.isfinite(covmean).all():
        off = np.eye(s1.shape[0]) * eps
        covmean = linalg.sqrtm((s1 + off).dot(s2 + off))
    if np.iscomplexobj(covmean):
        if not np.allclose(np.diagonal(covmean).imag, 0, atol=1e-3):
            raise ValueError(f"Imaginary component {np.max(np.abs(covmean.imag))}")
        covmean = covmean.real
    return float(diff.dot(diff) + np.trace(s1) + np.trace(s2) - 2.0 * np.trace(covmean))

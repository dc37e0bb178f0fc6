"""Synthetic registration pairs.

``ref``: N points uniform in an extent box; random SO(3) rotation R (QR of a
Gaussian matrix, det fixed to +1) and translation t ~ U(-0.5, 0.5)^3;
``src = (ref - t) @ R`` (so ``R @ src_i + t = ref_i``), then row-permuted so
that the prefix sub-sampling of the KNN pyramid (reference
dataloader/data_base.py:166-168) is a random sub-sample, as the reference's
``Resampler`` guarantees for 3DMatch (dataloader/transformation.py:72-80).
Extra feature channels (reflectance for KITTI-shaped input) are U(0,1).
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np

EXTENTS = {
    "3dmatch": ((0.0, 3.0), (0.0, 3.0), (0.0, 3.0)),
    "kitti": ((-50.0, 50.0), (-50.0, 50.0), (-3.0, 3.0)),
}


def random_rotation(rng: np.random.Generator) -> np.ndarray:
    q, r = np.linalg.qr(rng.standard_normal((3, 3)))
    q = q * np.sign(np.diag(r))[None, :]
    if np.linalg.det(q) < 0:
        q[:, 2] = -q[:, 2]
    return q


def _uniform_cloud(rng, n, extent):
    lo = np.array([e[0] for e in extent])
    hi = np.array([e[1] for e in extent])
    return rng.uniform(0.0, 1.0, (n, 3)) * (hi - lo) + lo


def make_pair(n: int, seed: int, feat_len: int = 3, shape: str = "3dmatch",
              partial_overlap: bool = False) -> Dict[str, np.ndarray]:
    """One pair as the reference's collate would hand it over, batch dim = 1
    (reference dataloader/data_base.py:196-209): ``points_src/ref [1,N,feat_len]``
    fp32 and ``transform_gt [1,3,4]``."""
    rng = np.random.Generator(np.random.Philox(key=int(seed) + 0xC10D))
    extent = EXTENTS[shape]
    R = random_rotation(rng)
    t = rng.uniform(-0.5, 0.5, 3)
    if not partial_overlap:
        ref = _uniform_cloud(rng, n, extent)
        src = (ref - t) @ R
    else:
        # C5: two half-space crops keeping ~75 % each (=> ~50 % mutual overlap),
        # as RandomCrop (reference dataloader/transformation.py:134-145), plus
        # N(0,0.01) jitter clipped at 0.05 (:99-107), then resampled to n.
        base = _uniform_cloud(rng, 2 * n, extent)
        c = base.mean(0)

        def crop(p, keep=0.75):
            d = rng.standard_normal(3)
            d /= np.linalg.norm(d)
            proj = (p - c) @ d
            return p[proj > np.quantile(proj, 1.0 - keep)]

        def jitter(p):
            return p + np.clip(rng.standard_normal(p.shape) * 0.01, -0.05, 0.05)

        def resample(p):
            idx = rng.choice(len(p), n, replace=len(p) < n)
            return p[idx]

        ref = resample(jitter(crop(base)))
        src = resample(jitter((crop(base) - t) @ R))
    src = src[rng.permutation(n)]
    ref = ref[rng.permutation(n)]
    extra_s = rng.uniform(0.0, 1.0, (n, max(feat_len - 3, 0)))
    extra_r = rng.uniform(0.0, 1.0, (n, max(feat_len - 3, 0)))
    T = np.concatenate([R, t[:, None]], axis=1)
    return {
        "points_src": np.concatenate([src, extra_s], 1)[None].astype(np.float32),
        "points_ref": np.concatenate([ref, extra_r], 1)[None].astype(np.float32),
        "transform_gt": T[None].astype(np.float32),
    }


def make_batch(n: int, seeds, feat_len: int = 3, shape: str = "3dmatch",
               partial_overlap: bool = False) -> Dict[str, np.ndarray]:
    """Stack several pairs along the batch dimension."""
    pairs = [make_pair(n, s, feat_len, shape, partial_overlap) for s in seeds]
    return {k: np.concatenate([p[k] for p in pairs], 0) for k in pairs[0]}

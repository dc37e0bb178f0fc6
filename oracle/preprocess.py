"""Oracle: pre-processing in front of the KNN pyramid (TEST INFRASTRUCTURE ONLY).

Restates, in numpy, what the reference does before ``nn_search``:

* ``process_point_cloud`` range / height crop            (reference dataloader/data_base.py:299-312)
* open3d ``voxel_down_sample`` (voxel average)           (call sites threeDMatch_loader.py:168-175, kitti_loader.py:335-338)
* ``Resampler._resample`` / ``FixedResampler._resample`` (dataloader/transformation.py:72-93)

open3d is a third-party dependency that is not installed here (version unpinned by the reference), and the
ORDER of its voxel_down_sample output is the iteration order of a hash map; numpy's global RNG drives the
reference's resampling.  Neither is pinned by any reference test, so parity at this boundary is "unpinned".
This module DEFINES the rule the engine owns (deepsir_amd/csrc/preprocess.hip must match it bit for bit):

* voxel index = floor((p - (min_bound - voxel/2)) / voxel) evaluated in float64 (open3d's published rule),
  min_bound over the points that survive the crop;
* voxels are emitted in ascending (ix, iy, iz); a voxel's channels are summed in float64 in ascending input
  order and divided by the count, then rounded to float32;
* random resampling: key_i = splitmix64(seed ^ (cloud << 40) ^ i) >> 1, points taken in ascending (key, i);
  if the cloud has fewer than k points the rest are drawn as splitmix64(~seed ^ (cloud << 40) ^ j) % n.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np

_M64 = (1 << 64) - 1


def splitmix64(x: int) -> int:
    z = (x + 0x9E3779B97F4A7C15) & _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def crop_mask(points: np.ndarray, crop: Optional[Sequence[float]]) -> np.ndarray:
    if crop is None:
        return np.ones(len(points), bool)
    r_min, r_max, z_min, z_max = (np.float32(c) for c in crop)
    p = points[:, :3].astype(np.float32)
    r2 = (p[:, 0] * p[:, 0] + p[:, 1] * p[:, 1]) + p[:, 2] * p[:, 2]
    return (r2 <= r_max * r_max) & (r2 > r_min * r_min) & (p[:, 2] >= z_min) & (p[:, 2] <= z_max)


def voxel_downsample(points: np.ndarray, voxel: float, crop: Optional[Sequence[float]] = None) -> np.ndarray:
    """[n, C] float32 -> [m, C] float32 voxel averages in ascending voxel order."""
    pts = np.ascontiguousarray(points, dtype=np.float32)
    keep = crop_mask(pts, crop)
    idx = np.nonzero(keep)[0]
    if len(idx) == 0:
        return np.zeros((0, pts.shape[1]), np.float32)
    p = pts[idx]
    minb = p[:, :3].min(0).astype(np.float64)
    v = float(np.float32(voxel))
    q = np.floor((p[:, :3].astype(np.float64) - (minb - 0.5 * v)) / v).astype(np.int64)
    key = (q[:, 0] << 36) | (q[:, 1] << 18) | q[:, 2]
    order = np.argsort(key, kind="stable")
    ks = key[order]
    head = np.r_[True, ks[1:] != ks[:-1]]
    seg = np.cumsum(head) - 1
    sums = np.zeros((seg[-1] + 1, pts.shape[1]), np.float64)
    np.add.at(sums, seg, p[order].astype(np.float64))      # sequential, ascending input order inside a voxel
    cnt = np.bincount(seg).astype(np.float64)
    return (sums / cnt[:, None]).astype(np.float32)


def resample(points: np.ndarray, k: int, cloud: int, seed: int, mode: str = "random") -> np.ndarray:
    n = len(points)
    if n == 0:
        return np.zeros((k, points.shape[1]), np.float32)
    if mode == "fixed":
        return points[np.arange(k) % n]
    keys = np.array([splitmix64((seed ^ (cloud << 40) ^ i) & _M64) >> 1 for i in range(n)], dtype=np.uint64)
    perm = np.argsort(keys, kind="stable")
    take = list(perm[: min(k, n)])
    for j in range(n, k):
        take.append(splitmix64(((~seed) & _M64) ^ (cloud << 40) ^ j) % n)
    return points[np.asarray(take, dtype=np.int64)]


def preprocess(clouds: List[np.ndarray], voxel: float, k: int, seed: int, crop=None, mode: str = "random") -> Tuple[np.ndarray, List[int]]:
    """Ragged list of [n_i, C] clouds -> ([len(clouds), k, C] float32, voxel counts)."""
    down = [voxel_downsample(c, voxel, crop) for c in clouds]
    out = np.stack([resample(d, k, i, seed, mode) for i, d in enumerate(down)])
    return out.astype(np.float32), [len(d) for d in down]

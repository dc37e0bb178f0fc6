"""Oracle: exact brute-force KNN pyramid (TEST INFRASTRUCTURE ONLY).

Follows reference dataloader/data_base.py:153-183 (``DataBase.nn_search``):
per level l, ``knn(pc, pc, K)`` (support = 1st arg, queries = 2nd arg),
``pool = neigh[:n_l // ratio]``, ``sub = pc[:n_l // ratio]``,
``up = knn(sub, pc, 1)``, then ``pc = sub``; the four lists are concatenated
over levels.

The reference calls the third-party ``torch_points_kernels.knn`` (nanoflann
KD-tree; version unpinned, not installed here), whose tie order nothing pins.
This oracle DEFINES the rule the engine owns:

* squared distance in fp32, ``d = (dx*dx + dy*dy) + dz*dz`` with every product
  and sum rounded to fp32 (no FMA contraction), ``dx = s.x - q.x``;
* neighbours sorted by (d ascending, then support index ascending).

Integer/index work: the HIP kernel must match this bit-exactly.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np


def sqdist_f32(q: np.ndarray, s: np.ndarray) -> np.ndarray:
    """[Q,3] x [S,3] -> [Q,S] fp32, rounding order as documented above."""
    q = q.astype(np.float32, copy=False)
    s = s.astype(np.float32, copy=False)
    dx = s[None, :, 0] - q[:, None, 0]
    dy = s[None, :, 1] - q[:, None, 1]
    dz = s[None, :, 2] - q[:, None, 2]
    return (dx * dx + dy * dy) + dz * dz


def knn(support: np.ndarray, query: np.ndarray, k: int, chunk: int = 1024) -> np.ndarray:
    """k nearest support points of every query point -> int32 [Q,k]."""
    S = support.shape[0]
    if k > S:
        raise ValueError(f"k={k} > support size {S}")
    out = np.empty((query.shape[0], k), dtype=np.int32)
    for a in range(0, query.shape[0], chunk):
        d = sqdist_f32(query[a:a + chunk], support)
        if k == 1:
            out[a:a + chunk, 0] = np.argmin(d, axis=1)  # first occurrence == lowest index
            continue
        kth = np.partition(d, k - 1, axis=1)[:, k - 1]
        mask = d <= kth[:, None]
        cnt = mask.sum(1)
        easy = cnt == k
        if easy.any():
            rows = np.nonzero(easy)[0]
            cols = np.nonzero(mask[rows])[1].reshape(len(rows), k)  # ascending index
            dd = np.take_along_axis(d[rows], cols, 1)
            order = np.argsort(dd, axis=1, kind="stable")
            out[a + rows] = np.take_along_axis(cols, order, 1)
        for r in np.nonzero(~easy)[0]:  # ties on the k-th distance
            cols = np.nonzero(mask[r])[0]
            order = np.argsort(d[r, cols], kind="stable")[:k]
            out[a + r] = cols[order]
    return out


def knn_pyramid(xyz: np.ndarray, k: int, ratios: Sequence[int]) -> Dict[str, np.ndarray]:
    """One cloud [N,3] -> concatenated pyramids (no batch dim):
    ``xyz [sum n_l,3] f32``, ``neigh_idx [sum n_l,k]``, ``sub_idx [sum n_{l+1},k]``,
    ``interp_idx [sum n_l,1]`` (int32)."""
    pc = np.ascontiguousarray(xyz[:, :3], dtype=np.float32)
    pts: List[np.ndarray] = []
    neigh: List[np.ndarray] = []
    pool: List[np.ndarray] = []
    up: List[np.ndarray] = []
    for r in ratios:
        nb = knn(pc, pc, k)
        num = pc.shape[0] // int(r)
        sub = pc[:num]
        pts.append(pc)
        neigh.append(nb)
        pool.append(nb[:num])
        up.append(knn(sub, pc, 1))
        pc = sub
    return {
        "xyz": np.concatenate(pts, 0),
        "neigh_idx": np.concatenate(neigh, 0),
        "sub_idx": np.concatenate(pool, 0),
        "interp_idx": np.concatenate(up, 0),
    }


def add_pyramids(data: Dict[str, np.ndarray], k: int = 16, ratios: Sequence[int] = (4, 4, 4, 4),
                 index_dtype=np.int64) -> Dict[str, np.ndarray]:
    """Batch version of ``nn_search``: adds the 8 index/xyz entries per the
    reference's key names (data_base.py:178-181)."""
    out = dict(data)
    for key in ("points_src", "points_ref"):
        per = [knn_pyramid(c, k, ratios) for c in data[key]]
        out[key + "_xyz"] = np.stack([p["xyz"] for p in per]).astype(np.float32)
        for name in ("neigh_idx", "sub_idx", "interp_idx"):
            out[f"{key}_{name}"] = np.stack([p[name] for p in per]).astype(index_dtype)
    return out

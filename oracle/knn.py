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


def knn_fast(support: np.ndarray, query: np.ndarray, k: int, chunk: int = 2048) -> np.ndarray:
    """``knn`` for LARGE levels (65536 points: C5), the same rule on torch-CPU tensors so that the distance matrix of a query chunk
    is formed and ranked on every host core.  Every arithmetic step is one elementwise torch op on fp32 (IEEE, no contraction: the
    same roundings as ``sqdist_f32``); the k smallest VALUES per row come from ``topk`` (values are unique whatever order it breaks
    ties in), the index set and its order are then rebuilt exactly as ``knn`` does: all columns with d <= k-th value, ascending
    index, stable sort by distance; rows with a tie ON the k-th distance go through the same per-row branch.  Equality with
    ``knn`` on adversarial clouds: tests/test_knn_oracle.py."""
    import torch
    S = support.shape[0]
    if k > S:
        raise ValueError(f"k={k} > support size {S}")
    s = torch.from_numpy(np.ascontiguousarray(support[:, :3], dtype=np.float32))
    q_all = torch.from_numpy(np.ascontiguousarray(query[:, :3], dtype=np.float32))
    out = np.empty((query.shape[0], k), dtype=np.int32)
    for a in range(0, query.shape[0], chunk):
        q = q_all[a:a + chunk]
        d = s[None, :, 0] - q[:, None, 0]
        d.mul_(d)                                   # dx * dx
        t = s[None, :, 1] - q[:, None, 1]
        t.mul_(t)
        d.add_(t)                                   # (dx * dx + dy * dy)
        t = s[None, :, 2] - q[:, None, 2]
        t.mul_(t)
        d.add_(t)                                   # ... + dz * dz
        del t
        if k == 1:
            # first occurrence of the row minimum == lowest index (torch.argmin does not promise which of equal minima it returns)
            m = d.min(1).values
            first = torch.where(d == m[:, None], torch.arange(S)[None, :], torch.full((1, 1), S)).min(1).values
            out[a:a + chunk, 0] = first.numpy().astype(np.int32)
            continue
        top = torch.topk(d, k, dim=1, largest=False, sorted=True)
        kth = top.values[:, k - 1]
        cnt = (d <= kth[:, None]).sum(1)
        easy = cnt == k                             # no tie ON the k-th distance: topk's index SET is the answer's, whatever its order
        cols = torch.sort(top.indices, dim=1).values                           # ascending index ...
        dd = torch.gather(d, 1, cols)
        order = torch.argsort(dd, dim=1, stable=True)                          # ... then stable by distance = (d, index) order
        res = torch.gather(cols, 1, order).numpy().astype(np.int32)
        out[a:a + chunk] = res
        for r in torch.nonzero(~easy)[:, 0].tolist():                          # ties on the k-th distance: knn's per-row branch
            dn = d[r].numpy()
            cand = np.nonzero(dn <= float(kth[r]))[0]
            out[a + r] = cand[np.argsort(dn[cand], kind="stable")[:k]]
    return out


def knn_pyramid(xyz: np.ndarray, k: int, ratios: Sequence[int], fast_min: int = 32768) -> Dict[str, np.ndarray]:
    """One cloud [N,3] -> concatenated pyramids (no batch dim):
    ``xyz [sum n_l,3] f32``, ``neigh_idx [sum n_l,k]``, ``sub_idx [sum n_{l+1},k]``,
    ``interp_idx [sum n_l,1]`` (int32)."""
    pc = np.ascontiguousarray(xyz[:, :3], dtype=np.float32)
    pts: List[np.ndarray] = []
    neigh: List[np.ndarray] = []
    pool: List[np.ndarray] = []
    up: List[np.ndarray] = []
    for r in ratios:
        search = knn_fast if pc.shape[0] >= fast_min else knn      # same rule, same results (tests/test_knn_oracle.py)
        nb = search(pc, pc, k)
        num = pc.shape[0] // int(r)
        sub = pc[:num]
        pts.append(pc)
        neigh.append(nb)
        pool.append(nb[:num])
        up.append(search(sub, pc, 1))
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

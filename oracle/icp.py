"""Oracle: point-to-point ICP as open3d's ``registration_icp`` runs it (TEST INFRASTRUCTURE ONLY).

Restates the loop of open3d ``RegistrationICP`` (pipelines/registration/Registration.cpp, open3d 0.9-0.13, the call at
reference test.py:253-254): correspondences = nearest target point within ``max_correspondence_distance`` (KD-tree
radius-limited 1-NN == exact brute force), fitness = |corr| / |src|, inlier_rmse = sqrt(sum d^2 / |corr|),
update = Kabsch / Umeyama without scale over the correspondences, stop when both |d fitness| < relative_fitness and
|d rmse| < relative_rmse, at most ``max_iteration`` updates.  open3d is not installable here and the branch is
switched off in the reference (test.py:216): **parity unpinned**; this file is the rule the HIP path is held to.
float64 throughout except the distance that decides a correspondence, evaluated like the engine in fp32
((dx*dx + dy*dy) + dz*dz, ties to the lower index) so both sides pick the same neighbours."""
from __future__ import annotations

import numpy as np


def _correspondences(cur32: np.ndarray, tgt32: np.ndarray, r2: np.float32, chunk: int = 1024):
    idx = np.empty(len(cur32), np.int64)
    d2 = np.empty(len(cur32), np.float32)
    for a in range(0, len(cur32), chunk):
        q = cur32[a:a + chunk, None, :]
        dx, dy, dz = (tgt32[None, :, 0] - q[..., 0]), (tgt32[None, :, 1] - q[..., 1]), (tgt32[None, :, 2] - q[..., 2])
        d = (dx * dx + dy * dy) + dz * dz                      # fp32, same association as the kernel
        i = d.argmin(axis=1)                                   # first minimum = lower index
        idx[a:a + chunk] = i
        d2[a:a + chunk] = d[np.arange(len(i)), i]
    ok = d2 <= r2
    return np.where(ok, idx, -1), np.where(ok, d2, np.float32(0)), ok


def _kabsch(src: np.ndarray, tgt: np.ndarray) -> np.ndarray:
    if len(src) == 0:
        return np.hstack([np.eye(3), np.zeros((3, 1))])
    cs, ct = src.mean(0), tgt.mean(0)
    H = (src - cs).T @ (tgt - ct)
    U, _, Vt = np.linalg.svd(H)
    V = Vt.T
    d = 1.0 if np.linalg.det(V @ U.T) > 0 else -1.0
    R = V @ np.diag([1.0, 1.0, d]) @ U.T
    return np.hstack([R, (ct - R @ cs)[:, None]])


def icp(src: np.ndarray, tgt: np.ndarray, T_init: np.ndarray, max_corr_dist: float, max_iter: int = 30,
        rel_fitness: float = 1e-6, rel_rmse: float = 1e-6):
    """src [J,>=3], tgt [K,>=3] fp32, T_init [3,4] -> (T [3,4] float64, fitness, inlier_rmse, converged, iterations)."""
    src32, tgt32 = np.ascontiguousarray(src[:, :3], np.float32), np.ascontiguousarray(tgt[:, :3], np.float32)
    r2 = np.float32(max_corr_dist) * np.float32(max_corr_dist)
    T = np.asarray(T_init, np.float64).copy()
    cur = (src32.astype(np.float32) @ T[:, :3].astype(np.float32).T + T[:, 3].astype(np.float32)).astype(np.float32)

    def evaluate(c):
        idx, d2, ok = _correspondences(c, tgt32, r2)
        n = int(ok.sum())
        return idx, ok, n / len(c), (float(np.sqrt(d2[ok].astype(np.float64).sum() / n)) if n else 0.0)

    idx, ok, fitness, rmse = evaluate(cur)
    converged, iters = False, 0
    for _ in range(max_iter):
        upd = _kabsch(cur[ok].astype(np.float64), tgt32[idx[ok]].astype(np.float64))
        T = np.hstack([upd[:, :3] @ T[:, :3], (upd[:, :3] @ T[:, 3] + upd[:, 3])[:, None]])
        cur = (cur.astype(np.float64) @ upd[:, :3].T + upd[:, 3]).astype(np.float32)
        idx, ok, f2, r2_ = evaluate(cur)
        iters += 1
        done = abs(fitness - f2) < rel_fitness and abs(rmse - r2_) < rel_rmse
        fitness, rmse = f2, r2_
        if done:
            converged = True
            break
    return T, fitness, rmse, converged, iters

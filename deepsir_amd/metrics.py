"""Registration criteria of the evaluation harness (reference
common/metrics_util.py:13-24 ``rte_rre``; thresholds test.py:49-54)."""
import numpy as np

THRESHOLDS = {"3DMatch": (0.3, 15.0), "KITTI": (0.6, 5.0)}   # (RTE m, RRE deg)


def rte_rre(T_pred, T_gt, rte_thresh, rre_thresh, eps=1e-16):
    """-> array [success, rte (m), rre (deg)]; ``None`` prediction counts as a failure."""
    if T_pred is None:
        return np.array([0, np.inf, np.inf])
    T_pred, T_gt = np.asarray(T_pred), np.asarray(T_gt)
    rte = np.linalg.norm(T_pred[:3, 3] - T_gt[:3, 3])
    c = (np.trace(T_pred[:3, :3].T @ T_gt[:3, :3]) - 1) / 2
    rre = np.arccos(np.clip(c, -1 + eps, 1 - eps)) * 180 / np.pi
    return np.array([rte < rte_thresh and rre < rre_thresh, rte, rre])

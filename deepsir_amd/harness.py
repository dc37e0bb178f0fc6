"""Evaluation harness: the caller contract of the reference's
``test.py::inference_align`` (test.py:358-457) around the accelerated model.

Same iteration order, ``opt = (num_reg_iter, True)``, the last transform
duplicated as the "pose_optimized" entry (pose_optimization is the identity in
the reference, test.py:209-216), and one stats row ``[succ, rte, rre, time, seq]``
per pair (test.py:432-441).  Differences, both deliberate:

* pairs are taken ``batch`` at a time (the reference hard-codes batch 1,
  test.py:56) — the per-pair time is the batch time divided by its size;
* pairs shard across ranks (``deepsir_amd.dist``), results are gathered once.
"""
from __future__ import annotations

import time
from typing import Callable, Dict, Iterable, List, Optional, Sequence

import numpy as np
import torch

from .dist import gather_results, shard_range, shard_sizes
from .metrics import THRESHOLDS, rte_rre


@torch.no_grad()
def inference_align(pairs: Sequence[Dict[str, np.ndarray]], model, num_reg_iter: int = 5, dataset_type: str = "3DMatch",
                    batch: int = 1, device: Optional[torch.device] = None, dist=None):
    """pairs: sequence of dicts with ``points_src/points_ref [1,N,C]``, ``transform_gt [1,3,4]`` and optionally the
    pyramid tensors and ``others`` (as the reference's collate, data_base.py:196-219).
    Returns (pred_transforms_all [n_pairs, n_iter+1, 3, 4], stats [n_pairs, 5]) gathered over ranks."""
    device = device or torch.device("cuda", torch.cuda.current_device())
    rte_t, rre_t = THRESHOLDS[dataset_type]
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    mine = list(shard_range(len(pairs), rank, world))
    preds: List[np.ndarray] = []
    stats = np.zeros((len(mine), 5))
    opt = (num_reg_iter, True)
    for b0 in range(0, len(mine), batch):
        ids = mine[b0:b0 + batch]
        keys = [k for k in pairs[ids[0]] if k != "others" and isinstance(pairs[ids[0]][k], np.ndarray)]
        data = {k: torch.from_numpy(np.concatenate([pairs[i][k] for i in ids], 0)).to(device) for k in keys}
        torch.cuda.synchronize(device)
        t0 = time.time()
        transforms, endpoints = model(data, opt)
        torch.cuda.synchronize(device)
        dt = (time.time() - t0) / len(ids)
        transforms.append(transforms[-1].detach())            # pose_optimization == identity (test.py:406-408)
        T = torch.stack(transforms, dim=1).cpu().numpy()      # [B, n_iter+1, 3, 4]
        preds.append(T)
        gt = data["transform_gt"].cpu().numpy()
        for j, i in enumerate(ids):
            row = b0 + j
            stats[row, :3] = rte_rre(T[j, -1], gt[j], rte_t, rre_t)
            stats[row, 3] = dt
            others = pairs[i].get("others")
            stats[row, 4] = others[0]["seq"] if others else -1
    pred = np.concatenate(preds, 0) if preds else np.zeros((0, num_reg_iter + 1, 3, 4), np.float32)
    if world > 1:
        sizes = shard_sizes(len(pairs), world)
        pred = gather_results(torch.from_numpy(pred).to(device), dist, sizes).cpu().numpy()
        stats = gather_results(torch.from_numpy(stats).to(device), dist, sizes).cpu().numpy()
    return pred, stats


def summarize(stats: np.ndarray) -> Dict[str, float]:
    """Aggregate as the reference's print_stats: recall, mean RTE/RRE over successes, mean time."""
    ok = stats[:, 0] > 0
    return {
        "recall": float(ok.mean()) if len(stats) else 0.0,
        "rte_mean": float(stats[ok, 1].mean()) if ok.any() else float("nan"),
        "rre_mean": float(stats[ok, 2].mean()) if ok.any() else float("nan"),
        "time_mean": float(stats[:, 3].mean()) if len(stats) else 0.0,
        "pairs_per_sec": float(1.0 / stats[:, 3].mean()) if len(stats) and stats[:, 3].mean() > 0 else 0.0,
    }


@torch.no_grad()
def evaluate_align(pred_transforms: np.ndarray, pairs: Sequence[Dict[str, np.ndarray]], engine, dataset_type: str = "3DMatch",
                   batch: int = 64, device: Optional[torch.device] = None):
    """Counterpart of the reference's ``test.py::evaluate_align`` (test.py:308-355): per registration
    iteration, the metrics of ``compute_metrics`` (common/metrics_util.py:27-85) on the first 1024 points of
    each cloud (test.py:331-332), computed on device by ``dsir_eval_metrics``.

    pred_transforms [n_pairs, n_iter(+1), 3, 4]; pairs as for ``inference_align``; ``engine`` a
    ``deepsir_amd.engine.Engine``.  Returns (metrics_for_iter: list of dicts of arrays, summary of the last iteration)."""
    device = device or torch.device("cuda", torch.cuda.current_device())
    rte_t, rre_t = THRESHOLDS[dataset_type]
    pred = torch.from_numpy(np.ascontiguousarray(pred_transforms, dtype=np.float32)).to(device)
    if pred.dim() == 3:
        pred = pred[:, None]
    n_it = pred.shape[1]
    acc = [dict((k, []) for k in engine.METRIC_NAMES) for _ in range(n_it)]
    for b0 in range(0, len(pairs), batch):
        ids = range(b0, min(len(pairs), b0 + batch))
        src = torch.from_numpy(np.concatenate([pairs[i]["points_src"][:, :1024] for i in ids], 0)).to(device)
        ref = torch.from_numpy(np.concatenate([pairs[i]["points_ref"][:, :1024] for i in ids], 0)).to(device)
        gt = torch.from_numpy(np.concatenate([pairs[i]["transform_gt"] for i in ids], 0)).float().to(device)
        for it in range(n_it):
            m = engine.eval_metrics(pred[b0:b0 + len(ids), it], gt, src, ref, rte_t, rre_t)
            for k, v in m.items():
                acc[it][k].append(v.cpu().numpy())
    metrics_for_iter = [{k: np.concatenate(v) for k, v in a.items()} for a in acc]
    return metrics_for_iter, summarize_metrics(metrics_for_iter[-1])


def summarize_metrics(metrics: Dict[str, np.ndarray]) -> Dict[str, float]:
    """Mean over instances with the reference's naming (common/metrics_util.py:88-101):
    ``*mse`` -> ``*rmse``; ``err_*`` -> ``_mean`` and ``_rmse``; everything else -> mean."""
    out = {}
    for k, v in metrics.items():
        if k.endswith("mse"):
            out[k[:-3] + "rmse"] = float(np.sqrt(np.mean(v)))
        elif k.startswith("err"):
            out[k + "_mean"] = float(np.mean(v))
            out[k + "_rmse"] = float(np.sqrt(np.mean(v ** 2)))
        else:
            out[k] = float(np.mean(v))
    return out

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
                    batch: int = 1, device: Optional[torch.device] = None, dist=None, pose_opt: Optional[str] = None,
                    voxel_size: float = 0.3, in_flight: int = 1):
    """pairs: sequence of dicts with ``points_src/points_ref [1,N,C]``, ``transform_gt [1,3,4]`` and optionally the
    pyramid tensors and ``others`` (as the reference's collate, data_base.py:196-219).
    ``in_flight`` > 1: the reference's one-pair-per-call loop fed AHEAD - the pairs go one by one to a
    ``deepsir_amd.serve.PairServer`` that keeps that many requests outstanding (same results bit for bit; the per-pair
    time is then the shard's wall time divided by its size).
    Returns (pred_transforms_all [n_pairs, n_iter+1, 3, 4], stats [n_pairs, 5]) gathered over ranks."""
    device = device or torch.device("cuda", torch.cuda.current_device())
    rte_t, rre_t = THRESHOLDS[dataset_type]
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    mine = list(shard_range(len(pairs), rank, world))
    preds: List[np.ndarray] = []
    stats = np.zeros((len(mine), 5))
    opt = (num_reg_iter, True)
    if in_flight > 1:
        if pose_opt is not None:
            raise ValueError("in_flight > 1 serves the registration alone (pose_opt must be None)")
        if any(k.endswith("_neigh_idx") for i in mine for k in pairs[i]):
            # the served path builds the KNN pyramid on the device; a loader's own pyramid (whose tie rule may differ) is honoured by
            # the one-pair-per-call path only - say so instead of silently diverging from in_flight=1
            raise ValueError("in_flight > 1 recomputes the KNN pyramid on the device: pairs that carry their own pyramid tensors "
                             "must be evaluated with in_flight=1 (or have the pyramid keys removed)")
        keys = ("points_src", "points_ref", "transform_gt")
        datas = [_stack([{k: v for k, v in pairs[i].items() if k in keys}], [0], device) for i in mine]
        n_max = max([max(d["points_src"].shape[1], d["points_ref"].shape[1]) for d in datas], default=1024)
        srv = model.serve(max_points=n_max, max_in_flight=in_flight, n_iter=num_reg_iter, want_aux=False)
        torch.cuda.synchronize(device)
        t0 = time.time()
        res = srv.run_closed_loop(((d["points_src"].float(), d["points_ref"].float()) for d in datas), in_flight)
        torch.cuda.synchronize(device)
        dt = (time.time() - t0) / max(len(mine), 1)
        srv.close()
        for row, (i, d, r) in enumerate(zip(mine, datas, res)):
            T = r["transforms"].cpu().numpy()
            preds.append(np.concatenate([T, T[-1:]], 0)[None])            # pose_optimization == identity (test.py:215-216)
            stats[row, :3] = rte_rre(T[-1], d["transform_gt"].cpu().numpy()[0], rte_t, rre_t)
            stats[row, 3] = dt
            others = pairs[i].get("others")
            stats[row, 4] = _seq_id(others[0]["seq"]) if others else -1
        mine_batches = range(0)
    else:
        mine_batches = range(0, len(mine), batch)
    for b0 in mine_batches:
        ids = mine[b0:b0 + batch]
        data = _stack(pairs, ids, device)
        torch.cuda.synchronize(device)
        t0 = time.time()
        transforms, endpoints = model(data, opt)
        torch.cuda.synchronize(device)
        dt = (time.time() - t0) / len(ids)
        if pose_opt == "icp":
            # pose_optimization with use_icp (test.py:241-258; off in the reference): point-to-point ICP on the raw clouds,
            # correspondence radius 2 x voxel size (test.py:219), open3d's default criteria
            T_opt, _ = _aux_engine(model, data).icp_refine(data["points_src"].float(), data["points_ref"].float(),
                                                transforms[-1].contiguous(), 2.0 * voxel_size)
            transforms.append(T_opt)
        elif pose_opt == "tune":
            # pose_optimization with use_tune (test.py:218-239; off in the reference): Adam fine-tune of the 6-D-rotation
            # pose on the last iteration's correspondences, weights = sigmoid of its inlier logits, distances in units
            # of 2 x voxel size (test.py:219, :238)
            T_opt, _ = _aux_engine(model, data).pose_finetune(endpoints["pt_src"].float(), endpoints["pt_ref_new"].float(),
                                                   transforms[-1].contiguous(), weights=endpoints["perm_matrices"][-1],
                                                   weights_are_logits=True, quantization_size=2.0 * voxel_size)
            transforms.append(T_opt)
        elif pose_opt is None:
            transforms.append(transforms[-1].detach())        # pose_optimization == identity (test.py:215-216, :406-408)
        else:
            raise ValueError("pose_opt must be None, 'icp' or 'tune'")
        T = torch.stack(transforms, dim=1).cpu().numpy()      # [B, n_iter+1, 3, 4]
        preds.append(T)
        gt = data["transform_gt"].cpu().numpy()
        for j, i in enumerate(ids):
            row = b0 + j
            stats[row, :3] = rte_rre(T[j, -1], gt[j], rte_t, rre_t)
            stats[row, 3] = dt
            others = pairs[i].get("others")
            stats[row, 4] = _seq_id(others[0]["seq"]) if others else -1
    pred = np.concatenate(preds, 0) if preds else np.zeros((0, num_reg_iter + 1, 3, 4), np.float32)
    if world > 1:
        sizes = shard_sizes(len(pairs), world)
        pred = gather_results(torch.from_numpy(pred).to(device), dist, sizes).cpu().numpy()
        stats = gather_results(torch.from_numpy(stats).to(device), dist, sizes).cpu().numpy()
    return pred, stats


def _aux_engine(model, data):
    """The plain engine behind the drop-in model for the operators next to the path (ICP, fine-tune): ``forward`` may have run on
    the model's server or pool instead."""
    n = max(int(data["points_src"].shape[1]), int(data["points_ref"].shape[1]))
    return model._ensure_engine(n, int(data["points_src"].shape[0]))


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
        def head(k):   # numpy (reference collate) or device tensors (deepsir_amd.data)
            return torch.cat([torch.as_tensor(pairs[i][k][:, :1024]).to(device) for i in ids], 0).float().contiguous()
        src, ref = head("points_src"), head("points_ref")
        gt = torch.cat([torch.as_tensor(pairs[i]["transform_gt"]).to(device) for i in ids], 0).float()
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


# ----------------------------------------------------------------------------------------------------------------------
# 'feat' / 'label' pipelines (reference test.py:460-567): the caller side of Network.forward = forward_pair
class SemanticMetric:
    """Running semantic-segmentation statistics of the reference's ``SemanticLoss`` (network/loss.py:854-992):
    label 0 ('unlabeled') is ignored, label l > 0 is class l-1; per-class IoU = TP / (GT + P - TP) (0 for an absent
    class), mean IoU over all ``num_classes``, overall accuracy, and the inverse-frequency weighted cross entropy
    (class weights 1 / (freq + 0.02) from the SemanticKITTI point counts, loss.py:905-911)."""

    NUM_PER_CLASS = np.array([55437630, 320797, 541736, 2578735, 3274484, 552662, 184064, 78858, 240942562, 17294618,
                              170599734, 6369672, 230413074, 101130274, 476491114, 9833174, 129609852, 4506626, 1168181],
                             dtype=np.float64)

    def __init__(self, num_classes: int = 19):
        self.num_classes = num_classes
        w = self.NUM_PER_CLASS / self.NUM_PER_CLASS.sum()
        self.class_weights = (1.0 / (w + 0.02)).astype(np.float32)
        self.reset()

    def reset(self):
        self.gt = np.zeros(self.num_classes, np.int64)
        self.pos = np.zeros(self.num_classes, np.int64)
        self.tp = np.zeros(self.num_classes, np.int64)
        self.correct = 0
        self.seen = 0

    def add(self, logits: torch.Tensor, labels: torch.Tensor):
        """logits [B, num_classes, N], labels [B, N] (0 = ignored).  Returns (weighted CE loss, accuracy) of this
        batch (loss.py:929-958) and accumulates the confusion counts (:960-971)."""
        lg = logits.transpose(1, 2).reshape(-1, self.num_classes).float()
        lb = labels.reshape(-1).long()
        valid = lb != 0
        lg, lb = lg[valid], lb[valid] - 1
        if lb.numel() == 0:
            return float("nan"), float("nan")
        w = torch.from_numpy(self.class_weights).to(lg.device)
        loss = torch.nn.functional.cross_entropy(lg, lb, weight=w, reduction="mean")
        pred = lg.max(dim=1)[1]
        acc = (pred == lb).sum().float() / float(lb.shape[0])
        conf = torch.bincount(lb * self.num_classes + pred, minlength=self.num_classes ** 2)
        conf = conf.reshape(self.num_classes, self.num_classes).cpu().numpy()
        self.gt += conf.sum(1)
        self.pos += conf.sum(0)
        self.tp += np.diagonal(conf)
        self.correct += int((pred == lb).sum())
        self.seen += int(lb.numel())
        return float(loss), float(acc)

    def result(self):
        """(mean IoU, per-class IoU list, mean accuracy); resets like the reference (loss.py:973-987)."""
        denom = (self.gt + self.pos - self.tp).astype(np.float64)
        iou = np.where(denom != 0, self.tp / np.where(denom != 0, denom, 1.0), 0.0)
        out = float(iou.sum() / self.num_classes), [float(x) for x in iou], self.correct / float(max(self.seen, 1))
        self.reset()
        return out


def _stack(pairs, ids, device):
    """Batch the array entries of the pair dicts `ids` on the device: numpy (the reference's collate output) or torch
    tensors (deepsir_amd.data: already resident), each with a leading batch dimension of 1."""
    out = {}
    for k, v in pairs[ids[0]].items():
        if k == "others":
            continue
        if isinstance(v, np.ndarray):
            out[k] = torch.from_numpy(np.concatenate([pairs[i][k] for i in ids], 0)).to(device)
        elif isinstance(v, torch.Tensor):
            out[k] = torch.cat([pairs[i][k].to(device) for i in ids], 0)
    return out


_SEQ_IDS: Dict[str, int] = {}


def _seq_id(seq) -> float:
    """The sequence column of the stats table (test.py:439) is numeric; scene NAMES (3DMatch) get a running index."""
    try:
        return float(seq)
    except (TypeError, ValueError):
        return float(_SEQ_IDS.setdefault(str(seq), len(_SEQ_IDS)))


def _pair_batches(pairs, batch, device):
    for b0 in range(0, len(pairs), batch):
        ids = list(range(b0, min(b0 + batch, len(pairs))))
        yield ids, _stack(pairs, ids, device)


@torch.no_grad()
def inference_feat(pairs: Sequence[Dict[str, np.ndarray]], model, batch: int = 1, device: Optional[torch.device] = None):
    """test.py::inference_feat (:460-505): key points + saliency per cloud.  Returns a list (one entry per pair) of
    {'pt_src' [M,3], 'score_src' [M], 'feat_src' [M,64], ... same for ref} as numpy, and the total model time."""
    device = device or torch.device("cuda", torch.cuda.current_device())
    out, total = [], 0.0
    for ids, data in _pair_batches(pairs, batch, device):
        torch.cuda.synchronize(device)
        t0 = time.time()
        _, ep = model(data)
        torch.cuda.synchronize(device)
        total += time.time() - t0
        for j in range(len(ids)):
            rec = {}
            for s in ("src", "ref"):
                rec[f"pt_{s}"] = ep[f"pt_{s}"][j].t().cpu().numpy()
                rec[f"feat_{s}"] = ep[f"feat_{s}"][j].t().cpu().numpy()
                rec[f"score_{s}"] = ep[f"score_{s}"][j].cpu().numpy()
            out.append(rec)
    return out, total


@torch.no_grad()
def inference_label(pairs: Sequence[Dict[str, np.ndarray]], model, batch: int = 1, device: Optional[torch.device] = None):
    """test.py::inference_label (:508-567): semantic head over every pair; ``labels_src`` / ``labels_ref`` [1,N]
    (0 = unlabeled) give accuracy / IoU.  Returns (per-pair predicted labels (1-based like the reference's export),
    {'mean_iou', 'iou', 'mean_acc', 'loss'}, total model time)."""
    device = device or torch.device("cuda", torch.cuda.current_device())
    metric = SemanticMetric(model.cfg.num_classes)
    preds, losses, total = [], [], 0.0
    for ids, data in _pair_batches(pairs, batch, device):
        torch.cuda.synchronize(device)
        t0 = time.time()
        _, ep = model(data)
        torch.cuda.synchronize(device)
        total += time.time() - t0
        if "labels_src" in data:
            l_s, _ = metric.add(ep["logits_src"], data["labels_src"])
            l_r, _ = metric.add(ep["logits_ref"], data["labels_ref"])
            losses.append(l_s + l_r)
        for j in range(len(ids)):
            preds.append({s: (torch.argmax(ep[f"logits_{s}"][j], dim=0) + 1).cpu().numpy() for s in ("src", "ref")})
    mean_iou, iou, mean_acc = metric.result()
    return preds, {"mean_iou": mean_iou, "iou": iou, "mean_acc": mean_acc,
                   "loss": float(np.nanmean(losses)) if losses else float("nan")}, total

"""Oracle: the training loss of the `align` pipeline and its gradient down to the inlier logits
(TEST INFRASTRUCTURE ONLY).

In ``forward_align_4`` the descriptor matching runs under ``torch.no_grad()`` (model.py:556) and the source cloud is moved
by ``R_t.detach()`` (model.py:590), so ``ScanAlignmentLoss`` (network/loss.py:705-851, called at train.py:401) reaches the
network parameters ONLY through the inlier logits: ``weights = logit.sigmoid()`` -> ``compute_rigid_transform_2``
(model.py:22-66, an SVD) -> ``se3_torch.concatenate`` (cumulative transforms, model.py:595) -> L1 point-distance loss,
plus the BCE-with-logits "correspondence confidence" term on the same logits.  This file restates that slice with torch
autograd: forward = the loss dictionary, backward = d total / d logits, the tensor the inlier RandLA's backward pass
consumes.  Pinned by tests/golden/align_loss_cases.npz, generated from the imported reference by
oracle/gen_golden_align_loss.py.
"""
from __future__ import annotations

from typing import Dict, List, Sequence

import numpy as np
import torch
import torch.nn.functional as F

from oracle.network import OracleNet


def find_correct_correspondence(pos_pairs: np.ndarray, pred_pairs: np.ndarray, hash_seed: int) -> np.ndarray:
    """[N',2], [N,2] int -> bool [N]: is the predicted (src, ref) pair among the ground-truth matches
    (loss.py:723-749 with ``_hash``, :280-294)."""
    key = lambda a: a[:, 0].astype(np.int64) + a[:, 1].astype(np.int64) * int(hash_seed)
    return np.isin(key(np.asarray(pred_pairs)), key(np.asarray(pos_pairs)))


def replay(pt_src: torch.Tensor, pt_ref: torch.Tensor, idx: Sequence[torch.Tensor], logits: Sequence[torch.Tensor]):
    """The differentiable tail of forward_align_4 (model.py:571-595) from the logits on:
    pt_src [B,J,3], pt_ref [B,K,3], idx[i] [B,J] int64, logits[i] [B,J] -> cumulative transforms, list of [B,3,4]."""
    xyz = pt_src
    out: List[torch.Tensor] = []
    for i in range(len(logits)):
        ref_new = torch.gather(pt_ref, 1, idx[i][:, :, None].expand(-1, -1, 3))
        w = logits[i].sigmoid()[:, :, None]
        T, _ = OracleNet.kabsch(xyz, ref_new, w)
        xyz = OracleNet.se3_apply(T.detach(), xyz)
        out.append(T if i == 0 else OracleNet.se3_compose(T, out[-1]))
    return out


def scan_alignment_loss(pt_src, transform_pred, transform_gt, logits, labels, loss_type="mae", wt_ptDist=1.0, wt_inlier=1.0,
                        discount=0.5) -> Dict[str, torch.Tensor]:
    """ScanAlignmentLoss.forward with reduction='mean' and wt_pose_loss = 0 (the defaults of arguments.py:51-61).
    labels[i] [B,J] float = find_correct_correspondence per iteration."""
    n = len(transform_pred)
    d: Dict[str, torch.Tensor] = {}
    gt = OracleNet.se3_apply(transform_gt, pt_src)
    for i in range(n):
        pred = OracleNet.se3_apply(transform_pred[i], pt_src)
        if wt_ptDist > 0:
            d[f"{loss_type}_{i}"] = F.l1_loss(pred, gt) if loss_type == "mae" else F.mse_loss(pred, gt)
    if wt_inlier > 0 and labels is not None:
        for i in range(n):
            d[f"outlier_{i}"] = F.binary_cross_entropy_with_logits(logits[i], labels[i]) * wt_inlier
    d["total"] = torch.sum(torch.stack([v * discount ** (n - int(k[k.rfind("_") + 1:]) - 1) for k, v in d.items()]), dim=0)
    return d


def loss_and_grad(pt_src: np.ndarray, pt_ref: np.ndarray, idx: np.ndarray, logits: np.ndarray, labels: np.ndarray,
                  transform_gt: np.ndarray, loss_type="mae", wt_ptDist=1.0, wt_inlier=1.0, discount=0.5):
    """numpy in / out: idx, logits, labels [n_iter,B,J] -> (dict of loss values, d total / d logits [n_iter,B,J])."""
    ps, pr = torch.from_numpy(pt_src).float(), torch.from_numpy(pt_ref).float()
    lg = [torch.from_numpy(l).float().requires_grad_(True) for l in logits]
    ix = [torch.from_numpy(i.astype(np.int64)) for i in idx]
    lb = [torch.from_numpy(l).float() for l in labels]
    T = replay(ps, pr, ix, lg)
    d = scan_alignment_loss(ps, T, torch.from_numpy(transform_gt).float(), lg, lb, loss_type, wt_ptDist, wt_inlier, discount)
    d["total"].backward()
    return {k: float(v.detach()) for k, v in d.items()}, np.stack([l.grad.numpy() for l in lg]), np.stack([t.detach().numpy() for t in T], 1)

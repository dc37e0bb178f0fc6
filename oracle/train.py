"""Oracle: the inlier model's TRAINING pass under torch autograd (TEST INFRASTRUCTURE ONLY - see oracle/__init__.py).

Restates what ``my_model.train()`` + ``loss.backward()`` + ``optimizer.step()`` (reference train.py:379-448) do to the one
sub-network the alignment loss reaches (network/model.py:556-588: matching under no_grad, ``R_t.detach()``): ``RandLA.forward``
(network/RandLANet.py:311-372) with ``fc_label``'s BatchNorm1d on batch statistics (:44) and Dropout(0.5) (:366) given as an
explicit keep mask, on the functional layers of ``oracle/network.py``.  Pinned against the imported reference's own autograd by
``oracle/gen_golden_train.py`` -> ``tests/golden/train_cases.npz``.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F

from oracle.network import OracleNet, _gather_nbr, _gather_pts
from deepsir_amd.arch import level_sizes


def trainable(net: OracleNet, prefix: str = "inlier_model") -> Dict[str, torch.Tensor]:
    """Marks the parameters under ``prefix`` as autograd leaves (BatchNorm running statistics stay buffers)."""
    out = {}
    for k, v in net.p.items():
        if k.startswith(prefix + ".") and v.dtype == torch.float32 and not k.endswith(("running_mean", "running_var")):
            v.requires_grad_(True)
            out[k] = v
    return out


def fc_label_train(net: OracleNet, prefix: str, x: torch.Tensor, update_running: bool = True) -> torch.Tensor:
    """MLP([64, 64, 32, ncls]) in training mode: Conv1d + BatchNorm1d(batch statistics, momentum 0.1) + LeakyReLU(0.2) twice,
    then Conv1d (RandLANet.py:34-55, :272)."""
    pos = 0
    for i in range(3):
        x = F.conv1d(x, net.p[f"{prefix}.{pos}.weight"], net.p[f"{prefix}.{pos}.bias"])
        pos += 1
        if i < 2:
            rm, rv = net.p[f"{prefix}.{pos}.running_mean"], net.p[f"{prefix}.{pos}.running_var"]
            if not update_running:
                rm, rv = rm.clone(), rv.clone()
            x = F.batch_norm(x, rm, rv, net.p[f"{prefix}.{pos}.weight"], net.p[f"{prefix}.{pos}.bias"], True, 0.1, 1e-5)
            x = F.leaky_relu(x, 0.2)
            pos += 2
    return x


def randla_train(net: OracleNet, prefix: str, features: torch.Tensor, xyz_multi: torch.Tensor, neigh_idx: torch.Tensor,
                 sub_idx: torch.Tensor, interp_idx: torch.Tensor, keep_mask: Optional[torch.Tensor] = None,
                 update_running: bool = True, return_feat: bool = False):
    """RandLA.forward in training mode -> logits [B, ncls, N] (and the 64-d features before the dropout when asked).
    keep_mask [B, 64, N] bool (None: dropout off)."""
    L = len(net.cfg.d_out)
    N = features.shape[1]
    n = level_sizes(N, net.cfg.sub_sampling_ratio)
    off = np.concatenate([[0], np.cumsum(n[:L])])
    soff = np.concatenate([[0], np.cumsum(n[1:L + 1])])
    xyz = xyz_multi.permute(0, 2, 1).contiguous()
    x = net.mlp2d(prefix + ".mlp_pre", features.permute(0, 2, 1).contiguous().unsqueeze(3))
    skips = []
    for l in range(L):
        a, b = int(off[l]), int(off[l + 1])
        enc = net.res_block(f"{prefix}.dilated_res_blocks.{l}", x, xyz[:, :, a:b], neigh_idx[:, a:b])
        x = _gather_nbr(enc.squeeze(3), sub_idx[:, int(soff[l]):int(soff[l + 1])]).max(dim=3, keepdim=True)[0]
        if l == 0:
            skips.append(enc)
        skips.append(x)
    x = net.mlp2d(prefix + ".mlp_mid", skips[-1])
    for j in range(L):
        a, b = int(off[L - j - 1]), int(off[L - j])
        up = _gather_pts(x.squeeze(3), interp_idx[:, a:b, 0]).unsqueeze(3)
        x = net.mlp2d(f"{prefix}.decoder_blocks.{j}", torch.cat([skips[-j - 2], up], dim=1))
    feat = F.conv2d(x, net.p[prefix + ".mlp_out.weight"]).squeeze(3)
    dropped = feat if keep_mask is None else feat * keep_mask.to(feat.dtype) * 2.0   # nn.Dropout(0.5) with its mask made explicit
    logits = fc_label_train(net, prefix + ".fc_label", dropped, update_running)
    return (logits, feat) if return_feat else logits


def adam_reference(params: Dict[str, torch.Tensor], lr: float = 1e-3) -> torch.optim.Adam:
    """The reference's optimiser (train.py:323)."""
    return torch.optim.Adam(list(params.values()), lr=lr)


def semantic_loss(logits: torch.Tensor, labels: torch.Tensor, class_weights) -> torch.Tensor:
    """SemanticLoss.compute_loss (loss.py:930-960): points labelled 0 are dropped, class = label - 1 (the `reducing_list`
    gather), weighted cross entropy with reduction 'mean' (:919-928).  logits [B, C, N], labels [B, N] int64.
    PARITY UNPINNED: the reference hands F.cross_entropy a [1, C] weight tensor, which the torch of this image rejects
    ("weight tensor should be defined either for all 19 classes or no classes"), so its own loss cannot be run here; this is
    the same call with the weights as the [C] vector the API documents."""
    C = logits.shape[1]
    lg = logits.transpose(1, 2).reshape(-1, C)
    lb = labels.reshape(-1)
    keep = lb != 0
    w = torch.as_tensor(class_weights, dtype=torch.float32).reshape(-1)
    return F.cross_entropy(lg[keep], lb[keep] - 1, weight=w, reduction="mean")


# ------------------------------------------------------------------ `feat` pipeline: aggregation MLPs under DetDesLoss
def mlp1d_train(net: OracleNet, prefix: str, x: torch.Tensor, n_layers: int, update_running: bool = True) -> torch.Tensor:
    """MLP (RandLANet.py:34-55) in training mode: Conv1d + BatchNorm1d(batch statistics) + LeakyReLU(0.2), last layer bare."""
    pos = 0
    for i in range(n_layers):
        x = F.conv1d(x, net.p[f"{prefix}.{pos}.weight"], net.p[f"{prefix}.{pos}.bias"])
        pos += 1
        if i < n_layers - 1:
            rm, rv = net.p[f"{prefix}.{pos}.running_mean"], net.p[f"{prefix}.{pos}.running_var"]
            if not update_running:
                rm, rv = rm.clone(), rv.clone()
            x = F.batch_norm(x, rm, rv, net.p[f"{prefix}.{pos}.weight"], net.p[f"{prefix}.{pos}.bias"], True, 0.1, 1e-5)
            x = F.leaky_relu(x, 0.2)
            pos += 2
    return x


def aggregate_train(net: OracleNet, xyz: torch.Tensor, feat0: torch.Tensor, score: torch.Tensor, update_running: bool = True):
    """One cloud batch's half of ``aggregation`` in training mode (model.py:209-235) followed by forward_pair's second
    F.normalize (:651-652).  xyz [B,3,M], feat0 [B,64,M], score [B,M] -> descriptors [B,64,M]."""
    g = torch.cat((xyz, score[:, None, :]), dim=1)
    # the reference calls mlp_feat on src and ref before mlp_att on src and ref (running statistics update order); one side here
    d = mlp1d_train(net, "mlp_feat", feat0, 3, update_running) + mlp1d_train(net, "mlp_att", g, 5, update_running)
    d = F.normalize(mlp1d_train(net, "mlp_proj", d, 1, update_running), p=2, dim=1)
    return F.normalize(d, p=2, dim=1)


def det_des_loss(feat_src, feat_ref, pt_src, pt_ref, score_ref, transform_gt, thres_radius: float, det_loss_weight: float = 1.0):
    """DetDesLoss.forward (loss.py:667-702) = CircleLoss.forward(feat_ref, feat_src, pt_ref, T_gt pt_src, score_ref, score_src)
    (:500-571), restated line by line INCLUDING what the arithmetic really does: ``dist_min`` is the minimum of
    ``dist_pc * false_negative`` - zero as soon as one column of the row is outside the radius -, so ``pos_mask`` marks only
    exact coincidences and the positive set is the false-negative ball; masked entries enter both log-sum-exps with
    exponent 0 (weight 0), i.e. as exp(0) = 1; the accuracy sums over the batch but divides by the row count.
    feat_* [B,C,M], pt_* [B,3,M], score_ref [B,M], transform_gt [B,3,4] -> (total, accuracy)."""
    eps, log_scale, pos_margin, neg_margin = 1e5, 10.0, 0.1, 1.4
    pos_pc = transform_gt[:, :3, :3] @ pt_src + transform_gt[:, :3, 3][:, :, None]                  # se3_torch.transform_V2
    anc_feat, pos_feat, anc_pc = feat_ref, feat_src, pt_ref
    anc_score = score_ref / torch.sum(score_ref, dim=1, keepdim=True)
    dist_pc = torch.norm(anc_pc.unsqueeze(3) - pos_pc.unsqueeze(2), dim=1)
    sq = -2 * torch.matmul(anc_feat.permute(0, 2, 1).contiguous(), pos_feat)                         # square_distance_V2 (matchnet.py:96-113)
    sq = sq + torch.sum(anc_feat ** 2, dim=1)[:, :, None] + torch.sum(pos_feat ** 2, dim=1)[:, None, :]
    dist_feat = torch.sqrt(sq + 1e-16)
    fn = dist_pc < thres_radius
    dist_min = torch.min(dist_pc * fn.float(), dim=2, keepdim=True)[0]
    pos_mask = torch.eq(dist_pc, dist_min)
    neg_mask = torch.logical_not(pos_mask | fn)
    pos = dist_feat - eps * neg_mask.float()
    pos_w = torch.clamp((pos - pos_margin).detach(), min=0)
    lse_pos = torch.logsumexp(log_scale * (pos - pos_margin) * pos_w, dim=-1)
    neg = dist_feat + eps * (~neg_mask).float()
    neg_w = torch.clamp((neg_margin - neg).detach(), min=0)
    neg_weighted = log_scale * (neg_margin - neg) * neg_w
    loss_col = F.softplus(lse_pos + torch.logsumexp(neg_weighted, dim=-1)) / log_scale
    loss_row = F.softplus(lse_pos + torch.logsumexp(neg_weighted, dim=-2)) / log_scale
    loss_feat = torch.mean(loss_col + loss_row)
    furthest_positive = torch.max(dist_feat * pos_mask.float(), dim=-1)[0]
    closest_negative = torch.min(dist_feat + eps * pos_mask.float(), dim=-1)[0]
    diff = furthest_positive - closest_negative
    accuracy = (diff < 0).sum() * 100.0 / diff.shape[1]
    loss_det = torch.mean(diff * anc_score)
    return loss_feat + loss_det * det_loss_weight, accuracy


def register_train(net: OracleNet, data: Dict[str, torch.Tensor], n_iter: int, masks: Optional[dict] = None):
    """forward_align_4 (model.py:520-607) with the whole network in training mode, as train.py:379 runs it: BatchNorm on batch
    statistics and Dropout on in the frozen sub-networks as well.  masks: {'fe_src', 'fe_ref': [B,64,N] bool,
    'inlier': list of n_iter [B,64,N] bool} (None entries: dropout off).  -> (cumulative transforms, idx list, logits list)."""
    masks = masks or {}
    side = {}
    for s_ in ("src", "ref"):
        k = "points_" + s_
        with torch.no_grad():                                                                # requires_grad False (freeze_model)
            logits, feat = randla_train(net, "feat_extractor", data[k], data[k + "_xyz"], data[k + "_neigh_idx"], data[k + "_sub_idx"],
                                        data[k + "_interp_idx"], masks.get("fe_" + s_), True, True)
            prob, label = torch.max(logits, dim=1, keepdim=True)
            N = data[k].shape[1]
            xyz = data[k + "_xyz"].permute(0, 2, 1)[:, :, :N].contiguous()
            score = net.score(feat, xyz, prob, label, data[k + "_neigh_idx"][:, :N])
        side[s_] = (xyz, feat, score)
    x_s, f_s, sc_s = side["src"]
    x_r, f_r, sc_r = side["ref"]
    xyz = x_s
    transforms, idxs, lgs = [], [], []
    for it in range(n_iter):
        with torch.no_grad():
            # per module the src call precedes the ref call (model.py:217-224); aggregate_train keeps that order per module
            d_s = aggregate_train_first(net, xyz, f_s, sc_s)
            d_r = aggregate_train_first(net, x_r, f_r, sc_r)
            idx = OracleNet.nn_match(d_s, d_r)
        ref_new = torch.gather(x_r, 2, idx[:, None, :].expand(-1, 3, -1))
        cat = torch.cat((xyz, ref_new), dim=1).permute(0, 2, 1).contiguous()
        m = masks.get("inlier")
        logit = randla_train(net, "inlier_model", cat, data["points_src_xyz"], data["points_src_neigh_idx"], data["points_src_sub_idx"],
                             data["points_src_interp_idx"], None if m is None else m[it]).squeeze(1)
        p_s, p_r = xyz.permute(0, 2, 1).contiguous(), ref_new.permute(0, 2, 1).contiguous()
        T, _ = OracleNet.kabsch(p_s, p_r, logit.sigmoid()[:, :, None])
        xyz = OracleNet.se3_apply(T.detach(), p_s).permute(0, 2, 1).contiguous()
        transforms.append(T if it == 0 else OracleNet.se3_compose(T, transforms[-1]))
        idxs.append(idx); lgs.append(logit)
    return transforms, idxs, lgs


def aggregate_train_first(net: OracleNet, xyz, feat0, score):
    """``aggregation`` (model.py:209-235) for one side in training mode, WITHOUT forward_pair's second normalize (the align
    pipeline matches on aggregation's own output, model.py:552-566)."""
    g = torch.cat((xyz, score[:, None, :]), dim=1)
    d = mlp1d_train(net, "mlp_feat", feat0, 3) + mlp1d_train(net, "mlp_att", g, 5)
    return F.normalize(mlp1d_train(net, "mlp_proj", d, 1), p=2, dim=1)

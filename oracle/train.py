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
                 update_running: bool = True) -> torch.Tensor:
    """RandLA.forward in training mode -> logits [B, ncls, N].  keep_mask [B, 64, N] bool (None: dropout off)."""
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
    if keep_mask is not None:
        feat = feat * keep_mask.to(feat.dtype) * 2.0          # nn.Dropout(0.5) with its mask made explicit
    return fc_label_train(net, prefix + ".fc_label", feat, update_running)


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

"""Oracle: the Adam fine-tune branch of the reference's ``pose_optimization`` (TEST INFRASTRUCTURE ONLY).

Restates ``transformation_finetune`` (reference test.py:159-207) with its helpers ``HighDimSmoothL1Loss``
(test.py:103-131), ``Transformation`` and ``ortho2rotation`` (network/DGR.py:60-132): the pose is re-parametrised as a 6-D
rotation (first two columns of R, orthonormalised by Gram-Schmidt) plus a translation and fitted to the network's last
correspondences with a weighted robust-L1 ("smooth L1 on the point distance in units of `quantization_size`") loss by
Adam (lr 0.1, ExponentialLR gamma 0.999), stopping when the loss falls below 1e-7, after `max_iter` steps, or after the
relative loss change was below `break_threshold_ratio` `max_break_count` times.

**Parity unpinned**: the branch is switched off in the reference (``use_tune = False``, test.py:215) and neither
``test.py`` nor ``network/DGR.py`` can be imported here (both import open3d).  The restatement below makes the same
library calls (torch autograd, ``torch.optim.Adam``, ``ExponentialLR``) on the same float32 tensors, so it is what the
reference would compute; ``csrc/finetune.hip`` is held to it.
"""
from __future__ import annotations

import numpy as np
import torch

_EPS32 = float(np.finfo(np.float32).eps)


def ortho2rotation(poses: torch.Tensor) -> torch.Tensor:
    """[B,6] -> [B,3,3]: columns x, y, z (DGR.py:60-108)."""
    def normalize(v):
        return v / torch.clamp(torch.sqrt((v ** 2).sum(1, keepdim=True)), min=1e-8)

    x_raw, y_raw = poses[:, 0:3], poses[:, 3:6]
    x = normalize(x_raw)
    proj = ((x * y_raw).sum(1, keepdim=True) / torch.clamp((x ** 2).sum(1, keepdim=True), min=1e-8)) * x
    y = normalize(y_raw - proj)
    z = torch.cross(x, y, dim=1)
    return torch.stack((x, y, z), 2)


def smooth_l1(X: torch.Tensor, Y: torch.Tensor, weights, q: float, delta: float = 1.0) -> torch.Tensor:
    """HighDimSmoothL1Loss.__call__ (test.py:112-131). X, Y [1,N,3]; weights [1,N,1] or None."""
    sq = torch.sum(((X - Y) / q) ** 2, dim=2, keepdim=True)
    half = 0.5 * (sq < delta).float()
    loss = (0.5 - half) * (torch.sqrt(sq + _EPS32) - 0.5 * delta ** 2) + half * sq
    return loss.mean() if weights is None else (loss * weights).sum() / weights.sum()


def transformation_finetune(xyz_src: np.ndarray, xyz_ref: np.ndarray, pose: np.ndarray, weights=None, quantization_size: float = 1.0,
                            max_iter: int = 1000, break_threshold_ratio: float = 1e-4, max_break_count: int = 20):
    """xyz_* [N,3] matched points, pose [3,4], weights [N] or None -> (pose [3,4] float32, dict(iterations, loss, break_count))."""
    X = torch.from_numpy(np.ascontiguousarray(xyz_src, np.float32))[None]
    Y = torch.from_numpy(np.ascontiguousarray(xyz_ref, np.float32))[None]
    W = None if weights is None else torch.from_numpy(np.ascontiguousarray(weights, np.float32)).reshape(1, -1, 1)
    P = torch.from_numpy(np.ascontiguousarray(pose, np.float32))
    rot6d = torch.nn.Parameter(torch.cat([P[:3, 0], P[:3, 1]])[None].clone())      # Transformation.__init__ (DGR.py:111-123)
    trans = torch.nn.Parameter(P[:3, 3][None].clone())

    def forward():
        R = ortho2rotation(rot6d)[0]
        return (X[0] @ R.t() + trans[0])[None]

    opt = torch.optim.Adam([rot6d, trans], lr=1e-1)
    sched = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=0.999)
    loss_prev = smooth_l1(forward(), Y, W, quantization_size).item()
    brk, i, loss = 0, -1, None
    for i in range(max_iter):
        loss = smooth_l1(forward(), Y, W, quantization_size)
        if loss.item() < 1e-7:
            break
        opt.zero_grad()
        loss.backward()
        opt.step()
        sched.step()
        if abs(loss_prev - loss.item()) < loss_prev * break_threshold_ratio:
            brk += 1
            if brk >= max_break_count:
                break
        loss_prev = loss.item()
    out = np.zeros((3, 4), np.float32)
    out[:, :3] = ortho2rotation(rot6d.detach())[0].numpy()
    out[:, 3] = trans[0].detach().numpy()
    return out, {"iterations": i, "loss": float(loss.item()) if loss is not None else float(loss_prev), "break_count": brk}

"""SE(3) helpers on ``[B,3,4]`` tensors — host-side mirror of the reference's
common/math/se3_torch.py:6-77 (identity / inverse / concatenate / transform),
used by the evaluation harness.  Inside the registration loop the same algebra
runs in the Kabsch kernel (csrc/kabsch.hip)."""
import torch


def identity(batch_size: int) -> torch.Tensor:
    return torch.eye(3, 4)[None].repeat(batch_size, 1, 1)


def inverse(Rt: torch.Tensor) -> torch.Tensor:
    R, t = Rt[..., :3, :3], Rt[..., :3, 3]
    Rt_ = R.transpose(-1, -2)
    return torch.cat([Rt_, Rt_ @ -t[..., None]], dim=-1)


def concatenate(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """a o b : apply b first, then a."""
    Ra, ta, Rb, tb = a[..., :3, :3], a[..., :3, 3], b[..., :3, :3], b[..., :3, 3]
    return torch.cat([Ra @ Rb, Ra @ tb[..., None] + ta[..., None]], dim=-1)


def transform(Rt: torch.Tensor, pts: torch.Tensor) -> torch.Tensor:
    """pts [B,N,3] -> pts R^T + t."""
    return pts @ Rt[..., :3, :3].transpose(-1, -2) + Rt[..., :3, 3][..., None, :]

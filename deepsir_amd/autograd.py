"""The reference's training loop, unchanged, on the drop-in ``Network`` (reference train.py:396-446):

    optimizer.zero_grad()
    pred_transforms, endpoints = my_model(train_data, opt_tuple)
    endpoints['transform_gt'] = ...; endpoints['matches'] = ...
    loss = my_model.loss_align_fun(endpoints, reduction='mean')['total']      # or loss_feat_fun / loss_label_fun
    loss.backward()
    optimizer.step()

Nothing here computes: the training forward, the losses and every backward operator are the HIP kernels behind
include/dsir_train.h and ``dsir_align_loss_backward`` (deepsir_amd/train.py composes them in the reference's module order).  This
file is the seam to torch's autograd the loop needs - two kinds of ``torch.autograd.Function``:

* ``_Taped``: the module's trainable ``nn.Parameter``s go in, the network outputs the loss will read come out; its backward hands
  the incoming gradient to the trainer's HIP backward pass (the tape kept by the forward) and returns every parameter's gradient,
  which autograd accumulates into ``param.grad`` - what ``optimizer.step()`` and the loop's NaN check (train.py:436-441) read;
* the loss functions' own nodes (``_AlignLoss``, ``_WeightedCE``, ``_DetDes``): the HIP loss operator returns the loss AND its
  gradient with respect to the network output in one call; the node scales it by the incoming gradient.

``ScanAlignmentLoss`` / ``SemanticLoss`` / ``DetDesLoss`` restate the CALL CONTRACT of the reference's loss modules
(network/loss.py:705-851, :854-995, :652-702: argument dict, keys of the result, reduction) over those operators; their values and
gradients are pinned by the imported reference's own (tests/golden/align_loss_cases.npz, train_cases.npz; tests/test_train.py).
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch


class _Taped(torch.autograd.Function):
    """forward(run, n_out, *params): ``run()`` executes the HIP training forward and returns (outputs, backward_fn);
    backward_fn(grad_outputs) runs the HIP backward and returns one gradient (or None) per parameter, in order."""

    @staticmethod
    def forward(ctx, run: Callable, n_out: int, *params):
        outs, back = run()
        assert len(outs) == n_out
        ctx.back = back
        ctx.n_params = len(params)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grad_outs):
        back, ctx.back = ctx.back, None
        if back is None:
            raise RuntimeError("the training forward's tape has been used: call forward again before a second backward")
        grads = back([None if g is None else g.contiguous() for g in grad_outs])
        assert len(grads) == ctx.n_params
        return (None, None) + tuple(grads)


def run_taped(params: Sequence[torch.nn.Parameter], run: Callable, n_out: int):
    """Outputs of ``run`` as tensors with a grad_fn that leads into ``params``."""
    return _Taped.apply(run, n_out, *params)


# ----------------------------------------------------------------------------------------------------------- align
class _AlignLoss(torch.autograd.Function):
    """total = sum_i discount^(n - 1 - i) (dist_i + outlier_i) of ScanAlignmentLoss (reduction='mean'), with d total / d logits."""

    @staticmethod
    def forward(ctx, logits, engine, pt_src, pt_ref, idx, labels, transform_gt, kw):
        out = engine.align_loss_backward(pt_src, pt_ref, idx, logits.detach().contiguous(), labels, transform_gt, **kw)
        ctx.save_for_backward(out["grad_logits"])
        ctx.terms = out["losses"]
        ctx.transforms = out["transforms"]
        return torch.tensor(out["losses"]["total"], dtype=torch.float32, device=logits.device)

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None, None, None, None, None, None, None


class ScanAlignmentLoss:
    """``my_model.loss_align_fun`` (reference network/loss.py:705-851; wt_pose_loss = 0, its default).  data: the endpoints of an
    `align` forward plus 'transform_gt' [B,3,4] and optionally 'matches' (per pair an int [n',2] array, as the reference's loader
    gives) for the correspondence-confidence term.  reduction 'mean': scalar tensors, 'total' carries the gradient back into the
    inlier model (through the forward's tape when the forward ran in training mode); 'none': per-pair values [B] (validation,
    train.py:136), no gradient."""

    def __init__(self, net, args):
        self._net = net
        self.loss_type = getattr(args, "loss_type", "mae")
        self.wt_ptDist_loss = float(getattr(args, "wt_ptDist_loss", 1.0))
        self.wt_inlier_loss = float(getattr(args, "wt_inlier_loss", 1.0))
        self.wt_pose_loss = float(getattr(args, "wt_pose_loss", 0.0))
        self.discount_factor = float(getattr(args, "loss_discount_factor", 0.5))
        if self.loss_type not in ("mae", "mse"):
            raise AssertionError("loss_type must be 'mae' or 'mse' (reference loss.py:721)")
        if self.wt_pose_loss > 0:
            raise NotImplementedError("wt_pose_loss > 0 (off by default, arguments.py:57) is outside the accelerated path")

    def __call__(self, data: Dict, reduction=None):
        from .train import find_correct_correspondence
        if reduction not in ("mean", "none"):
            raise AssertionError("reduction must be 'mean' or 'none' (reference loss.py:765)")
        tr = data.get("_train")
        pt_src, pt_ref = data["pt_src"].float().contiguous(), data["pt_ref"].float().contiguous()
        dev = pt_src.device
        B, J, _ = pt_src.shape
        perm = data["perm_matrices"]
        n_iter = len(perm)
        logits = tr["logits"] if tr is not None else torch.stack([p.reshape(B, J) for p in perm]).contiguous()
        pp = data["pred_pairs"]
        idx = tr["idx"] if tr is not None else getattr(pp, "_idx", None)
        if idx is None:     # materialised CPU pairs [B, J, 2] per iteration (the reference's own layout)
            idx = torch.stack([torch.as_tensor(p)[:, :, 1] for p in pp]).to(torch.int32).to(dev)
        idx = idx.to(torch.int32).contiguous()
        labels = None
        if self.wt_inlier_loss > 0 and "matches" in data:
            labels = torch.from_numpy(find_correct_correspondence(data["matches"], idx, J)).to(dev)
        T_gt = data["transform_gt"].float().to(dev).contiguous()
        eng = tr["engine"] if tr is not None else self._net._ensure_engine(max(J, pt_ref.shape[1]), B)
        kw = dict(loss_type=self.loss_type, wt_ptDist_loss=self.wt_ptDist_loss, wt_inlier_loss=self.wt_inlier_loss,
                  loss_discount_factor=self.discount_factor)
        if reduction == "none":
            out = eng.align_loss_backward(pt_src, pt_ref, idx, logits.detach().contiguous(), labels, T_gt, per_pair=True, **kw)
            return {k: torch.from_numpy(np.asarray(v, np.float32)).to(dev) for k, v in out["losses_per_pair"].items()}
        total = _AlignLoss.apply(logits, eng, pt_src, pt_ref, idx, labels, T_gt, kw)
        terms = total.grad_fn.terms if total.grad_fn is not None else eng.align_loss_backward(pt_src, pt_ref, idx, logits.detach().contiguous(), labels, T_gt, **kw)["losses"]
        res = {k: torch.tensor(v, dtype=torch.float32, device=dev) for k, v in terms.items() if k != "total"}
        res["total"] = total
        return res

    forward = __call__


# ----------------------------------------------------------------------------------------------------------- label
class _WeightedCE(torch.autograd.Function):
    """F.cross_entropy(weight = class weights) over the labelled points of one cloud batch, loss.py:930-960."""

    @staticmethod
    def forward(ctx, logits2d, ops, labels, class_weights):
        ops.begin()
        d, out = ops.weighted_ce(logits2d.detach().contiguous(), labels, class_weights)
        ctx.save_for_backward(d)
        ctx.stats = out
        return out[0].clone()

    @staticmethod
    def backward(ctx, g):
        (d,) = ctx.saved_tensors
        return d * g, None, None, None


class SemanticLoss:
    """``my_model.loss_label_fun`` (reference network/loss.py:854-995): (loss_src + loss_ref, acc_src + acc_ref) from
    endpoints['logits_src' / 'logits_ref'] [B,C,N] and endpoints['labels_src' / 'labels_ref'] [B,N] (0 = unlabeled, ignored)."""

    def __init__(self, net, args=None):
        self._net = net

    def __call__(self, endpoints: Dict):
        from .train import _Ops, semantic_class_weights
        dev = endpoints["logits_src"].device
        ops = _Ops(dev)
        cw = torch.tensor(semantic_class_weights(), dtype=torch.float32, device=dev)
        loss, acc = None, None
        for side in ("src", "ref"):
            lg = endpoints[f"logits_{side}"]                       # [B, C, N]
            B, C_, N = lg.shape
            l2 = lg.permute(0, 2, 1).reshape(B * N, C_)
            labels = endpoints[f"labels_{side}"].to(torch.int32).to(dev).reshape(-1).contiguous()
            v = _WeightedCE.apply(l2, ops, labels, cw)
            st = v.grad_fn.stats if v.grad_fn is not None else ops.weighted_ce(l2.detach().contiguous(), labels, cw)[1]
            a = st[2] / torch.clamp(st[3], min=1.0)
            loss = v if loss is None else loss + v
            acc = a if acc is None else acc + a
        return loss, acc

    forward = __call__


# ----------------------------------------------------------------------------------------------------------- feat
class _DetDes(torch.autograd.Function):
    """DetDesLoss = CircleLoss on the key-point descriptors + the detection term (loss.py:483-571, :652-702)."""

    @staticmethod
    def forward(ctx, d_ref, d_src, ops, pt_ref, pt_src, score_ref, transform_gt, thres_radius, det_loss_weight):
        ops.begin()
        out, g_ref, g_src = ops.det_des_loss(d_ref.detach().contiguous(), d_src.detach().contiguous(), pt_ref, pt_src, score_ref, transform_gt,
                                             thres_radius, det_loss_weight)
        ctx.save_for_backward(g_ref, g_src)
        ctx.vals = out
        return out[0].clone()

    @staticmethod
    def backward(ctx, g):
        g_ref, g_src = ctx.saved_tensors
        return g_ref * g, g_src * g, None, None, None, None, None, None, None


class DetDesLoss:
    """``my_model.loss_feat_fun`` (reference network/loss.py:652-702): (loss_feat + det_loss_weight * loss_det, acc) from
    endpoints feat_src / feat_ref [B,C,M], pt_src / pt_ref [B,3,M], score_src / score_ref [B,M] and 'transform_gt' [B,3,4]."""

    def __init__(self, net, args):
        self._net = net
        self.thres_radius = float(getattr(args, "thres_radius", -1.0))
        self.det_loss_weight = float(getattr(args, "det_loss_weight", 1.0))

    def __call__(self, data: Dict):
        from .train import _Ops
        d_src = data["feat_src"].permute(0, 2, 1)                  # [B, M, C] point-major, as the operator takes them
        d_ref = data["feat_ref"].permute(0, 2, 1)
        dev = d_src.device
        pm = lambda t: t.permute(0, 2, 1).float().contiguous()
        if self.thres_radius <= 0:
            raise AssertionError("args.thres_radius must be set (the reference's loaders set it, threeDMatch_loader.py:50)")
        ops = _Ops(dev)
        v = _DetDes.apply(d_ref, d_src, ops, pm(data["pt_ref"]), pm(data["pt_src"]), data["score_ref"].float().contiguous(),
                          data["transform_gt"].float().to(dev).contiguous(), self.thres_radius, self.det_loss_weight)
        vals = v.grad_fn.vals if v.grad_fn is not None else ops.det_des_loss(d_ref.contiguous(), d_src.contiguous(), pm(data["pt_ref"]), pm(data["pt_src"]),
                                                                              data["score_ref"].float().contiguous(),
                                                                              data["transform_gt"].float().to(dev).contiguous(),
                                                                              self.thres_radius, self.det_loss_weight)[0]
        return v, vals[3]

    forward = __call__

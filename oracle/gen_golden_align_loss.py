"""Golden vectors for the training slice (loss + gradient down to the inlier logits), produced by the REFERENCE's own
``ScanAlignmentLoss`` (network/loss.py:705-851), ``compute_rigid_transform_2`` (network/model.py:22-66) and
``se3_torch`` (common/math/se3_torch.py) under torch autograd, imported from /root/reference (build container only;
TEST INFRASTRUCTURE).

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/oracle/gen_golden_align_loss.py

The differentiable tail of forward_align_4 (model.py:571-595: sigmoid -> weighted Kabsch -> detach-transform ->
concatenate) is replayed from given logits / correspondences with the reference's functions; ``.cuda()`` calls inside the
loss (it moves the BCE term to the GPU, loss.py:819) are met by running it with reduction='mean' on a build where
``Tensor.cuda`` is the identity for the duration of the call (in memory, on this process's tensors only: there is no GPU
in the build container).  Nothing of the reference is written anywhere."""
import os
import sys
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepsir_amd.synth import make_pair  # noqa: E402


def make_case(rng, B, J, K, n_iter, outliers):
    """Matched-point problem as the loss sees it: src clouds, ref clouds, per-iteration correspondences (mostly right,
    some wrong), inlier logits that mostly (not perfectly) tell them apart, ground-truth pose and match list."""
    src, ref, gt, matches = [], [], [], []
    for b in range(B):
        p = make_pair(max(J, K), int(rng.integers(1, 10_000)), 3)
        T = p["transform_gt"][0].astype(np.float64)
        s = p["points_src"][0, :J].astype(np.float32)
        r_exact = (s.astype(np.float64) @ T[:, :3].T + T[:, 3]).astype(np.float32)
        r = np.concatenate([r_exact, rng.uniform(0, 3, (K - J, 3)).astype(np.float32)], 0) if K > J else r_exact[:K]
        perm = rng.permutation(K)
        r = r[perm]
        inv = np.empty(K, np.int64); inv[perm] = np.arange(K)
        src.append(s); ref.append(r); gt.append(T.astype(np.float32))
        m = np.stack([np.arange(min(J, K)), inv[:min(J, K)]], 1)
        matches.append(m[rng.random(len(m)) < 0.9])                      # the match list misses a few true pairs
    idx = np.zeros((n_iter, B, J), np.int64)
    logits = np.zeros((n_iter, B, J), np.float32)
    for i in range(n_iter):
        for b in range(B):
            true = np.full(J, -1, np.int64)
            mm = matches[b]
            true[mm[:, 0]] = mm[:, 1]
            wrong = (rng.random(J) < outliers * (1.0 - 0.2 * i)) | (true < 0)
            idx[i, b] = np.where(wrong, rng.integers(0, K, J), true)
            logits[i, b] = np.where(wrong, rng.normal(-1.0, 1.5, J), rng.normal(1.5, 1.5, J)).astype(np.float32)
    return np.stack(src), np.stack(ref), np.stack(gt), matches, idx, logits


def main(ref_root="/root/reference"):
    warnings.filterwarnings("ignore")
    sys.path.insert(0, ref_root)
    import arguments  # type: ignore
    import network.model as ref_model  # type: ignore
    from common.math import se3_torch  # type: ignore
    from network.loss import ScanAlignmentLoss  # type: ignore

    args = arguments.train_arguments().parse_args([]) if hasattr(arguments, "train_arguments") else arguments.eval_arguments().parse_args([])
    rng = np.random.Generator(np.random.Philox(key=2024))
    out = {}
    cases = [(2, 600, 600, 3, 0.25, "mae"), (1, 1500, 1700, 5, 0.4, "mae"), (3, 257, 300, 2, 0.1, "mse")]
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self                       # no GPU here; see the module docstring
    try:
        for c, (B, J, K, n_iter, outl, ltype) in enumerate(cases):
            src, ref, gt, matches, idx, logits = make_case(rng, B, J, K, n_iter, outl)
            args.loss_type, args.wt_ptDist_loss, args.wt_inlier_loss, args.wt_pose_loss, args.loss_discount_factor = ltype, 1.0, 1.0, 0.0, 0.5
            loss_fn = ScanAlignmentLoss(args)
            ps, pr = torch.from_numpy(src), torch.from_numpy(ref)
            lg = [torch.from_numpy(logits[i]).requires_grad_(True) for i in range(n_iter)]
            xyz = ps
            transforms, pred_pairs = [], []
            for i in range(n_iter):
                ix = torch.from_numpy(idx[i])
                ref_new = torch.gather(pr, 1, ix[:, :, None].expand(-1, -1, 3))
                R_t, bad = ref_model.compute_rigid_transform_2(xyz, ref_new, weights=lg[i].sigmoid()[:, :, None])
                assert not bad
                xyz = se3_torch.transform(R_t.detach(), xyz)
                transforms.append(R_t if i == 0 else se3_torch.concatenate(R_t, transforms[-1]))
                ar = torch.arange(J)[None, :, None].expand(B, J, 1).int()
                pred_pairs.append(torch.cat([ar, ix.int()[:, :, None]], dim=2))
            data = {"pt_src": ps, "perm_matrices": lg, "transform_pred": transforms, "transform_gt": torch.from_numpy(gt),
                    "pred_pairs": pred_pairs, "matches": [torch.from_numpy(m) for m in matches]}
            d = loss_fn(data, reduction="mean")
            d["total"].backward()
            labels = np.stack([loss_fn.find_correct_correspondence(data["matches"], pred_pairs[i], hash_seed=J) for i in range(n_iter)])
            out[f"c{c}_src"], out[f"c{c}_ref"], out[f"c{c}_gt"] = src, ref, gt
            out[f"c{c}_idx"], out[f"c{c}_logits"], out[f"c{c}_labels"] = idx.astype(np.int32), logits, labels.astype(np.float32)
            for b, m in enumerate(matches):
                out[f"c{c}_matches{b}"] = m.astype(np.int32)
            out[f"c{c}_transforms"] = np.stack([t.detach().numpy() for t in transforms], 1)
            out[f"c{c}_grad_logits"] = np.stack([l.grad.numpy() for l in lg])
            out[f"c{c}_loss_names"] = np.array(sorted(d.keys()))
            out[f"c{c}_loss_values"] = np.array([float(d[k]) for k in sorted(d.keys())], np.float64)
            out[f"c{c}_loss_type"] = np.array(ltype)
            print(f"case {c}: total {float(d['total']):.6f}, |grad| max {np.abs(out[f'c{c}_grad_logits']).max():.3e}")
    finally:
        torch.Tensor.cuda = orig_cuda
    out["n_cases"] = np.asarray(len(cases))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "align_loss_cases.npz"), **out)


if __name__ == "__main__":
    main()

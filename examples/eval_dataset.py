"""Evaluate a checkpoint on the reference's test splits read from disk (the reference's `test.py` main for
`--pipeline align`, test.py:386-457): dataset -> device pre-processing -> registration -> RTE / RRE / success.

    python examples/eval_dataset.py --dataset 3DMatch --root /data/3DMatch --num-points 5000 [--ckpt model.pth] [--limit 50]
    python examples/eval_dataset.py --dataset KITTI   --root /data/kitti   --num-points 16384 --voxel-size 0.3

Directory layouts are the reference's (dataloader/threeDMatch_loader.py, kitti_loader.py); see deepsir_amd/data.py.
Without --ckpt a seeded random state-dict is used (plumbing check only)."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsir_amd import data as D  # noqa: E402
from deepsir_amd.arch import NetConfig  # noqa: E402
from deepsir_amd.harness import evaluate_align, inference_align, summarize  # noqa: E402
from deepsir_amd.model import Network  # noqa: E402
from deepsir_amd.weights import generate_state_dict, to_torch_state_dict  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dataset", choices=["3DMatch", "KITTI"], required=True)
    ap.add_argument("--root", required=True)
    ap.add_argument("--ckpt", default="")
    ap.add_argument("--num-points", type=int, default=5000)
    ap.add_argument("--voxel-size", type=float, default=None)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--limit", type=int, default=0)
    ap.add_argument("--pose-opt", choices=["none", "icp"], default="none")
    a = ap.parse_args()
    kitti = a.dataset == "KITTI"
    cfg = NetConfig(feat_len=4 if kitti else 3)
    args = argparse.Namespace(pipeline="align", num_sub=-1, num_knn=16, out_feat_dim=64, clip_weight_thresh=0.0,
                              feat_len=cfg.feat_len, d_out=[16, 64, 128, 256], num_points=a.num_points,
                              sub_sampling_ratio=[4, 4, 4, 4], use_ppf=False)
    model = Network(args)
    sd = torch.load(a.ckpt, map_location="cpu") if a.ckpt else to_torch_state_dict(generate_state_dict(cfg, 0))
    model.load_state_dict(sd["state_dict"] if "state_dict" in sd else sd)
    model = model.cuda().eval()
    eng = model._ensure_engine(max(a.num_points, 1024), a.batch)   # the engine also runs the datasets' device steps
    voxel = a.voxel_size or (0.3 if kitti else 0.03)
    ds = (D.KittiOdometryTest(a.root, eng, voxel_size=voxel, feat_len=cfg.feat_len, num_points=a.num_points) if kitti
          else D.ThreeDMatchTest(a.root, eng, voxel_size=voxel, num_points=a.num_points))
    n = min(len(ds), a.limit) if a.limit else len(ds)
    pairs = [D.as_batch(ds[i]) for i in range(n)]
    pred, stats = inference_align(pairs, model, a.iters, a.dataset, batch=a.batch,
                                  pose_opt=None if a.pose_opt == "none" else "icp", voxel_size=voxel)
    print({k: round(float(v), 4) for k, v in summarize(stats).items()})
    _, summary = evaluate_align(pred, pairs, eng, a.dataset)
    print({k: round(float(v), 5) for k, v in summary.items()})


if __name__ == "__main__":
    main()

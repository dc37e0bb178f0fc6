"""The optimisation loop of the reference's train.py (:379-489) for its three pipelines, on the device and without autograd:

    python examples/train_pipelines.py --pipeline align --pairs 8 --points 2048 --steps 20
    python examples/train_pipelines.py --pipeline label --pairs 4 --points 2048 --steps 20
    python examples/train_pipelines.py --pipeline feat  --pairs 4 --points 2048 --num-sub 512 --steps 20

`align` trains the inlier model through ScanAlignmentLoss (the only sub-network that loss reaches: matching runs under
no_grad, model.py:556), `label` the feature extractor's semantic head path through SemanticLoss, `feat` the aggregation MLPs
through DetDesLoss (the extractor is frozen there, model.py:136).  Data here is synthetic (random rigid pairs, random
weights, random class labels): the script shows the calls a training driver makes, the data loaders are the reference's own.
Learning-rate decay as `update_learning_rate` (train.py:38-48): x ratio per `--decay-every` steps, clipped at 1e-4."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsir_amd.arch import NetConfig  # noqa: E402
from deepsir_amd.engine import Engine  # noqa: E402
from deepsir_amd.synth import make_pair  # noqa: E402
from deepsir_amd.train import (AggregationTrainer, AlignTrainStep, RandlaTrainer, feat_pipeline_inputs, train_step_feat,  # noqa: E402
                               train_step_label)
from deepsir_amd.weights import generate_state_dict  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pipeline", choices=("align", "feat", "label"), default="align")
    ap.add_argument("--pairs", type=int, default=8)
    ap.add_argument("--points", type=int, default=2048)
    ap.add_argument("--iters", type=int, default=3, help="num_train_reg_iter (align)")
    ap.add_argument("--num-sub", type=int, default=512, help="key points per cloud (feat)")
    ap.add_argument("--thres-radius", type=float, default=0.15)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--decay-every", type=int, default=10)
    ap.add_argument("--decay-ratio", type=float, default=0.95)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = NetConfig(feat_len=3, pipeline=a.pipeline, num_sub=a.num_sub if a.pipeline == "feat" else -1)
    sd = generate_state_dict(cfg, 1, "plain" if a.pipeline == "label" else "separated")
    eng = Engine(cfg, max_points=a.points, max_pairs=a.pairs)
    eng.load_state_dict(sd)
    raws = [make_pair(a.points, 1000 + b, 3) for b in range(a.pairs)]
    src = torch.from_numpy(np.concatenate([r["points_src"] for r in raws])).to(dev)
    ref = torch.from_numpy(np.concatenate([r["points_ref"] for r in raws])).to(dev)
    gt = torch.from_numpy(np.concatenate([r["transform_gt"] for r in raws]).astype(np.float32)).to(dev)
    batch = {"points_src": src, "points_ref": ref}
    for s, pts in (("src", src), ("ref", ref)):
        batch[f"{s}_xyz"], batch[f"{s}_neigh"], batch[f"{s}_sub"], batch[f"{s}_interp"] = eng.knn_pyramid(pts)
    lr = a.lr
    if a.pipeline == "align":
        tr = RandlaTrainer(cfg, sd, "inlier_model", 6, 1, dev)
        stepper = AlignTrainStep(eng, tr, a.pairs, a.points, a.points, a.iters)
    elif a.pipeline == "label":
        tr = RandlaTrainer(cfg, sd, "feat_extractor", cfg.feat_len, cfg.num_classes, dev)
        g = torch.Generator().manual_seed(0)
        labels = {s: torch.randint(0, 20, (a.pairs, a.points), generator=g).int().to(dev) for s in ("src", "ref")}
    else:
        tr = AggregationTrainer(cfg, sd, dev)
        gt = gt.clone()
        gt[:, :, 3] += 2e-3          # rigid synthetic copies: keep CircleLoss's exact-equality pos_mask off its knife edge
    t0 = time.perf_counter()
    for step in range(1, a.steps + 1):
        if a.pipeline == "align":
            res = eng.register(src, ref, n_iter=a.iters)                        # the no_grad half with the current weights
            out = stepper.step(batch, res, gt, labels=None, lr=lr, dropout_seed=step)
            loss = out["losses"]["total"]
            eng.load_state_dict({**sd, **tr.state_dict()})                      # the updated inlier model serves the next step
        elif a.pipeline == "label":
            loss = train_step_label(tr, batch, labels["src"], labels["ref"], lr=lr, dropout_seed=step)["loss"]
        else:
            inp = feat_pipeline_inputs(eng, batch, a.num_sub)                   # frozen half (selection is loop invariant here)
            loss = train_step_feat(tr, inp, gt, a.thres_radius, 1.0, lr=lr)["loss"]
        if step % a.decay_every == 0:
            lr = max(lr * a.decay_ratio, 1e-4)
        print(f"step {step:3d}  loss {loss:.5f}  lr {lr:.2e}", flush=True)
    torch.cuda.synchronize()
    print(f"{a.steps} steps of the {a.pipeline} pipeline in {time.perf_counter() - t0:.2f} s")


if __name__ == "__main__":
    main()

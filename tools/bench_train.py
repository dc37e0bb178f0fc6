"""tools/bench_train.py [--pairs P] [--points N] [--iters I] [--steps K]: time of one `align` training step of the inlier model
(deepsir_amd.train.train_step_align: 5 training-mode forwards, loss + gradient, 5 backwards, Adam) on one GPU; the
inference half (Engine.register) is timed separately.  Prints one JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsir_amd.arch import NetConfig  # noqa: E402
from deepsir_amd.engine import Engine  # noqa: E402
from deepsir_amd.synth import make_pair  # noqa: E402
from deepsir_amd.train import AggregationTrainer, AlignTrainStep, RandlaTrainer, train_step_align_full  # noqa: E402
from deepsir_amd.weights import generate_state_dict  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--pairs", type=int, default=8)
ap.add_argument("--points", type=int, default=5000)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--full", action="store_true", help="whole network in training mode (train_step_align_full), as train.py runs it")
ap.add_argument("--eager", action="store_true", help="launch every operator from the host (no hipGraph replay)")
a = ap.parse_args()
cfg = NetConfig(feat_len=3)
sd = generate_state_dict(cfg, 3, "plain")
dev = torch.device("cuda:0")
eng = Engine(cfg, max_points=a.points, max_pairs=a.pairs)
eng.load_state_dict(sd)
raws = [make_pair(a.points, 100 + b, 3) for b in range(a.pairs)]
src = torch.from_numpy(np.concatenate([r["points_src"] for r in raws])).to(dev)
ref = torch.from_numpy(np.concatenate([r["points_ref"] for r in raws])).to(dev)
gt = torch.from_numpy(np.concatenate([r["transform_gt"] for r in raws]).astype(np.float32)).to(dev)
sx, sn, ss, si = eng.knn_pyramid(src)
batch = {"points_src": src, "points_ref": ref, "src_xyz": sx, "src_neigh": sn, "src_sub": ss, "src_interp": si}
tr = RandlaTrainer(cfg, sd, "inlier_model", 6, 1, dev)
labels = (torch.rand(a.iters, a.pairs, a.points) < 0.5).float().to(dev)
if a.full:
    fe, ag = RandlaTrainer(cfg, sd, "feat_extractor", cfg.feat_len, cfg.num_classes, dev), AggregationTrainer(cfg, sd, dev)
    rx, rn, rs, ri = eng.knn_pyramid(ref)
    batch.update({"ref_xyz": rx, "ref_neigh": rn, "ref_sub": rs, "ref_interp": ri})
stepper = AlignTrainStep(eng, tr, a.pairs, a.points, a.points, a.iters, dropout=True, use_graph=not a.eager)
t_inf, t_train, losses = [], [], []
for s in range(a.steps + 2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res = eng.register(src, ref, n_iter=a.iters)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    if a.full:
        g_ = torch.Generator(device=dev).manual_seed(s)
        keep = lambda *sh: (torch.rand(*sh, generator=g_, device=dev) >= 0.5).to(torch.uint8)
        masks = {"fe_src": keep(a.pairs, a.points, 64), "fe_ref": keep(a.pairs, a.points, 64), "inlier": keep(a.iters, a.pairs, a.points, 64)}
        out = train_step_align_full(eng, tr, fe, ag, batch, gt, a.iters, lambda idx: labels, lr=1e-3, masks=masks)
    else:
        out = stepper.step(batch, res, gt, labels=labels, lr=1e-3, dropout_seed=s)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    eng.load_state_dict({**sd, **tr.state_dict()})            # the updated inlier model serves the next step's inference
    if s >= 2:                                              # step 0 runs eagerly, step 1 captures
        t_inf.append(t1 - t0); t_train.append(t2 - t1)
    losses.append(out["losses"]["total"])
print(json.dumps({"mode": "whole network in training mode (eager)" if a.full else "eager" if a.eager else "hipGraph replay", "pairs": a.pairs, "points": a.points, "iters": a.iters, "inference_ms": round(1e3 * float(np.median(t_inf)), 2),
                  "train_step_ms": round(1e3 * float(np.median(t_train)), 2),
                  "train_pairs_per_s": round(a.pairs / float(np.median(t_train)), 2), "losses": [round(l, 5) for l in losses],
                  "peak_mem_gb": round(torch.cuda.max_memory_allocated() / 2**30, 2)}))

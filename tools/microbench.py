"""Stage-level micro-benchmarks through the C ABI (for rocprofv3 A/B runs).

    python tools/microbench.py aggregate --clouds 16 --n 5000 --reps 20
    python tools/microbench.py randla    --clouds 32 --n 5000 --reps 10
    python tools/microbench.py match     --clouds 16 --n 5000 --reps 20
    python tools/microbench.py knn       --clouds 32 --n 5000 --reps 10
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsir_amd.arch import NetConfig
from deepsir_amd.engine import Engine
from deepsir_amd.weights import generate_state_dict

ap = argparse.ArgumentParser()
ap.add_argument("what", choices=["aggregate", "randla", "inlier", "match", "match_screened", "knn", "score"])
ap.add_argument("--clouds", type=int, default=16)
ap.add_argument("--n", type=int, default=5000)
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()

cfg = NetConfig(feat_len=3)
eng = Engine(cfg, 0, max_points=a.n, max_pairs=max(1, (a.clouds + 1) // 2))
eng.load_state_dict(generate_state_dict(cfg, 0))
dev = torch.device("cuda", 0)
g = torch.Generator(device="cpu").manual_seed(0)
pts = (torch.rand(a.clouds, a.n, 3, generator=g) * 3).to(dev)
xyz, neigh, sub, interp = eng.knn_pyramid(pts)
feat = torch.randn(a.clouds, a.n, 64, generator=g).to(dev)
score = torch.rand(a.clouds, a.n, generator=g).to(dev)
desc = torch.nn.functional.normalize(torch.randn(a.clouds, a.n, 64, generator=g), dim=2).to(dev)


def run():
    if a.what == "aggregate":
        eng.aggregate(xyz[:, :a.n].contiguous(), feat, score)
    elif a.what == "randla":
        eng.randla_forward("feat_extractor", pts, xyz, neigh, sub, interp)
    elif a.what == "inlier":
        eng.randla_forward("inlier_model", torch.cat([pts, pts], 2), xyz, neigh, sub, interp)
    elif a.what == "match":
        eng.nn_match(desc, desc.flip(0))
    elif a.what == "match_screened":
        eng.nn_match_screened(desc, desc.flip(0), want_stats=False)
    elif a.what == "knn":
        eng.knn_pyramid(pts)
    elif a.what == "score":
        eng.score(feat, torch.randn(a.clouds, a.n, 19, device=dev), xyz, neigh)


run()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.reps):
    run()
torch.cuda.synchronize()
print(f"{a.what}: {(time.perf_counter() - t0) / a.reps * 1e3:.3f} ms per call (clouds={a.clouds}, n={a.n})")

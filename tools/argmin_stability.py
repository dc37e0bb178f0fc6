"""How stable is the nearest-descriptor arg-min from one registration iteration to the next?  For every iteration it >= 1:
share of rows whose arg-min is unchanged, and the rank of the previous match among the new distances (number of ref
columns at least as close): decides whether D(j, previous match) is a useful screening threshold."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsir_amd.arch import NetConfig
from deepsir_amd.engine import Engine
from deepsir_amd.synth import make_batch
from deepsir_amd.weights import generate_state_dict

P, N = int(sys.argv[1]) if len(sys.argv) > 1 else 4, int(sys.argv[2]) if len(sys.argv) > 2 else 5000
SHAPE = sys.argv[3] if len(sys.argv) > 3 else "3dmatch"
FL = 4 if SHAPE == "kitti" else 3
cfg = NetConfig(feat_len=FL)
eng = Engine(cfg, 0, max_points=N, max_pairs=P)
eng.load_state_dict(generate_state_dict(cfg, 0))
b = make_batch(N, list(range(10_000, 10_000 + P)), FL, SHAPE)
src, ref = torch.from_numpy(b["points_src"]).cuda(), torch.from_numpy(b["points_ref"]).cuda()
out = eng.register(src, ref, 5)
T, idx = out["transforms"], out["idx"]
pts = torch.cat([src, ref], 0)
xyz, neigh, sub, interp = eng.knn_pyramid(pts)
feat, logits = eng.randla_forward("feat_extractor", pts, xyz, neigh, sub, interp)
score, _ = eng.score(feat, logits, xyz, neigh)
dr = eng.aggregate(xyz[P:, :N].contiguous(), feat[P:].contiguous(), score[P:].contiguous())
for it in range(5):
    xs = src[:, :, :3]
    if it > 0:
        R, t = T[:, it - 1, :, :3], T[:, it - 1, :, 3]
        xs = xs @ R.transpose(1, 2) + t[:, None]
    ds = eng.aggregate(xs.contiguous(), feat[:P].contiguous(), score[:P].contiguous())
    mine = eng.nn_match(ds, dr)
    same_engine = (mine == idx[it]).float().mean().item()
    if it == 0:
        print(f"it 0: recomputed arg-min equals the registration's on {100 * same_engine:.2f} % of the rows")
        continue
    D = (ds * ds).sum(2, keepdim=True) + (dr * dr).sum(2)[:, None] - 2 * ds @ dr.transpose(1, 2)
    prev = idx[it - 1].long()
    dprev = torch.gather(D, 2, prev[..., None])
    rank = (D <= dprev).sum(2).float()
    unchanged = (idx[it] == idx[it - 1]).float().mean().item()
    q = [(rank <= k).float().mean().item() for k in (1, 2, 4, 16, 64)]
    print(f"it {it}: recomputed == registration {100 * same_engine:.2f} %; arg-min unchanged {100 * unchanged:.1f} %; "
          f"rank of the previous match <= 1/2/4/16/64: " + " ".join(f"{100 * v:.1f}" for v in q) + f" %; mean rank {rank.mean().item():.1f}")

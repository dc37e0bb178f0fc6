"""Candidate statistics of the screened arg-min (csrc/nn_screen.hip) on descriptors the engine itself produces."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsir_amd.arch import NetConfig
from deepsir_amd.engine import Engine
from deepsir_amd.synth import make_batch
from deepsir_amd.weights import generate_state_dict

P, N = int(sys.argv[1]) if len(sys.argv) > 1 else 8, int(sys.argv[2]) if len(sys.argv) > 2 else 5000
SHAPE = sys.argv[3] if len(sys.argv) > 3 else "3dmatch"
FL = 4 if SHAPE == "kitti" else 3
cfg = NetConfig(feat_len=FL)
eng = Engine(cfg, 0, max_points=N, max_pairs=P)
eng.load_state_dict(generate_state_dict(cfg, 0))
b = make_batch(N, list(range(10_000, 10_000 + P)), FL, SHAPE)
pts = torch.cat([torch.from_numpy(b["points_src"]), torch.from_numpy(b["points_ref"])], 0).cuda()
xyz, neigh, sub, interp = eng.knn_pyramid(pts)
feat, logits = eng.randla_forward("feat_extractor", pts, xyz, neigh, sub, interp)
score, _ = eng.score(feat, logits, xyz, neigh)
desc = eng.aggregate(xyz[:, :N].contiguous(), feat, score)
ds, dr = desc[:P].contiguous(), desc[P:].contiguous()
idx, (ncand, nexh) = eng.nn_match_screened(ds, dr)
exact = eng.nn_match(ds, dr)
print(f"{SHAPE} pairs {P} x {N}: {ncand / (P * N):.3f} candidates per row, {nexh} of {P * N} rows exhaustive, "
      f"equal to the exhaustive kernel: {bool(torch.equal(idx, exact))}")

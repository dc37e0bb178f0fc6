"""Register P synthetic pairs of N points and dump (transforms, idx, logits) to an .npz: used by
tests/test_gpu_parity.py::test_register_screened_equals_exhaustive and tests/test_gpu_large_configs.py to compare two
processes that differ only in an environment switch (the switches are read once per process).

    register_dump.py OUT PAIRS POINTS ITERS [FEAT_LEN [SHAPE [PARTIAL]]]      SHAPE: 3dmatch | kitti, PARTIAL: 0 | 1"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsir_amd.arch import NetConfig
from deepsir_amd.engine import Engine
from deepsir_amd.synth import make_batch
from deepsir_amd.weights import generate_state_dict, to_torch_state_dict

out, P, N, iters = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
feat_len = int(sys.argv[5]) if len(sys.argv) > 5 else 3
shape = sys.argv[6] if len(sys.argv) > 6 else "3dmatch"
partial = bool(int(sys.argv[7])) if len(sys.argv) > 7 else False
cfg = NetConfig(feat_len=feat_len)
eng = Engine(cfg, 0, max_points=max(N, 1024), max_pairs=P)
eng.load_state_dict(to_torch_state_dict(generate_state_dict(cfg, 3)))
eng.set_prune_thresholds(8192, 1)     # the pruned search for every launch on clouds of 8192 points and more, however few rows it has
b = make_batch(N, list(range(500, 500 + P)), feat_len, shape, partial)
o = eng.register(torch.from_numpy(b["points_src"]).cuda(), torch.from_numpy(b["points_ref"]).cuda(), iters)
pyr = eng.knn_pyramid(torch.from_numpy(b["points_src"]).cuda())      # the src clouds' pyramid as its own operator (same launches as inside register at P clouds)
st = eng.screen_stats()
kept, total = eng.prune_stats()
np.savez(out, tiles_visited=np.int64(kept), tiles_unpruned=np.int64(total),
         screened_searches=np.int64(st["screened_searches"]), rows_undecided=np.int64(st["rows_undecided"]),
         pairs_exhaustive=np.int64(st["pairs_exhaustive"]), neigh=pyr[1].cpu().numpy(), sub=pyr[2].cpu().numpy(), interp=pyr[3].cpu().numpy(),
         **{k: o[k].cpu().numpy() for k in ("transforms", "idx", "logits")})
eng.close()

"""Register P synthetic pairs of N points and dump (transforms, idx, logits) to an .npz: used by
tests/test_gpu_parity.py::test_register_screened_equals_exhaustive to compare two processes that differ only in an
environment switch (the switches are read once per process)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsir_amd.arch import NetConfig
from deepsir_amd.engine import Engine
from deepsir_amd.synth import make_batch
from deepsir_amd.weights import generate_state_dict, to_torch_state_dict

out, P, N, iters = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
cfg = NetConfig()
eng = Engine(cfg, 0, max_points=max(N, 1024), max_pairs=P)
eng.load_state_dict(to_torch_state_dict(generate_state_dict(cfg, 3)))
b = make_batch(N, list(range(500, 500 + P)), 3)
o = eng.register(torch.from_numpy(b["points_src"]).cuda(), torch.from_numpy(b["points_ref"]).cuda(), iters)
np.savez(out, **{k: o[k].cpu().numpy() for k in ("transforms", "idx", "logits")})
eng.close()

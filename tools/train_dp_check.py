"""tools/train_dp_check.py (run as N ranks: python -m torch.distributed.run --nproc-per-node 2 ... tools/train_dp_check.py):
data-parallel training step of the inlier model: every rank registers and trains on its own pairs, the flat gradient buffer
is all-reduced once, the optimiser step is identical on all ranks.  Prints one JSON line from rank 0:
  grad_is_mean: the reduced gradient equals the mean of the ranks' local gradients; params_identical: after two steps every
  rank holds the same weights."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsir_amd.arch import NetConfig  # noqa: E402
from deepsir_amd.engine import Engine  # noqa: E402
from deepsir_amd.synth import make_pair  # noqa: E402
from deepsir_amd.train import RandlaTrainer, train_step_align  # noqa: E402
from deepsir_amd.weights import generate_state_dict  # noqa: E402

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
torch.cuda.set_device(local)
backend = os.environ.get("DSIR_BENCH_BACKEND", "nccl")      # gloo: several ranks rehearsing on ONE GPU (RCCL refuses that)
dist.init_process_group(backend, rank=rank, world_size=world)
dev = torch.device("cuda", local)
cfg = NetConfig(feat_len=3)
sd = generate_state_dict(cfg, 3, "plain")
n, P, n_iter = 1024, 2, 2
eng = Engine(cfg, local, max_points=n, max_pairs=P)
eng.load_state_dict(sd)
raws = [make_pair(n, 700 + rank * P + b, 3) for b in range(P)]          # every rank its own pairs
src = torch.from_numpy(np.concatenate([r["points_src"] for r in raws])).to(dev)
ref = torch.from_numpy(np.concatenate([r["points_ref"] for r in raws])).to(dev)
gt = torch.from_numpy(np.concatenate([r["transform_gt"] for r in raws]).astype(np.float32)).to(dev)
sx, sn, ss, si = eng.knn_pyramid(src)
batch = {"points_src": src, "points_ref": ref, "src_xyz": sx, "src_neigh": sn, "src_sub": ss, "src_interp": si}
res = eng.register(src, ref, n_iter=n_iter)
tr = RandlaTrainer(cfg, sd, "inlier_model", 6, 1, dev)
train_step_align(eng, tr, batch, res, gt, apply=False)                   # local gradient
local_g = tr.flat_g.clone()
gathered = [torch.empty_like(local_g) for _ in range(world)]
dist.all_gather(gathered, local_g)
train_step_align(eng, tr, batch, res, gt, apply=False, dist=dist)        # the same step, gradients reduced
mean = torch.stack(gathered).mean(0)
grad_ok = bool(torch.allclose(tr.flat_g, mean, rtol=1e-4, atol=1e-7 * float(mean.abs().max())))
for _ in range(2):
    train_step_align(eng, tr, batch, res, gt, lr=1e-3, dist=dist)
ps = [torch.empty_like(tr.flat_p) for _ in range(world)]
dist.all_gather(ps, tr.flat_p)
same = all(bool(torch.equal(ps[0], p)) for p in ps[1:])
if rank == 0:
    print(json.dumps({"world_size": world, "backend": backend, "grad_is_mean": grad_ok, "params_identical": same}))
dist.barrier()
dist.destroy_process_group()

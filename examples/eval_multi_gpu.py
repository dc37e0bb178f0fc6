"""Sharded evaluation over the GPUs of a node: the multi-GPU analogue of the reference's per-pair loop
(test.py:386-450).  One process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/eval_multi_gpu.py \\
        --pairs 64 --points 5000 --out /tmp/eval.npz

Every rank builds the same drop-in `Network`, registers its contiguous block of the (here: synthetic) pair list and the
results meet in one all_gather (RCCL on GPUs; `--backend gloo` only to rehearse several ranks on one GPU)."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsir_amd.arch import NetConfig  # noqa: E402
from deepsir_amd.harness import inference_align, summarize  # noqa: E402
from deepsir_amd.model import Network  # noqa: E402
from deepsir_amd.synth import make_pair  # noqa: E402
from deepsir_amd.weights import generate_state_dict, to_torch_state_dict  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=16)
    ap.add_argument("--points", type=int, default=2048)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import torch.distributed as dist  # noqa: F811
        kw = {"device_id": torch.device("cuda", local)} if a.backend == "nccl" else {}
        dist.init_process_group(a.backend, rank=rank, world_size=world, **kw)
    args = argparse.Namespace(pipeline="align", num_sub=-1, num_knn=16, out_feat_dim=64, clip_weight_thresh=0.0, feat_len=3,
                              d_out=[16, 64, 128, 256], num_points=a.points, sub_sampling_ratio=[4, 4, 4, 4], use_ppf=False)
    net = Network(args)
    net.load_state_dict(to_torch_state_dict(generate_state_dict(NetConfig(), 0)))   # a real run: torch.load(ckpt)['state_dict']
    net = net.cuda().eval()
    pairs = [make_pair(a.points, 1000 + i, 3) for i in range(a.pairs)]
    pred, stats = inference_align(pairs, net, a.iters, batch=a.batch, dist=dist)
    if rank == 0:
        print("pairs", len(pred), "summary", summarize(stats))
        if a.out:
            np.savez(a.out, pred=pred, stats=stats)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

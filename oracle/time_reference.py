"""the REFERENCE itself, timed in the build container (TEST / MEASUREMENT INFRASTRUCTURE ONLY;
/root/reference does not exist on the GPU box, so this never runs there).

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/oracle/time_reference.py [--sizes 2048 5000 16384] [--runs 5]

`network.model.Network.forward` imported from /root/reference, weights = the build's seeded generated state-dict, inputs =
the build's synthetic pairs with the oracle's exact KNN pyramid (the reference's torch_points_kernels is not installable
here), `torch.set_num_threads(nproc)`; median of >= 5 timed runs after one warm-up, in both windows: model-only (the
reference's own, test.py:399-402) and model + the build's CPU KNN.  The oracle (oracle/network.py) is timed beside it on
the same inputs - it is what travels to the GPU box as `cpu_baseline`.  Prints a markdown table + one JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from deepsir_amd.arch import NetConfig  # noqa: E402
from deepsir_amd.synth import make_pair  # noqa: E402
from deepsir_amd.weights import generate_state_dict, to_torch_state_dict  # noqa: E402
from oracle.knn import add_pyramids  # noqa: E402
from oracle.network import OracleNet, to_torch  # noqa: E402


def median_time(fn, runs):
    fn()   # warm-up
    ts = []
    for _ in range(runs):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--sizes", type=int, nargs="+", default=[2048, 5000, 16384])
    ap.add_argument("--runs", type=int, default=5)
    a = ap.parse_args()
    warnings.filterwarnings("ignore")
    nproc = len(os.sched_getaffinity(0))
    torch.set_num_threads(nproc)
    sys.path.insert(0, a.ref)
    import arguments  # type: ignore
    import network.model as ref_model  # type: ignore
    rows = []
    for n in a.sizes:
        feat_len = 4 if n >= 16384 else 3                      # the 16k case is KITTI-shaped (xyz + reflectance)
        shape = "kitti" if n >= 16384 else "3dmatch"
        cfg = NetConfig(feat_len=feat_len)
        args = arguments.eval_arguments().parse_args([])
        args.pipeline, args.feat_len, args.num_sub = "align", feat_len, -1
        net = ref_model.Network(args).eval()
        sd = generate_state_dict(cfg, 0)
        net.load_state_dict(to_torch_state_dict(sd), strict=True)
        raw = make_pair(n, 1001, feat_len, shape)
        t0 = time.perf_counter()
        data_np = add_pyramids(raw, cfg.num_knn, cfg.sub_sampling_ratio)
        knn_s = time.perf_counter() - t0
        data = to_torch(data_np)
        runs = a.runs if n <= 5000 else max(3, a.runs - 2)
        with torch.no_grad():
            ref_s = median_time(lambda: net(data, (5, True)), runs)
        orc = OracleNet(cfg, sd)
        orc_s = median_time(lambda: orc.register(data, 5), runs)
        rows.append({"n": n, "feat_len": feat_len, "runs": runs, "reference_s": ref_s, "oracle_s": orc_s, "knn_s": knn_s})
        print(f"N={n}: reference {ref_s:.3f} s/pair, oracle {orc_s:.3f} s/pair, CPU KNN pyramid {knn_s:.3f} s", flush=True)
    print(f"\nnproc = {nproc}, torch.get_num_threads() = {torch.get_num_threads()}, torch {torch.__version__}\n")
    print("| N pts/cloud | feat_len | reference `Network.forward` s/pair (model-only) | pairs/s | + build's CPU KNN: s/pair | pairs/s | oracle s/pair (model-only) | pairs/s |")
    print("|---|---|---|---|---|---|---|---|")
    for r in rows:
        print(f"| {r['n']} | {r['feat_len']} | {r['reference_s']:.3f} | {1 / r['reference_s']:.2f} | {r['reference_s'] + r['knn_s']:.3f} | "
              f"{1 / (r['reference_s'] + r['knn_s']):.2f} | {r['oracle_s']:.3f} | {1 / r['oracle_s']:.2f} |")
    print(json.dumps({"nproc": nproc, "threads": torch.get_num_threads(), "rows": rows}))


if __name__ == "__main__":
    main()

"""How selective would a COARSER screening of the nearest-descriptor search be?  (round-3 experiment)

Registers a few pairs, takes the engine's own descriptors of every iteration (aux output desc_src / desc_ref) and counts,
per src row, the ref columns whose exact distance lies within delta of the row minimum - the survivors a screening with a
bound of total width ~delta must hand to the exact fp32 decision.  delta = 1.2e-4 is today's three-product bound (2 d at
unit norm), 2e-3 / 4e-3 / 8e-3 bracket a single-product (ah.bh only) bound of width 2^-9 .. 2^-8.

Measurement aid: torch on the GPU computes the distance matrices here; nothing of this is on the product path.

    python tools/survivor_stats.py [POINTS [PAIRS [WEIGHTS [FEAT_LEN SHAPE PARTIAL]]]]"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsir_amd.arch import NetConfig
from deepsir_amd.engine import Engine
from deepsir_amd.synth import make_batch
from deepsir_amd.weights import generate_state_dict

N = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
P = int(sys.argv[2]) if len(sys.argv) > 2 else 4
variant = sys.argv[3] if len(sys.argv) > 3 else "plain"
feat_len = int(sys.argv[4]) if len(sys.argv) > 4 else 3
shape = sys.argv[5] if len(sys.argv) > 5 else "3dmatch"
partial = bool(int(sys.argv[6])) if len(sys.argv) > 6 else False
cfg = NetConfig(feat_len=feat_len)
eng = Engine(cfg, 0, max_points=N, max_pairs=P)
eng.load_state_dict(generate_state_dict(cfg, 0, variant))
b = make_batch(N, list(range(10_000, 10_000 + P)), feat_len, shape, partial)
o = eng.register(torch.from_numpy(b["points_src"]).cuda(), torch.from_numpy(b["points_ref"]).cuda(), 5, want_desc=True)
deltas = [1.2e-4, 5e-4, 1e-3, 2e-3, 4e-3, 8e-3, 1.6e-2]
print(f"points {N} pairs {P} weights {variant} shape {shape} partial {partial}")
for it in range(5):
    cnt = np.zeros((len(deltas), 0))
    per = []
    gaps = []
    for p in range(P):
        a, r = o["desc_src"][it, p], o["desc_ref"][p]
        sb = (r * r).sum(1)
        rows = []
        for c0 in range(0, N, 4096):
            x = a[c0:c0 + 4096]
            d = (x * x).sum(1)[:, None] + sb[None, :] - 2.0 * (x @ r.t())
            top2 = torch.topk(d, 2, dim=1, largest=False)[0]
            gaps.append((top2[:, 1] - top2[:, 0]).cpu().numpy())
            dmin = top2[:, :1]
            rows.append(torch.stack([(d <= dmin + dl).sum(1) for dl in deltas], 0).cpu().numpy())
        per.append(np.concatenate(rows, 1))
    c = np.concatenate(per, 1).astype(np.float64)
    g = np.concatenate(gaps)
    line = "  ".join(f"d={dl:.1e}: mean {c[i].mean():7.2f} p50 {np.median(c[i]):5.0f} p99 {np.percentile(c[i], 99):6.0f} max {c[i].max():6.0f} >16: {100 * (c[i] > 16).mean():5.1f}%"
                     for i, dl in enumerate(deltas))
    print(f"iter {it}: top-2 gap median {np.median(g):.2e} p10 {np.percentile(g, 10):.2e} | {line}")
eng.close()

"""Probe: K single-pair registrations in flight (K engines, each one captured hipGraph on its own stream).
python tools/serve_probe.py [--points 5000] [--reps 64]   -> pairs/s per K, driven by one host thread and by K threads."""
import argparse
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsir_amd.arch import NetConfig
from deepsir_amd.engine import Engine
from deepsir_amd.synth import make_batch
from deepsir_amd.weights import generate_state_dict

ap = argparse.ArgumentParser()
ap.add_argument("--points", type=int, default=5000)
ap.add_argument("--reps", type=int, default=64)
ap.add_argument("--kmax", type=int, default=8)
a = ap.parse_args()
cfg = NetConfig(feat_len=3)
sd = generate_state_dict(cfg, 0)
dev = torch.device("cuda", 0)
b = make_batch(a.points, [10_000 + i for i in range(a.kmax)], 3, "3dmatch", False)
src = torch.from_numpy(b["points_src"]).to(dev)
ref = torch.from_numpy(b["points_ref"]).to(dev)
engs = []
for k in range(a.kmax):
    e = Engine(cfg, 0, max_points=a.points, max_pairs=1)
    e.load_state_dict(sd)
    e.enable_graph(True)
    engs.append(e)
outs = [e.register(src[k:k + 1], ref[k:k + 1], 5, want_aux=False) for k, e in enumerate(engs)]
ins = [(src[k:k + 1].contiguous(), ref[k:k + 1].contiguous()) for k in range(a.kmax)]


def drive(k, n):
    e = engs[k]
    for _ in range(n):
        e.register(ins[k][0], ins[k][1], 5, want_aux=False, sync=False, out={"transforms": outs[k]["transforms"]})


for K in (1, 2, 4, 8):
    if K > a.kmax:
        break
    for k in range(K):
        drive(k, 2); engs[k].sync()
    # one host thread, round robin
    t0 = time.perf_counter()
    for _ in range(a.reps):
        for k in range(K):
            drive(k, 1)
    t_sub = time.perf_counter() - t0
    for k in range(K):
        engs[k].sync()
    t1 = time.perf_counter() - t0
    # K host threads
    th = [threading.Thread(target=drive, args=(k, a.reps)) for k in range(K)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    for k in range(K):
        engs[k].sync()
    t2 = time.perf_counter() - t0
    n = K * a.reps
    print(f"K={K}: one thread {n / t1:.1f} pairs/s (submit {t_sub / n * 1e3:.3f} ms/pair host), {K} threads {n / t2:.1f} pairs/s", flush=True)

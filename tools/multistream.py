"""Experiment: S engines (own stream + workspace each) registering P pairs each, concurrently."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsir_amd.arch import NetConfig
from deepsir_amd.engine import Engine
from deepsir_amd.synth import make_batch
from deepsir_amd.weights import generate_state_dict
ap = argparse.ArgumentParser()
ap.add_argument("--streams", type=int, default=2)
ap.add_argument("--pairs", type=int, default=32)
ap.add_argument("--steps", type=int, default=10)
a = ap.parse_args()
cfg = NetConfig(feat_len=3); sd = generate_state_dict(cfg, 0)
dev = torch.device("cuda", 0)
engs, ins, outs = [], [], []
for s in range(a.streams):
    e = Engine(cfg, 0, max_points=5000, max_pairs=a.pairs); e.load_state_dict(sd); engs.append(e)
    b = make_batch(5000, [100 * s + i for i in range(a.pairs)], 3)
    ins.append((torch.from_numpy(b["points_src"]).to(dev), torch.from_numpy(b["points_ref"]).to(dev)))
    outs.append(e.register(*ins[-1], 5, want_aux=False))
def step():
    for e, (s, r), o in zip(engs, ins, outs):
        e.register(s, r, 5, want_aux=False, sync=False, out={"transforms": o["transforms"]})
for _ in range(2): step()
for e in engs: e.sync()
t0 = time.perf_counter()
for _ in range(a.steps): step()
for e in engs: e.sync()
dt = time.perf_counter() - t0
print(f"streams={a.streams} pairs/stream={a.pairs}: {a.streams * a.pairs * a.steps / dt:.1f} pairs/s")

"""Per-phase timing inside the deep-level walker (csrc/walk.hip) for one pair in flight.

    DSIR_TUNING=1 DSIR_WALK_TRACE=1 [DSIR_WALK_WPC=n] python3 tools/walk_trace.py [points] [pairs] [out.txt]

Replays a captured registration a few times, then prints for every program (pass) of the last replay and every phase: tiles, the
time from the previous phase's last publish to this phase's first tile past its wait (the hand-off), the span of its tile bodies and
the publish tail - all from device-clock stamps of cloud 0 (include/dsir.h, dsir_walk_trace)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import numpy as np
import torch
import deepsir_amd  # noqa: F401
from deepsir_amd.arch import NetConfig
from deepsir_amd.engine import Engine
from deepsir_amd.synth import make_batch
from deepsir_amd.weights import generate_state_dict

points = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 1
out_path = sys.argv[3] if len(sys.argv) > 3 else None
cfg = NetConfig(feat_len=3)
eng = Engine(cfg, 0, max_points=points, max_pairs=pairs)
eng.load_state_dict(generate_state_dict(cfg, 0))
b = make_batch(points, [10_000 + i for i in range(pairs)], 3)
src, ref = torch.from_numpy(b["points_src"]).cuda(), torch.from_numpy(b["points_ref"]).cuda()
out = {"transforms": torch.empty((pairs, 5, 3, 4), device="cuda")}
eng.enable_graph(True)
for _ in range(4):
    eng.register(src, ref, 5, want_aux=False, out=out)
buf = (C.c_int64 * (12 * 32 * 4))()
khz = C.c_int64()
eng._call(eng.lib.dsir_walk_trace(eng.h, 1, None, C.byref(khz)))
eng.register(src, ref, 5, want_aux=False, out=out)
eng._call(eng.lib.dsir_walk_trace(eng.h, 0, buf, C.byref(khz)))
t = np.frombuffer(buf, dtype=np.int64).reshape(12, 32, 4).astype(np.float64) / (khz.value / 1e3)     # microseconds
lines = [f"walker phases, {pairs} pair(s) x {points} points, wpc {os.environ.get('DSIR_WALK_WPC', 'default')}; us from the program's first stamp"]
for s in range(12):
    ph = [p for p in range(32) if t[s, p, 3] > 0]
    if not ph:
        continue
    t0 = t[s, ph[0], 0]
    lines.append(f"program {s}: {len(ph)} phases, span {t[s, ph[-1], 3] - t0:.1f} us")
    prev = t0
    for p in ph:
        a, w, e, d = t[s, p] - t0
        lines.append(f"  phase {p:2d}: first pick {a:7.1f}  past wait {w:7.1f} (hand-off {w - (prev - t0):5.1f})  bodies end {e:7.1f} (span {e - w:5.1f})  published {d:7.1f} (+{d - e:4.1f})")
        prev = t[s, p, 3]
txt = "\n".join(lines)
print(txt)
if out_path:
    open(out_path, "w").write(txt + "\n")

"""Do two engines overlap better when their registrations are half a step out of phase?  python tools/stagger_probe.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepsir_amd  # noqa: F401,E402
import torch  # noqa: E402
from deepsir_amd.arch import NetConfig  # noqa: E402
from deepsir_amd.engine import EnginePool  # noqa: E402
from deepsir_amd.synth import make_batch  # noqa: E402
from deepsir_amd.weights import generate_state_dict  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
cfg = NetConfig()
sd = generate_state_dict(cfg, 0)
P, N = 256, 5000
b = make_batch(N, [10_000 + i for i in range(P)], 3)
src, ref = torch.from_numpy(b["points_src"]).cuda(), torch.from_numpy(b["points_ref"]).cuda()
pool = EnginePool(cfg, 0, N, P, 2)
pool.load_state_dict(sd)
out = torch.empty((P, 5, 3, 4), device="cuda")
e0, e1 = pool.engines
h = P // 2


def run(offset_pairs):
    torch.cuda.synchronize()
    t = time.perf_counter()
    done = 0
    if offset_pairs:
        e1.register(src[h:h + offset_pairs], ref[h:h + offset_pairs], 5, want_aux=False, sync=False, out={"transforms": out[h:h + offset_pairs]})
        done += offset_pairs
    for _ in range(steps):
        e0.register(src[:h], ref[:h], 5, want_aux=False, sync=False, out={"transforms": out[:h]})
        e1.register(src[h:], ref[h:], 5, want_aux=False, sync=False, out={"transforms": out[h:]})
        done += P
    pool.sync()
    torch.cuda.synchronize()
    return done / (time.perf_counter() - t)


run(0)
for off in (0, 32, 64, 96, 0, 64):
    print(f"offset {off:3d} pairs on engine 1: {run(off):8.1f} pairs/s", flush=True)

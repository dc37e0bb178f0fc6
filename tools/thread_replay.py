"""Do graph replays issued from SEPARATE host threads (one per engine) overlap?  E engines x batch b, R replays each.
    [GPU_MAX_HW_QUEUES=8] python3 tools/thread_replay.py E b [threads=1|0]"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
import deepsir_amd  # noqa: F401
from deepsir_amd.arch import NetConfig
from deepsir_amd.engine import Engine
from deepsir_amd.synth import make_batch
from deepsir_amd.weights import generate_state_dict

E, b = int(sys.argv[1]), int(sys.argv[2])
threaded = bool(int(sys.argv[3])) if len(sys.argv) > 3 else True
N, R = 5000, 30
cfg = NetConfig(feat_len=3)
sd = generate_state_dict(cfg, 0)
bt = make_batch(N, list(range(10_000, 10_000 + b)), 3)
src, ref = torch.from_numpy(bt["points_src"]).cuda(), torch.from_numpy(bt["points_ref"]).cuda()
engs, outs, streams = [], [], []
for e in range(E):
    st = torch.cuda.Stream()
    eng = Engine(cfg, 0, max_points=N, max_pairs=b)
    eng.load_state_dict(sd)
    with torch.cuda.stream(st):
        eng.use_torch_stream(True)
    eng.enable_graph(True)
    if os.environ.get("WALK"):
        eng.enable_walk(True)
    out = {"transforms": torch.empty((b, 5, 3, 4), device="cuda")}
    with torch.cuda.stream(st):
        for _ in range(3):
            eng.register(src, ref, 5, want_aux=False, sync=False, out=out)
    st.synchronize()
    engs.append(eng); outs.append(out); streams.append(st)

def work(e, reps):
    with torch.cuda.stream(streams[e]):
        for _ in range(reps):
            engs[e].register(src, ref, 5, want_aux=False, sync=False, out=outs[e])
    streams[e].synchronize()

torch.cuda.synchronize()
t = time.perf_counter()
if threaded:
    th = [threading.Thread(target=work, args=(e, R)) for e in range(E)]
    for x in th: x.start()
    for x in th: x.join()
else:
    for _ in range(R):
        for e in range(E):
            with torch.cuda.stream(streams[e]):
                engs[e].register(src, ref, 5, want_aux=False, sync=False, out=outs[e])
    t_enq = time.perf_counter() - t
    torch.cuda.synchronize()
    print(f"  host time in the {E * R} register() calls (graph launches): {t_enq / (E * R) * 1e3:.3f} ms each; {engs[0].graph_stats()}")
t = time.perf_counter() - t
same = all(torch.equal(outs[0]["transforms"], o["transforms"]) for o in outs)
print(f"{'walker ' if os.environ.get('WALK') else ''}HWQ {os.environ.get('GPU_MAX_HW_QUEUES', 'dflt')} engines {E} batch {b} {'threads' if threaded else 'one thread'}: {E * R * b / t:8.1f} pairs/s  "
      f"({t / R * 1e3:.2f} ms per round of {E} x {b})  same bits across engines: {same}", flush=True)
for eng in engs:
    eng.close()

"""K = 8 single-pair requests in flight (deepsir_amd/serve.py) against the number of engines they are spread over.
    [GPU_MAX_HW_QUEUES=8] python3 tools/k8_sweep.py [K]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
import torch
import deepsir_amd  # noqa: F401
from deepsir_amd.arch import NetConfig
from deepsir_amd.serve import PairServer
from deepsir_amd.synth import make_batch
from deepsir_amd.weights import generate_state_dict

K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = 5000
cfg = NetConfig(feat_len=3)
sd = generate_state_dict(cfg, 0)
b = make_batch(N, list(range(10_000, 10_016)), 3)
src, ref = torch.from_numpy(b["points_src"]).cuda(), torch.from_numpy(b["points_ref"]).cuda()
L = src.shape[0]
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"))
ref_out = None
for E in (1, 2, 4, 8):
    if E > K:
        continue
    srv = PairServer(cfg, sd, 0, max_points=N, max_in_flight=K, engines=E, n_iter=5, want_aux=False)
    nreq = 256
    reqs = [(src[i % L], ref[i % L]) for i in range(nreq)]
    srv.run_closed_loop(reqs[: 2 * K], K)
    torch.cuda.synchronize()
    t = time.perf_counter()
    res = srv.run_closed_loop(reqs, K)
    torch.cuda.synchronize()
    t = time.perf_counter() - t
    outs = torch.stack([r["transforms"] for r in res[:L]])
    same = True if ref_out is None else bool(torch.equal(outs, ref_out))
    ref_out = outs if ref_out is None else ref_out
    print(f"K {K} engines {E} pairs/batch {srv.max_batch}: {nreq / t:8.1f} pairs/s  batches {srv.batches_dispatched}  same bits as engines=1: {same}", flush=True)
    srv.close()

"""Closed-loop serving rate: python tools/serve_bench.py [K] [engines] [requests] [points]  -> pairs/s with K single-pair requests in flight."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import deepsir_amd  # noqa: F401,E402
import torch  # noqa: E402
from deepsir_amd.arch import NetConfig  # noqa: E402
from deepsir_amd.serve import PairServer  # noqa: E402
from deepsir_amd.synth import make_batch  # noqa: E402
from deepsir_amd.weights import generate_state_dict  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
E = int(sys.argv[2]) if len(sys.argv) > 2 else 2
R = int(sys.argv[3]) if len(sys.argv) > 3 else 96
N = int(sys.argv[4]) if len(sys.argv) > 4 else 5000
cfg = NetConfig()
sd = generate_state_dict(cfg, 0)
b = make_batch(N, [10_000 + i for i in range(32)], 3)
src, ref = torch.from_numpy(b["points_src"]).cuda(), torch.from_numpy(b["points_ref"]).cuda()
srv = PairServer(cfg, sd, 0, max_points=N, max_in_flight=K, engines=E, n_iter=5, want_aux=False)
reqs = [(src[i % 32], ref[i % 32]) for i in range(R)]
srv.run_closed_loop(reqs[: 2 * K], K)
rates = []
for _ in range(4):
    torch.cuda.synchronize()
    t = time.perf_counter()
    srv.run_closed_loop(reqs, K)
    torch.cuda.synchronize()
    rates.append(R / (time.perf_counter() - t))
print(f"K={K} engines={E} batch={srv.max_batch} screen_min={os.environ.get('DSIR_SCREEN_MIN_WORK', '-')}: " + " ".join(f"{r:.0f}" for r in rates) + " pairs/s")
srv.close()

import sys, numpy as np, torch
sys.path.insert(0, '.')
from deepsir_amd.arch import NetConfig
from deepsir_amd.weights import generate_state_dict, to_torch_state_dict
from deepsir_amd.engine import Engine
from deepsir_amd.synth import make_batch
outp = sys.argv[1]
pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 2      # more pairs: launch-size dependent choices (grids, kernels) come into play
res = {}
for fl, n in ((3, 5000), (4, 2048), (3, 1500)):
    cfg = NetConfig(feat_len=fl)
    sd = to_torch_state_dict(generate_state_dict(cfg, 1))
    eng = Engine(cfg, max_points=max(n, 1024), max_pairs=pairs); eng.load_state_dict(sd)
    b = make_batch(n, [7 + i for i in range(pairs)], fl)
    o = eng.register(torch.from_numpy(b["points_src"]).cuda(), torch.from_numpy(b["points_ref"]).cuda(), 3)
    for k in ("transforms", "idx", "logits"):
        res[f"{k}_{fl}_{n}"] = o[k].cpu().numpy()
np.savez(outp, **res)

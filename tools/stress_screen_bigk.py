"""tools/stress_screen_bigk.py: the screened arg-min against the exhaustive kernel on LONG ref ranges (the launches whose ref
range is split 2 - 8 ways): any differing index is a bug."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsir_amd.arch import NetConfig
from deepsir_amd.engine import Engine
rng = np.random.Generator(np.random.Philox(key=5))
eng = Engine(NetConfig(), 0, max_points=70000, max_pairs=2)
bad = 0
for (P, J, K, regime) in [(2, 3000, 20000, "random"), (1, 9000, 40000, "near"), (2, 1500, 65536, "cluster"), (1, 16384, 16384, "near"),
                          (1, 700, 33000, "dup")]:
    a = rng.standard_normal((P, J, 64)).astype(np.float32)
    b = rng.standard_normal((P, K, 64)).astype(np.float32)
    m = min(J, K)
    if regime == "near":
        b[:, :m] = a[:, :m] + rng.standard_normal((P, m, 64)).astype(np.float32) * 1e-3
    elif regime == "cluster":
        cen = rng.standard_normal((P, 6, 64)).astype(np.float32)
        b = np.take_along_axis(cen, rng.integers(0, 6, (P, K, 1)).repeat(64, 2), 1) + 1e-2 * rng.standard_normal((P, K, 64)).astype(np.float32)
        a = np.take_along_axis(cen, rng.integers(0, 6, (P, J, 1)).repeat(64, 2), 1) + 1e-2 * rng.standard_normal((P, J, 64)).astype(np.float32)
    elif regime == "dup":
        b[:, K // 2:K // 2 + m // 2] = a[:, :m // 2]
        b[:, :m // 2] = a[:, :m // 2]
    a /= np.linalg.norm(a, axis=2, keepdims=True); b /= np.linalg.norm(b, axis=2, keepdims=True)
    ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    ex = eng.nn_match(ta, tb).cpu().numpy()
    sc, (ncand, nexh) = eng.nn_match_screened(ta, tb)
    d = int((sc.cpu().numpy() != ex).sum())
    bad += d
    print(f"P {P} J {J} K {K} {regime}: differing {d}, entries {ncand}, rows to the exhaustive kernel {nexh}")
print("mismatching total", bad)
sys.exit(1 if bad else 0)

"""Randomised stress of the pruned nearest-descriptor search (csrc/nn_prune.hip) against the exhaustive exact-fp32 arg-min:
whole registrations (dsir_register, 3 - 5 iterations) on clouds of 8192 ... 24000 points with ragged src / ref sizes, KITTI- and
3DMatch-shaped clouds, full and 50 % overlap, plain / separated / clustered descriptor weights, duplicated ref points; any
difference in idx / logits / transforms is a bug (pruning only removes products that cannot hold a row's minimum).
    python tools/stress_prune.py [CASES [SEED]]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsir_amd.arch import NetConfig
from deepsir_amd.engine import Engine
from deepsir_amd.synth import make_batch
from deepsir_amd.weights import generate_state_dict

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 12
rng = np.random.Generator(np.random.Philox(key=int(sys.argv[2]) if len(sys.argv) > 2 else 5))
NMAX, PMAX = 24000, 4
engines = {}
bad = 0
t0 = time.time()
for c in range(cases):
    feat_len = int(rng.choice([3, 4]))
    variant = str(rng.choice(["plain", "separated", "clustered:0.03", "clustered:0.01"]))
    key = (feat_len, variant)
    if key not in engines:
        if len(engines) >= 3:
            engines.pop(next(iter(engines))).close()
        cfg = NetConfig(feat_len=feat_len)
        e = Engine(cfg, 0, max_points=NMAX, max_pairs=PMAX)
        e.load_state_dict(generate_state_dict(cfg, 7, variant))
        engines[key] = e
    eng = engines[key]
    P = int(rng.integers(1, PMAX + 1))
    J, K = int(rng.integers(8192, NMAX + 1)), int(rng.integers(8192, NMAX + 1))
    while P * J * K < 2.05e8:                       # the screened path is what is under test (P J K >= 2e8)
        P = min(PMAX, P + 1) if P < PMAX else P
        J, K = min(NMAX, J + 2000), min(NMAX, K + 2000)
    shape = "kitti" if feat_len == 4 else "3dmatch"
    partial = bool(rng.integers(0, 2))
    iters = int(rng.integers(3, 6))
    raw = make_batch(max(J, K), [int(rng.integers(0, 1 << 30)) for _ in range(P)], feat_len, shape, partial)
    src = torch.from_numpy(np.ascontiguousarray(raw["points_src"][:, :J])).cuda()
    ref_np = np.ascontiguousarray(raw["points_ref"][:, :K]).copy()
    if rng.random() < 0.5:
        a, n = int(rng.integers(0, K - 200)), int(rng.integers(2, 130))
        ref_np[:, a:a + n] = ref_np[:, a:a + 1]     # a run of identical ref points
    ref = torch.from_numpy(ref_np).cuda()
    outs = []
    for pruned in (True, False):
        eng.set_prune_thresholds(64 if pruned else 0, 1)
        eng.enable_screen(pruned)
        eng.prune_stats(reset=True)
        o = eng.register(src, ref, iters)
        kept, total = eng.prune_stats()
        outs.append({k: o[k].cpu().numpy() for k in ("idx", "logits", "transforms", "invalid")})
        if pruned:
            share = kept / max(total, 1)
    diff = [k for k in outs[0] if not np.array_equal(outs[0][k], outs[1][k], equal_nan=True)]
    bad += bool(diff)
    print(f"case {c}: P {P} J {J} K {K} feat {feat_len} {variant} partial {int(partial)} iters {iters}: products visited {share:.3f}"
          f"{' DIFFERS in ' + ','.join(diff) if diff else ' ok'}", flush=True)
for e in engines.values():
    e.close()
print(f"{cases} cases, {bad} mismatching, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)

"""Randomised stress of the screened arg-min against the exhaustive exact-fp32 kernel: shapes, magnitudes and degeneracy
regimes drawn at random; any difference is a bug (both must return the same indices)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deepsir_amd.arch import NetConfig
from deepsir_amd.engine import Engine

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.Generator(np.random.Philox(key=int(sys.argv[2]) if len(sys.argv) > 2 else 7))
eng = Engine(NetConfig(), 0, max_points=8192, max_pairs=4)
bad = 0
t0 = time.time()
for c in range(cases):
    P = int(rng.integers(1, 5))
    J = int(rng.integers(1, 6001)) if rng.random() < 0.8 else int(rng.integers(1, 70))
    K = int(rng.integers(1, 6001)) if rng.random() < 0.8 else int(rng.integers(1, 70))
    regime = rng.choice(["random", "near", "cluster", "dup", "tiny", "big", "mixed_scale", "unnormalised", "same_sign", "same_sign_near",
                         "one_hot", "constant"])
    a = rng.standard_normal((P, J, 64)).astype(np.float32)
    b = rng.standard_normal((P, K, 64)).astype(np.float32)
    m = min(J, K)
    if regime in ("same_sign", "same_sign_near"):
        # no cancellation inside the accumulator: a truncating adder's rounding errors would all point the same way
        sa_, sb_ = rng.choice([-1.0, 1.0]), rng.choice([-1.0, 1.0])
        a, b = sa_ * np.abs(a), sb_ * np.abs(b)
        if regime == "same_sign_near":
            b[:, :m] = a[:, :m] * (sb_ / sa_) + np.abs(rng.standard_normal((P, m, 64))).astype(np.float32) * sb_ * 10.0 ** rng.uniform(-7, -2)
    elif regime == "one_hot":
        # a single non-zero component per vector (many exactly equal distances: ties must go to the lower index)
        a = np.zeros((P, J, 64), np.float32); b = np.zeros((P, K, 64), np.float32)
        np.put_along_axis(a, rng.integers(0, 64, (P, J, 1)), rng.uniform(0.25, 4.0, (P, J, 1)).astype(np.float32) * rng.choice([-1, 1], (P, J, 1)), 2)
        np.put_along_axis(b, rng.integers(0, 64, (P, K, 1)), rng.uniform(0.25, 4.0, (P, K, 1)).astype(np.float32) * rng.choice([-1, 1], (P, K, 1)), 2)
    elif regime == "constant":
        # 64 identical products per dot product
        a = (rng.uniform(0.05, 0.2, (P, J, 1)) * np.ones((1, 1, 64))).astype(np.float32)
        b = (rng.uniform(0.05, 0.2, (P, K, 1)) * np.ones((1, 1, 64))).astype(np.float32)
    if regime == "near":
        b[:, :m] = a[:, :m] + rng.standard_normal((P, m, 64)).astype(np.float32) * 10.0 ** rng.uniform(-7, -2)
    elif regime == "cluster":
        nc = int(rng.integers(1, 9))
        cen = rng.standard_normal((P, nc, 64)).astype(np.float32)
        s = 10.0 ** rng.uniform(-6, -1)
        b = np.take_along_axis(cen, rng.integers(0, nc, (P, K, 1)).repeat(64, 2), 1) + s * rng.standard_normal((P, K, 64)).astype(np.float32)
        a = np.take_along_axis(cen, rng.integers(0, nc, (P, J, 1)).repeat(64, 2), 1) + s * rng.standard_normal((P, J, 64)).astype(np.float32)
    elif regime == "dup":
        b[:, rng.integers(0, K, max(1, K // 3))] = b[:, rng.integers(0, K, 1)]
    if regime not in ("unnormalised", "one_hot", "constant"):
        a /= np.linalg.norm(a, axis=2, keepdims=True)
        b /= np.linalg.norm(b, axis=2, keepdims=True)
    if regime == "tiny":
        a *= 10.0 ** rng.uniform(-12, -3); b *= 10.0 ** rng.uniform(-12, -3)
    elif regime == "big":
        a *= 10.0 ** rng.uniform(1, 4); b *= 10.0 ** rng.uniform(1, 4)
    elif regime == "mixed_scale":
        a *= (10.0 ** rng.uniform(-4, 2, (P, J, 1))).astype(np.float32)
        b *= (10.0 ** rng.uniform(-4, 2, (P, K, 1))).astype(np.float32)
    ta, tb = torch.from_numpy(a.astype(np.float32)).cuda(), torch.from_numpy(b.astype(np.float32)).cuda()
    ex = eng.nn_match(ta, tb)
    sc, (nc_, nexh) = eng.nn_match_screened(ta, tb)
    if not torch.equal(ex, sc):
        bad += 1
        d = (ex != sc).sum().item()
        print(f"MISMATCH case {c}: P {P} J {J} K {K} regime {regime}: {d} rows differ")
print(f"{cases} cases, {bad} mismatching, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)

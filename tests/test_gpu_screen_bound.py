"""The fp16 screening of the nearest-descriptor search (csrc/nn_screen.hip) on the matrix core itself.

The reference decides every arg-min in fp32 (network/matchnet.py:96-113); the engine first discards columns by a lower
bound L computed with fp16 MFMAs and decides among the survivors in exact fp32.  That is exact iff, for EVERY (row, column),

        L <= D <= U = L + 2 d          (D: the exact fp32 distance of dsir_nn_match)

This file asserts it entry by entry through ``dsir_screen_bounds`` - the product kernel's MFMA chain on the product's fp16
operands - on inputs chosen against the bound rather than drawn from a Gaussian: same-sign components (no cancellation in
the accumulator: rounding errors of a truncating adder would all point the same way), constant vectors and 64 identical
products (longest carry chains), components on fp16 rounding boundaries (largest low parts), |x| = 16 (edge of the domain),
norms from 1e-3 to 30, one-hot and sparse vectors, components below the fp16 normal range.

Bar: every entry inside [L, U], AND with a measured margin of 2: |D - (L + U)/2| <= (U - L)/4.  The accumulation error
of the matrix core is measured beside it (fp64 sum of the same fp16 products), in units of one fp32 rounding of the sum of
magnitudes, and printed: the bound's header budgets 198 of those.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _engine():
    from deepsir_amd.arch import NetConfig
    from deepsir_amd.engine import Engine
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU")
    global _ENG
    try:
        return _ENG
    except NameError:
        _ENG = Engine(NetConfig(), 0, max_points=8192, max_pairs=1)
        return _ENG


def split16(x):
    """numpy restatement of split4 (nn_screen.hip): x -> (2^11 xh, xl) as float64 arrays of fp16-representable values."""
    x = x.astype(np.float32)
    t = x.astype(np.float16)
    t = np.where(np.abs(t.astype(np.float32)) < np.float32(2.0 ** -14), np.float16(0), t)
    lo = ((x - t.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
    return t.astype(np.float64) * 2048.0, lo.astype(np.float64)


def unit(v):
    return v / np.maximum(np.linalg.norm(v, axis=-1, keepdims=True), 1e-30)


def fp16_ties(rng, shape, lo_exp=-6, hi_exp=0):
    """values half way between two adjacent fp16 numbers (the conversion is a tie; the low part is as large as it gets),
    nudged by -1 / 0 / +1 fp32 ulp"""
    e = rng.integers(lo_exp, hi_exp, shape)
    m = rng.integers(0, 1024, shape)
    h = (1.0 + m / 1024.0) * 2.0 ** e                   # an fp16 value
    x = (h + 2.0 ** (e - 11)).astype(np.float32)         # + half an fp16 ulp: exactly representable in fp32
    nudge = rng.integers(-1, 2, shape)
    x = np.where(nudge < 0, np.nextafter(x, np.float32(0)), np.where(nudge > 0, np.nextafter(x, np.float32(np.inf)), x))
    return (x * rng.choice([-1.0, 1.0], shape)).astype(np.float32)


def regimes(rng, J, K):
    g = lambda n: rng.standard_normal((n, 64))
    out = {}
    out["gaussian_unit"] = (unit(g(J)), unit(g(K)))
    out["all_positive_unit"] = (unit(np.abs(g(J))), unit(np.abs(g(K))))
    out["pos_vs_neg_unit"] = (unit(np.abs(g(J))), -unit(np.abs(g(K))))
    out["all_negative_unit"] = (-unit(np.abs(g(J))), -unit(np.abs(g(K))))
    # constant vectors: 64 identical products per dot product (values with long mantissas so that every partial sum rounds)
    ca = rng.uniform(0.05, 0.2, (J, 1)) * np.ones((1, 64))
    cb = rng.uniform(0.05, 0.2, (K, 1)) * np.ones((1, 64))
    out["constant_vectors"] = (ca, cb)
    out["constant_unit_1_8"] = (np.full((J, 64), 0.125) * (1 + 2.0 ** -10 * rng.integers(0, 8, (J, 1))),
                                np.full((K, 64), 0.125) * (1 + 2.0 ** -10 * rng.integers(0, 8, (K, 1))))
    out["fp16_ties"] = (fp16_ties(rng, (J, 64)), fp16_ties(rng, (K, 64)))
    out["fp16_ties_positive_unit"] = (unit(np.abs(fp16_ties(rng, (J, 64)))), np.abs(fp16_ties(rng, (K, 64), -4, -2)))
    out["abs_16_all"] = (16.0 * rng.choice([-1.0, 1.0], (J, 64)), 16.0 * rng.choice([-1.0, 1.0], (K, 64)))
    out["abs_16_same_sign"] = (np.full((J, 64), 16.0) - 2.0 ** -8 * rng.integers(0, 64, (J, 64)),
                               np.full((K, 64), 16.0) - 2.0 ** -8 * rng.integers(0, 64, (K, 64)))
    a = unit(g(J)); b = unit(g(K))
    a[:, :3] = 16.0 * np.sign(a[:, :3]); b[:, :3] = 16.0 * np.sign(b[:, :3])
    out["abs_16_some_rest_small"] = (a, b)
    out["norms_1e-3_to_30"] = (unit(g(J)) * 10.0 ** rng.uniform(-3, np.log10(30.0), (J, 1)),
                               unit(g(K)) * 10.0 ** rng.uniform(-3, np.log10(30.0), (K, 1)))
    out["norms_positive_1e-3_to_30"] = (unit(np.abs(g(J))) * 10.0 ** rng.uniform(-3, np.log10(30.0), (J, 1)),
                                        unit(np.abs(g(K))) * 10.0 ** rng.uniform(-3, np.log10(30.0), (K, 1)))
    a = np.zeros((J, 64)); a[np.arange(J), rng.integers(0, 64, J)] = rng.uniform(0.5, 16.0, J) * rng.choice([-1, 1], J)
    b = np.zeros((K, 64)); b[np.arange(K), rng.integers(0, 64, K)] = rng.uniform(0.5, 16.0, K) * rng.choice([-1, 1], K)
    out["one_hot"] = (a, b)
    a = g(J) * (rng.random((J, 64)) < 0.1); b = g(K) * (rng.random((K, 64)) < 0.1)
    a[:, 0] += 1e-3; b[:, 0] += 1e-3
    out["sparse_unit"] = (unit(a), unit(b))
    out["below_fp16_normal"] = (g(J) * 10.0 ** rng.uniform(-9, -4, (J, 64)), g(K) * 10.0 ** rng.uniform(-9, -4, (K, 64)))
    a = unit(g(J)); b = unit(g(K)); m = min(J, K)
    b[:m] = unit(a[:m] + 10.0 ** rng.uniform(-7, -3, (m, 1)) * g(m))
    out["near_duplicates_unit"] = (a, b)
    a = unit(np.abs(g(J))); b = unit(np.abs(g(K)))
    b[:m] = unit(a[:m] + 10.0 ** rng.uniform(-7, -3, (m, 1)) * np.abs(g(m)))
    out["near_duplicates_positive_unit"] = (a, b)
    # geometric decay: a few dominant products and a long tail (alignment shifts inside the adder)
    dec = 2.0 ** -np.arange(64) * 8.0
    out["geometric_decay_positive"] = (dec * rng.uniform(0.5, 1.0, (J, 64)), dec * rng.uniform(0.5, 1.0, (K, 64)))
    return {k: (v[0].astype(np.float32), v[1].astype(np.float32)) for k, v in out.items()}


def check(eng, name, a, b, in_domain=True):
    J, K = a.shape[0], b.shape[0]
    o = eng.screen_bounds(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda())
    L = o["lower"].cpu().numpy().astype(np.float64); U = o["upper"].cpu().numpy().astype(np.float64)
    D = o["exact"].cpu().numpy().astype(np.float64)
    z = o["zacc"].cpu().numpy().astype(np.float64)
    assert int(o["out_of_domain"].item()) == (0 if in_domain else 1), name
    assert np.isfinite(L).all() and np.isfinite(U).all() and np.isfinite(D).all(), name
    # --- the property itself, every entry
    below = int((D < L).sum()); above = int((D > U).sum())
    half = (U - L) / 2.0
    ratio = np.abs(D - (L + U) / 2.0) / half                 # 1.0 = on the edge of [L, U]
    # --- accumulation error of the six chained MFMAs against an fp64 sum of the same fp16 products
    ah, al = split16(a); bh, bl = split16(b)
    # the seed 2^22 c of each column is whatever the kernel used: an all-zero src row leaves it in the accumulator
    z0 = eng.screen_bounds(torch.zeros((1, 64), device="cuda"), torch.from_numpy(b).cuda())["zacc"].cpu().numpy().astype(np.float64)[0]
    dot = ah @ bh.T + ah @ bl.T + al @ bh.T               # = 2^22 (ah.bh + 2^-11 (ah.bl + al.bh)) in the kernel's scaling
    mag = np.abs(ah) @ np.abs(bh).T + np.abs(ah) @ np.abs(bl).T + np.abs(al) @ np.abs(bh).T + np.abs(z0)[None, :]
    err = np.abs(z - (z0[None, :] + dot))
    roundings = err / np.maximum(mag * 2.0 ** -24, 1e-300)
    # --- the product path on the same input: its entries carry this kernel's L, its threshold covers the minimum, its
    # answer is the exact arg-min (ties to the lower index)
    idx = o["idx"].cpu().numpy(); T = o["thresh"].cpu().numpy().astype(np.float64)
    cnt = o["cand_count"].cpu().numpy(); code = o["cand_code"].cpu().numpy(); cl = o["cand_lower"].cpu().numpy()
    cap = code.shape[1]
    Dmin = D.min(1)
    want = D.argmin(1)                                      # first minimum = lowest index
    L32 = o["lower"].cpu().numpy()
    mism = 0; uncovered = 0
    for j in range(J):
        n = min(int(cnt[j]), cap)
        cols = code[j, :n]
        real = cols >= 0
        mism += int((cl[j, :n][real] != L32[j, cols[real]]).sum())
        if cnt[j] > cap:
            continue                                        # overflowed list: the row went to the exhaustive kernel
        live = cl[j, :n].astype(np.float64) <= T[j]
        k = want[j]
        hit = bool(((cols == k) & live).any()) or bool((((-cols - 1) % 16 == k % 16) & (cols < 0) & live).any())
        uncovered += 0 if hit else 1
    print(f"[screen-bound] {name:32s} {J}x{K}: D<L {below}, D>U {above}, worst |D-mid|/half {ratio.max():.4f}, "
          f"MFMA accumulation error {roundings.max():6.2f} fp32 roundings of the magnitude sum (median {np.median(roundings):.3f}); "
          f"entries/row {np.minimum(cnt, cap).mean():.2f}, overflowed rows {(cnt > cap).sum()}")
    assert below == 0 and above == 0, f"{name}: the screening bound is VIOLATED on this hardware ({below} below, {above} above)"
    assert (T >= Dmin).all(), f"{name}: a row's threshold is below its minimum distance"
    assert mism == 0, f"{name}: {mism} product entries do not carry the diagnostic kernel's lower bound"
    assert uncovered == 0, f"{name}: {uncovered} rows whose arg-min column is neither an entry nor covered by a class"
    assert (idx == want).all(), f"{name}: {(idx != want).sum()} rows differ from the exact arg-min"
    return float(ratio.max()), float(roundings.max())


@pytest.mark.parametrize("J,K,seed", [(256, 1024, 11), (100, 333, 12), (512, 2048, 13)])
def test_bound_holds_entrywise_on_adversarial_inputs(J, K, seed):
    eng = _engine()
    rng = np.random.Generator(np.random.Philox(key=seed))
    worst = {}
    for name, (a, b) in regimes(rng, J, K).items():
        worst[name] = check(eng, name, a, b)
    r = max(v[0] for v in worst.values()); n = max(v[1] for v in worst.values())
    print(f"[screen-bound] worst over regimes: |D-mid|/half = {r:.4f} (bar 0.5), MFMA accumulation = {n:.2f} roundings (budget 198)")
    # measured margin of 2 on the bound: nothing may use more than half of the half-width
    assert r <= 0.5, worst


def test_out_of_domain_is_flagged_not_screened():
    """|x| > 16: split4 raises the flag (the product path then searches exhaustively; dsir_screen_bounds screens anyway so
    that the flag and the arithmetic can be looked at separately - here only the flag and the final answer are checked)."""
    eng = _engine()
    rng = np.random.Generator(np.random.Philox(key=5))
    a = rng.standard_normal((64, 64)).astype(np.float32) * 20
    b = rng.standard_normal((128, 64)).astype(np.float32) * 20
    o = eng.screen_bounds(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda())
    assert int(o["out_of_domain"].item()) == 1
    idx, _ = eng.nn_match_screened(torch.from_numpy(a[None]).cuda(), torch.from_numpy(b[None]).cuda())
    ex = eng.nn_match(torch.from_numpy(a[None]).cuda(), torch.from_numpy(b[None]).cuda())
    assert torch.equal(idx, ex)

"""Properties of the KNN-pyramid oracle (the reference's nanoflann KNN is not
installable: parity at this boundary is 'unpinned', so the oracle is checked
against an independent fp64 brute force and its own tie rule)."""
import numpy as np
import pytest

from oracle.knn import add_pyramids, knn, knn_fast, knn_pyramid, sqdist_f32


def test_knn_matches_stable_sort():
    rng = np.random.default_rng(0)
    pts = rng.uniform(0, 3, (700, 3)).astype(np.float32)
    pts[100:110] = pts[5]          # exact duplicates -> distance ties
    pts[300] = pts[299]
    idx = knn(pts, pts, 16)
    d = sqdist_f32(pts, pts)
    ref = np.argsort(d, axis=1, kind="stable")[:, :16]
    assert np.array_equal(idx, ref.astype(np.int32))
    assert idx[0, 0] == 0 and idx[105, 0] == 5     # lowest index wins a zero-distance tie


def test_knn_against_float64():
    rng = np.random.default_rng(1)
    s = rng.uniform(0, 3, (500, 3)).astype(np.float32)
    q = rng.uniform(0, 3, (64, 3)).astype(np.float32)
    idx = knn(s, q, 8)
    d64 = ((q[:, None].astype(np.float64) - s[None].astype(np.float64)) ** 2).sum(-1)
    srt = np.sort(d64, axis=1)[:, :8]
    got = np.take_along_axis(d64, idx.astype(np.int64), 1)
    np.testing.assert_allclose(got, srt, rtol=1e-5, atol=1e-9)


def test_pyramid_shapes_and_prefix_rule():
    rng = np.random.default_rng(2)
    n = 1100
    pts = rng.uniform(0, 3, (n, 3)).astype(np.float32)
    p = knn_pyramid(pts, 16, (4, 4, 4, 4))
    sizes = [1100, 275, 68, 17, 4]
    assert p["xyz"].shape == (sum(sizes[:4]), 3)
    assert p["neigh_idx"].shape == (sum(sizes[:4]), 16)
    assert p["sub_idx"].shape == (sum(sizes[1:5]), 16)
    assert p["interp_idx"].shape == (sum(sizes[:4]), 1)
    # level 1 points are the first n/4 points of level 0
    assert np.array_equal(p["xyz"][1100:1100 + 275], pts[:275])
    # pooling indices = neighbours of the kept prefix
    assert np.array_equal(p["sub_idx"][:275], p["neigh_idx"][:275])
    # every kept point interpolates from itself
    assert np.array_equal(p["interp_idx"][:275, 0], np.arange(275))
    assert p["interp_idx"][:1100].max() < 275


def test_too_few_support_points():
    with pytest.raises(ValueError):
        knn(np.zeros((4, 3), np.float32), np.zeros((4, 3), np.float32), 16)


def test_batch_keys():
    rng = np.random.default_rng(3)
    data = {"points_src": rng.uniform(0, 3, (2, 1024, 3)).astype(np.float32),
            "points_ref": rng.uniform(0, 3, (2, 1024, 4)).astype(np.float32)}
    out = add_pyramids(data)
    for k in ("points_src", "points_ref"):
        assert out[k + "_xyz"].shape == (2, 1360, 3)
        assert out[k + "_neigh_idx"].dtype == np.int64
        assert out[k + "_sub_idx"].shape == (2, 340, 16)


def test_fast_variant_is_the_same_rule():
    """``knn_fast`` (torch, every host core: the C5 levels of tests/test_gpu_parity.py) against ``knn`` on clouds with exact ties:
    duplicates, a lattice, all points identical, queries that are not support points - 16 neighbours and the 1-NN form."""
    rng = np.random.default_rng(5)
    n = 1500
    uni = rng.uniform(0, 3, (n, 3)).astype(np.float32)
    uni[100:110] = uni[5]; uni[700] = uni[699]
    lattice = np.stack(np.meshgrid(*[np.arange(12, dtype=np.float32)] * 3, indexing="ij"), -1).reshape(-1, 3)[rng.permutation(12 ** 3)[:n]]
    same = np.tile(np.array([[1.25, -2.5, 0.75]], np.float32), (n, 1))
    for name, pts in (("uniform+duplicates", uni), ("lattice", lattice.astype(np.float32)), ("identical", same)):
        assert np.array_equal(knn_fast(pts, pts, 16, chunk=256), knn(pts, pts, 16)), name
        assert np.array_equal(knn_fast(pts[: n // 4], pts, 1, chunk=256), knn(pts[: n // 4], pts, 1)), name
    a = knn_pyramid(uni, 16, (4, 4, 4, 4), fast_min=1)
    b = knn_pyramid(uni, 16, (4, 4, 4, 4))
    for k in a:
        assert np.array_equal(a[k], b[k]), k

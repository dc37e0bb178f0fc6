"""GPU parity tests: the HIP engine (through the C ABI of include/dsir.h) against
the CPU oracle and against the golden fixtures captured from the imported
reference.  Run on the MI355X box with ``pytest -m gpu``.

Tolerances (BASELINE.json north_star): (R, t) within 1e-4 rad / 1e-4 m of the
reference on identical inputs.  Index/integer work (KNN pyramid) is bit-exact.
Per-stage float tensors are compared at rtol/atol 2e-4 against values of O(1)
(fp32 summation-order noise of ~35 chained layers).
"""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLD, build_case, load_golden

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a GPU (run with -m 'not gpu' on CPU-only hosts)")
    return torch.device("cuda", 0)


_ENGINES = {}


def engine_for(cfg, sd, key, max_points=8192, max_pairs=2):
    from deepsir_amd.engine import Engine
    k = (key, cfg.feat_len, max_points, max_pairs)
    if k not in _ENGINES:
        e = Engine(cfg, 0, max_points=max_points, max_pairs=max_pairs)
        e.load_state_dict(sd)
        _ENGINES[k] = e
    return _ENGINES[k]


def cu(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(_dev())


def rot_angle(Ra, Rb):
    """Rounding-robust rotation distance (rad): |vee(skew(Ra^T Rb))| in fp64."""
    D = Ra.astype(np.float64).T @ Rb.astype(np.float64)
    v = 0.5 * np.array([D[2, 1] - D[1, 2], D[0, 2] - D[2, 0], D[1, 0] - D[0, 1]])
    s = np.linalg.norm(v)
    c = 0.5 * (np.trace(D) - 1.0)
    return float(np.arctan2(s, c))


def assert_pose_close(T, Tref, tol_rad=1e-4, tol_m=1e-4, msg=""):
    T, Tref = np.asarray(T), np.asarray(Tref)
    worst_a = worst_d = 0.0
    for t, r in zip(T.reshape(-1, 3, 4), Tref.reshape(-1, 3, 4)):
        a = rot_angle(t[:, :3], r[:, :3])
        d = float(np.linalg.norm(t[:, 3].astype(np.float64) - r[:, 3].astype(np.float64)))
        worst_a, worst_d = max(worst_a, a), max(worst_d, d)
        assert a < tol_rad and d < tol_m, f"{msg} rot diff {a:.3e} rad, trans diff {d:.3e} m"
    print(f"[pose] {msg}: worst rot diff {worst_a:.2e} rad, worst trans diff {worst_d:.2e} m (tol {tol_rad:g}/{tol_m:g})")


# --------------------------------------------------------------------------- KNN pyramid (bit-exact)
@pytest.mark.parametrize("n,seed", [(1024, 0), (1100, 1), (2048, 2), (4999, 3), (5000, 4)])
def test_knn_pyramid_bit_exact(n, seed):
    from deepsir_amd.arch import NetConfig
    from deepsir_amd.weights import generate_state_dict
    from oracle.knn import knn_pyramid
    cfg = NetConfig(feat_len=3)
    eng = engine_for(cfg, generate_state_dict(cfg, 0), "w0")
    rng = np.random.default_rng(seed)
    pts = rng.uniform(0, 3, (2, n, 4)).astype(np.float32)
    pts[0, 7:12, :3] = pts[0, 3, :3]          # duplicates -> exact distance ties (padded clouds have them)
    pts[1, n // 2, :3] = pts[1, 0, :3]
    xyz, neigh, sub, interp = eng.knn_pyramid(cu(pts))
    for c in range(2):
        ref = knn_pyramid(pts[c], 16, cfg.sub_sampling_ratio)
        assert np.array_equal(xyz[c].cpu().numpy(), ref["xyz"])
        assert np.array_equal(neigh[c].cpu().numpy(), ref["neigh_idx"])
        assert np.array_equal(sub[c].cpu().numpy(), ref["sub_idx"])
        assert np.array_equal(interp[c].cpu().numpy(), ref["interp_idx"])


def test_interp_search_through_the_grid_bit_exact():
    """The interpolation search (nearest point of the next level) walks that level's grid when the launch holds >= 65536
    queries (csrc/knn_grid.hip::grid_nn1_kernel): 16 clouds of 4224 points (levels 4224 / 1056 / 264 / 66: level 0 searches
    level 1's grid) of shapes that stress the walk - uniform, planar, collinear, clusters with far outliers among the QUERY-only
    points (queries far outside the support's bounding box), all points identical, exact duplicates of support points,
    a lattice (exact ties: the lower index wins) - against the brute-force oracle, every level, bit for bit."""
    from deepsir_amd.arch import NetConfig
    from deepsir_amd.weights import generate_state_dict
    from oracle.knn import knn_pyramid
    cfg = NetConfig(feat_len=3)
    eng = engine_for(cfg, generate_state_dict(cfg, 0), "w0nn1", max_points=4224, max_pairs=16)
    rng = np.random.default_rng(11)
    n = 4224
    clouds = []
    for c in range(10):
        clouds.append(rng.uniform(0, 3, (n, 3)).astype(np.float32))
    planar = rng.uniform(0, 3, (n, 3)).astype(np.float32); planar[:, 2] = 0.5
    line = np.zeros((n, 3), np.float32); line[:, 0] = rng.uniform(-5, 5, n)
    blobs = (rng.standard_normal((n, 3)) * 0.05 + rng.integers(0, 4, (n, 1)) * 2.0).astype(np.float32)
    blobs[2000:2016] = rng.uniform(-500, 500, (16, 3))                    # outliers that are queries only (index >= 1056)
    same = np.tile(np.array([[1.25, -2.5, 0.75]], np.float32), (n, 1))
    dup = rng.uniform(0, 3, (n, 3)).astype(np.float32); dup[1056:2112] = dup[:1056]; dup[5] = dup[900]   # queries ON support points; a support duplicate
    lattice = np.stack(np.meshgrid(*[np.arange(17, dtype=np.float32)] * 3, indexing="ij"), -1).reshape(-1, 3)[rng.permutation(17 ** 3)[:n]]
    clouds += [planar, line, blobs, same, dup, lattice.astype(np.float32)]
    pts = np.stack(clouds)
    assert pts.shape == (16, n, 3)
    xyz, neigh, sub, interp = eng.knn_pyramid(cu(pts))
    for c in range(16):
        ref = knn_pyramid(pts[c], 16, cfg.sub_sampling_ratio)
        assert np.array_equal(interp[c].cpu().numpy(), ref["interp_idx"]), f"interp_idx of cloud {c}"
        assert np.array_equal(neigh[c].cpu().numpy(), ref["neigh_idx"]), f"neigh_idx of cloud {c}"


def test_knn_grid_adversarial_clouds_bit_exact():
    """The grid-pruned search (levels with >= 2048 points) must equal the brute-force oracle bit for bit
    on clouds that stress the grid: planar, collinear, clustered with far outliers, all points identical,
    KITTI-shaped extent."""
    from deepsir_amd.arch import NetConfig
    from deepsir_amd.weights import generate_state_dict
    from oracle.knn import knn
    cfg = NetConfig(feat_len=3)
    eng = engine_for(cfg, generate_state_dict(cfg, 0), "w0knn", max_points=16384, max_pairs=1)
    rng = np.random.default_rng(7)
    n = 4096
    planar = rng.uniform(0, 3, (n, 3)).astype(np.float32); planar[:, 2] = 0.5
    line = np.zeros((n, 3), np.float32); line[:, 0] = rng.uniform(-5, 5, n)
    blobs = (rng.standard_normal((n, 3)) * 0.05 + rng.integers(0, 4, (n, 1)) * 2.0).astype(np.float32)
    blobs[:8] = rng.uniform(-500, 500, (8, 3))
    same = np.tile(np.array([[1.25, -2.5, 0.75]], np.float32), (n, 1))
    lattice = np.stack(np.meshgrid(*[np.arange(16, dtype=np.float32)] * 3, indexing="ij"), -1).reshape(-1, 3)  # exact ties
    for name, pts in (("planar", planar), ("line", line), ("blobs", blobs), ("same", same), ("lattice", lattice)):
        _, neigh, _, _ = eng.knn_pyramid(cu(pts[None]))
        got = neigh[0, :n].cpu().numpy()
        assert np.array_equal(got, knn(pts, pts, 16)), name
    kitti = np.concatenate([rng.uniform(-50, 50, (16384, 2)), rng.uniform(-3, 3, (16384, 1))], 1).astype(np.float32)
    _, neigh, _, _ = eng.knn_pyramid(cu(kitti[None]))
    assert np.array_equal(neigh[0, :16384].cpu().numpy(), knn(kitti, kitti, 16))
    assert np.array_equal(neigh[0, 16384:16384 + 4096].cpu().numpy(), knn(kitti[:4096], kitti[:4096], 16))


def test_knn_pyramid_65536_points_bit_exact():
    """C5's pyramid (65536 / 16384 / 4096 / 1024 points: three grid levels, the interpolation search through the grid, one brute-force
    level) against the brute-force oracle, ALL four levels, neigh / sub / interp bit for bit - the other large-cloud tests feed the
    engine's own pyramid to the oracle and would not see a wrong 16th neighbour.  The cloud is C5's: crop +
    jitter + resampling WITH replacement, i.e. it holds exact duplicates (distance ties)."""
    from deepsir_amd.arch import NetConfig
    from deepsir_amd.synth import make_pair
    from deepsir_amd.weights import generate_state_dict
    from oracle.knn import knn_pyramid
    cfg = NetConfig(feat_len=3)
    n = 65536
    from deepsir_amd.engine import Engine
    eng = Engine(cfg, 0, max_points=n, max_pairs=1)
    eng.load_state_dict(generate_state_dict(cfg, 0))
    pts = make_pair(n, 9005, 3, "3dmatch", True)["points_src"]          # [1, n, 3]
    xyz, neigh, sub, interp = eng.knn_pyramid(cu(pts))
    eng.close()
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    ref = knn_pyramid(pts[0], 16, cfg.sub_sampling_ratio)
    assert np.array_equal(xyz[0].cpu().numpy(), ref["xyz"])
    off = 0
    for l, nl in enumerate((65536, 16384, 4096, 1024)):
        assert np.array_equal(neigh[0, off:off + nl].cpu().numpy(), ref["neigh_idx"][off:off + nl]), f"neigh_idx, level {l}"
        assert np.array_equal(interp[0, off:off + nl].cpu().numpy(), ref["interp_idx"][off:off + nl]), f"interp_idx, level {l}"
        off += nl
    assert np.array_equal(sub[0].cpu().numpy(), ref["sub_idx"])


def test_knn_rejects_small_cloud():
    from deepsir_amd.arch import NetConfig
    from deepsir_amd.engine import EngineError
    from deepsir_amd.weights import generate_state_dict
    cfg = NetConfig(feat_len=3)
    eng = engine_for(cfg, generate_state_dict(cfg, 0), "w0")
    with pytest.raises(EngineError):
        eng.knn_pyramid(cu(np.zeros((1, 512, 3), np.float32)))


# --------------------------------------------------------------------------- stage parity on the golden stage cases
@pytest.mark.parametrize("name", ["stage_n1024_s1", "stage_n1024_s2_sep"])
def test_stages_vs_oracle_and_golden(name):
    from oracle.network import OracleNet, to_torch
    torch.set_num_threads(1)
    g, m, cfg, sd, data = build_case(name)
    eng = engine_for(cfg, sd, name)
    net = OracleNet(cfg, sd)
    d = to_torch(data)
    N = m["n"]
    # ---- RandLA.forward (feature extractor), src and ref as one batch of two clouds
    feats = cu(np.concatenate([data["points_src"], data["points_ref"]], 0))
    xyz = cu(np.concatenate([data["points_src_xyz"], data["points_ref_xyz"]], 0))
    neigh = cu(np.concatenate([data["points_src_neigh_idx"], data["points_ref_neigh_idx"]], 0), torch.int32)
    sub = cu(np.concatenate([data["points_src_sub_idx"], data["points_ref_sub_idx"]], 0), torch.int32)
    interp = cu(np.concatenate([data["points_src_interp_idx"], data["points_ref_interp_idx"]], 0), torch.int32)
    feat, logits = eng.randla_forward("feat_extractor", feats, xyz, neigh, sub, interp)
    f_s, x_s, lab_s, sc_s, f_r, x_r, lab_r, sc_r = net.forward_pair(d)
    feat_np = feat.cpu().numpy()
    np.testing.assert_allclose(feat_np[0].T, g["feat_src"][0], rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(feat_np[1].T, g["feat_ref"][0], rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(feat_np[0].T, f_s[0].numpy(), rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(logits[0].cpu().numpy().T, g["logits_src"][0], rtol=2e-4, atol=2e-4)
    # ---- score_fun: teacher-forced on the reference's feat/logits
    _, _, lg_r = net.randla("feat_extractor", d["points_ref"], d["points_ref_xyz"], d["points_ref_neigh_idx"],
                            d["points_ref_sub_idx"], d["points_ref_interp_idx"])
    feat_ref_in = cu(np.stack([g["feat_src"][0].T, g["feat_ref"][0].T]))
    logit_in = cu(np.stack([g["logits_src"][0].T, lg_r[0].numpy().T]))
    score, label = eng.score(feat_ref_in, logit_in, xyz, neigh)
    np.testing.assert_allclose(score[0].cpu().numpy(), g["score_src"][0], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(score[1].cpu().numpy(), g["score_ref"][0], rtol=1e-4, atol=1e-6)
    assert np.array_equal(label[0].cpu().numpy().astype(np.int8), g["label_src"][0, 0])
    assert np.array_equal(label[1].cpu().numpy().astype(np.int8), g["label_ref"][0, 0])
    # ---- aggregation: teacher-forced inputs
    sc_in = cu(np.stack([g["score_src"][0], g["score_ref"][0]]))
    desc = eng.aggregate(xyz[:, :N].contiguous(), feat_ref_in, sc_in)
    np.testing.assert_allclose(desc[0].cpu().numpy().T, g["desc_src0"][0], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(desc[1].cpu().numpy().T, g["desc_ref0"][0], rtol=1e-4, atol=2e-5)
    # ---- nearest-descriptor match on the reference's own descriptors
    ds, dr = torch.from_numpy(g["desc_src0"]), torch.from_numpy(g["desc_ref0"])
    idx = eng.nn_match(cu(g["desc_src0"][0].T[None]), cu(g["desc_ref0"][0].T[None]))[0].cpu().numpy()
    ref_idx = g["idx0_direct"][0].astype(np.int64)
    bad = np.nonzero(idx != ref_idx)[0]
    if len(bad):   # every disagreeing row must be a near tie in fp64
        best, second, _ = OracleNet.nn_gap(ds, dr)
        d64 = -2 * (ds.double()[0].T @ dr.double()[0]) + (ds.double()[0] ** 2).sum(0)[:, None] + (dr.double()[0] ** 2).sum(0)[None]
        for r in bad:
            assert abs(float(d64[r, idx[r]] - d64[r, ref_idx[r]])) < 1e-6, f"row {r}: not a tie"
    assert len(bad) <= 0.005 * len(idx)
    # ---- inlier RandLA, teacher-forced on the reference's iteration-0 correspondences
    i0 = g["idx"][0, 0].astype(np.int64)
    cat = np.concatenate([data["points_src_xyz"][0, :N], data["points_ref_xyz"][0, :N][i0]], 1)[None]
    _, lg_in = eng.randla_forward("inlier_model", cu(cat), xyz[:1], neigh[:1], sub[:1], interp[:1])
    np.testing.assert_allclose(lg_in[0, :, 0].cpu().numpy(), g["inlier_logit0"][0, 0], rtol=5e-4, atol=5e-4)
    # ---- Kabsch on the reference's weights
    w = 1.0 / (1.0 + np.exp(-g["inlier_logit0"][0, 0].astype(np.float64)))
    T, badflag = eng.kabsch(cu(data["points_src_xyz"][:, :N]), cu(data["points_ref_xyz"][:, :N][:, i0]),
                            cu(w[None].astype(np.float32)))
    assert_pose_close(T.cpu().numpy(), g["kabsch_T0"], 2e-6, 2e-6, "kabsch")
    assert int(badflag[0]) == 0


def test_kabsch_golden_cases():
    from deepsir_amd.arch import NetConfig
    from deepsir_amd.weights import generate_state_dict
    cfg = NetConfig(feat_len=3)
    eng = engine_for(cfg, generate_state_dict(cfg, 0), "w0")
    g = np.load(GOLD + "/kabsch_cases.npz")
    names = sorted({k[:-4] for k in g.files if k.endswith("_src")})
    for n in names:
        T, bad = eng.kabsch(cu(g[n + "_src"]), cu(g[n + "_tgt"]), cu(g[n + "_w"][..., 0]))
        assert bool(bad.any().item()) == bool(g[n + "_invalid"]), n
        assert_pose_close(T.cpu().numpy(), g[n + "_T"], 5e-6, 5e-6, n)
        R = T.cpu().numpy()[:, :, :3].astype(np.float64)
        assert np.all(np.linalg.det(R) > 0.999), n


@pytest.mark.parametrize("m", [16384, 20001, 65536])
def test_kabsch_chunked_reduction_on_large_clouds(m):
    """Clouds of >= 16384 points are reduced in chunks of 4096 points by several workgroups per pair (csrc/kabsch.hip): against
    the oracle's solve (fp64 SVD, model.py:22-66) within 5e-6 rad / 5e-6 m, and against the one-workgroup kernel on the same
    input (dsir_set_kabsch_chunked_min moves the threshold) within 1e-6 - same formulas, the fp64 partial sums in another order."""
    import os
    from deepsir_amd.arch import NetConfig
    from deepsir_amd.weights import generate_state_dict
    from oracle.network import OracleNet
    cfg = NetConfig(feat_len=3)
    eng = engine_for(cfg, generate_state_dict(cfg, 0), "w0kab", max_points=65536, max_pairs=2)
    rng = np.random.default_rng(m)
    P = 2
    src = rng.uniform(-2, 2, (P, m, 3)).astype(np.float32)
    ang = rng.uniform(-0.5, 0.5, (P, 3))
    tgt = np.empty_like(src)
    for p in range(P):
        cx, cy, cz = np.cos(ang[p]); sx, sy, sz = np.sin(ang[p])
        R = np.array([[cz * cy, cz * sy * sx - sz * cx, cz * sy * cx + sz * sx], [sz * cy, sz * sy * sx + cz * cx, sz * sy * cx - cz * sx],
                      [-sy, cy * sx, cy * cx]])
        tgt[p] = (src[p].astype(np.float64) @ R.T + rng.uniform(-1, 1, 3)).astype(np.float32)
    tgt += rng.normal(0, 0.01, tgt.shape).astype(np.float32)
    w = rng.uniform(0, 1, (P, m)).astype(np.float32)
    w[0, : m // 3] = 0.0                                   # a third of the points of pair 0 without weight
    T_chunked, bad = eng.kabsch(cu(src), cu(tgt), cu(w))
    assert not bool(bad.any())
    eng.set_kabsch_chunked_min(1 << 30)
    try:
        T_single, bad1 = eng.kabsch(cu(src), cu(tgt), cu(w))
    finally:
        eng.set_kabsch_chunked_min(0)
    assert not bool(bad1.any())
    assert_pose_close(T_chunked.cpu().numpy(), T_single.cpu().numpy(), 1e-6, 1e-6, f"chunked vs one workgroup, m = {m}")
    ref = np.stack([OracleNet.kabsch(torch.from_numpy(src[p:p + 1]), torch.from_numpy(tgt[p:p + 1]), torch.from_numpy(w[p:p + 1, :, None]))[0].numpy()[0]
                    for p in range(P)])
    assert_pose_close(T_chunked.cpu().numpy(), ref, 5e-6, 5e-6, f"chunked vs oracle, m = {m}")


# --------------------------------------------------------------------------- whole path
@pytest.mark.parametrize("name", ["stage_n1024_s1", "stage_n1024_s2_sep", "e2e_n2048_s3", "e2e_n2048_s4_sep",
                                  "e2e_n2048_s6_f4", "e2e_n5000_s5"])
def test_register_teacher_forced_matches_reference(name):
    """forward_align_4 with the reference's own correspondences forced: every
    stage but the arg-min runs on the engine; (R,t) per iteration within 1e-4."""
    g, m, cfg, sd, data = build_case(name)
    eng = engine_for(cfg, sd, name)
    forced = cu(np.transpose(g["idx"].astype(np.int32), (1, 0, 2)))           # [n_iter, P, J]
    out = eng.register(cu(data["points_src"]), cu(data["points_ref"]), m["n_iter"], forced_idx=forced)
    assert_pose_close(out["transforms"].cpu().numpy(), g["transforms"], 1e-4, 1e-4, name)
    lg = out["logits"].cpu().numpy()[:, 0]
    np.testing.assert_allclose(lg, g["logits"][0], rtol=2e-3, atol=2e-3)
    assert int(out["invalid"][0]) == int(bool(g["invalid"]))
    np.testing.assert_allclose(out["pt_ref_new"].cpu().numpy(), g["pt_ref_new"], rtol=0, atol=0)


@pytest.mark.parametrize("name", ["stage_n1024_s1", "stage_n1024_s2_sep", "e2e_n2048_s3", "e2e_n2048_s4_sep",
                                  "e2e_n2048_s6_f4", "e2e_n5000_s5"])
def test_register_free_running(name):
    """No teacher forcing, pyramids built on device.  The arg-min over fp32
    near-ties is not reproducible even by the reference against itself
    (SURVEY §8c), so: report the agreement rate; when it is 100 % through
    iteration i, (R,t) must be within 1e-4 through iteration i."""
    g, m, cfg, sd, data = build_case(name)
    eng = engine_for(cfg, sd, name)
    out = eng.register(cu(data["points_src"]), cu(data["points_ref"]), m["n_iter"])
    idx = out["idx"].cpu().numpy()[:, 0]
    agree = [(idx[i] == g["idx"][0, i]).mean() for i in range(m["n_iter"])]
    print(f"{name}: arg-min agreement per iteration {['%.4f' % a for a in agree]}")
    assert agree[0] > 0.99
    T = out["transforms"].cpu().numpy()[0]
    for i in range(m["n_iter"]):
        if all(a == 1.0 for a in agree[: i + 1]):
            assert_pose_close(T[i], g["transforms"][0, i], 1e-4, 1e-4, f"{name} iter {i}")
    first = next((i for i, a in enumerate(agree) if a < 1.0), None)
    if first is not None:
        _assert_flips_are_near_ties(name, g, m, cfg, sd, data, eng, idx, first)
    # whatever the flips, the result stays a rigid transform close to the reference's
    for i in range(m["n_iter"]):
        R = T[i][:, :3].astype(np.float64)
        np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-5)
    assert rot_angle(T[-1][:, :3], g["transforms"][0, -1][:, :3]) < 2e-2


def _assert_flips_are_near_ties(name, g, m, cfg, sd, data, eng, idx, it):
    """at the FIRST iteration whose arg-mins disagree with the reference's, every disagreeing row must
    be an fp64 near-tie.  Up to that iteration both sides saw the same correspondences, so the oracle (bit-identical to
    the reference on this fixture), teacher-forced with the reference's own indices, reproduces the reference's
    descriptors of that iteration; D = their fp64 distance matrix.

      reference picked k_r (fp32 evaluation, error <= e32), engine picked k_e on ITS descriptors, which differ from the
      oracle's by eps = max row |desc_engine - desc_oracle|_2 (measured below through the engine's own stage kernels):
      |D_engine - D| <= eta = 2 (eps_src + eps_ref) + e32   =>   D[k_e] - D[k_r] <= 2 eta + e32 is all a correct
      arg-min can do; a wrong arg-min (a bug) lands on a column whose distance exceeds the minimum by the typical
      top-2 spread, orders of magnitude more (printed)."""
    from oracle.network import OracleNet, to_torch
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    net = OracleNet(cfg, sd)
    d = to_torch(data)
    taps = {}
    forced = [torch.from_numpy(g["idx"][:, i].astype(np.int64)) for i in range(it + 1)]
    T_or, _ = net.register(d, it + 1, forced_idx=forced, taps=taps)
    ds, dr = taps["desc_src"][it], taps["desc_ref"][it]                       # [1,64,J], [1,64,K] (the reference's)
    a64, b64 = ds.double()[0].T, dr.double()[0]
    bad = np.nonzero(idx[it] != g["idx"][0, it])[0]
    D = -2 * (a64[bad] @ b64) + (a64[bad] ** 2).sum(1)[:, None] + (b64 ** 2).sum(0)[None]
    dmin = D.min(1).values.numpy()
    d_e = D[torch.arange(len(bad)), torch.from_numpy(idx[it][bad].astype(np.int64))].numpy()
    d_r = D[torch.arange(len(bad)), torch.from_numpy(g["idx"][0, it][bad].astype(np.int64))].numpy()
    # the engine's descriptors of the same state, through its own stage kernels (C ABI)
    N = m["n"]
    feats = cu(np.concatenate([data["points_src"], data["points_ref"]], 0))
    xyz = cu(np.concatenate([data["points_src_xyz"], data["points_ref_xyz"]], 0))
    neigh = cu(np.concatenate([data["points_src_neigh_idx"], data["points_ref_neigh_idx"]], 0), torch.int32)
    sub = cu(np.concatenate([data["points_src_sub_idx"], data["points_ref_sub_idx"]], 0), torch.int32)
    interp = cu(np.concatenate([data["points_src_interp_idx"], data["points_ref_interp_idx"]], 0), torch.int32)
    feat, logits = eng.randla_forward("feat_extractor", feats, xyz, neigh, sub, interp)
    score, _ = eng.score(feat, logits, xyz, neigh)
    xyz0 = xyz[:, :N].contiguous().clone()
    if it > 0:   # src coordinates after the reference's first `it` updates
        Tc = torch.from_numpy(g["transforms"][0, it - 1]).double()
        p = torch.from_numpy(data["points_src_xyz"][0, :N]).double()
        xyz0[0] = (p @ Tc[:, :3].T + Tc[:, 3]).float().to(xyz0.device)
    desc = eng.aggregate(xyz0, feat, score).cpu().double()
    eps_s = float((desc[0] - a64).norm(dim=1).max())
    eps_r = float((desc[1] - b64.T).norm(dim=1).max())
    e32 = 2e-6                                                               # fp32 evaluation of a distance in [0, 4]
    eta = 2.0 * (eps_s + eps_r) + e32
    best, second, _ = OracleNet.nn_gap(ds, dr)
    spread = float((second - best)[0].median())
    print(f"[near-tie] {name} iteration {it}: {len(bad)} of {N} rows differ; fp64 excess of the engine's pick over the "
          f"row minimum: max {float((d_e - dmin).max()):.3e}, of the reference's pick: max {float((d_r - dmin).max()):.3e}; "
          f"descriptor deviation eps_src {eps_s:.2e} eps_ref {eps_r:.2e} -> admissible {2 * eta + e32:.2e}; "
          f"median top-2 gap of all rows {spread:.3e}")
    assert eps_s < 2e-4 and eps_r < 2e-4, "engine descriptors drifted from the reference's"
    assert np.all(d_r - dmin <= e32 * (1.0 + np.abs(dmin))), "the reference's own pick is not an fp64 near-minimum"
    assert np.all(d_e - d_r <= (2 * eta + e32) * (1.0 + np.abs(dmin))), "a disagreeing row is not a near-tie"
    # and the reference's bar wherever it is meaningful: most flips are ties at fp32 resolution
    tight = d_e - dmin <= 1e-6 * (1.0 + np.abs(dmin))
    print(f"[near-tie] {int(tight.sum())} of {len(bad)} flips within 1e-6 (1 + |d|) of the fp64 minimum")
    # measured on MI355X: every flip on every fixture sits within 4e-7 of the fp64 minimum (one row per case), so SURVEY's
    # own bar holds as written; the derived bound above stays as the principled ceiling
    assert tight.all(), "a disagreeing row is further than 1e-6 (1 + |d|) from the fp64 minimum"


def test_register_with_supplied_pyramids_and_batch():
    """Caller-supplied int64 pyramids (the reference's data dict) and a batch of 2 pairs
    give the same answer as device-built pyramids / single pairs (bitwise)."""
    g, m, cfg, sd, data = build_case("e2e_n2048_s3")
    g2, m2, _, _, data2 = build_case("e2e_n2048_s4_sep")   # different weights in the fixture; only its points are used
    eng = engine_for(cfg, sd, "e2e_n2048_s3")
    src = cu(np.concatenate([data["points_src"], data2["points_src"]], 0))
    ref = cu(np.concatenate([data["points_ref"], data2["points_ref"]], 0))
    both = eng.register(src, ref, 3)
    one = eng.register(src[:1], ref[:1], 3)
    two = eng.register(src[1:], ref[1:], 3)
    assert torch.equal(both["transforms"][0], one["transforms"][0])
    assert torch.equal(both["transforms"][1], two["transforms"][0])
    assert torch.equal(both["idx"][:, 0], one["idx"][:, 0])
    pyr = {k: cu(v) for k, v in data.items() if k.endswith(("_xyz", "_idx"))}
    sup = eng.register(src[:1], ref[:1], 3, pyramids=pyr)
    assert torch.equal(sup["transforms"], one["transforms"])
    assert torch.equal(sup["idx"], one["idx"])


def test_determinism_and_full_size_properties():
    """BASELINE full size (5000-pt pairs): run twice -> bitwise identical; the
    cumulative transforms are rotations; the identity-alignment property:
    registering a cloud against itself with forced identity correspondences
    returns the identity pose."""
    from deepsir_amd.arch import NetConfig
    from deepsir_amd.synth import make_batch
    from deepsir_amd.weights import generate_state_dict
    cfg = NetConfig(feat_len=3)
    sd = generate_state_dict(cfg, 0)
    eng = engine_for(cfg, sd, "w0")
    b = make_batch(5000, [11, 12], 3)
    src, ref = cu(b["points_src"]), cu(b["points_ref"])
    o1 = eng.register(src, ref, 5)
    o2 = eng.register(src, ref, 5)
    assert torch.equal(o1["transforms"], o2["transforms"])
    assert torch.equal(o1["idx"], o2["idx"])
    assert torch.equal(o1["logits"], o2["logits"])
    T = o1["transforms"].cpu().numpy().astype(np.float64)
    for R in T.reshape(-1, 3, 4)[:, :, :3]:
        np.testing.assert_allclose(R @ R.T, np.eye(3), atol=2e-5)
        assert np.linalg.det(R) > 0.999
    ident = torch.arange(5000, dtype=torch.int32, device=src.device)[None, None].expand(5, 2, 5000).contiguous()
    o3 = eng.register(src, src, 5, forced_idx=ident)
    Ti = o3["transforms"].cpu().numpy()
    eye = np.tile(np.eye(3, 4, dtype=np.float32), (2, 5, 1, 1))
    np.testing.assert_allclose(Ti, eye, atol=5e-6)


def test_graph_replay_is_bitwise_identical():
    """dsir_enable_graph: captured-and-replayed launches give the same bits as eager launches,
    also after the inputs in the (same) buffers change."""
    from deepsir_amd.arch import NetConfig
    from deepsir_amd.engine import Engine
    from deepsir_amd.synth import make_batch
    from deepsir_amd.weights import generate_state_dict
    cfg = NetConfig(feat_len=3)
    eng = Engine(cfg, 0, max_points=2048, max_pairs=2)
    eng.load_state_dict(generate_state_dict(cfg, 0))
    b1, b2 = make_batch(2048, [31, 32], 3), make_batch(2048, [33, 34], 3)
    src, ref = cu(b1["points_src"]), cu(b1["points_ref"])
    eager1 = {k: v.clone() for k, v in eng.register(src, ref, 5).items() if isinstance(v, torch.Tensor)}
    eng.enable_graph(True)
    out = eng.register(src, ref, 5)                       # capture + first replay
    keep = {k: v for k, v in out.items() if isinstance(v, torch.Tensor)}
    for k in eager1:
        assert torch.equal(eager1[k], keep[k]), k
    src.copy_(cu(b2["points_src"])); ref.copy_(cu(b2["points_ref"]))
    out2 = eng.register(src, ref, 5, out=keep)            # pure replay on new data in the same buffers
    eng.enable_graph(False)
    eager2 = eng.register(src, ref, 5)
    for k in eager1:
        assert torch.equal(eager2[k], out2[k]), k
    assert not torch.equal(eager1["transforms"], eager2["transforms"])
    eng.close()


def test_engine_pool_matches_single_engine():
    """Two engines on two HIP streams registering halves of a batch concurrently == one engine, bitwise."""
    from deepsir_amd.arch import NetConfig
    from deepsir_amd.engine import Engine, EnginePool
    from deepsir_amd.synth import make_batch
    from deepsir_amd.weights import generate_state_dict
    cfg = NetConfig(feat_len=3)
    sd = generate_state_dict(cfg, 0)
    b = make_batch(2048, [41, 42, 43], 3)
    src, ref = cu(b["points_src"]), cu(b["points_ref"])
    one = Engine(cfg, 0, max_points=2048, max_pairs=3)
    one.load_state_dict(sd)
    pool = EnginePool(cfg, 0, max_points=2048, max_pairs=3, streams=2)
    pool.load_state_dict(sd)
    a = one.register(src, ref, 5)
    p = pool.register(src, ref, 5)
    for k in ("transforms", "idx", "logits", "pt_ref_new", "invalid"):
        assert torch.equal(a[k], p[k]), k
    q = pool.register(src, ref, 5, want_aux=False)
    assert torch.equal(q["transforms"], a["transforms"])
    one.close(); pool.close()


def test_nn_match_properties_full_size():
    """5000 x 5000 and ragged 4999 x 5003: exact agreement with a float64 arg-min
    wherever the fp64 top-2 gap exceeds fp32 resolution; self-match is the identity."""
    from deepsir_amd.arch import NetConfig
    from deepsir_amd.weights import generate_state_dict
    cfg = NetConfig(feat_len=3)
    eng = engine_for(cfg, generate_state_dict(cfg, 0), "w0")
    rng = np.random.default_rng(5)
    for J, K in ((5000, 5000), (4999, 5003), (1, 64), (70, 63)):
        a = rng.standard_normal((2, J, 64)).astype(np.float32)
        b = rng.standard_normal((2, K, 64)).astype(np.float32)
        a /= np.linalg.norm(a, axis=2, keepdims=True)
        b /= np.linalg.norm(b, axis=2, keepdims=True)
        idx = eng.nn_match(cu(a), cu(b)).cpu().numpy()
        for p in range(2):
            d = -2 * a[p].astype(np.float64) @ b[p].astype(np.float64).T + (b[p].astype(np.float64) ** 2).sum(1)[None]
            ref = d.argmin(1)
            part = np.partition(d, 1, axis=1) if K > 1 else d
            gap = part[:, 1] - part[:, 0] if K > 1 else np.ones(J)
            clear = gap > 1e-5
            assert np.array_equal(idx[p][clear], ref[clear])
            got = np.take_along_axis(d, idx[p][:, None].astype(np.int64), 1)[:, 0]
            assert np.all(got - d.min(1) < 1e-5)
    a = rng.standard_normal((1, 3000, 64)).astype(np.float32)
    a /= np.linalg.norm(a, axis=2, keepdims=True)
    idx = eng.nn_match(cu(a), cu(a)).cpu().numpy()[0]
    assert np.array_equal(idx, np.arange(3000))
    # exact duplicates in ref: the lower index wins (torch.min / oracle tie rule)
    b = np.concatenate([a[0, :100], a[0, :100]], 0)[None]
    idx = eng.nn_match(cu(a[:, :100]), cu(b)).cpu().numpy()[0]
    assert np.array_equal(idx, np.arange(100))


def test_network_dropin_api():
    """The reference's host API: Network(args), load_state_dict, .cuda(), .eval(), net(data, opt)."""
    import argparse
    from deepsir_amd.model import Network
    from deepsir_amd.weights import to_torch_state_dict
    g, m, cfg, sd, data = build_case("e2e_n2048_s3")
    args = argparse.Namespace(pipeline="align", num_sub=-1, num_knn=16, out_feat_dim=64, clip_weight_thresh=0.0,
                              feat_len=3, d_out=[16, 64, 128, 256], num_points=2048, sub_sampling_ratio=[4, 4, 4, 4],
                              use_ppf=False)
    net = Network(args)
    assert len(net.state_dict()) == 370
    net.load_state_dict(to_torch_state_dict(sd))
    net = net.cuda().eval()
    d = {k: cu(v) for k, v in data.items()}
    transforms, ep = net(d, (5, True))
    assert len(transforms) == 5 and tuple(transforms[0].shape) == (1, 3, 4) and transforms[0].is_cuda
    assert tuple(ep["pred_pairs"][0].shape) == (1, 2048, 2) and ep["pred_pairs"][0].dtype == torch.int32
    assert not ep["pred_pairs"][0].is_cuda and tuple(ep["perm_matrices"][0].shape) == (1, 2048)
    assert bool(ep["invalid_gradient"]) is False          # read lazily from the device (deepsir_amd/model.py, _LazyFlag)
    agree = (ep["pred_pairs"][0][0, :, 1].numpy() == g["idx"][0, 0]).mean()
    assert agree > 0.99
    if all((ep["pred_pairs"][i][0, :, 1].numpy() == g["idx"][0, i]).all() for i in range(5)):
        assert_pose_close(torch.stack(transforms, 1).cpu().numpy(), g["transforms"], 1e-4, 1e-4)
    with pytest.raises(RuntimeError):
        net.load_state_dict({"bogus": torch.zeros(1)})


@pytest.mark.parametrize("n_src,n_ref,seed", [(1357, 1357, 21), (1100, 1999, 22), (3001, 2048, 23)])
def test_ragged_sizes_vs_oracle(n_src, n_ref, seed):
    """Sizes that are multiples of nothing (row tiles of 16, GEMM blocks of 64/128, 4-wide vectors) and unequal src / ref
    clouds (separate pyramids, no joint batch): the engine's correspondences forced into the oracle, poses within 1e-4;
    arg-min checked against the oracle's own descriptors where the fp64 gap is clear."""
    from deepsir_amd.arch import NetConfig
    from deepsir_amd.engine import Engine
    from deepsir_amd.synth import make_pair
    from deepsir_amd.weights import generate_state_dict
    from oracle.knn import add_pyramids
    from oracle.network import OracleNet, to_torch
    cfg = NetConfig(feat_len=3)
    sd = generate_state_dict(cfg, 2, "separated")
    a, b = make_pair(n_src, seed, 3), make_pair(n_ref, seed + 100, 3)
    raw = {"points_src": a["points_src"], "points_ref": b["points_ref"]}
    eng = Engine(cfg, 0, max_points=max(n_src, n_ref), max_pairs=1)
    eng.load_state_dict(sd)
    out = eng.register(cu(raw["points_src"]), cu(raw["points_ref"]), 3)
    assert not bool(out["invalid"].any())
    data = to_torch(add_pyramids(raw, cfg.num_knn, cfg.sub_sampling_ratio))
    idx = out["idx"].cpu()
    assert int(idx.min()) >= 0 and int(idx.max()) < n_ref and tuple(idx.shape) == (3, 1, n_src)
    taps = {}
    T_forced, ep = OracleNet(cfg, sd).register(data, 3, forced_idx=[idx[i].long() for i in range(3)], taps=taps)
    assert_pose_close(out["transforms"].cpu().numpy()[0], np.stack([t.numpy()[0] for t in T_forced]), 1e-4, 1e-4,
                      f"ragged {n_src}x{n_ref}")
    np.testing.assert_allclose(out["logits"].cpu().numpy()[:, 0], np.stack([l.numpy()[0] for l in ep["perm_matrices"]]),
                               rtol=2e-3, atol=2e-3)
    best, second, arg = OracleNet.nn_gap(taps["desc_src"][0], taps["desc_ref"][0])
    clear = ((second - best) > 1e-4 * (1.0 + best.abs()))[0].numpy()
    assert clear.mean() > 0.5
    assert np.array_equal(idx[0, 0].numpy()[clear], arg[0].numpy()[clear])
    eng.close()


def test_kitti_shaped_16k_feat4_vs_oracle():
    """BASELINE config 3 shape: 16384-point clouds with reflectance (feat_len 4), KITTI-like extent.
    The engine's own correspondences are forced into the CPU oracle: (R,t) of every iteration within 1e-4."""
    from deepsir_amd.arch import NetConfig
    from deepsir_amd.engine import Engine
    from deepsir_amd.synth import make_pair
    from deepsir_amd.weights import generate_state_dict
    from oracle.knn import add_pyramids
    from oracle.network import OracleNet, to_torch
    cfg = NetConfig(feat_len=4)
    sd = generate_state_dict(cfg, 1)
    raw = make_pair(16384, 77, 4, shape="kitti")
    eng = Engine(cfg, 0, max_points=16384, max_pairs=1)
    eng.load_state_dict(sd)
    out = eng.register(cu(raw["points_src"]), cu(raw["points_ref"]), 3)
    data = to_torch(add_pyramids(raw, cfg.num_knn, cfg.sub_sampling_ratio))
    # pyramid built on device == oracle pyramid (bit-exact), checked through a second call with supplied pyramids
    sup = eng.register(cu(raw["points_src"]), cu(raw["points_ref"]), 3, pyramids={k: v.cuda() for k, v in data.items() if k.endswith(("_xyz", "_idx"))})
    assert torch.equal(sup["transforms"], out["transforms"]) and torch.equal(sup["idx"], out["idx"])
    torch.set_num_threads(max(1, min(16, torch.get_num_threads())))
    idx = out["idx"].cpu()
    T_forced, ep = OracleNet(cfg, sd).register(data, 3, forced_idx=[idx[i].long() for i in range(3)])
    assert_pose_close(out["transforms"].cpu().numpy()[0], np.stack([t.numpy()[0] for t in T_forced]), 1e-4, 1e-4, "16k")
    np.testing.assert_allclose(out["logits"].cpu().numpy()[:, 0], np.stack([l.numpy()[0] for l in ep["perm_matrices"]]), rtol=2e-3, atol=2e-3)
    eng.close()


def test_64k_partial_overlap_properties():
    """BASELINE config 5 shape: 65536-point clouds, 50 % overlap + jitter.  Size-independent properties:
    bitwise determinism, rotations stay in SO(3), arg-min indices in range, finite logits."""
    from deepsir_amd.arch import NetConfig
    from deepsir_amd.engine import Engine
    from deepsir_amd.synth import make_pair
    from deepsir_amd.weights import generate_state_dict
    cfg = NetConfig(feat_len=3)
    eng = Engine(cfg, 0, max_points=65536, max_pairs=1)
    eng.load_state_dict(generate_state_dict(cfg, 0))
    raw = make_pair(65536, 5, 3, partial_overlap=True)
    src, ref = cu(raw["points_src"]), cu(raw["points_ref"])
    a = eng.register(src, ref, 2)
    b = eng.register(src, ref, 2)
    assert torch.equal(a["transforms"], b["transforms"]) and torch.equal(a["idx"], b["idx"])
    T = a["transforms"].cpu().numpy().astype(np.float64)[0]
    for R in T[:, :, :3]:
        np.testing.assert_allclose(R @ R.T, np.eye(3), atol=5e-5)
    idx = a["idx"].cpu().numpy()
    assert idx.min() >= 0 and idx.max() < 65536 and np.isfinite(a["logits"].cpu().numpy()).all()
    assert int(a["invalid"][0]) == 0
    eng.close()


def test_error_paths_fail_loudly():
    """Unsupported / inconsistent arguments are rejected with a message (never a silent fallback, never a fault)."""
    from deepsir_amd.arch import NetConfig
    from deepsir_amd.engine import Engine, EngineError
    from deepsir_amd.weights import generate_state_dict
    cfg = NetConfig(feat_len=3)
    sd = generate_state_dict(cfg, 0)
    eng = Engine(cfg, 0, max_points=2048, max_pairs=2)
    x = torch.rand(1, 2048, 3, device=_dev())
    with pytest.raises(EngineError, match="not finalized"):
        eng.register(x, x, 2)                                          # weights missing
    bad = dict(sd); bad["mlp_proj.0.weight"] = np.zeros((64, 32), np.float32)
    with pytest.raises(EngineError, match="size mismatch"):
        eng.load_state_dict(bad)
    short = dict(sd); short.pop("mlp_att.12.bias")
    with pytest.raises(EngineError, match="missing keys"):
        eng.load_state_dict(short)
    eng.load_state_dict(sd)
    with pytest.raises(EngineError, match="max_points"):
        eng.register(torch.rand(1, 4096, 3, device=_dev()), torch.rand(1, 4096, 3, device=_dev()), 2)
    with pytest.raises(EngineError, match="pairs="):
        eng.register(torch.rand(3, 2048, 3, device=_dev()), torch.rand(3, 2048, 3, device=_dev()), 2)
    with pytest.raises(EngineError, match="n_iter|null argument"):
        eng.register(x, x, 0)
    with pytest.raises(EngineError, match="channels"):
        eng.register(torch.rand(1, 2048, 4, device=_dev()), torch.rand(1, 2048, 4, device=_dev()), 2)
    with pytest.raises(EngineError, match="CUDA tensor"):
        eng.register(x.cpu(), x.cpu(), 2)
    with pytest.raises(EngineError, match="too small|at least"):
        eng.register(torch.rand(1, 512, 3, device=_dev()), torch.rand(1, 512, 3, device=_dev()), 2)
    with pytest.raises(EngineError, match="feat pipeline"):
        eng.forward_pair(x, x, 128)                                    # top-k selection on an align context
    with pytest.raises(EngineError, match="bad arguments"):
        eng.icp_refine(x, x, torch.eye(3, 4, device=_dev())[None], -1.0)
    # after all the rejected calls the context is still healthy
    out = eng.register(x, x, 2)
    assert torch.isfinite(out["transforms"]).all()
    eng.close()


@pytest.mark.parametrize("kind", ["random", "near_matches", "duplicates", "near_ties", "clustered"])
def test_screened_argmin_equals_exhaustive(kind):
    """csrc/nn_screen.hip (fp16-split screening under a rigorous bound, exact fp32 decision) must return the arg-min of
    the exhaustive exact-fp32 kernel on every row — including exact duplicates (tie -> lower index), distances closer than
    fp32 resolution of the screening values, and candidate-list overflow (exhaustive fallback)."""
    from deepsir_amd.arch import NetConfig
    from deepsir_amd.engine import Engine
    rng = np.random.Generator(np.random.Philox(key=["random", "near_matches", "duplicates", "near_ties", "clustered"].index(kind) + 77))
    P, J, K = 2, 3001, 2777
    a = rng.standard_normal((P, J, 64)).astype(np.float32)
    b = rng.standard_normal((P, K, 64)).astype(np.float32)
    if kind == "near_matches":
        b[:, :2000] = a[:, :2000] + rng.standard_normal((P, 2000, 64)).astype(np.float32) * 1e-3
    elif kind == "duplicates":
        b[:, 100:600] = b[:, 1100:1600]                     # exact ties between far-apart indices
        b[:, 2000:2040] = b[:, 1999:2000]                   # 41 identical columns: more than the candidate cap
        a[:, :40] = b[:, 1999:2000]
    elif kind == "near_ties":
        base = rng.standard_normal((P, 1, 64)).astype(np.float32)
        b[:, :1500] = base + rng.standard_normal((P, 1500, 64)).astype(np.float32) * 3e-6   # 1500 columns within ~1e-5
        a[:, :500] = base + rng.standard_normal((P, 500, 64)).astype(np.float32) * 3e-6
    elif kind == "clustered":
        centers = rng.standard_normal((P, 8, 64)).astype(np.float32)
        b = centers[:, rng.integers(0, 8, K)][np.arange(P)[:, None], np.arange(K)[None, :]] if False else \
            centers[:, rng.integers(0, 8, K)] + rng.standard_normal((P, K, 64)).astype(np.float32) * 1e-2
        a = centers[:, rng.integers(0, 8, J)] + rng.standard_normal((P, J, 64)).astype(np.float32) * 1e-2
    a /= np.linalg.norm(a, axis=2, keepdims=True)
    b /= np.linalg.norm(b, axis=2, keepdims=True)
    eng = Engine(NetConfig(), 0, max_points=4096, max_pairs=2)
    ta, tb = cu(a.astype(np.float32)), cu(b.astype(np.float32))
    exact = eng.nn_match(ta, tb).cpu().numpy()
    scr, (ncand, nexh) = eng.nn_match_screened(ta, tb)
    scr = scr.cpu().numpy()
    print(f"[screen] {kind}: {ncand / (P * J):.2f} candidates per row, {nexh} of {P * J} rows scanned exhaustively")
    assert np.array_equal(scr, exact)
    if kind in ("duplicates", "near_ties"):
        assert nexh > 0            # the overflow path is exercised
    eng.close()


def test_register_screened_equals_exhaustive(tmp_path):
    """A whole registration large enough for the screened arg-min (P*J*K >= 2e8, csrc/engine.hip) must produce the same
    bits - correspondences, inlier logits, transforms - as the same registration with the exhaustive exact-fp32 kernel
    (DSIR_NO_SCREEN=1; the switch is read once per process, hence two child processes)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for name, extra in (("screened", {}), ("exhaustive", {"DSIR_TUNING": "1", "DSIR_NO_SCREEN": "1"})):
        out = str(tmp_path / f"{name}.npz")
        env = dict(os.environ, **extra)
        env.pop("DSIR_SCREEN_OVF_MIN", None)
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "register_dump.py"), out, "10", "5000", "4"], env=env,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(out))
    a, b = outs
    for k in ("idx", "logits", "transforms"):
        assert np.array_equal(a[k], b[k]), f"{k} differs between the screened and the exhaustive arg-min"


@pytest.mark.parametrize("pairs,points", [(10, 5000), (1, 2048), (3, 1357)])
def test_search_operands_from_the_aggregation_epilogue_are_the_same_bits(tmp_path, pairs, points):
    """Round 5: the aggregation chain's epilogue writes what the descriptor search needs of the descriptors - |desc|^2, the screening's
    fp16 operand pair, the exhaustive search's preset slots - instead of a second pass over them (split_norm_kernel / sqnorm_kernel).
    Same arithmetic in the same order: a registration equals, bit for bit, the one whose search prepares its operands itself
    (DSIR_NO_AGG_EXTRAS=1; read once per process, hence two processes) - in the screened regime (10 x 5000) and in the exhaustive
    one (small launches), ragged row counts included."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for name, extra in (("fused", {}), ("separate", {"DSIR_TUNING": "1", "DSIR_NO_AGG_EXTRAS": "1"})):
        out = str(tmp_path / f"{name}.npz")
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "register_dump.py"), out, str(pairs), str(points), "4"],
                           env=dict(os.environ, **extra), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(out))
    a, b = outs
    assert int(a["screened_searches"]) == int(b["screened_searches"]) == (4 if pairs == 10 else 0)
    for k in ("idx", "logits", "transforms"):
        assert np.array_equal(a[k], b[k]), f"{k} differs"


@pytest.mark.parametrize("pairs,points", [(1, 5000), (2, 2048), (1, 1357), (3, 4100), (9, 1357), (8, 5000), (20, 5000), (2, 16384)])
def test_pyramid_launch_merges_are_the_same_bits(tmp_path, pairs, points):
    """Round 5: the KNN pyramid as three launches (the grids of the large levels; their searches; the interpolation searches of all
    levels with the 16-NN of the levels without a grid; plus one per interpolation search that walks a grid) instead of a chain of
    ten, and the pose solve with its points held in registers between its passes.  Same kernel bodies, same per-thread order: the
    pyramid and the whole registration equal, bit for bit, those of the separate launches / the streaming solve
    (DSIR_NO_PYRAMID_MERGE=1, DSIR_KABSCH_STREAM=1; read once per process, hence two processes).  The cases cover every form a
    level takes: four lanes / one lane per query on the grid, one wave / one lane per query without it, interpolation by brute force /
    through the grid (20 x 5000, 2 x 16384), clouds beyond the register-resident solve (2 x 16384)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for name, extra in (("merged", {}), ("separate", {"DSIR_TUNING": "1", "DSIR_NO_PYRAMID_MERGE": "1", "DSIR_KABSCH_STREAM": "1"})):
        out = str(tmp_path / f"{name}.npz")
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "register_dump.py"), out, str(pairs), str(points), "4"],
                           env=dict(os.environ, **extra), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(out))
    a, b = outs
    for k in ("neigh", "sub", "interp", "idx", "logits", "transforms"):
        assert np.array_equal(a[k], b[k]), f"{k} differs"


def test_screened_argmin_out_of_domain_inputs():
    """Elements beyond the fp16 range or not finite void the screening bound: split16 raises its flag and every pair is
    searched by the exhaustive kernel, so the result still equals dsir_nn_match."""
    from deepsir_amd.arch import NetConfig
    from deepsir_amd.engine import Engine
    rng = np.random.Generator(np.random.Philox(key=91))
    P, J, K = 2, 1500, 1700
    a = rng.standard_normal((P, J, 64)).astype(np.float32)
    b = rng.standard_normal((P, K, 64)).astype(np.float32)
    b[1, 17, 3] = 1.0e5                                        # beyond fp16
    a[0, 5, :] *= 1.0e-9                                       # below the fp16 subnormals: covered by the bound's constant
    eng = Engine(NetConfig(), 0, max_points=2048, max_pairs=2)
    ta, tb = cu(a), cu(b)
    exact = eng.nn_match(ta, tb).cpu().numpy()
    scr, (ncand, nexh) = eng.nn_match_screened(ta, tb)
    assert np.array_equal(scr.cpu().numpy(), exact)
    assert nexh == P * J                                       # every row went to the exhaustive kernel
    eng.close()


@pytest.mark.parametrize("P,J,K", [(1, 1, 1), (3, 17, 63), (2, 257, 65), (1, 64, 64), (2, 255, 129), (1, 1000, 4097), (5, 300, 20),
                                   (1, 4096, 16), (2, 513, 1025)])
def test_screened_argmin_shapes(P, J, K):
    """Ragged shapes of the screened arg-min (rows not a multiple of the 256-row workgroup, columns not a multiple of the
    64-column tile, fewer columns than one tile, a single row / column): equal to the exhaustive kernel."""
    from deepsir_amd.arch import NetConfig
    from deepsir_amd.engine import Engine
    rng = np.random.Generator(np.random.Philox(key=1000 * P + 10 * J + K))
    a = rng.standard_normal((P, J, 64)).astype(np.float32)
    b = rng.standard_normal((P, K, 64)).astype(np.float32)
    if K > 3:
        b[:, K // 2] = b[:, K // 3]                           # an exact tie between two columns
    m = min(J, K)
    a[:, :m:5] = b[:, :m:5] + 1e-4 * rng.standard_normal((P, len(range(0, m, 5)), 64)).astype(np.float32)   # near matches
    a /= np.linalg.norm(a, axis=2, keepdims=True)
    b /= np.linalg.norm(b, axis=2, keepdims=True)
    eng = Engine(NetConfig(), 0, max_points=max(1024, J, K), max_pairs=P)
    ta, tb = cu(a), cu(b)
    exact = eng.nn_match(ta, tb).cpu().numpy()
    scr, _ = eng.nn_match_screened(ta, tb)
    assert np.array_equal(scr.cpu().numpy(), exact)
    eng.close()


@pytest.mark.parametrize("n,clouds,extent,wseed,variant", [(5000, 6, 3.0, 0, "plain"), (1357, 2, 3.0, 2, "separated"), (16384, 2, 50.0, 1, "plain"),
                                                           (700, 1, 3.0, 3, "plain")])
def test_agg_chain_split_matches_fp32_chain(n, clouds, extent, wseed, variant):
    """csrc/agg_chain_h.hip (the aggregation chain's wide layers as fp16-split products: three fp16 MFMAs per fp32 product,
    fp32 accumulation) against csrc/agg_chain.hip (exact-fp32 MFMA, bit-identical to the layer-by-layer launches) on the same
    inputs through dsir_aggregate: unit descriptors within 2e-6 (measured ~1e-7: the size of fp32's own summation-order noise
    on these 64..256-term sums, and 10 x below the 2e-5 at which the descriptors are compared with the reference's,
    test_stages_vs_oracle_and_golden).  Both large (128 points per block) and small launches, ragged n."""
    from deepsir_amd.arch import NetConfig
    from deepsir_amd.engine import Engine
    from deepsir_amd.weights import generate_state_dict
    cfg = NetConfig(feat_len=3)
    eng = Engine(cfg, 0, max_points=max(n, 1024), max_pairs=max(1, (clouds + 1) // 2))
    eng.load_state_dict(generate_state_dict(cfg, wseed, variant))
    rng = np.random.Generator(np.random.Philox(key=n + clouds))
    xyz = cu(rng.uniform(-extent if extent > 3 else 0.0, extent, (clouds, n, 3)).astype(np.float32))
    feat = cu(rng.standard_normal((clouds, n, 64)).astype(np.float32))
    score = cu(rng.uniform(0, 1, (clouds, n)).astype(np.float32))
    eng.enable_agg_split(False)
    ref = eng.aggregate(xyz, feat, score).clone()
    eng.enable_agg_split(True)
    got = eng.aggregate(xyz, feat, score)
    d = (got.double() - ref.double()).abs()
    norms = got.double().norm(dim=2)
    print(f"[agg-split] n {n} x {clouds} clouds ({variant}): max |d desc| {float(d.max()):.2e}, mean {float(d.mean()):.2e}, "
          f"row-L2 max {float((got.double() - ref.double()).norm(dim=2).max()):.2e}; |desc| in [{float(norms.min()):.7f}, {float(norms.max()):.7f}]")
    assert torch.isfinite(got).all()
    assert float(d.max()) <= 2e-6
    assert not torch.equal(got, ref)          # different arithmetic: identical bits would mean the switch is dead
    eng.close()


@pytest.mark.parametrize("name,which", [("stage_n1024_s1", "feat_extractor"), ("e2e_n5000_s5", "feat_extractor"), ("e2e_n2048_s6_f4", "inlier_model")])
def test_head_split_matches_fp32_head(name, which):
    """csrc/head_mlp_h.hip (mlp_out + fc_label as fp16-split products, three fp16 MFMAs per fp32 product) against
    csrc/head_mlp.hip (exact fp32, bit-identical to the four separate launches) through dsir_randla_forward: the 64-d
    feature and the logits within 1e-5 of their scale (measured ~1e-6), for the feature extractor (19 classes) and the
    inlier model (1 logit)."""
    g, m, cfg, sd, data = build_case(name)
    eng = engine_for(cfg, sd, name + "_head", max_points=max(8192, m["n"]))
    n = m["n"]
    if which == "feat_extractor":
        feats = cu(np.concatenate([data["points_src"], data["points_ref"]], 0))
    else:
        rng = np.random.Generator(np.random.Philox(key=n))
        feats = cu(rng.uniform(0, 3, (2, n, 6)).astype(np.float32))
    xyz = cu(np.concatenate([data["points_src_xyz"], data["points_ref_xyz"]], 0))
    neigh = cu(np.concatenate([data["points_src_neigh_idx"], data["points_ref_neigh_idx"]], 0), torch.int32)
    sub = cu(np.concatenate([data["points_src_sub_idx"], data["points_ref_sub_idx"]], 0), torch.int32)
    interp = cu(np.concatenate([data["points_src_interp_idx"], data["points_ref_interp_idx"]], 0), torch.int32)
    eng.enable_agg_split(False)
    f0, l0 = eng.randla_forward(which, feats, xyz, neigh, sub, interp)
    f0, l0 = f0.clone(), l0.clone()
    eng.enable_agg_split(True)
    f1, l1 = eng.randla_forward(which, feats, xyz, neigh, sub, interp)
    ef = float((f1.double() - f0.double()).abs().max() / f0.double().abs().max())
    el = float((l1.double() - l0.double()).abs().max() / l0.double().abs().max())
    print(f"[head-split] {name} {which}: feat rel err {ef:.2e} (scale {float(f0.abs().max()):.2f}), logits rel err {el:.2e} (scale {float(l0.abs().max()):.2f})")
    assert torch.isfinite(f1).all() and torch.isfinite(l1).all()
    assert ef <= 1e-5 and el <= 1e-5
    assert not torch.equal(l0, l1)            # different arithmetic: identical bits would mean the switch is dead

"""Adam fine-tune of the pose (reference test.py:159-207 + network/DGR.py:60-132, disabled in the
reference and not importable offline -> parity unpinned).  CPU: the oracle's own properties.  GPU: dsir_pose_finetune
against the oracle on the same matched points."""
import numpy as np
import pytest

from deepsir_amd.synth import make_pair, random_rotation
from oracle.finetune import ortho2rotation, transformation_finetune


def _case(n, seed, ang_deg=2.0, shift=0.03, noise=0.002, outliers=0.2):
    """Matched point pairs under a ground-truth pose with noise and a share of gross outliers, soft weights that mostly
    (not perfectly) tell them apart, and a perturbed initial pose: what the fine-tune sees after the last iteration."""
    rng = np.random.default_rng(seed)
    p = make_pair(n, seed, 3)
    src = p["points_src"][0].astype(np.float32)
    T_gt = p["transform_gt"][0].astype(np.float64)
    ref = (src.astype(np.float64) @ T_gt[:, :3].T + T_gt[:, 3] + rng.normal(0, noise, (n, 3)))
    bad = rng.random(n) < outliers
    ref[bad] = rng.uniform(0, 3, (int(bad.sum()), 3))
    w = np.where(bad, rng.uniform(0.0, 0.3, n), rng.uniform(0.5, 1.0, n)).astype(np.float32)
    ax = rng.standard_normal(3); ax /= np.linalg.norm(ax)
    a = np.deg2rad(ang_deg)
    Kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    dR = np.eye(3) + np.sin(a) * Kx + (1 - np.cos(a)) * Kx @ Kx
    T0 = np.hstack([dR @ T_gt[:, :3], (T_gt[:, 3] + rng.uniform(-shift, shift, 3))[:, None]]).astype(np.float32)
    return src, ref.astype(np.float32), w, T0, T_gt


def _rot_err(Ra, Rb):
    D = np.asarray(Ra, np.float64).T @ np.asarray(Rb, np.float64)
    v = 0.5 * np.array([D[2, 1] - D[1, 2], D[0, 2] - D[2, 0], D[1, 0] - D[0, 1]])
    return float(np.arctan2(np.linalg.norm(v), 0.5 * (np.trace(D) - 1.0)))


def _loss(src, ref, w, T, q):
    s = (((src.astype(np.float64) @ T[:, :3].T.astype(np.float64) + T[:, 3] - ref) / q) ** 2).sum(1)
    l = np.where(s < 1.0, 0.5 * s, 0.5 * (np.sqrt(s + np.finfo(np.float32).eps) - 0.5))
    return float((l * w).sum() / w.sum())


def test_ortho2rotation_is_a_rotation_and_keeps_an_orthonormal_input():
    import torch
    rng = np.random.default_rng(0)
    R = random_rotation(rng)
    six = torch.from_numpy(np.concatenate([R[:, 0], R[:, 1]])[None].astype(np.float32))
    np.testing.assert_allclose(ortho2rotation(six)[0].numpy(), R, atol=1e-6)
    M = ortho2rotation(torch.from_numpy(rng.standard_normal((5, 6)).astype(np.float32))).numpy().astype(np.float64)
    for m in M:
        np.testing.assert_allclose(m.T @ m, np.eye(3), atol=1e-5)
        assert np.linalg.det(m) > 0.999


def test_oracle_finetune_improves_the_pose():
    src, ref, w, T0, T_gt = _case(1500, 3)
    T, res = transformation_finetune(src, ref, T0, w, quantization_size=0.06)
    assert 20 <= res["iterations"] < 1000 and res["break_count"] >= 1
    # the loss floor is set by the outliers (robust, not squared, beyond one quantisation unit): the fit must reach the
    # loss of the ground-truth pose and pull the rotation error down by more than an order of magnitude
    l0, l1, lg = (_loss(src, ref, w, t.astype(np.float64), 0.06) for t in (T0, T, T_gt))
    assert l1 < l0 and l1 <= lg * 1.0005
    assert _rot_err(T[:, :3], T_gt[:, :3]) < 0.05 * _rot_err(T0[:, :3], T_gt[:, :3])
    np.testing.assert_allclose(T[:, :3].astype(np.float64).T @ T[:, :3].astype(np.float64), np.eye(3), atol=1e-5)
    # exact correspondences already at the optimum: the loss is below 1e-7 at once and nothing moves
    clean = (src.astype(np.float64) @ T_gt[:, :3].T + T_gt[:, 3]).astype(np.float32)
    T2, res2 = transformation_finetune(src, clean, T_gt.astype(np.float32), None, quantization_size=0.06)
    assert res2["iterations"] == 0 and res2["loss"] < 1e-7
    np.testing.assert_allclose(T2, T_gt, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("n,seeds,weighted", [(1500, (3, 4, 5), True), (5000, (6, 7), True), (1024, (8, 9), False)])
def test_gpu_finetune_matches_oracle(n, seeds, weighted):
    import torch
    from deepsir_amd.arch import NetConfig
    from deepsir_amd.engine import Engine
    cases = [_case(n, s) for s in seeds]
    eng = Engine(NetConfig(), 0, max_points=max(n, 1024), max_pairs=len(cases))
    src = torch.from_numpy(np.stack([c[0] for c in cases])).cuda()
    ref = torch.from_numpy(np.stack([c[1] for c in cases])).cuda()
    w = torch.from_numpy(np.stack([c[2] for c in cases])).cuda() if weighted else None
    T0 = torch.from_numpy(np.stack([c[3] for c in cases])).cuda()
    q = 0.06
    T, stats = eng.pose_finetune(src, ref, T0, weights=w, quantization_size=q)
    T, stats = T.cpu().numpy(), stats.cpu().numpy()
    for k, (s, r, wk, t0, t_gt) in enumerate(cases):
        To, res = transformation_finetune(s, r, t0, wk if weighted else None, quantization_size=q)
        ww = wk if weighted else np.ones(n, np.float32)
        l_hip, l_or, l_0 = _loss(s, r, ww, T[k].astype(np.float64), q), _loss(s, r, ww, To.astype(np.float64), q), _loss(s, r, ww, t0.astype(np.float64), q)
        print(f"[finetune] n={n} case {k}: iterations hip {int(stats[k, 0])} / oracle {res['iterations']}, loss {l_0:.5f} -> hip {l_hip:.6f} / "
              f"oracle {l_or:.6f}, pose diff {_rot_err(T[k][:, :3], To[:, :3]):.2e} rad {np.linalg.norm(T[k][:, 3] - To[:, 3]):.2e} m")
        # the same optimisation: same plateau (the stopping rule watches 1e-4 relative changes), same pose to well below
        # the correspondence noise, a rotation matrix, and the reported loss is the loss of the step before the last update
        assert abs(l_hip - l_or) <= 2e-3 * l_or + 1e-7
        assert _rot_err(T[k][:, :3], To[:, :3]) < 2e-3 and np.linalg.norm(T[k][:, 3] - To[:, 3]) < 2e-3
        assert abs(int(stats[k, 0]) - res["iterations"]) <= max(25, res["iterations"] // 4)
        assert l_hip < l_0 and l_hip <= _loss(s, r, ww, t_gt, q) * 1.0005
        np.testing.assert_allclose(T[k][:, :3].astype(np.float64).T @ T[k][:, :3].astype(np.float64), np.eye(3), atol=1e-5)
        assert abs(stats[k, 1] - l_hip) < 0.02 * l_hip + 1e-6 and 1 <= stats[k, 2] <= 20
    # sigmoid of logits on the fly == explicit weights
    if weighted:
        logit = torch.log(w / (1 - w).clamp_min(1e-6))
        T2, _ = eng.pose_finetune(src, ref, T0, weights=logit, weights_are_logits=True, quantization_size=q)
        assert float((T2.cpu() - torch.from_numpy(T)).abs().max()) < 2e-3
    # already optimal, exact correspondences: untouched, zero iterations
    clean = torch.from_numpy(np.stack([(c[0].astype(np.float64) @ c[4][:, :3].T + c[4][:, 3]).astype(np.float32) for c in cases])).cuda()
    Tg = torch.from_numpy(np.stack([c[4].astype(np.float32) for c in cases])).cuda()
    T3, st3 = eng.pose_finetune(src, clean, Tg, quantization_size=q)
    assert float((T3 - Tg).abs().max()) < 1e-6 and float(st3[:, 0].max()) == 0.0 and float(st3[:, 1].max()) < 1e-7
    eng.close()


@pytest.mark.gpu
def test_gpu_harness_pose_opt_tune():
    """inference_align(pose_opt='tune'): the fine-tuned pose is appended as the last entry (reference test.py:406-408) and
    equals the oracle's fine-tune of the same correspondences."""
    import argparse
    import torch
    from deepsir_amd.arch import NetConfig
    from deepsir_amd.harness import inference_align
    from deepsir_amd.model import Network
    from deepsir_amd.weights import generate_state_dict, to_torch_state_dict
    args = argparse.Namespace(pipeline="align", num_sub=-1, num_knn=16, out_feat_dim=64, clip_weight_thresh=0.0, feat_len=3,
                              d_out=[16, 64, 128, 256], num_points=2048, sub_sampling_ratio=[4, 4, 4, 4], use_ppf=False)
    net = Network(args)
    net.load_state_dict(to_torch_state_dict(generate_state_dict(NetConfig(), 0)))
    net = net.cuda().eval()
    pairs = [make_pair(2048, s, 3) for s in (51, 52)]
    plain, _ = inference_align(pairs, net, 3, batch=2)
    tuned, _ = inference_align(pairs, net, 3, batch=2, pose_opt="tune", voxel_size=0.05)
    assert plain.shape == tuned.shape == (2, 4, 3, 4)
    assert np.array_equal(plain[:, :3], tuned[:, :3])
    for k, p in enumerate(pairs):
        d = {kk: torch.from_numpy(v[None] if v.ndim == 2 else v).cuda() for kk, v in p.items() if kk.startswith("points_")}
        _, ep = net({"points_src": d["points_src"], "points_ref": d["points_ref"]}, (3, True))
        To, _ = transformation_finetune(ep["pt_src"][0].cpu().numpy(), ep["pt_ref_new"][0].cpu().numpy(), plain[k, 2],
                                        torch.sigmoid(ep["perm_matrices"][-1][0]).cpu().numpy(), quantization_size=0.1)
        assert _rot_err(tuned[k, 3][:, :3], To[:, :3]) < 5e-3 and np.linalg.norm(tuned[k, 3][:, 3] - To[:, 3]) < 5e-3

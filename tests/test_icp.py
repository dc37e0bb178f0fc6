"""Point-to-point ICP refinement (reference test.py:241-258, open3d registration_icp — disabled in
the reference and unpinned): the oracle's own sanity on CPU, the HIP path (dsir_icp_refine) against the oracle on GPU."""
import numpy as np
import pytest

from deepsir_amd.synth import make_pair, random_rotation
from oracle.icp import icp


def _perturbed(n, seed, ang_deg=3.0, shift=0.05, noise=0.0):
    p = make_pair(n, seed, 3)
    rng = np.random.default_rng(seed)
    T_gt = p["transform_gt"][0].astype(np.float64)
    ax = rng.standard_normal(3); ax /= np.linalg.norm(ax)
    a = np.deg2rad(ang_deg)
    Kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    dR = np.eye(3) + np.sin(a) * Kx + (1 - np.cos(a)) * Kx @ Kx
    T0 = np.hstack([dR @ T_gt[:, :3], (dR @ T_gt[:, 3] + rng.uniform(-shift, shift, 3))[:, None]])
    src, ref = p["points_src"][0].copy(), p["points_ref"][0].copy()
    if noise:
        ref = (ref + rng.normal(0, noise, ref.shape)).astype(np.float32)
    return src, ref, T0.astype(np.float32), T_gt


def _rot_err(Ra, Rb):
    """rounding-robust rotation distance (rad): atan2(|vee(skew(D))|, (tr D - 1) / 2), D = Ra^T Rb in fp64"""
    D = np.asarray(Ra, np.float64).T @ np.asarray(Rb, np.float64)
    v = 0.5 * np.array([D[2, 1] - D[1, 2], D[0, 2] - D[2, 0], D[1, 0] - D[0, 1]])
    return float(np.arctan2(np.linalg.norm(v), 0.5 * (np.trace(D) - 1.0)))


def test_oracle_icp_recovers_ground_truth():
    src, ref, T0, T_gt = _perturbed(1024, 5)
    T, fitness, rmse, converged, iters = icp(src, ref, T0, 0.1)
    assert converged and 1 <= iters <= 30 and fitness > 0.99 and rmse < 1e-4
    assert _rot_err(T[:, :3], T_gt[:, :3]) < 1e-4 and np.linalg.norm(T[:, 3] - T_gt[:, 3]) < 1e-4
    # a radius that admits nothing: no correspondences, identity updates, init returned
    T2, f2, r2, c2, i2 = icp(src, ref + 100.0, T0, 0.05)
    assert f2 == 0.0 and np.allclose(T2, T0, atol=1e-6) and c2


@pytest.mark.gpu
@pytest.mark.parametrize("n,seed,noise", [(1024, 7, 0.0), (2500, 8, 0.004), (1357, 9, 0.002)])
def test_gpu_icp_matches_oracle(n, seed, noise):
    import torch
    from deepsir_amd.arch import NetConfig
    from deepsir_amd.engine import Engine
    cases = [_perturbed(n, seed + 10 * k, noise=noise) for k in range(3)]
    eng = Engine(NetConfig(), 0, max_points=max(n, 1024), max_pairs=3)
    src = torch.from_numpy(np.stack([c[0] for c in cases])).cuda()
    ref = torch.from_numpy(np.stack([c[1] for c in cases])).cuda()
    T0 = torch.from_numpy(np.stack([c[2] for c in cases])).cuda()
    T, stats = eng.icp_refine(src, ref, T0, 0.1)
    T, stats = T.cpu().numpy(), stats.cpu().numpy()
    for k, (s, r, t0, t_gt) in enumerate(cases):
        To, fitness, rmse, converged, iters = icp(s, r, t0, 0.1)
        assert abs(stats[k, 0] - fitness) < 2e-3 and abs(stats[k, 1] - rmse) < 1e-5, (stats[k], fitness, rmse)
        assert _rot_err(T[k][:, :3].astype(np.float64), To[:, :3]) < 2e-5
        assert np.linalg.norm(T[k][:, 3] - To[:, 3]) < 2e-5
        assert stats[k, 2] == 1.0 and abs(stats[k, 3] - iters) <= 2
        if noise == 0.0:
            assert _rot_err(T[k][:, :3].astype(np.float64), t_gt[:, :3]) < 1e-4
    # nothing within reach: the initial pose comes back
    T2, st2 = eng.icp_refine(src, ref + 100.0, T0, 0.05)
    assert torch.allclose(T2, T0, atol=1e-6) and float(st2[:, 0].abs().max()) == 0.0
    eng.close()


@pytest.mark.gpu
def test_gpu_harness_pose_opt_icp():
    """inference_align(pose_opt='icp'): the ICP-refined pose is appended as the last entry (reference test.py:406-408)."""
    import argparse
    import torch
    from deepsir_amd.harness import inference_align
    from deepsir_amd.model import Network
    from deepsir_amd.weights import generate_state_dict, to_torch_state_dict
    from deepsir_amd.arch import NetConfig
    args = argparse.Namespace(pipeline="align", num_sub=-1, num_knn=16, out_feat_dim=64, clip_weight_thresh=0.0, feat_len=3,
                              d_out=[16, 64, 128, 256], num_points=2048, sub_sampling_ratio=[4, 4, 4, 4], use_ppf=False)
    net = Network(args)
    net.load_state_dict(to_torch_state_dict(generate_state_dict(NetConfig(), 0)))
    net = net.cuda().eval()
    pairs = [make_pair(2048, s, 3) for s in (41, 42)]
    plain, _ = inference_align(pairs, net, 3, batch=2)
    icp_T, _ = inference_align(pairs, net, 3, batch=2, pose_opt="icp", voxel_size=0.05)
    assert plain.shape == icp_T.shape == (2, 4, 3, 4)
    assert np.array_equal(plain[:, :3], icp_T[:, :3]) and np.array_equal(plain[:, 3], plain[:, 2])
    for k, p in enumerate(pairs):
        To, *_ = icp(p["points_src"][0], p["points_ref"][0], plain[k, 2], 0.1)
        assert _rot_err(icp_T[k, 3][:, :3], To[:, :3]) < 5e-5 and np.linalg.norm(icp_T[k, 3][:, 3] - To[:, 3]) < 5e-5

"""Dataset front ends (deepsir_amd/data.py; reference dataloader/threeDMatch_loader.py, kitti_loader.py test branches).
CPU: the file parsers and the pair-selection / pose arithmetic against plain numpy restatements of the reference lines.
GPU: tiny synthetic datasets written to disk in the reference's directory layout, walked end to end."""
import os
import struct

import numpy as np
import pytest
import torch

from deepsir_amd import data as D


# --------------------------------------------------------------------------------------------- helpers
def write_gt_log(path, records):
    with open(path, "w") as f:
        for (i, j, n), T in records:
            f.write(f"{i}\t {j}\t {n}\n")
            for r in range(4):
                f.write(" ".join(f"{v:.8e}" for v in T[r]) + "\n")


def write_ply(path, xyz, fmt="binary_little_endian", extra=True):
    n = len(xyz)
    props = ["property float x", "property float y", "property float z"]
    if extra:
        props = ["property float nx"] + props[:1] + ["property uchar red"] + props[1:] + ["property double curvature"]
    hdr = ["ply", f"format {fmt} 1.0", "comment written by the test", f"element vertex {n}"] + props + \
          ["element face 0", "property list uchar int vertex_indices", "end_header"]
    with open(path, "wb") as f:
        f.write(("\n".join(hdr) + "\n").encode("ascii"))
        if fmt == "ascii":
            for p in xyz:
                row = [0.5, p[0], 7, p[1], p[2], 0.25] if extra else list(p)
                f.write((" ".join(repr(float(v)) if not (extra and k == 2) else str(int(v)) for k, v in enumerate(row)) + "\n").encode())
        else:
            e = "<" if fmt == "binary_little_endian" else ">"
            for p in xyz:
                if extra:
                    f.write(struct.pack(e + "ffBffd", 0.5, p[0], 7, p[1], p[2], 0.25))
                else:
                    f.write(struct.pack(e + "fff", *p))


def rot_z(a):
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, -s, 0, 0], [s, c, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1.0]])


# --------------------------------------------------------------------------------------------- CPU
def test_read_trajectory_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    recs = [((i, i + 2, 37), np.vstack([rng.standard_normal((3, 4)), [0, 0, 0, 1]])) for i in range(5)]
    p = tmp_path / "gt.log"
    write_gt_log(p, recs)
    got = D.read_trajectory(str(p))
    assert [m for m, _ in got] == [m for m, _ in recs]
    for (_, a), (_, b) in zip(got, recs):
        np.testing.assert_allclose(a, b, rtol=1e-8)
    with open(p, "a") as f:
        f.write("1 2 3\n")
    with pytest.raises(ValueError):
        D.read_trajectory(str(p))


@pytest.mark.parametrize("fmt", ["ascii", "binary_little_endian", "binary_big_endian"])
@pytest.mark.parametrize("extra", [False, True])
def test_read_ply_xyz(tmp_path, fmt, extra):
    xyz = np.random.default_rng(1).standard_normal((57, 3)).astype(np.float32)
    p = tmp_path / "c.ply"
    write_ply(p, xyz, fmt, extra)
    got = D.read_ply_xyz(str(p))
    assert got.dtype == np.float32 and got.shape == (57, 3)
    np.testing.assert_array_equal(got, xyz)


def test_read_ply_rejects_garbage(tmp_path):
    p = tmp_path / "x.ply"
    p.write_bytes(b"not a ply\n")
    with pytest.raises(ValueError):
        D.read_ply_xyz(str(p))
    write_ply(p, np.zeros((4, 3), np.float32))
    raw = p.read_bytes()
    p.write_bytes(raw[:-5])
    with pytest.raises(ValueError):
        D.read_ply_xyz(str(p))


def make_kitti(root, drive, n_scans, step, pts_per_scan=0, rng=None, scene=None):
    """poses/<drive>.txt with a vehicle driving along camera z (= velodyne x) with a slow yaw; optional scans of a static
    scene expressed in each scan's velodyne frame."""
    seq = os.path.join(root, "dataset", "sequences", "%02d" % drive, "velodyne")
    os.makedirs(seq, exist_ok=True)
    os.makedirs(os.path.join(root, "dataset", "poses"), exist_ok=True)
    V = D._VELO2CAM
    poses = []
    for t in range(n_scans):
        Tw = rot_z(0.0)
        a = 0.004 * t
        Tw[:3, :3] = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])   # yaw about camera y
        Tw[:3, 3] = [0.02 * t, 0.0, step * t]
        poses.append(Tw)
        if scene is not None:
            # world (= camera-0 frame of scan 0) -> velodyne frame of scan t
            to_velo = np.linalg.inv(V) @ np.linalg.inv(Tw)
            p = scene @ to_velo[:3, :3].T + to_velo[:3, 3]
            refl = rng.random((len(p), 1))
            np.concatenate([p, refl], 1).astype(np.float32).tofile(os.path.join(seq, "%06d.bin" % t))
        else:
            np.zeros((1, 4), np.float32).tofile(os.path.join(seq, "%06d.bin" % t))
    np.savetxt(os.path.join(root, "dataset", "poses", "%02d.txt" % drive), np.array([p[:3].reshape(-1) for p in poses]))
    return poses


def reference_pairs(all_pos, inames, min_dist=10.0):
    """kitti_loader.py:98-127 restated line by line (the branch that cannot terminate there is not reachable here)."""
    Ts = all_pos[:, :3, 3]
    pdist = ((Ts.reshape(1, -1, 3) - Ts.reshape(-1, 1, 3)) ** 2).sum(-1)
    more = pdist > min_dist ** 2
    files, curr = [], inames[0]
    while curr in inames:
        nxt = np.where(more[curr][curr:curr + 100])[0]
        if len(nxt) == 0:
            curr += 1
            continue
        nxt = nxt[0] + curr - 1
        if nxt in inames:
            files.append((curr, int(nxt)))
            curr = nxt + 1
    return files


def test_kitti_pairs_and_odometry_pose(tmp_path):
    poses = make_kitti(str(tmp_path), 9, 140, 0.9)
    ds = D.KittiOdometryTest(str(tmp_path), engine=None, sequences=[9])
    want = reference_pairs(np.array(poses), list(range(140)))
    assert [(t0, t1) for _, t0, t1 in ds.files] == want and len(want) > 5
    # scan t0 -> scan t1: a static world point seen from both scans must map onto itself
    V = D._VELO2CAM
    w = np.array([3.0, -1.0, 20.0, 1.0])
    _, t0, t1 = ds.files[2]
    x0 = np.linalg.inv(V) @ np.linalg.inv(poses[t0]) @ w
    x1 = np.linalg.inv(V) @ np.linalg.inv(poses[t1]) @ w
    M = ds.odometry_pose(9, t0, t1)
    np.testing.assert_allclose(M @ x0, x1, atol=1e-9)


def test_kitti_drops_the_known_bad_pair(tmp_path, monkeypatch):
    """kitti_loader.py:129-131: the test split removes pair (8, 15, 58)."""
    make_kitti(str(tmp_path), 8, 4, 1.0)
    monkeypatch.setattr(D.KittiOdometryTest, "pairs", lambda self, drive: [(drive, 0, 11), (drive, 15, 58), (drive, 59, 70)])
    ds = D.KittiOdometryTest(str(tmp_path), engine=None, sequences=[8])
    assert ds.files == [(8, 0, 11), (8, 59, 70)]


def test_read_semantic_labels(tmp_path):
    raw = np.array([0, 1, 10, 252, 40, 60, 99, 259, 81], dtype=np.uint32)
    inst = (np.arange(len(raw), dtype=np.uint32) + 3) << 16            # instance ids in the upper half are dropped
    p = tmp_path / "000000.label"
    (raw | inst).tofile(p)
    got = D.read_semantic_labels(str(p), len(raw))
    assert got.dtype == np.uint8 and got.tolist() == [0, 0, 1, 1, 9, 9, 0, 5, 19]
    with pytest.raises(ValueError):
        D.read_semantic_labels(str(p), 5)
    np.array([7], dtype=np.uint32).tofile(p)                           # 7 is not a SemanticKITTI label
    with pytest.raises(KeyError):
        D.read_semantic_labels(str(p))


# --------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_threedmatch_test_split_end_to_end(tmp_path):
    from deepsir_amd.arch import NetConfig
    from deepsir_amd.engine import Engine
    from oracle.preprocess import voxel_downsample
    rng = np.random.default_rng(3)
    root = tmp_path / "3dmatch"
    scene = "7-scenes-redkitchen"
    os.makedirs(root / "test" / scene)
    os.makedirs(root / "test" / (scene + "-evaluation"))
    clouds = [(rng.random((6000, 3)) * [2.0, 1.5, 1.0]).astype(np.float32) for _ in range(3)]
    for k, c in enumerate(clouds):
        write_ply(root / "test" / scene / f"cloud_bin_{k}.ply", c, "binary_little_endian", extra=(k == 1))
    recs = [((0, 1, 3), rot_z(0.3)), ((0, 2, 3), rot_z(-0.2)), ((1, 2, 3), rot_z(0.1))]
    write_gt_log(root / "test" / (scene + "-evaluation") / "gt.log", recs)
    eng = Engine(NetConfig(), 0, max_points=8192, max_pairs=2)
    ds = D.ThreeDMatchTest(str(root), eng, scenes=[scene])
    assert len(ds) == 3
    item = ds[1]
    assert item["others"] == {"seq": scene, "id_ref": 0, "id_src": 2}
    np.testing.assert_allclose(item["transform_gt"], recs[1][1][:3], rtol=1e-6)
    for key, c in (("points_ref", clouds[0]), ("points_src", clouds[2])):
        want = voxel_downsample(c, 0.03)
        got = item[key].cpu().numpy()
        assert got.shape == want.shape
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-6)
    fixed = D.ThreeDMatchTest(str(root), eng, scenes=[scene], num_points=2048)[0]
    assert tuple(fixed["points_src"].shape) == (2048, 3) and tuple(fixed["points_ref"].shape) == (2048, 3)
    eng.close()


@pytest.mark.gpu
def test_kitti_test_split_end_to_end(tmp_path):
    from deepsir_amd.arch import NetConfig
    from deepsir_amd.engine import Engine
    rng = np.random.default_rng(4)
    # a static scene: ground patches and walls in the world (camera-0) frame, x right, y down, z forward
    g = np.stack([rng.uniform(-25, 25, 30000), np.full(30000, 1.6) + 0.02 * rng.standard_normal(30000), rng.uniform(-10, 60, 30000)], 1)
    wl = np.stack([np.full(12000, -8.0) + 0.02 * rng.standard_normal(12000), rng.uniform(-3, 1.6, 12000), rng.uniform(-10, 60, 12000)], 1)
    wr = np.stack([rng.uniform(5, 9, 8000), rng.uniform(-3, 1.6, 8000), 30 + 3 * np.sin(rng.uniform(0, 6.28, 8000))], 1)
    scene = np.concatenate([g, wl, wr], 0)
    root = str(tmp_path / "kitti")
    poses = make_kitti(root, 10, 30, 1.2, rng=rng, scene=scene)
    eng = Engine(NetConfig(feat_len=4), 0, max_points=65536, max_pairs=1)
    ds = D.KittiOdometryTest(root, eng, sequences=[10], voxel_size=0.3)
    assert len(ds) >= 2
    drive, t0, t1 = ds.files[0]
    item = ds[0]
    assert item["others"] == {"seq": 10, "id_src": t0, "id_ref": t1}
    assert item["points_src"].shape[1] == 4 and item["points_src"].shape[0] > 1000
    # crop: 3 m < r <= 60 m, -3 <= z <= 10 (voxel centroids stay inside the convex part of it)
    p = item["points_src"][:, :3].cpu().numpy()
    r = np.linalg.norm(p, axis=1)
    assert r.max() <= 60.0 + 1e-3 and p[:, 2].min() >= -3.0 - 1e-3 and p[:, 2].max() <= 10.0 + 1e-3
    # ground truth: the ICP-refined pose stays within a few centimetres of the (exact, synthetic) odometry pose and is cached
    M = ds.odometry_pose(10, t0, t1)
    T = item["transform_gt"]
    assert np.abs(T - M[:3]).max() < 0.05
    # written to the engine's OWN cache directory: its ICP is not pinned against open3d, so the reference's cache
    # (icp_opti_pose/) is never written by this side
    fn = os.path.join(root, "icp_opti_pose_dsir", f"10_{t0}_{t1}.npy")
    ref_fn = os.path.join(root, "icp_opti_pose", f"10_{t0}_{t1}.npy")
    assert os.path.exists(fn) and not os.path.exists(ref_fn)
    np.save(fn, np.eye(4) * 2.0)                      # the cache is authoritative, like the reference's
    assert np.array_equal(ds[0]["transform_gt"], (np.eye(4) * 2.0)[:3].astype(np.float32))
    os.makedirs(os.path.dirname(ref_fn), exist_ok=True)
    np.save(ref_fn, np.eye(4) * 3.0)                  # a pose the reference cached wins over ours, and is left alone
    assert np.array_equal(ds[0]["transform_gt"], (np.eye(4) * 3.0)[:3].astype(np.float32))
    # the pair registers: src moved by T_gt lands on ref (nearest-neighbour distance of the voxel centroids)
    os.remove(fn); os.remove(ref_fn)
    item = ds[0]
    Tg = torch.from_numpy(item["transform_gt"]).to(item["points_src"].device)
    moved = item["points_src"][:, :3] @ Tg[:, :3].T + Tg[:, 3]
    d = torch.cdist(moved[:2000], item["points_ref"][:, :3]).min(1)[0]
    assert float(d.median()) < 0.25
    eng.close()


@pytest.mark.gpu
def test_dataset_to_metrics_through_the_harness(tmp_path):
    """data.ThreeDMatchTest -> as_batch -> harness.inference_align / evaluate_align with device-resident points."""
    import argparse
    from deepsir_amd.arch import NetConfig
    from deepsir_amd.harness import evaluate_align, inference_align
    from deepsir_amd.model import Network
    from deepsir_amd.weights import generate_state_dict, to_torch_state_dict
    rng = np.random.default_rng(5)
    root = tmp_path / "3dmatch"
    scene = "sun3d-hotel_uc-scan3"
    os.makedirs(root / "test" / scene)
    os.makedirs(root / "test" / (scene + "-evaluation"))
    base = (rng.random((20000, 3)) * [3.0, 2.0, 1.5]).astype(np.float32)
    T = rot_z(0.4)
    T[:3, 3] = [0.3, -0.2, 0.1]
    moved = (base - T[:3, 3]) @ T[:3, :3]            # T maps `moved` (src) onto `base` (ref)
    write_ply(root / "test" / scene / "cloud_bin_0.ply", base, extra=False)
    write_ply(root / "test" / scene / "cloud_bin_1.ply", moved.astype(np.float32), extra=False)
    write_gt_log(root / "test" / (scene + "-evaluation") / "gt.log", [((0, 1, 2), T)])
    args = argparse.Namespace(pipeline="align", num_sub=-1, num_knn=16, out_feat_dim=64, clip_weight_thresh=0.0, feat_len=3,
                              d_out=[16, 64, 128, 256], num_points=2048, sub_sampling_ratio=[4, 4, 4, 4], use_ppf=False)
    net = Network(args)
    net.load_state_dict(to_torch_state_dict(generate_state_dict(NetConfig(), 0)))
    net = net.cuda().eval()
    eng = net._ensure_engine(2048, 1)
    ds = D.ThreeDMatchTest(str(root), eng, scenes=[scene], num_points=2048)
    pairs = [D.as_batch(ds[0])]
    assert pairs[0]["points_src"].is_cuda and tuple(pairs[0]["points_src"].shape) == (1, 2048, 3)
    pred, stats = inference_align(pairs, net, 2, "3DMatch", batch=1)
    assert pred.shape == (1, 3, 3, 4) and np.isfinite(pred).all() and stats.shape == (1, 5)
    metrics, summary = evaluate_align(pred, pairs, eng, "3DMatch")
    assert len(metrics) == 3 and all(np.isfinite(v).all() for v in metrics[-1].values())


@pytest.mark.gpu
def test_kitti_labels_and_fixed_equalisation(tmp_path):
    """SemanticKITTIPair's test branch: labels ride through the voxel average as a channel and are truncated; the cloud
    with fewer voxels is tiled to the size of the other (FixedResampler)."""
    from deepsir_amd.arch import NetConfig
    from deepsir_amd.engine import Engine
    from oracle.preprocess import voxel_downsample
    rng = np.random.default_rng(6)
    scene = np.stack([rng.uniform(-30, 30, 40000), rng.uniform(-2, 1.6, 40000), rng.uniform(-10, 70, 40000)], 1)
    root = str(tmp_path / "kitti")
    make_kitti(root, 9, 14, 1.0, rng=rng, scene=scene)
    lab_dir = os.path.join(root, "dataset", "sequences", "09", "labels")
    os.makedirs(lab_dir)
    keys = np.array(sorted(D.SEMANTIC_KITTI_LEARNING_MAP), dtype=np.uint32)
    raw_labels = {}
    for t in range(14):
        raw_labels[t] = keys[rng.integers(0, len(keys), len(scene))] | (rng.integers(0, 900, len(scene)).astype(np.uint32) << 16)
        raw_labels[t].tofile(os.path.join(lab_dir, "%06d.label" % t))
    eng = Engine(NetConfig(feat_len=4), 0, max_points=65536, max_pairs=1)
    ds = D.KittiOdometryTest(root, eng, sequences=[9], voxel_size=0.4, with_labels=True, refine_pose=False)
    assert len(ds) >= 1
    _, t0, t1 = ds.files[0]
    item = ds[0]
    n = item["points_src"].shape[0]
    assert item["points_ref"].shape[0] == n and item["labels_src"].shape == (n,) and item["labels_src"].dtype == torch.int32
    crop = (3.0, 60.0, -3.0, 10.0)
    sizes = []
    for side, t in (("src", t0), ("ref", t1)):
        scan = D.read_velodyne(os.path.join(root, "dataset", "sequences", "09", "velodyne", "%06d.bin" % t))
        cls = D.read_semantic_labels(os.path.join(lab_dir, "%06d.label" % t), len(scan)).astype(np.float32)
        want = voxel_downsample(np.concatenate([scan, cls[:, None]], 1), 0.4, crop)
        m = len(want)
        sizes.append(m)
        got_p = item[f"points_{side}"].cpu().numpy()
        got_l = item[f"labels_{side}"].cpu().numpy()
        np.testing.assert_allclose(got_p[:m], want[:, :4], rtol=0, atol=1e-6)
        assert np.array_equal(got_l[:m], want[:, 4].astype(np.int32))
        # FixedResampler: np.tile(points, (k // m, 1)) then the first k % m rows
        assert np.array_equal(got_p[m:], got_p[np.arange(m, n) % m]) and np.array_equal(got_l[m:], got_l[np.arange(m, n) % m])
    assert n == max(sizes)
    batch = D.as_batch(item)
    assert tuple(batch["labels_ref"].shape) == (1, n)
    eng.close()

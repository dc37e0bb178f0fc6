"""Pre-processing in front of the path: crop + voxel average + resample.
open3d / numpy's global RNG are unpinned in the reference, so the rule is ours (oracle/preprocess.py);
the HIP path must reproduce the oracle bit for bit (integer / index work and float64 averages)."""
import numpy as np
import pytest
import torch

from oracle.preprocess import crop_mask, preprocess, resample, splitmix64, voxel_downsample


def _raw_cloud(rng, n, extent=3.0, c=3):
    p = rng.uniform(0, extent, (n, 3))
    extra = rng.uniform(0, 1, (n, c - 3))
    return np.concatenate([p, extra], 1).astype(np.float32)


def test_splitmix64_known_answers():
    # reference values of the published splitmix64 generator seeded with 0 (first two outputs)
    assert splitmix64(0) == 0xE220A8397B1DCDAF
    assert splitmix64(0x9E3779B97F4A7C15) == 0x6E789E6AA1B965F4


def test_voxel_downsample_properties():
    rng = np.random.default_rng(0)
    pts = _raw_cloud(rng, 20000, c=4)
    out = voxel_downsample(pts, 0.1)
    assert out.dtype == np.float32 and out.shape[1] == 4 and 0 < len(out) < len(pts)
    # every output lies inside the bounding box and the weighted mean of the cloud is preserved
    assert (out[:, :3] >= pts[:, :3].min(0) - 1e-6).all() and (out[:, :3] <= pts[:, :3].max(0) + 1e-6).all()
    q = np.floor((pts[:, :3].astype(np.float64) - (pts[:, :3].min(0).astype(np.float64) - 0.05)) / float(np.float32(0.1))).astype(np.int64)
    _, cnt = np.unique(q, axis=0, return_counts=True)
    assert len(cnt) == len(out)
    np.testing.assert_allclose((out.astype(np.float64) * cnt[:, None]).sum(0) / len(pts), pts.astype(np.float64).mean(0), rtol=1e-6)
    # idempotent up to voxel membership: down-sampling twice changes nothing much in count
    assert len(voxel_downsample(out, 0.1)) <= len(out)
    assert len(voxel_downsample(pts, 0.1, crop=(0.5, 2.5, 0.2, 2.0))) < len(out)
    assert crop_mask(pts, (0.5, 2.5, 0.2, 2.0)).sum() < len(pts)


def test_resample_rules():
    rng = np.random.default_rng(1)
    pts = _raw_cloud(rng, 300)
    a = resample(pts, 200, 0, 7)
    assert len(np.unique(a, axis=0)) == 200                      # no repeats while points last
    b = resample(pts, 500, 0, 7)
    assert len(np.unique(b[:300], axis=0)) == 300                # all points present, in random order
    assert not np.array_equal(b[:300], pts)
    assert np.array_equal(resample(pts, 200, 0, 7), a) and not np.array_equal(resample(pts, 200, 1, 7), a)
    f = resample(pts, 700, 0, 7, mode="fixed")
    assert np.array_equal(f[:300], pts) and np.array_equal(f[600:], pts[:100])


@pytest.mark.gpu
def test_hip_preprocess_bit_exact():
    from deepsir_amd.arch import NetConfig
    from deepsir_amd.engine import Engine
    cfg = NetConfig(feat_len=3)
    eng = Engine(cfg, 0, max_points=8192, max_pairs=4)
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(2)
    raw = [_raw_cloud(rng, n, c=4) for n in (60000, 35000, 90000, 1200)]
    raw[1][:, :3] = raw[1][:, :3] * 20 - 30          # KITTI-like extent for the crop
    for crop, voxel in ((None, 0.06), ((3.0, 28.0, -25.0, 25.0), 0.5)):
        vox, counts = eng.voxel_downsample([torch.from_numpy(r).to(dev) for r in raw], voxel, crop)
        counts = counts.cpu().numpy()
        for i, r in enumerate(raw):
            ref = voxel_downsample(r, voxel, crop)
            assert counts[i] == len(ref), (i, counts[i], len(ref))
            assert np.array_equal(vox[i, :len(ref)].cpu().numpy(), ref), i
        for mode in ("random", "fixed"):
            out = eng.resample(vox, torch.from_numpy(counts).to(dev), 5000, seed=1234, mode=mode).cpu().numpy()
            for i, r in enumerate(raw):
                ref = resample(voxel_downsample(r, voxel, crop), 5000, i, 1234, mode)
                assert np.array_equal(out[i], ref), (i, mode)
    # end to end: raw clouds -> network input -> pyramid (the random order feeds the prefix sub-sampling)
    inp, counts = eng.preprocess([torch.from_numpy(r[:, :3].copy()).to(dev) for r in raw[:2]], 0.05, 5000, seed=9)
    ref, n_ref = preprocess([r[:, :3] for r in raw[:2]], 0.05, 5000, 9)
    assert np.array_equal(inp.cpu().numpy(), ref) and counts.cpu().numpy().tolist() == n_ref
    xyz, neigh, sub, interp = eng.knn_pyramid(inp)
    assert tuple(neigh.shape) == (2, 6640, 16)
    eng.close()

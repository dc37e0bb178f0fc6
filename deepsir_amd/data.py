"""Dataset front ends of the registration path: the on-disk formats of the reference's TEST splits -> the pair dicts
`harness.inference_align` consumes (the callers immediately in front of the path).

  reference                                              here
  threeDMatch_loader.read_trajectory (:15-37)            read_trajectory
  o3d.io.read_point_cloud (:165-166)                     read_ply_xyz  (vertex x, y, z of ascii / binary PLY)
  ThreeDMatch.prepare_test / get_data test branch        ThreeDMatchTest
      (:118-175): gt.log pairs, cloud_bin_{i}.ply,
      voxel_down_sample(0.03)
  KITTIPair.prepare_kitti_test (:98-131)                 KittiOdometryTest.pairs
  KITTIPair.get_data (:299-346): velodyne .bin,          KittiOdometryTest.__getitem__
      process_point_cloud crop, pose_refine (ICP,
      cached as icp_opti_pose/<drive>_<t0>_<t1>.npy),
      voxel_down_sample(voxel_size)

File parsing is host IO in numpy.  Every per-point step runs on the device through the engine: crop + voxel average
(csrc/preprocess.hip, `Engine.voxel_downsample`), the point-to-point ICP of pose_refine (csrc/icp.hip,
`Engine.icp_refine`), resampling to a fixed size (`Engine.resample`).  The voxel output order is this engine's own
(ascending voxel index; open3d's hash order is not reproducible without open3d, see include/dsir.h), which is
immaterial to the network: its input is a point SET that the resampler permutes anyway.

The train / val branches (augmentation, pickled 3DMatch fragments, SemanticKITTI labels) feed training, which is out of
scope."""
from __future__ import annotations

import glob
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

# the reference's split files (dataloader/split/test_3dmatch.txt, test_kitti.txt): the standard public test splits
THREEDMATCH_TEST_SCENES = (
    "7-scenes-redkitchen", "sun3d-home_at-home_at_scan1_2013_jan_1", "sun3d-home_md-home_md_scan9_2012_sep_30",
    "sun3d-hotel_uc-scan3", "sun3d-hotel_umd-maryland_hotel1", "sun3d-hotel_umd-maryland_hotel3",
    "sun3d-mit_76_studyroom-76-1studyroom2", "sun3d-mit_lab_hj-lab_hj_tea_nov_2_2012_scan1_erika")
KITTI_TEST_SEQUENCES = (8, 9, 10)

# SemanticKITTI `learning_map` (raw label -> training class 0..19, 0 = unlabeled): the public SemanticKITTI API table the
# reference reads from dataloader/semantic-kitti.yaml (kitti_loader.py:358-362)
SEMANTIC_KITTI_LEARNING_MAP = {0: 0, 1: 0, 10: 1, 11: 2, 13: 5, 15: 3, 16: 5, 18: 4, 20: 5, 30: 6, 31: 7, 32: 8, 40: 9, 44: 10,
                               48: 11, 49: 12, 50: 13, 51: 14, 52: 0, 60: 9, 70: 15, 71: 16, 72: 17, 80: 18, 81: 19, 99: 0,
                               252: 1, 253: 7, 254: 6, 255: 8, 256: 5, 257: 5, 258: 4, 259: 5}

# velodyne -> camera-0 calibration the reference hard-codes (kitti_loader.py:148-159)
_VELO2CAM = np.array([[7.533745e-03, -9.999714e-01, -6.166020e-04, -4.069766e-03],
                      [1.480249e-02, 7.280733e-04, -9.998902e-01, -7.631618e-02],
                      [9.998621e-01, 7.523790e-03, 1.480755e-02, -2.717806e-01],
                      [0.0, 0.0, 0.0, 1.0]])


def read_trajectory(path: str, dim: int = 4) -> List[Tuple[Tuple[int, ...], np.ndarray]]:
    """`gt.log` of a 3DMatch evaluation folder: records of one metadata line (ints: i, j, n_fragments) and `dim` matrix
    rows -> [(metadata, pose [dim, dim] float64)]."""
    out = []
    with open(path, "r") as f:
        lines = [ln for ln in f.read().splitlines() if ln.strip()]
    if len(lines) % (dim + 1) != 0:
        raise ValueError(f"{path}: {len(lines)} non-empty lines are not a whole number of {dim + 1}-line records")
    for r in range(0, len(lines), dim + 1):
        meta = tuple(int(x) for x in lines[r].split())
        mat = np.array([[float(x) for x in lines[r + 1 + i].split()] for i in range(dim)], dtype=np.float64)
        if mat.shape != (dim, dim):
            raise ValueError(f"{path}: record {r // (dim + 1)} is not a {dim}x{dim} matrix")
        out.append((meta, mat))
    return out


_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2",
              "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4",
              "double": "f8", "float64": "f8"}


def read_ply_xyz(path: str) -> np.ndarray:
    """Vertex positions of a PLY file (ascii, binary_little_endian or binary_big_endian; any extra scalar vertex
    properties such as normals / colours are skipped) -> [n, 3] float32."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, n_vertex, props, in_vertex, seen_vertex = None, 0, [], False, False
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: unterminated PLY header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] == "comment" or tok[0] == "obj_info":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                if tok[1] == "vertex":
                    if seen_vertex:
                        raise ValueError(f"{path}: two vertex elements")
                    n_vertex, in_vertex, seen_vertex = int(tok[2]), True, True
                else:
                    if not seen_vertex:
                        raise ValueError(f"{path}: element '{tok[1]}' precedes the vertices (unsupported)")
                    in_vertex = False
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError(f"{path}: list property on vertices (unsupported)")
                if tok[1] not in _PLY_TYPES:
                    raise ValueError(f"{path}: unknown property type {tok[1]}")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        names = [p[0] for p in props]
        if not all(k in names for k in ("x", "y", "z")):
            raise ValueError(f"{path}: vertices without x, y, z")
        if fmt == "ascii":
            cols = [names.index(k) for k in ("x", "y", "z")]
            rows = np.loadtxt(f, dtype=np.float64, max_rows=n_vertex, ndmin=2) if n_vertex else np.zeros((0, len(names)))
            if rows.shape[0] != n_vertex:
                raise ValueError(f"{path}: {rows.shape[0]} of {n_vertex} vertices present")
            return np.ascontiguousarray(rows[:, cols], dtype=np.float32)
        if fmt not in ("binary_little_endian", "binary_big_endian"):
            raise ValueError(f"{path}: unknown PLY format {fmt}")
        order = "<" if fmt == "binary_little_endian" else ">"
        dt = np.dtype([(n, order + t) for n, t in props])
        raw = f.read(n_vertex * dt.itemsize)
        if len(raw) != n_vertex * dt.itemsize:
            raise ValueError(f"{path}: truncated vertex data")
        v = np.frombuffer(raw, dtype=dt, count=n_vertex)
        return np.stack([v["x"], v["y"], v["z"]], 1).astype(np.float32)


def read_velodyne(path: str) -> np.ndarray:
    """KITTI velodyne scan: float32 records (x, y, z, reflectance) -> [n, 4]."""
    a = np.fromfile(path, dtype=np.float32)
    if a.size % 4 != 0:
        raise ValueError(f"{path}: {a.size} floats are not a whole number of (x, y, z, reflectance) records")
    return a.reshape(-1, 4)


def read_semantic_labels(path: str, n: Optional[int] = None) -> np.ndarray:
    """SemanticKITTI `.label` file: one uint32 per point, semantic label in the lower 16 bits (instance id above), mapped
    through `learning_map` -> [n] uint8 training classes (kitti_loader.py:368-377)."""
    raw = np.fromfile(path, dtype=np.uint32) & 0xFFFF
    if n is not None and raw.size != n:
        raise ValueError(f"{path}: {raw.size} labels for {n} points")
    lut = np.zeros(65536, dtype=np.int32) - 1
    for k, v in SEMANTIC_KITTI_LEARNING_MAP.items():
        lut[k] = v
    out = lut[raw]
    if (out < 0).any():
        raise KeyError(f"{path}: label {int(raw[out < 0][0])} is not in the SemanticKITTI learning map")
    return out.astype(np.uint8)


def _to_device(engine, a: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(engine.device)


def _voxelize(engine, clouds: Sequence[np.ndarray], voxel_size: float, crop=None) -> List[torch.Tensor]:
    """crop + voxel average of a ragged list on the device -> list of [n_i, C] CUDA tensors."""
    dev = [_to_device(engine, c) for c in clouds]
    vox, counts = engine.voxel_downsample(dev, voxel_size, crop)
    n = counts.cpu().tolist()
    return [vox[i, :n[i]].contiguous() for i in range(len(clouds))]


def as_batch(item: Dict[str, object]) -> Dict[str, object]:
    """One dataset sample -> the batch-of-one dict of the reference's collate (data_base.py:196-219), points staying on
    the device: what `harness.inference_align` / `evaluate_align` take as an element of `pairs`."""
    out = {"points_src": item["points_src"][None].contiguous(), "points_ref": item["points_ref"][None].contiguous(),
           "transform_gt": np.asarray(item["transform_gt"], dtype=np.float32)[None], "others": [item["others"]]}
    for k in ("labels_src", "labels_ref"):
        if k in item:
            out[k] = item[k][None].contiguous()
    return out


class ThreeDMatchTest:
    """The 3DMatch test split as the reference walks it (threeDMatch_loader.py:118-175): for every record (i, j, T_gt) of
    `<root>/test/<scene>-evaluation/gt.log`, ref = `<scene>/cloud_bin_i.ply`, src = `<scene>/cloud_bin_j.ply`, both
    voxel-averaged at 0.03 m.  `num_points`: resample to a fixed size (what the collate needs for batches > 1;
    seeded, Resampler semantics); None keeps the ragged sizes."""

    def __init__(self, root: str, engine, scenes: Sequence[str] = THREEDMATCH_TEST_SCENES, voxel_size: float = 0.03,
                 num_points: Optional[int] = None, seed: int = 0):
        self.test_path = os.path.join(root, "test")
        if not os.path.isdir(self.test_path):
            raise FileNotFoundError(f"Invalid path: {self.test_path}")
        self.engine, self.voxel_size, self.num_points, self.seed = engine, float(voxel_size), num_points, int(seed)
        self.files: List[Tuple[str, int, int, np.ndarray]] = []
        for s in scenes:
            traj = os.path.join(self.test_path, s + "-evaluation", "gt.log")
            if not os.path.exists(traj):
                raise FileNotFoundError(traj)
            for meta, pose in read_trajectory(traj):
                self.files.append((s, meta[0], meta[1], pose))

    def __len__(self):
        return len(self.files)

    def __getitem__(self, index: int) -> Dict[str, object]:
        s, i, j, T_gt = self.files[index]
        ref = read_ply_xyz(os.path.join(self.test_path, s, f"cloud_bin_{i}.ply"))
        src = read_ply_xyz(os.path.join(self.test_path, s, f"cloud_bin_{j}.ply"))
        if self.num_points:
            pts, _ = self.engine.preprocess([_to_device(self.engine, src), _to_device(self.engine, ref)], self.voxel_size,
                                            int(self.num_points), seed=self.seed + index)
            vs, vr = pts[0], pts[1]
        else:
            vs, vr = _voxelize(self.engine, [src, ref], self.voxel_size)
        return {"points_src": vs, "points_ref": vr, "transform_gt": T_gt[:3, :].astype(np.float32),
                "others": {"seq": s, "id_ref": i, "id_src": j}}


class KittiOdometryTest:
    """KITTI odometry test pairs as the reference builds them (kitti_loader.py:98-131, :241-346): within a sequence,
    from the current scan the first one more than 10 m away (looking at most 100 scans ahead, minus one - the
    3DFeatNet convention), then continue after it; pair (8, 15, 58) dropped.  A sample = both scans cropped
    (3 m < r <= 60 m, -3 m <= z <= 10 m) and voxel-averaged at `voxel_size` with the reflectance as 4th channel, and
    the ground-truth pose: odometry poses through the velodyne calibration, refined by point-to-point ICP (0.2 m,
    <= 200 iterations, on 0.05 m voxels) and cached as `<root>/icp_opti_pose_dsir/<drive>_<t0>_<t1>.npy`; the
    reference's own cache `<root>/icp_opti_pose/` is read when present and never written.

    Sizes: `num_points` None = what the reference's test split does (SemanticKITTIPair.__getitem__ with fixed=True,
    apply_augment_V2, data_base.py:271-283): the cloud with fewer voxels is tiled (FixedResampler) to the size of the
    other; an int = seeded random resample of both to that size (Resampler).  `with_labels`: the SemanticKITTI class of
    every point rides through the voxel average as a 5th channel and is truncated to an integer afterwards, exactly the
    reference's treatment (kitti_loader.py:324-341, :401-402): `labels_src` / `labels_ref` [n] int32."""

    MIN_DIST = 10.0

    def __init__(self, root: str, engine, sequences: Sequence[int] = KITTI_TEST_SEQUENCES, voxel_size: float = 0.3,
                 feat_len: int = 4, num_points: Optional[int] = None, seed: int = 0, refine_pose: bool = True,
                 with_labels: bool = False):
        self.root_path = os.path.join(root, "dataset")
        if not os.path.isdir(self.root_path):
            raise FileNotFoundError(f"Invalid path: {self.root_path}")
        self.icp_path = os.path.join(root, "icp_opti_pose")
        self.engine, self.voxel_size, self.feat_len = engine, float(voxel_size), int(feat_len)
        self.num_points, self.seed, self.refine_pose = num_points, int(seed), bool(refine_pose)
        self.with_labels = bool(with_labels)
        self._poses: Dict[int, np.ndarray] = {}
        self.files: List[Tuple[int, int, int]] = []
        for drive in sequences:
            self.files.extend(self.pairs(int(drive)))
        if (8, 15, 58) in self.files:
            self.files.remove((8, 15, 58))

    # ---- index
    def scan_ids(self, drive: int) -> List[int]:
        names = glob.glob(os.path.join(self.root_path, "sequences", "%02d" % drive, "velodyne", "*.bin"))
        if not names:
            raise FileNotFoundError(f"Make sure that the path {self.root_path} has drive id: {drive}")
        return sorted(int(os.path.basename(n)[:-4]) for n in names)

    def poses(self, drive: int) -> np.ndarray:
        """[frames, 4, 4] camera-0 poses T_w_cam0 of `poses/<drive>.txt`."""
        if drive not in self._poses:
            a = np.atleast_2d(np.genfromtxt(os.path.join(self.root_path, "poses", "%02d.txt" % drive)))
            T = np.tile(np.eye(4), (a.shape[0], 1, 1))
            T[:, :3, :] = a.reshape(-1, 3, 4)
            self._poses[drive] = T
        return self._poses[drive]

    def pairs(self, drive: int) -> List[Tuple[int, int, int]]:
        inames = set(self.scan_ids(drive))
        pos = self.poses(drive)[:, :3, 3]
        out = []
        curr = min(inames)
        while curr in inames:
            d2 = ((pos[curr:curr + 100] - pos[curr]) ** 2).sum(-1)
            far = np.where(d2 > self.MIN_DIST ** 2)[0]
            if len(far) == 0:
                curr += 1
                continue
            nxt = int(far[0]) + curr - 1
            if nxt in inames:
                out.append((drive, curr, nxt))
                curr = nxt + 1
            # else: the reference loops on the same `curr` forever; a missing scan inside a sequence does not occur
            else:
                curr += 1
        return out

    def __len__(self):
        return len(self.files)

    # ---- ground truth
    def odometry_pose(self, drive: int, t0: int, t1: int) -> np.ndarray:
        """Scan t0 -> scan t1 from the provided poses (kitti_loader.py:257-259): [4, 4] float64."""
        p = self.poses(drive)
        M = (_VELO2CAM.T @ p[t0].T @ np.linalg.inv(p[t1].T) @ np.linalg.inv(_VELO2CAM.T)).T
        return M

    def gt_pose(self, drive: int, t0: int, t1: int, xyz0: np.ndarray, xyz1: np.ndarray) -> np.ndarray:
        key = "%d_%d_%d" % (drive, t0, t1)
        # The reference's own cache (open3d ICP) is READ when present, never written: this engine's refinement is not
        # pinned against open3d (voxel order, termination bookkeeping), so its poses go to a directory of their own and
        # neither side's ground truth silently becomes the other's.
        ref_fn = os.path.join(self.icp_path, key + ".npy")
        if os.path.exists(ref_fn):
            return np.load(ref_fn)
        fn = os.path.join(self.icp_path + "_dsir", key + ".npy")
        if os.path.exists(fn):
            return np.load(fn)
        M = self.odometry_pose(drive, t0, t1)
        if not self.refine_pose:
            return M
        # as the reference: scan 0 is moved by M first, ICP starts from the identity and its result T' is applied on the
        # RIGHT (M2 = M @ T', kitti_loader.py:264-271)
        v0, v1 = _voxelize(self.engine, [xyz0[:, :3], xyz1[:, :3]], 0.05)
        Md = torch.from_numpy(M.astype(np.float32)).to(self.engine.device)
        v0 = (v0 @ Md[:3, :3].T + Md[:3, 3]).contiguous()
        eye = torch.eye(4, device=self.engine.device)[None, :3, :].contiguous()
        T, _ = self.engine.icp_refine(v0[None].contiguous(), v1[None].contiguous(), eye, 0.2, max_iter=200)
        Tp = np.eye(4)
        Tp[:3, :] = T[0].double().cpu().numpy()
        M2 = M @ Tp
        os.makedirs(os.path.dirname(fn), exist_ok=True)
        np.save(fn, M2)
        return M2

    # ---- samples
    def __getitem__(self, index: int) -> Dict[str, object]:
        drive, t0, t1 = self.files[index]
        seq = os.path.join(self.root_path, "sequences", "%02d" % drive, "velodyne")
        xyz0 = read_velodyne(os.path.join(seq, "%06d.bin" % t0))
        xyz1 = read_velodyne(os.path.join(seq, "%06d.bin" % t1))
        crop = (3.0, 60.0, -3.0, 10.0)

        def crop_host(c):   # process_point_cloud (data_base.py:299-312), for the pose refinement's input
            r2 = (c[:, :3].astype(np.float64) ** 2).sum(1)
            return c[(r2 <= crop[1] ** 2) & (r2 > crop[0] ** 2) & (c[:, 2] >= crop[2]) & (c[:, 2] <= crop[3])]
        T_gt = self.gt_pose(drive, t0, t1, crop_host(xyz0), crop_host(xyz1))
        C = max(3, min(self.feat_len, 4))
        raw = [xyz0, xyz1]
        if self.with_labels:
            lab = os.path.join(self.root_path, "sequences", "%02d" % drive, "labels")
            raw = [np.concatenate([c, read_semantic_labels(os.path.join(lab, "%06d.label" % t), len(c))[:, None].astype(np.float32)], 1)
                   for c, t in ((xyz0, t0), (xyz1, t1))]          # [x, y, z, reflectance, class]
        else:
            raw = [c[:, :C] for c in raw]
        vox, counts = self.engine.voxel_downsample([_to_device(self.engine, c) for c in raw], self.voxel_size, crop)
        if self.num_points:
            pts = self.engine.resample(vox, counts, int(self.num_points), self.seed + index, "random")
        else:
            pts = self.engine.resample(vox, counts, int(counts.max().item()), 0, "fixed")
        out = {"points_src": pts[0, :, :C].contiguous(), "points_ref": pts[1, :, :C].contiguous(),
               "transform_gt": T_gt[:3, :].astype(np.float32), "others": {"seq": drive, "id_src": t0, "id_ref": t1}}
        if self.with_labels:
            out["labels_src"] = pts[0, :, 4].to(torch.int32)       # astype(np.int32): truncation of the voxel mean
            out["labels_ref"] = pts[1, :, 4].to(torch.int32)
        return out

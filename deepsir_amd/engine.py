"""Thin Python object over a ``dsir_ctx`` (include/dsir.h).

PyTorch is used for device memory only: every method takes / returns
``torch`` CUDA tensors and hands raw device pointers to the C ABI.  All
compute is in libdsir.so; nothing here falls back to torch ops or the CPU.

Activations crossing this boundary are point-major ``[clouds, n, C]``.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from .arch import PIPELINES, NetConfig, level_sizes


class EngineError(RuntimeError):
    pass


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _chk(t: torch.Tensor, dtype, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise EngineError(f"{name}: expected a CUDA tensor (the engine has no CPU path)")
    if t.dtype != dtype:
        raise EngineError(f"{name}: expected {dtype}, got {t.dtype}")
    return t.contiguous()


class Engine:
    def __init__(self, cfg: NetConfig, device: int = 0, max_points: int = 8192, max_pairs: int = 1):
        if cfg.use_ppf:
            raise EngineError("use_ppf=True is outside the hot path (reference default is False)")
        self.lib = _lib.load()
        self.cfg = cfg
        self.device = torch.device("cuda", device)
        self.max_points, self.max_pairs = int(max_points), int(max_pairs)
        c = _lib.dsir_cfg()
        c.feat_len, c.num_knn, c.num_layers = cfg.feat_len, cfg.num_knn, len(cfg.d_out)
        for i in range(4):
            c.sub_sampling_ratio[i] = cfg.sub_sampling_ratio[i]
            c.d_out[i] = cfg.d_out[i]
        c.out_feat_dim, c.num_classes = cfg.out_feat_dim, cfg.num_classes
        c.max_points, c.max_pairs = self.max_points, self.max_pairs
        c.pipeline = PIPELINES.index(cfg.pipeline)
        h = C.c_void_p()
        if self.lib.dsir_create(device, C.byref(c), C.byref(h)) != 0:
            raise EngineError("dsir_create: " + self.lib.dsir_last_error(None).decode())
        self.h = h
        self.weights_loaded = False
        self._shared_stream = False
        self._bound_stream = None

    def close(self):
        if getattr(self, "h", None):
            self.lib.dsir_destroy(self.h)
            self.h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ helpers
    def _call(self, rc: int):
        if rc != 0:
            raise EngineError(self.lib.dsir_last_error(self.h).decode())

    def _pre(self):
        if not self._shared_stream:
            torch.cuda.current_stream(self.device).synchronize()
            return
        # shared mode follows torch's current stream (torch.cuda.graph captures on a side stream): re-bind when it changed
        cur = torch.cuda.current_stream(self.device).cuda_stream
        if cur != self._bound_stream:
            self._call(self.lib.dsir_set_stream(self.h, C.c_void_p(cur), 0))
            self._bound_stream = cur

    def sync(self):
        if not self._shared_stream:     # on the caller's stream the results are stream-ordered with the caller's own kernels
            self._call(self.lib.dsir_sync(self.h))

    def use_torch_stream(self, on: bool = True):
        """Order the engine's operators on torch's CURRENT stream (include/dsir.h, dsir_set_stream) instead of the context's
        own: no host synchronisation around a call any more (results are stream-ordered with torch's kernels), which also makes
        the operators capturable by ``torch.cuda.graph``.  ``on=False``: back to the context's own stream (after a device
        synchronisation, so that nothing enqueued on torch's stream is overtaken)."""
        if on:
            self._call(self.lib.dsir_sync(self.h))
            self._bound_stream = torch.cuda.current_stream(self.device).cuda_stream
            self._call(self.lib.dsir_set_stream(self.h, C.c_void_p(self._bound_stream), 0))
            self._shared_stream = True
        else:
            torch.cuda.synchronize(self.device)
            self._call(self.lib.dsir_set_stream(self.h, None, 1))
            self._shared_stream = False
            self._bound_stream = None

    def _empty(self, shape, dtype=torch.float32):
        return torch.empty(shape, dtype=dtype, device=self.device)

    # ------------------------------------------------------------------ weights
    def expected_keys(self):
        n = self.lib.dsir_num_weights(self.h)
        out = []
        for i in range(n):
            numel = C.c_int64()
            out.append((self.lib.dsir_weight_name(self.h, i, C.byref(numel)).decode(), numel.value))
        return out

    def load_state_dict(self, sd: Dict[str, "np.ndarray | torch.Tensor"], strict: bool = True):
        """Network.load_state_dict counterpart (reference test.py:614): strict key/shape checking."""
        expected = [k for k, _ in self.expected_keys()]
        missing = [k for k in expected if k not in sd]
        unexpected = [k for k in sd if k not in set(expected)]
        if strict and (missing or unexpected):
            raise EngineError(f"Error(s) in loading state_dict: missing keys {missing[:5]}{'...' if len(missing) > 5 else ''}, "
                              f"unexpected keys {unexpected[:5]}{'...' if len(unexpected) > 5 else ''}")
        for k in expected:
            if k not in sd:
                continue
            v = sd[k]
            a = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
            if k.endswith("num_batches_tracked"):
                self._call(self.lib.dsir_load_weight(self.h, k.encode(), None, None, 0))
                continue
            a = np.ascontiguousarray(a, dtype=np.float32)
            shape = (C.c_int64 * a.ndim)(*a.shape)
            self._call(self.lib.dsir_load_weight(self.h, k.encode(), a.ctypes.data_as(C.c_void_p), shape, a.ndim))
        self._call(self.lib.dsir_finalize_weights(self.h))
        self.weights_loaded = True

    # ------------------------------------------------------------------ stages
    def pyramid_shapes(self, n: int) -> Tuple[int, int]:
        nl = level_sizes(n, self.cfg.sub_sampling_ratio)
        return sum(nl[:-1]), sum(nl[1:])

    def narrow(self, t: torch.Tensor) -> torch.Tensor:
        """int64 index tensor -> int32 (on device)."""
        if t.dtype == torch.int32:
            return t.contiguous()
        t = _chk(t, torch.int64, "index tensor")
        out = self._empty(t.shape, torch.int32)
        self._pre()
        self._call(self.lib.dsir_narrow_i64(self.h, _ptr(t), _ptr(out), t.numel()))
        return out

    def knn_pyramid(self, points: torch.Tensor):
        """DataBase.nn_search counterpart.  points [clouds, n, >=3] ->
        (xyz [c,S,3], neigh_idx [c,S,16] i32, sub_idx [c,S1,16] i32, interp_idx [c,S,1] i32)."""
        points = _chk(points, torch.float32, "points")
        c, n, stride = points.shape
        S, S1 = self.pyramid_shapes(n)
        xyz = self._empty((c, S, 3))
        neigh = self._empty((c, S, 16), torch.int32)
        sub = self._empty((c, S1, 16), torch.int32)
        interp = self._empty((c, S, 1), torch.int32)
        self._pre()
        self._call(self.lib.dsir_knn_pyramid(self.h, _ptr(points), stride, c, n, _ptr(xyz), _ptr(neigh), _ptr(sub), _ptr(interp)))
        self.sync()
        return xyz, neigh, sub, interp

    def randla_forward(self, which: str, features, xyz, neigh, sub, interp, want_logits=True):
        """RandLA.forward counterpart -> (feat [c,n,64], logits [c,n,ncls])."""
        features = _chk(features, torch.float32, "features")
        c, n, cin = features.shape
        w = {"feat_extractor": 0, "inlier_model": 1}[which]
        ncls = self.cfg.num_classes if w == 0 else 1
        feat = self._empty((c, n, 64))
        logits = self._empty((c, n, ncls)) if want_logits else None
        xyz, neigh, sub, interp = (_chk(xyz, torch.float32, "xyz"), _chk(neigh, torch.int32, "neigh_idx"),
                                   _chk(sub, torch.int32, "sub_idx"), _chk(interp, torch.int32, "interp_idx"))
        self._pre()
        self._call(self.lib.dsir_randla_forward(self.h, w, _ptr(features), cin, c, n, _ptr(xyz), _ptr(neigh), _ptr(sub),
                                                _ptr(interp), _ptr(feat), _ptr(logits)))
        self.sync()
        return feat, logits

    def score(self, feat, logits, xyz_multi, neigh_multi):
        """torch.max(logits) + score_fun counterpart -> (score [c,n], label [c,n] i32)."""
        feat, logits = _chk(feat, torch.float32, "feat"), _chk(logits, torch.float32, "logits")
        xyz_multi, neigh_multi = _chk(xyz_multi, torch.float32, "xyz"), _chk(neigh_multi, torch.int32, "neigh_idx")
        c, n, _ = feat.shape
        score = self._empty((c, n))
        label = self._empty((c, n), torch.int32)
        self._pre()
        self._call(self.lib.dsir_score(self.h, _ptr(feat), _ptr(logits), _ptr(xyz_multi), xyz_multi.shape[1] * 3,
                                       _ptr(neigh_multi), neigh_multi.shape[1] * 16, c, n, _ptr(score), _ptr(label)))
        self.sync()
        return score, label

    def aggregate(self, xyz, feat0, score):
        """One cloud-batch half of Network.aggregation -> desc [c,n,64]."""
        xyz, feat0, score = _chk(xyz, torch.float32, "xyz"), _chk(feat0, torch.float32, "feat0"), _chk(score, torch.float32, "score")
        c, n, _ = feat0.shape
        desc = self._empty((c, n, 64))
        self._pre()
        self._call(self.lib.dsir_aggregate(self.h, _ptr(xyz), xyz.shape[1] * 3, _ptr(feat0), _ptr(score), c, n, _ptr(desc)))
        self.sync()
        return desc

    def nn_match(self, desc_src, desc_ref, sync=True):
        """match_features_V2 + min(dim=2)[1] counterpart -> idx [p,J] i32."""
        desc_src, desc_ref = _chk(desc_src, torch.float32, "desc_src"), _chk(desc_ref, torch.float32, "desc_ref")
        p, J, _ = desc_src.shape
        K = desc_ref.shape[1]
        idx = self._empty((p, J), torch.int32)
        self._pre()
        self._call(self.lib.dsir_nn_match(self.h, _ptr(desc_src), _ptr(desc_ref), p, J, K, _ptr(idx)))
        if sync:
            self.sync()
        return idx

    def nn_match_screened(self, desc_src, desc_ref, want_stats=True):
        """Same arg-min as nn_match, computed the way dsir_register does (fp16-split screening + exact fp32 decision).
        Returns (idx [p,J] i32, (candidate entries emitted by the screening, rows left to the exhaustive kernel))."""
        desc_src, desc_ref = _chk(desc_src, torch.float32, "desc_src"), _chk(desc_ref, torch.float32, "desc_ref")
        p, J, _ = desc_src.shape
        K = desc_ref.shape[1]
        idx = self._empty((p, J), torch.int32)
        st = (C.c_int64 * 2)()
        self._pre()
        self._call(self.lib.dsir_nn_match_screened(self.h, _ptr(desc_src), _ptr(desc_ref), p, J, K, _ptr(idx),
                                                   st if want_stats else None))
        self.sync()
        return idx, (int(st[0]), int(st[1]))

    def screen_bounds(self, desc_src, desc_ref):
        """Diagnostics of the fp16 screening on one pair (include/dsir.h, dsir_screen_bounds): desc_src [J,64], desc_ref [K,64]
        -> dict(lower, upper, exact, zacc [J,K]; idx, thresh, cand_count [J]; cand_code, cand_lower [J,cap]; out_of_domain)."""
        a, b = _chk(desc_src, torch.float32, "desc_src"), _chk(desc_ref, torch.float32, "desc_ref")
        J, K = a.shape[0], b.shape[0]
        cap = int(self.lib.dsir_screen_cap())
        out = {"lower": self._empty((J, K)), "upper": self._empty((J, K)), "exact": self._empty((J, K)), "zacc": self._empty((J, K)),
               "idx": self._empty((J,), torch.int32), "thresh": self._empty((J,)), "cand_count": self._empty((J,), torch.int32),
               "cand_code": self._empty((J, cap), torch.int32), "cand_lower": self._empty((J, cap)),
               "out_of_domain": self._empty((1,), torch.int32)}
        self._pre()
        self._call(self.lib.dsir_screen_bounds(self.h, _ptr(a), _ptr(b), J, K, _ptr(out["lower"]), _ptr(out["upper"]),
                                               _ptr(out["exact"]), _ptr(out["zacc"]), _ptr(out["idx"]), _ptr(out["thresh"]), _ptr(out["cand_count"]),
                                               _ptr(out["cand_code"]), _ptr(out["cand_lower"]), _ptr(out["out_of_domain"])))
        self.sync()
        return out

    def kabsch(self, src, tgt, w):
        """compute_rigid_transform_2 counterpart -> (T [p,3,4], invalid [p] i32)."""
        src, tgt = _chk(src, torch.float32, "src"), _chk(tgt, torch.float32, "tgt")
        w = _chk(w.reshape(w.shape[0], -1), torch.float32, "weights")
        p, m, _ = src.shape
        T = self._empty((p, 3, 4))
        bad = self._empty((p,), torch.int32)
        self._pre()
        self._call(self.lib.dsir_kabsch(self.h, _ptr(src), _ptr(tgt), _ptr(w), p, m, _ptr(T), _ptr(bad)))
        self.sync()
        return T, bad

    # ------------------------------------------------------------------ the whole path
    def _pair_batch(self, points_src, points_ref, pyramids):
        """dsir_pair_batch of P pairs (+ the tensors that must stay alive while it is in use)."""
        points_src, points_ref = _chk(points_src, torch.float32, "points_src"), _chk(points_ref, torch.float32, "points_ref")
        P, J, cin = points_src.shape
        K = points_ref.shape[1]
        if cin != self.cfg.feat_len or points_ref.shape[2] != cin:
            raise EngineError(f"points have {cin} channels, engine was built for feat_len={self.cfg.feat_len}")
        b = _lib.dsir_pair_batch()
        b.pairs, b.n_src, b.n_ref = P, J, K
        b.points_src, b.points_ref = _ptr(points_src), _ptr(points_ref)
        keep = [points_src, points_ref]
        if pyramids is not None:
            for side, fld in (("src", "src"), ("ref", "ref")):
                x = _chk(pyramids[f"points_{side}_xyz"], torch.float32, "xyz")
                nb = self.narrow(pyramids[f"points_{side}_neigh_idx"])
                sb = self.narrow(pyramids[f"points_{side}_sub_idx"])
                ip = self.narrow(pyramids[f"points_{side}_interp_idx"])
                keep += [x, nb, sb, ip]
                setattr(b, fld + "_xyz", _ptr(x)); setattr(b, fld + "_neigh", _ptr(nb))
                setattr(b, fld + "_sub", _ptr(sb)); setattr(b, fld + "_interp", _ptr(ip))
        return b, keep, (P, J, K)

    def register(self, points_src, points_ref, n_iter: int = 5, pyramids: Optional[dict] = None,
                 forced_idx: Optional[torch.Tensor] = None, want_aux: bool = True, sync: bool = True, out: Optional[dict] = None,
                 want_desc: bool = False):
        """forward_align_4 counterpart for P pairs.

        points_* [P, N, feat_len].  pyramids: optional dict with the reference's
        data-dict keys (points_{src,ref}_{xyz,neigh_idx,sub_idx,interp_idx}),
        int32 or int64.  Returns dict(transforms [P,n_iter,3,4], idx [n_iter,P,J],
        logits [n_iter,P,J], pt_ref_new [P,J,3], invalid [P])."""
        b, keep, (P, J, K) = self._pair_batch(points_src, points_ref, pyramids)
        if forced_idx is not None:
            forced_idx = _chk(forced_idx, torch.int32, "forced_idx")
            assert tuple(forced_idx.shape) == (n_iter, P, J)
            b.forced_idx = _ptr(forced_idx)
        if out is None:
            out = {"transforms": self._empty((P, n_iter, 3, 4))}
            if want_aux:
                out["idx"] = self._empty((n_iter, P, J), torch.int32)
                out["logits"] = self._empty((n_iter, P, J))
                out["pt_ref_new"] = self._empty((P, J, 3))
                out["invalid"] = self._empty((P,), torch.int32)
            if want_desc:     # test aid: the descriptors every iteration's search ran on
                out["desc_src"] = self._empty((n_iter, P, J, 64))
                out["desc_ref"] = self._empty((P, K, 64))
        r = _lib.dsir_pair_result()
        r.transforms = _ptr(out["transforms"])
        r.idx, r.logits = _ptr(out.get("idx")), _ptr(out.get("logits"))
        r.pt_ref_new, r.invalid = _ptr(out.get("pt_ref_new")), _ptr(out.get("invalid"))
        r.desc_src, r.desc_ref = _ptr(out.get("desc_src")), _ptr(out.get("desc_ref"))
        self._pre()
        self._call(self.lib.dsir_register(self.h, C.byref(b), n_iter, C.byref(r)))
        if sync:
            self.sync()
        out["_keep"] = keep
        return out

    def forward_pair(self, points_src, points_ref, num_sub: Optional[int] = None, pyramids: Optional[dict] = None,
                     sync: bool = True):
        """Network.forward_pair counterpart (reference model.py:609-666) for P pairs, point-major:
        returns {side: {xyz [P,M,3], feat [P,M,64], logits [P,N,ncls], score [P,M], label [P,M], index [P,M]}}
        with M = num_sub if num_sub > 0 else N; score/label/index only where the pipeline produces them."""
        num_sub = self.cfg.num_sub if num_sub is None else int(num_sub)
        b, keep, (P, J, K) = self._pair_batch(points_src, points_ref, pyramids)
        res, structs = {}, []
        for side, n in (("src", J), ("ref", K)):
            M = num_sub if num_sub > 0 else n
            o = {"xyz": self._empty((P, M, 3)), "feat": self._empty((P, M, 64)),
                 "logits": self._empty((P, n, self.cfg.num_classes))}
            if self.cfg.pipeline != "label":
                o["score"] = self._empty((P, M))
                o["label"] = self._empty((P, M), torch.int32)
                if num_sub > 0:
                    o["index"] = self._empty((P, M), torch.int32)
            c = _lib.dsir_cloud_out()
            for k in ("xyz", "feat", "logits", "score", "label", "index"):
                setattr(c, k, _ptr(o.get(k)))
            res[side] = o
            structs.append(c)
        self._pre()
        self._call(self.lib.dsir_forward_pair(self.h, C.byref(b), int(num_sub), C.byref(structs[0]), C.byref(structs[1])))
        if sync:
            self.sync()
        res["_keep"] = keep
        return res

    # ------------------------------------------------------------------ in front of the path: pre-processing
    def voxel_downsample(self, clouds: Sequence[torch.Tensor], voxel_size: float, crop: Optional[Sequence[float]] = None,
                         cap: Optional[int] = None):
        """open3d voxel_down_sample (+ process_point_cloud crop) counterpart for a ragged list of [n_i, C] CUDA
        clouds -> (voxels [clouds, cap, C], counts [clouds] i32).  See include/dsir.h for the ordering rule."""
        pts = torch.cat([_chk(c, torch.float32, "cloud") for c in clouds], 0)
        n = [int(c.shape[0]) for c in clouds]
        stride = pts.shape[1]
        offs = (C.c_int64 * (len(n) + 1))(*np.concatenate([[0], np.cumsum(n)]).tolist())
        cap = int(cap or max(n))
        out = self._empty((len(n), cap, stride))
        counts = self._empty((len(n),), torch.int32)
        crop_arr = None if crop is None else (C.c_float * 4)(*[float(x) for x in crop])
        self._pre()
        self._call(self.lib.dsir_voxel_downsample(self.h, _ptr(pts), offs, len(n), stride, float(voxel_size), crop_arr, cap,
                                                  _ptr(out), _ptr(counts)))
        return out, counts

    def resample(self, voxels: torch.Tensor, counts: torch.Tensor, k: int, seed: int = 0, mode: str = "random"):
        """Resampler / FixedResampler counterpart: [clouds, cap, C] + counts -> [clouds, k, C]."""
        voxels, counts = _chk(voxels, torch.float32, "voxels"), _chk(counts, torch.int32, "counts")
        c, cap, stride = voxels.shape
        out = self._empty((c, k, stride))
        self._pre()
        self._call(self.lib.dsir_resample(self.h, _ptr(voxels), _ptr(counts), c, cap, stride, int(k),
                                          {"random": 0, "fixed": 1}[mode], int(seed) & ((1 << 64) - 1), _ptr(out)))
        self.sync()
        return out

    def preprocess(self, clouds: Sequence[torch.Tensor], voxel_size: float, k: int, seed: int = 0,
                   crop: Optional[Sequence[float]] = None, mode: str = "random"):
        """Raw clouds -> [clouds, k, C] network input (crop, voxel average, resample in random order)."""
        vox, counts = self.voxel_downsample(clouds, voxel_size, crop)
        return self.resample(vox, counts, k, seed, mode), counts

    # ------------------------------------------------------------------ after the path: metrics
    METRIC_NAMES = ("r_mse", "r_mae", "t_mse", "t_mae", "err_r_deg", "err_t", "succ", "chamfer_dist")

    def eval_metrics(self, pred, gt, points_src, points_ref, rte_thresh: float, rre_thresh: float):
        """compute_metrics counterpart (reference common/metrics_util.py:27-85).
        pred [P,3,4] (may be a strided view such as transforms[:, i]), gt [P,3,4], points_* [P,N,>=3]
        -> dict of float64 tensors [P] keyed by METRIC_NAMES."""
        points_src, points_ref = _chk(points_src, torch.float32, "points_src"), _chk(points_ref, torch.float32, "points_ref")
        gt = _chk(gt, torch.float32, "transform_gt")
        if not pred.is_cuda or pred.dtype != torch.float32 or pred.stride(-1) != 1 or pred.stride(-2) != 4:
            pred = pred.float().contiguous().to(self.device)
        P, n, stride = points_src.shape
        out = self._empty((P, 8), torch.float64)
        self._pre()
        self._call(self.lib.dsir_eval_metrics(self.h, _ptr(pred), pred.stride(0) if P > 1 else 12, _ptr(gt), _ptr(points_src),
                                              _ptr(points_ref), P, n, stride, float(rte_thresh), float(rre_thresh), _ptr(out)))
        self.sync()
        return {k: out[:, i] for i, k in enumerate(self.METRIC_NAMES)}

    # ------------------------------------------------------------------ measurement hooks
    def icp_refine(self, points_src, points_ref, T_init, max_corr_dist: float, max_iter: int = 30,
                   rel_fitness: float = 1e-6, rel_rmse: float = 1e-6):
        """open3d registration_icp (point-to-point) counterpart for P pairs (reference test.py:241-258, disabled there).
        points_* [P,N,>=3] CUDA fp32, T_init [P,3,4] -> (T [P,3,4], stats [P,4] f64: fitness, inlier_rmse, converged, iters)."""
        points_src, points_ref = _chk(points_src, torch.float32, "points_src"), _chk(points_ref, torch.float32, "points_ref")
        T_init = _chk(T_init, torch.float32, "T_init")
        P, J, stride = points_src.shape
        K = points_ref.shape[1]
        assert points_ref.shape[0] == P and points_ref.shape[2] == stride and tuple(T_init.shape) == (P, 3, 4)
        T = self._empty((P, 3, 4))
        stats = self._empty((P, 4), torch.float64)
        self._pre()
        self._call(self.lib.dsir_icp_refine(self.h, _ptr(points_src), _ptr(points_ref), P, J, K, stride, float(max_corr_dist),
                                            int(max_iter), float(rel_fitness), float(rel_rmse), _ptr(T_init), _ptr(T), _ptr(stats)))
        self.sync()
        return T, stats

    def pose_finetune(self, xyz_src, xyz_ref, T_init, weights=None, weights_are_logits: bool = False,
                      quantization_size: float = 1.0, max_iter: int = 1000, break_threshold_ratio: float = 1e-4,
                      max_break_count: int = 20):
        """transformation_finetune counterpart for P pairs (reference test.py:159-207, disabled there).
        xyz_src / xyz_ref [P,m,3] matched points, T_init [P,3,4], weights [P,m] or None
        -> (T [P,3,4], stats [P,3] f64: iterations, loss, break_count)."""
        xyz_src, xyz_ref = _chk(xyz_src, torch.float32, "xyz_src"), _chk(xyz_ref, torch.float32, "xyz_ref")
        T_init = _chk(T_init, torch.float32, "T_init")
        P, m, three = xyz_src.shape
        if three != 3 or tuple(xyz_ref.shape) != (P, m, 3) or tuple(T_init.shape) != (P, 3, 4):
            raise EngineError("pose_finetune: xyz_src / xyz_ref must be [P,m,3] and T_init [P,3,4]")
        if weights is not None:
            weights = _chk(weights.reshape(P, -1), torch.float32, "weights")
            if weights.shape[1] != m:
                raise EngineError("pose_finetune: weights must be [P,m]")
        T = self._empty((P, 3, 4))
        stats = self._empty((P, 3), torch.float64)
        self._pre()
        self._call(self.lib.dsir_pose_finetune(self.h, _ptr(xyz_src), _ptr(xyz_ref), _ptr(weights), 1 if weights_are_logits else 0,
                                               P, m, _ptr(T_init), float(quantization_size), int(max_iter),
                                               float(break_threshold_ratio), int(max_break_count), _ptr(T), _ptr(stats)))
        self.sync()
        return T, stats

    def align_loss_backward(self, pt_src, pt_ref, idx, logits, labels, transform_gt, loss_type: str = "mae",
                            wt_ptDist_loss: float = 1.0, wt_inlier_loss: float = 1.0, loss_discount_factor: float = 0.5,
                            per_pair: bool = False):
        """ScanAlignmentLoss (reduction='mean') + its gradient down to the inlier logits (include/dsir.h).
        pt_src [P,J,3], pt_ref [P,K,3], idx [n,P,J] i32, logits [n,P,J], labels [n,P,J] or None, transform_gt [P,3,4]
        -> dict(losses {mae_i|mse_i, outlier_i, total}, grad_logits [n,P,J], transforms [P,n,3,4])."""
        pt_src, pt_ref = _chk(pt_src, torch.float32, "pt_src"), _chk(pt_ref, torch.float32, "pt_ref")
        idx, logits = _chk(idx, torch.int32, "idx"), _chk(logits, torch.float32, "logits")
        transform_gt = _chk(transform_gt, torch.float32, "transform_gt")
        if labels is not None:
            labels = _chk(labels, torch.float32, "labels")
        n, P, J = logits.shape
        K = pt_ref.shape[1]
        if tuple(pt_src.shape) != (P, J, 3) or tuple(idx.shape) != (n, P, J) or tuple(transform_gt.shape) != (P, 3, 4) or \
                (labels is not None and tuple(labels.shape) != (n, P, J)):
            raise EngineError("align_loss_backward: inconsistent shapes")
        lt = {"mae": 0, "mse": 1}[loss_type]
        T = self._empty((P, n, 3, 4))
        grad = self._empty((n, P, J))
        losses = (C.c_double * (2 * n))()
        pp = (C.c_double * (2 * n * P))() if per_pair else None
        self._pre()
        self._call(self.lib.dsir_align_loss_backward2(self.h, _ptr(pt_src), _ptr(pt_ref), _ptr(idx), _ptr(logits), _ptr(labels),
                                                      _ptr(transform_gt), P, J, K, n, lt, float(wt_ptDist_loss), float(wt_inlier_loss),
                                                      float(loss_discount_factor), _ptr(T), losses, _ptr(grad), pp))
        self.sync()
        d, total = {}, 0.0
        for i in range(n):
            disc = loss_discount_factor ** (n - i - 1)
            if wt_ptDist_loss > 0:
                d[f"{loss_type}_{i}"] = losses[2 * i]; total += disc * losses[2 * i]
            if wt_inlier_loss > 0 and labels is not None:
                d[f"outlier_{i}"] = losses[2 * i + 1]; total += disc * losses[2 * i + 1]
        d["total"] = total
        out = {"losses": d, "grad_logits": grad, "transforms": T}
        if per_pair:      # reduction='none' (loss.py:779, :836): every pair's own terms, [P] each
            a = np.frombuffer(pp, dtype=np.float64).reshape(P, n, 2)
            dd, tot = {}, np.zeros(P)
            for i in range(n):
                disc = loss_discount_factor ** (n - i - 1)
                if wt_ptDist_loss > 0:
                    dd[f"{loss_type}_{i}"] = a[:, i, 0].copy(); tot += disc * a[:, i, 0]
                if wt_inlier_loss > 0 and labels is not None:
                    dd[f"outlier_{i}"] = a[:, i, 1].copy(); tot += disc * a[:, i, 1]
            dd["total"] = tot
            out["losses_per_pair"] = dd
        return out

    def enable_graph(self, on=True):
        """Replay dsir_register through a captured hipGraph (same buffers on every call; one graph per call signature)."""
        if on:
            from . import graph_replay_safe
            if not graph_replay_safe():
                raise EngineError("hipGraph replay needs DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 in the environment BEFORE the process "
                                  "first touches the GPU (this ROCm's graph packet capture replays wrongly; deepsir_amd/__init__.py): "
                                  "import deepsir_amd before creating CUDA tensors, or export the variable")
        self._call(self.lib.dsir_enable_graph(self.h, 1 if on else 0))

    def graph_stats(self) -> Dict[str, int]:
        """Launch census of the registration this engine captured last (include/dsir.h, dsir_graph_stats)."""
        out = (C.c_int64 * 4)()
        self._call(self.lib.dsir_graph_stats(self.h, out))
        return {"nodes": int(out[0]), "kernels": int(out[1]), "memsets": int(out[2]), "memcpys": int(out[3])}

    def enable_fork(self, on=True):
        """A/B switch: independent branches of a small launch's schedule on auxiliary streams, or (default: measured faster) everything
        on one stream.  Same bits either way (include/dsir.h, dsir_enable_fork)."""
        self._call(self.lib.dsir_enable_fork(self.h, 1 if on else 0))

    def enable_walk(self, on=True):
        """A/B switch: the deep pyramid levels of a RandLA pass as one persistent launch (csrc/walk.hip; up to 16 clouds per launch;
        OFF by default - fewer launches, but measured slower) or every layer its own launch.  Same bits either way (include/dsir.h)."""
        self._call(self.lib.dsir_enable_walk(self.h, 1 if on else 0))

    def enable_match_timer(self, on=True):
        self._call(self.lib.dsir_enable_match_timer(self.h, 1 if on else 0))

    def match_timer(self, reset=True):
        ms, n = C.c_double(), C.c_int64()
        self._call(self.lib.dsir_match_timer(self.h, 1 if reset else 0, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def match_timer2(self, reset=True):
        """(operation ms, dominant-kernel ms, launches) of the timed arg-min searches (HIP events on the engine's stream)."""
        op, k, n = C.c_double(), C.c_double(), C.c_int64()
        self._call(self.lib.dsir_match_timer2(self.h, 1 if reset else 0, C.byref(op), C.byref(k), C.byref(n)))
        return op.value, k.value, n.value

    def enable_screen(self, on=True):
        """A/B switch: off = the exhaustive exact-fp32 arg-min kernel throughout (same bits, include/dsir.h)."""
        self._call(self.lib.dsir_enable_screen(self.h, 1 if on else 0))

    def enable_agg_split(self, on=True):
        """A/B switch: fp16-split aggregation chain (default) vs the exact-fp32 chain (include/dsir.h, dsir_enable_agg_split)."""
        self._call(self.lib.dsir_enable_agg_split(self.h, 1 if on else 0))

    SCREEN_STAT_NAMES = ("screened_searches", "rows_searched", "rows_undecided", "pairs_exhaustive", "exhaustive_searches")

    def screen_stats(self, reset=True) -> Dict[str, int]:
        """Which arg-min path dsir_register took since the last reset and how selective the screening was
        (include/dsir.h, dsir_screen_stats)."""
        out = (C.c_int64 * 5)()
        self._call(self.lib.dsir_screen_stats(self.h, 1 if reset else 0, out))
        return {k: int(out[i]) for i, k in enumerate(self.SCREEN_STAT_NAMES)}

    def prune_stats(self, reset=True) -> Tuple[int, int]:
        """(tile products visited, tile products without pruning) of the pruned search since the last reset (dsir_prune_stats)."""
        out = (C.c_int64 * 2)()
        self._call(self.lib.dsir_prune_stats(self.h, 1 if reset else 0, out))
        return int(out[0]), int(out[1])

    def set_prune_thresholds(self, min_points: int = 8192, min_rows: int = 65536):
        """A/B switch: the pruned search runs for ref clouds of min_points points and more (0 = never) in launches of min_rows src
        rows and more; same bits either way (include/dsir.h, dsir_set_prune_thresholds)."""
        self._call(self.lib.dsir_set_prune_thresholds(self.h, int(min_points), int(min_rows)))

    def set_kabsch_chunked_min(self, min_points: int = 0):
        """A/B switch: clouds of min_points points and more solve their pose in chunks (include/dsir.h); <= 0: the default."""
        self._call(self.lib.dsir_set_kabsch_chunked_min(self.h, int(min_points)))

    def match_timer_device(self, reset=True):
        """(total ms, launches) of the timed nn_match launches on the device clock (first wave start .. last wave end)."""
        ms, n = C.c_double(), C.c_int64()
        self._call(self.lib.dsir_match_timer_device(self.h, 1 if reset else 0, C.byref(ms), C.byref(n)))
        return ms.value, n.value


class EnginePool:
    """S engines (each with its own HIP stream and workspace) registering disjoint slices of a batch
    concurrently.  Kernels of one registration are serialised on their stream; two streams let the GPU
    overlap one slice's latency-/memory-bound RandLA kernels with the other's MFMA-bound matching
    (+8-10 % pairs/s on MI355X at 64 pairs, tools/multistream.py).  Results are identical to a single
    engine: pairs are independent and every kernel's tiling depends on the per-cloud shape only."""

    def __init__(self, cfg: NetConfig, device: int = 0, max_points: int = 8192, max_pairs: int = 2, streams: int = 2):
        self.streams = max(1, int(streams))
        self.per = (int(max_pairs) + self.streams - 1) // self.streams
        self.engines = [Engine(cfg, device, max_points, self.per) for _ in range(self.streams)]
        self.cfg, self.device = cfg, self.engines[0].device

    def load_state_dict(self, sd, strict: bool = True):
        for e in self.engines:
            e.load_state_dict(sd, strict)

    def close(self):
        for e in self.engines:
            e.close()

    def sync(self):
        for e in self.engines:
            e.sync()

    def _slices(self, P):
        per = (P + self.streams - 1) // self.streams
        return [(a, min(P, a + per)) for a in range(0, P, per)]

    def register(self, points_src, points_ref, n_iter: int = 5, want_aux: bool = True, sync: bool = True,
                 out: Optional[dict] = None, pyramids: Optional[dict] = None, want_desc: bool = False):
        P = points_src.shape[0]
        if P > self.per * self.streams:
            raise EngineError(f"pairs={P} exceeds the pool's max_pairs={self.per * self.streams}")
        sl = self._slices(P)
        if out is None:
            out = {"transforms": torch.empty((P, n_iter, 3, 4), dtype=torch.float32, device=self.device)}
        parts = []
        for e, (a, b) in zip(self.engines, sl):
            o = {"transforms": out["transforms"][a:b]} if not want_aux else None
            pyr = None if pyramids is None else {k: v[a:b] for k, v in pyramids.items()}
            parts.append(e.register(points_src[a:b], points_ref[a:b], n_iter, want_aux=want_aux, sync=False, out=o,
                                    pyramids=pyr, want_desc=want_desc and want_aux))
        if sync or want_aux:
            self.sync()
        if want_aux:
            out["transforms"] = torch.cat([p["transforms"] for p in parts], 0)
            out["idx"] = torch.cat([p["idx"] for p in parts], 1)
            out["logits"] = torch.cat([p["logits"] for p in parts], 1)
            out["pt_ref_new"] = torch.cat([p["pt_ref_new"] for p in parts], 0)
            out["invalid"] = torch.cat([p["invalid"] for p in parts], 0)
            if want_desc:
                out["desc_src"] = torch.cat([p["desc_src"] for p in parts], 1)
                out["desc_ref"] = torch.cat([p["desc_ref"] for p in parts], 0)
        out["_parts"] = parts
        return out

    def enable_match_timer(self, on=True):
        for e in self.engines:
            e.enable_match_timer(on)

    def match_timer(self, reset=True):
        ms = n = 0
        for e in self.engines:
            m, k = e.match_timer(reset)
            ms, n = ms + m, n + k
        return ms, n

    def match_timer_device(self, reset=True):
        ms = n = 0
        for e in self.engines:
            m, k = e.match_timer_device(reset)
            ms, n = ms + m, n + k
        return ms, n

    def enable_graph(self, on=True):
        for e in self.engines:
            e.enable_graph(on)

    def match_timer2(self, reset=True):
        op = k = n = 0
        for e in self.engines:
            a, b, c = e.match_timer2(reset)
            op, k, n = op + a, k + b, n + c
        return op, k, n

    def enable_screen(self, on=True):
        for e in self.engines:
            e.enable_screen(on)

    def enable_agg_split(self, on=True):
        for e in self.engines:
            e.enable_agg_split(on)

    def screen_stats(self, reset=True) -> Dict[str, int]:
        tot: Dict[str, int] = {}
        for e in self.engines:
            for k, v in e.screen_stats(reset).items():
                tot[k] = tot.get(k, 0) + v
        return tot

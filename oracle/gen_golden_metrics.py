"""Golden vectors for the evaluation metrics, produced by the REFERENCE's own
``common.metrics_util.compute_metrics`` (metrics_util.py:27-85) imported from /root/reference
(build container only; TEST INFRASTRUCTURE).

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden_metrics.py

The reference calls ``scipy.spatial.transform.Rotation.from_dcm`` (common/math/so3.py:23), an API scipy
renamed to ``from_matrix`` in 1.4 and removed in 1.6; the installed scipy is newer, so this script
rebinds the name ``Rotation`` inside the imported ``common.math.so3`` module (in memory) to a two-line
adapter that spells ``from_dcm`` as ``from_matrix``.  That is the only deviation; no reference file is touched.
"""
import os
import sys
import warnings

import numpy as np
import torch
from scipy.spatial.transform import Rotation

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepsir_amd.synth import make_pair, random_rotation  # noqa: E402



class _RotationCompat:
    """scipy < 1.6 spelling used by the reference (from_dcm) on top of the installed scipy (from_matrix)."""
    from_dcm = staticmethod(Rotation.from_matrix)


def main(ref_root="/root/reference"):
    warnings.filterwarnings("ignore")
    sys.path.insert(0, ref_root)
    import common.math.so3 as ref_so3  # type: ignore
    if not hasattr(Rotation, "from_dcm"):
        ref_so3.Rotation = _RotationCompat      # module global of the imported reference, in memory only
    from common.metrics_util import compute_metrics, rte_rre  # type: ignore

    rng = np.random.Generator(np.random.Philox(key=4242))
    out = {}
    cases = []
    for i, (n, noise_r, noise_t) in enumerate([(2048, 0.02, 0.01), (1024, 0.3, 0.2), (1500, 1.5, 1.0), (2048, 0.0, 0.0),
                                              (3000, 0.1, 0.4), (1024, 3.0, 0.05)]):
        p = make_pair(n, 900 + i, 3)
        gt = p["transform_gt"][0].astype(np.float64)
        # prediction = ground truth perturbed by a small random rotation / translation
        ax = rng.standard_normal(3); ax /= np.linalg.norm(ax)
        dR = Rotation.from_rotvec(ax * noise_r).as_matrix()
        pred = np.concatenate([dR @ gt[:, :3], (gt[:, 3] + rng.standard_normal(3) * noise_t)[:, None]], 1)
        cases.append((p, pred.astype(np.float32)))
    # one batch of 2 (the reference evaluates [:1024] slices, test.py:331-332; compute_metrics itself slices [:2048])
    for i, (p, pred) in enumerate(cases):
        data = {"transform_gt": torch.from_numpy(p["transform_gt"]), "points_src": torch.from_numpy(p["points_src"]),
                "points_ref": torch.from_numpy(p["points_ref"])}
        m = compute_metrics(data, torch.from_numpy(pred[None]), 0.3, 15.0)
        out[f"c{i}_src"], out[f"c{i}_ref"] = p["points_src"], p["points_ref"]
        out[f"c{i}_gt"], out[f"c{i}_pred"] = p["transform_gt"], pred[None]
        for k, v in m.items():
            out[f"c{i}_{k}"] = np.asarray(v, dtype=np.float64)
        # the harness's own criterion (metrics_util.py:13-24, called at test.py:432-441) on the same poses, fp32 as the
        # harness holds them, under both threshold sets (test.py:49-54)
        out[f"c{i}_rte_rre_3dmatch"] = np.asarray(rte_rre(pred, p["transform_gt"][0], 0.3, 15.0), dtype=np.float64)
        out[f"c{i}_rte_rre_kitti"] = np.asarray(rte_rre(pred, p["transform_gt"][0], 0.6, 5.0), dtype=np.float64)
    out["rte_rre_none"] = np.asarray(rte_rre(None, cases[0][0]["transform_gt"][0], 0.3, 15.0), dtype=np.float64)
    out["n_cases"] = np.asarray(len(cases))
    out["thresholds"] = np.asarray([0.3, 15.0])
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "metrics_cases.npz"), **out)
    print("wrote metrics_cases.npz:", {k: out[k] for k in out if k.startswith("c0_") and out[k].size < 4})


if __name__ == "__main__":
    main()

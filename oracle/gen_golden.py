"""Generate the golden fixtures under tests/golden/ by importing the REFERENCE
(TEST INFRASTRUCTURE ONLY; runs in the build container, where /root/reference
is mounted — never on the GPU box).

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py [--ref /root/reference]

What is committed is data only: seeds, small input checksums and the
reference's outputs.  Inputs are re-created from the seeds by
``deepsir_amd.synth`` / ``deepsir_amd.weights`` / ``oracle.knn`` (the KNN
pyramid is ours: the reference's ``torch_points_kernels`` is not installable
here).  The reference is called through its own public entry
points: ``Network(args)``, ``load_state_dict``, ``net(data, (n_iter, True))``
(network/model.py:297-298,520-607) and, for per-stage vectors,
``net.feat_extractor`` (RandLANet.py:311), ``net.forward_pair`` (model.py:609),
``net.aggregation`` (:209), ``match_features_V2`` (matchnet.py:116),
``net.inlier_model`` and ``compute_rigid_transform_2`` (model.py:22).
"""
from __future__ import annotations

import argparse
import copy
import hashlib
import json
import os
import sys
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from deepsir_amd.arch import NetConfig  # noqa: E402
from deepsir_amd.synth import make_pair  # noqa: E402
from deepsir_amd.weights import generate_state_dict, to_torch_state_dict  # noqa: E402
from oracle.knn import add_pyramids  # noqa: E402
from oracle.network import to_torch  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def digest(*arrays) -> str:
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:16]


def build_reference(ref_root: str, feat_len: int):
    sys.path.insert(0, ref_root)
    warnings.filterwarnings("ignore")
    import arguments  # type: ignore
    import network.model as ref_model  # type: ignore
    import network.matchnet as ref_match  # type: ignore

    args = arguments.eval_arguments().parse_args([])
    args.pipeline, args.feat_len, args.num_sub = "align", feat_len, -1
    net = ref_model.Network(args).eval()
    return net, ref_model, ref_match, args


def np_(t):
    return t.detach().cpu().numpy()


def run_case(net, ref_model, ref_match, cfg, n, seed, wseed, variant, full, n_iter=5):
    sd = generate_state_dict(cfg, wseed, variant)
    net.load_state_dict(to_torch_state_dict(sd), strict=True)
    raw = make_pair(n, seed, cfg.feat_len)
    data_np = add_pyramids(raw, cfg.num_knn, cfg.sub_sampling_ratio)
    data = to_torch(data_np)
    out = {
        "meta": json.dumps(dict(n=n, seed=seed, wseed=wseed, variant=variant, feat_len=cfg.feat_len, n_iter=n_iter,
                                threads=torch.get_num_threads(), torch=torch.__version__)),
        "input_digest": digest(data_np["points_src"], data_np["points_ref"], data_np["points_src_neigh_idx"],
                               data_np["points_ref_interp_idx"]),
        "weights_digest": digest(*[v for v in sd.values()]),
    }
    with torch.no_grad():
        transforms, ep = net(data, (n_iter, True))
        out["transforms"] = np.stack([np_(t) for t in transforms], 1)              # [B,n_iter,3,4]
        idx = np.stack([np_(p)[:, :, 1] for p in ep["pred_pairs"]], 1)             # [B,n_iter,J]
        out["idx"] = idx.astype(np.int16 if n < 32768 else np.int32)
        out["logits"] = np.stack([np_(l) for l in ep["perm_matrices"]], 1)         # [B,n_iter,J]
        out["invalid"] = np.asarray(bool(ep["invalid_gradient"]))
        out["pt_ref_new"] = np_(ep["pt_ref_new"])
        f_s, x_s, lab_s, sc_s, f_r, x_r, lab_r, sc_r = net.forward_pair(data)
        out["score_src"], out["score_ref"] = np_(sc_s), np_(sc_r)
        out["label_src"], out["label_ref"] = np_(lab_s).astype(np.int8), np_(lab_r).astype(np.int8)
        out["feat_src_sum"] = np.array([f_s.double().sum().item(), (f_s.double() ** 2).sum().item()])
        out["feat_ref_sum"] = np.array([f_r.double().sum().item(), (f_r.double() ** 2).sum().item()])
        if full:
            out["feat_src"], out["feat_ref"] = np_(f_s), np_(f_r)
            _, _, lg = net.feat_extractor(data["points_src"], data["points_src_xyz"], data["points_src_neigh_idx"],
                                          data["points_src_sub_idx"], data["points_src_interp_idx"])
            out["logits_src"] = np_(lg)
            d_s, d_r = net.aggregation(x_s, x_r, f_s, f_r, lab_s, lab_r, sc_s, sc_r)
            out["desc_src0"], out["desc_ref0"] = np_(d_s), np_(d_r)
            m = ref_match.match_features_V2(d_s, d_r)
            out["idx0_direct"] = np_(m.min(dim=2)[1]).astype(np.int16)
            # one inlier pass + Kabsch, teacher-forced on the reference's own iteration-0 indices
            i0 = torch.from_numpy(idx[:, 0].astype(np.int64))
            ref_new = torch.gather(x_r, 2, i0[:, None, :].expand(-1, 3, -1))
            cat = torch.cat((x_s, ref_new), 1).permute(0, 2, 1).contiguous()
            f_in, _, lg_in = net.inlier_model(cat, data["points_src_xyz"], data["points_src_neigh_idx"],
                                              data["points_src_sub_idx"], data["points_src_interp_idx"])
            out["inlier_feat0"] = np_(f_in)
            out["inlier_logit0"] = np_(lg_in)
            T0, bad = ref_model.compute_rigid_transform_2(x_s.permute(0, 2, 1).contiguous(),
                                                          ref_new.permute(0, 2, 1).contiguous(),
                                                          lg_in.squeeze(1).sigmoid()[:, :, None])
            out["kabsch_T0"] = np_(T0)
    return out


def kabsch_cases(ref_model):
    """Isolated vectors for compute_rigid_transform_2 incl. a reflection case
    (det(V U^T) < 0) and a non-finite case (identity + flag)."""
    rng = np.random.Generator(np.random.Philox(key=777))
    cases = {}

    def add(name, src, tgt, w):
        T, bad = ref_model.compute_rigid_transform_2(torch.from_numpy(src), torch.from_numpy(tgt), torch.from_numpy(w))
        cases[name + "_src"], cases[name + "_tgt"], cases[name + "_w"] = src, tgt, w
        cases[name + "_T"], cases[name + "_invalid"] = np_(T), np.asarray(bool(bad))

    from deepsir_amd.synth import random_rotation
    for i, m in enumerate((64, 1000, 5000)):
        src = rng.uniform(0, 3, (1, m, 3)).astype(np.float32)
        R, t = random_rotation(rng), rng.uniform(-0.5, 0.5, 3)
        tgt = (src @ R.T + t + rng.standard_normal((1, m, 3)) * 0.02).astype(np.float32)
        w = rng.uniform(0.0, 1.0, (1, m, 1)).astype(np.float32)
        add(f"rigid{i}", src, tgt, w)
    # reflection: tgt is a mirrored copy -> det(V U^T) < 0 -> V[:,2] flipped
    src = rng.uniform(-1, 1, (1, 500, 3)).astype(np.float32)
    tgt = (src * np.array([1, 1, -1], np.float32)).astype(np.float32)
    add("mirror", src, tgt, np.ones((1, 500, 1), np.float32))
    # random (unrelated) clouds with signed weights
    add("random", rng.standard_normal((1, 300, 3)).astype(np.float32), rng.standard_normal((1, 300, 3)).astype(np.float32),
        rng.standard_normal((1, 300, 1)).astype(np.float32))
    # two clouds in a batch
    add("batch2", rng.standard_normal((2, 200, 3)).astype(np.float32), rng.standard_normal((2, 200, 3)).astype(np.float32),
        rng.uniform(0, 1, (2, 200, 1)).astype(np.float32))
    # non-finite input -> SVD raises -> identity + invalid flag (model.py:61-64)
    bad = rng.standard_normal((1, 50, 3)).astype(np.float32)
    bad[0, 3, 1] = np.nan
    import logging
    logging.disable(logging.CRITICAL)
    add("nan", bad, rng.standard_normal((1, 50, 3)).astype(np.float32), np.ones((1, 50, 1), np.float32))
    logging.disable(logging.NOTSET)
    return cases


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--threads", type=int, default=1)
    ap.add_argument("--keys-only", action="store_true", help="only (re)write state_dict_keys.json (keys, shapes, parameter list)")
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    os.makedirs(GOLD, exist_ok=True)
    cfg = NetConfig(feat_len=3)
    net, ref_model, ref_match, args = build_reference(a.ref, 3)

    keys = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in net.state_dict().items()]
    # what ``optim.Adam(my_model.parameters(), lr)`` (train.py:323) is handed, per pipeline: names in order + requires_grad
    # (freeze_model / freeze_model_2, model.py:196-207)
    params = {}
    for pipe in ("align", "feat", "label"):
        pargs = copy.copy(args)
        pargs.pipeline, pargs.num_sub = pipe, (-1 if pipe == "align" else 512)
        pargs.thres_radius = 1.0   # only read by the training loss that Network.__init__ constructs (loss.py:498)
        pnet = ref_model.Network(pargs)
        params[pipe] = [[k, bool(v.requires_grad)] for k, v in pnet.named_parameters()]
    with open(os.path.join(GOLD, "state_dict_keys.json"), "w") as f:
        json.dump({"feat_len": 3, "keys": keys, "parameters": params}, f, indent=0)
    if a.keys_only:
        return

    np.savez_compressed(os.path.join(GOLD, "kabsch_cases.npz"), **kabsch_cases(ref_model))

    plan = [  # (name, n, seed, wseed, variant, full)
        ("stage_n1024_s1", 1024, 1, 0, "plain", True),
        ("stage_n1024_s2_sep", 1024, 2, 0, "separated", True),
        ("e2e_n2048_s3", 2048, 3, 0, "plain", False),
        ("e2e_n2048_s4_sep", 2048, 4, 0, "separated", False),
        ("e2e_n5000_s5", 5000, 5, 0, "plain", False),
    ]
    for name, n, seed, wseed, variant, full in plan:
        out = run_case(net, ref_model, ref_match, cfg, n, seed, wseed, variant, full)
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
        print(name, "T_last =", out["transforms"][0, -1].round(4).tolist())

    # KITTI-shaped (feat_len 4) compact case
    cfg4 = NetConfig(feat_len=4)
    net4, ref_model4, ref_match4, _ = build_reference(a.ref, 4)
    out = run_case(net4, ref_model4, ref_match4, cfg4, 2048, 6, 1, "plain", False)
    np.savez_compressed(os.path.join(GOLD, "e2e_n2048_s6_f4.npz"), **out)
    print("done")


if __name__ == "__main__":
    main()

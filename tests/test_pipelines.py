"""'feat' / 'label' pipelines of Network (reference model.py:173-179, :609-697):
oracle pinned against vectors from the imported reference (oracle/gen_golden_pipelines.py); HIP path
(dsir_forward_pair through the C ABI) against the oracle and the same vectors."""
import numpy as np
import pytest
import torch

from conftest import build_case
from oracle.network import OracleNet, to_torch

CASES = ["label_n1024_s11", "feat_n1024_s12", "feat_n2048_s13_sub512", "feat_n2048_s14_sub300_f4"]


def _check_selection(pt, score, g_pt, g_score, tag, num_sub):
    """torch.topk leaves the order of equal scores open: equal multisets of scores, and identical points wherever a
    score is not tied with a neighbour."""
    np.testing.assert_allclose(score, g_score, rtol=1e-5, atol=1e-7, err_msg=tag)
    if num_sub <= 0:
        np.testing.assert_array_equal(pt, g_pt, err_msg=tag)
        return
    s = g_score[0]
    strict = np.ones(len(s), bool)
    strict[1:] &= s[1:] < s[:-1] * (1 - 1e-5)
    strict[:-1] &= s[:-1] * (1 - 1e-5) > s[1:]
    assert strict.mean() > 0.5, "fixture is degenerate"
    np.testing.assert_array_equal(pt[0][:, strict], g_pt[0][:, strict], err_msg=tag)


@pytest.mark.parametrize("name", CASES)
def test_oracle_pipelines_match_reference(name):
    torch.set_num_threads(1)
    g, m, cfg, sd, data = build_case(name)
    ep = OracleNet(cfg, sd).forward_endpoints(to_torch(data), m["pipeline"], m["num_sub"])
    for s in ("src", "ref"):
        np.testing.assert_allclose(ep[f"logits_{s}"].numpy(), g[f"logits_{s}"], rtol=1e-5, atol=1e-6)
        if m["pipeline"] == "feat":
            _check_selection(ep[f"pt_{s}"].numpy(), ep[f"score_{s}"].numpy(), g[f"pt_{s}"], g[f"score_{s}"], name + s, m["num_sub"])
        same = np.all(ep[f"pt_{s}"].numpy() == g[f"pt_{s}"], axis=1)[0]          # rows where the selection coincides
        assert same.mean() > 0.9
        np.testing.assert_allclose(ep[f"feat_{s}"].numpy()[0][:, same], g[f"feat_{s}"][0][:, same], rtol=1e-5, atol=1e-6)
    assert sorted(k for k in ep if not k.startswith(("label_", "index_"))) == m["keys"]


def test_oracle_topk_tie_rule():
    """equal scores (incl. +0 / -0) are taken in ascending index"""
    sc = torch.tensor([[0.5, 0.0, 0.7, -0.0, 0.5, 0.0, 0.7]])
    idx = torch.sort(sc + 0.0, dim=-1, descending=True, stable=True)[1]
    assert idx.tolist() == [[2, 6, 0, 4, 1, 3, 5]]


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_gpu_pipelines(name):
    from deepsir_amd.engine import Engine
    torch.set_num_threads(1)
    g, m, cfg, sd, data = build_case(name)
    eng = Engine(cfg, max_points=m["n"], max_pairs=1)
    eng.load_state_dict(sd)
    assert len(eng.expected_keys()) == len(sd)
    src, ref = torch.from_numpy(data["points_src"]).cuda(), torch.from_numpy(data["points_ref"]).cuda()
    out = eng.forward_pair(src, ref, m["num_sub"])          # pyramids built on device
    ep = OracleNet(cfg, sd).forward_endpoints(to_torch(data), m["pipeline"], m["num_sub"])
    for s in ("src", "ref"):
        o = out[s]
        logits = o["logits"].permute(0, 2, 1).cpu().numpy()
        np.testing.assert_allclose(logits, g[f"logits_{s}"], rtol=1e-3, atol=2e-4)
        pt = o["xyz"].permute(0, 2, 1).cpu().numpy()
        feat = o["feat"].permute(0, 2, 1).cpu().numpy()
        if m["pipeline"] == "feat":
            score = o["score"].cpu().numpy()
            _check_selection(pt, score, g[f"pt_{s}"], g[f"score_{s}"], name + s, m["num_sub"])
            np.testing.assert_allclose(score, ep[f"score_{s}"].numpy(), rtol=1e-5, atol=1e-7)
            if m["num_sub"] > 0:
                # index / label / xyz are consistent with each other and with the input cloud
                idx = o["index"].cpu().numpy().astype(np.int64)
                assert len(np.unique(idx[0])) == m["num_sub"]
                np.testing.assert_array_equal(pt[0].T, data[f"points_{s}"][0][idx[0], :3])
                agree = (idx == ep[f"index_{s}"].numpy()).mean()
                assert agree > 0.98, agree
            lab = o["label"].cpu().numpy()
            assert (lab == ep[f"label_{s}"].numpy()[:, 0]).mean() > 0.995
        same = np.all(pt == g[f"pt_{s}"], axis=1)[0]
        assert same.mean() > 0.9
        np.testing.assert_allclose(feat[0][:, same], g[f"feat_{s}"][0][:, same], rtol=1e-3, atol=1e-4)
        np.testing.assert_allclose(np.linalg.norm(feat[0], axis=0), 1.0, atol=1e-5)


@pytest.mark.gpu
def test_gpu_network_feat_label_dropin():
    """deepsir_amd.model.Network with args.pipeline in {'feat','label'}: reference calling convention and shapes."""
    from types import SimpleNamespace
    from deepsir_amd.model import Network
    from deepsir_amd.weights import to_torch_state_dict
    for name in ("label_n1024_s11", "feat_n2048_s13_sub512"):
        g, m, cfg, sd, data = build_case(name)
        args = SimpleNamespace(pipeline=m["pipeline"], num_sub=m["num_sub"], feat_len=m["feat_len"], num_knn=16,
                               out_feat_dim=64, d_out=[16, 64, 128, 256], sub_sampling_ratio=[4, 4, 4, 4],
                               clip_weight_thresh=0.0, use_ppf=False)
        net = Network(args)
        net.load_state_dict(to_torch_state_dict(sd), strict=True)
        net = net.cuda().eval()
        batch = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in data.items() if hasattr(v, "shape")}
        none, ep = net(batch)                                # caller-supplied int64 pyramids are honoured
        assert none is None
        assert sorted(ep.keys()) == m["keys"]
        for k in m["keys"]:
            assert tuple(ep[k].shape) == g[k].shape, k
        np.testing.assert_allclose(ep["logits_src"].cpu().numpy(), g["logits_src"], rtol=1e-3, atol=2e-4)


@pytest.mark.gpu
def test_gpu_pipeline_guards():
    from deepsir_amd.arch import NetConfig
    from deepsir_amd.engine import Engine, EngineError
    from deepsir_amd.weights import generate_state_dict
    cfg = NetConfig(pipeline="label")
    eng = Engine(cfg, max_points=1024, max_pairs=1)
    eng.load_state_dict(generate_state_dict(cfg, 0))
    x = torch.rand(1, 1024, 3, device="cuda")
    with pytest.raises(EngineError, match="align"):
        eng.register(x, x, 2)
    with pytest.raises(EngineError, match="feat pipeline"):
        eng.forward_pair(x, x, 100)
    full = generate_state_dict(NetConfig(), 0)
    with pytest.raises(Exception, match="unexpected key"):
        eng.load_state_dict(full)


@pytest.mark.gpu
def test_gpu_harness_feat_and_label():
    """inference_feat / inference_label (reference test.py:460-567) around the drop-in Network."""
    from types import SimpleNamespace
    from deepsir_amd.harness import inference_feat, inference_label
    from deepsir_amd.model import Network
    from deepsir_amd.weights import to_torch_state_dict

    def build(name):
        g, m, cfg, sd, data = build_case(name)
        args = SimpleNamespace(pipeline=m["pipeline"], num_sub=m["num_sub"], feat_len=m["feat_len"], num_knn=16,
                               out_feat_dim=64, d_out=[16, 64, 128, 256], sub_sampling_ratio=[4, 4, 4, 4],
                               clip_weight_thresh=0.0, use_ppf=False)
        net = Network(args)
        net.load_state_dict(to_torch_state_dict(sd), strict=True)
        return g, m, data, net.cuda().eval()

    g, m, data, net = build("feat_n2048_s13_sub512")
    pair = {k: v for k, v in data.items() if k.startswith("points_") and k.count("_") == 1}   # clouds only: KNN on device
    recs, t = inference_feat([pair, pair], net, batch=2)
    assert len(recs) == 2 and recs[0]["pt_src"].shape == (512, 3) and recs[0]["feat_ref"].shape == (512, 64) and t > 0
    np.testing.assert_allclose(recs[0]["score_src"], g["score_src"][0], rtol=1e-5, atol=1e-7)
    np.testing.assert_array_equal(recs[0]["score_src"], recs[1]["score_src"])

    g, m, data, net = build("label_n1024_s11")
    pair = {k: v for k, v in data.items() if k.startswith("points_") and k.count("_") == 1}
    want = {s: g[f"logits_{s}"][0].argmax(0) + 1 for s in ("src", "ref")}
    pair["labels_src"], pair["labels_ref"] = want["src"][None].astype(np.int64), want["ref"][None].astype(np.int64)
    preds, met, t = inference_label([pair], net)
    agree = np.mean(preds[0]["src"] == want["src"])
    assert agree > 0.995 and met["mean_acc"] > 0.995 and 0.0 < met["mean_iou"] <= 1.0 and np.isfinite(met["loss"])

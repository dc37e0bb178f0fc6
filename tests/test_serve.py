"""Batch-1 serving (deepsir_amd/serve.py): K single-pair registrations in flight behind submit()/result() and behind
``Network.forward`` - results bit-identical to the batched path, whatever shares a request's batch."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(n_pairs, n_points, seed0=500):
    from deepsir_amd.arch import NetConfig
    from deepsir_amd.synth import make_batch
    from deepsir_amd.weights import generate_state_dict
    cfg = NetConfig(feat_len=3)
    sd = generate_state_dict(cfg, 0)
    b = make_batch(n_points, [seed0 + i for i in range(n_pairs)], 3)
    return cfg, sd, torch.from_numpy(b["points_src"]).cuda(), torch.from_numpy(b["points_ref"]).cuda(), b


@pytest.mark.parametrize("in_flight,engines", [(8, 2), (3, 1), (1, 1)])
def test_served_single_pairs_equal_the_batched_path(in_flight, engines):
    """19 single-pair requests through a PairServer (ragged last batch, two engines in turn, hipGraph replay per batch size)
    against ONE batched Engine.register of the same pairs: transforms, correspondences, logits - equal bits."""
    from deepsir_amd.engine import Engine
    from deepsir_amd.serve import PairServer
    n, N, n_iter = 19, 2048, 3
    cfg, sd, src, ref, _ = _setup(n, N)
    eng = Engine(cfg, 0, max_points=N, max_pairs=n)
    eng.load_state_dict(sd)
    want = eng.register(src, ref, n_iter)
    eng.close()
    srv = PairServer(cfg, sd, 0, max_points=N, max_in_flight=in_flight, engines=engines, n_iter=n_iter)
    res = srv.run_closed_loop(((src[i], ref[i]) for i in range(n)), in_flight)
    assert srv.pairs_dispatched == n and srv.batches_dispatched >= (n + srv.max_batch - 1) // srv.max_batch
    for i, r in enumerate(res):
        assert torch.equal(r["transforms"], want["transforms"][i]), i
        assert torch.equal(r["idx"], want["idx"][:, i]) and torch.equal(r["logits"], want["logits"][:, i]), i
        assert torch.equal(r["pt_ref_new"], want["pt_ref_new"][i]) and int(r["invalid"]) == int(want["invalid"][i])
    # a second round re-uses the captured graphs and the static buffers
    res2 = srv.run_closed_loop(((src[i], ref[i]) for i in reversed(range(n))), in_flight)
    for i, r in zip(reversed(range(n)), res2):
        assert torch.equal(r["transforms"], want["transforms"][i]), i
    srv.close()


def test_server_groups_requests_by_shape_and_takes_host_tensors():
    from deepsir_amd.engine import Engine
    from deepsir_amd.serve import PairServer
    cfg, sd, src, ref, _ = _setup(6, 2048, 700)
    srv = PairServer(cfg, sd, 0, max_points=2048, max_in_flight=4, engines=2, n_iter=2, want_aux=False)
    eng = Engine(cfg, 0, max_points=2048, max_pairs=1)
    eng.load_state_dict(sd)
    reqs = [(src[0], ref[0]), (src[1, :1500], ref[1]), (src[2].cpu(), ref[2].cpu()), (src[3, :1500], ref[3]), (src[4], ref[4, :1800])]
    futs = [srv.submit(s, r) for s, r in reqs]
    for (s, r), f in zip(reqs, futs):
        want = eng.register(s.cuda()[None].contiguous(), r.cuda()[None].contiguous(), 2, want_aux=False)["transforms"][0]
        assert torch.equal(f.result()["transforms"], want)
    srv.close(); eng.close()


def test_network_forward_small_batches_are_asynchronous_and_exact():
    """``Network.forward`` with the reference's batch of ONE (test.py:56): served from a captured graph, no host synchronisation
    inside forward (the results are ordered on torch's stream), ``pred_pairs`` copied to the host only when read - and the same
    bits as the engine's batched registration; an optimiser constructs on ``parameters()`` and an in-place weight change is seen."""
    import argparse
    from deepsir_amd.engine import Engine
    from deepsir_amd.model import Network, _LazyPredPairs
    from deepsir_amd.weights import to_torch_state_dict
    cfg, sd, src, ref, _ = _setup(4, 2048, 900)
    args = argparse.Namespace(pipeline="align", num_sub=-1, num_knn=16, out_feat_dim=64, clip_weight_thresh=0.0, feat_len=3,
                              d_out=[16, 64, 128, 256], num_points=2048, sub_sampling_ratio=[4, 4, 4, 4], use_ppf=False, num_reg_iter=3)
    net = Network(args)
    net.load_state_dict(to_torch_state_dict(sd))
    net = net.cuda().eval()
    eng = Engine(cfg, 0, max_points=2048, max_pairs=4)
    eng.load_state_dict(sd)
    want = eng.register(src, ref, 3)
    outs = [net({"points_src": src[i:i + 1], "points_ref": ref[i:i + 1]}, (3, True)) for i in range(4)]      # four calls, nothing read yet
    for i, (T, ep) in enumerate(outs):
        assert isinstance(ep["pred_pairs"], _LazyPredPairs) and ep["pred_pairs"]._items is None
        assert torch.equal(torch.stack(T, 1), want["transforms"][i:i + 1])
        assert torch.equal(ep["pred_pairs"][2][0, :, 1], want["idx"][2, i].cpu()) and not ep["pred_pairs"][0].is_cuda
        assert torch.equal(ep["perm_matrices"][1], want["logits"][1, i:i + 1]) and not ep["invalid_gradient"]
    T4, _ = net({"points_src": src, "points_ref": ref}, (3, True))                                           # one call of four pairs
    assert torch.equal(torch.stack(T4, 1), want["transforms"])
    opt = torch.optim.SGD([p for p in net.parameters() if p.requires_grad], lr=0.1)
    with torch.no_grad():
        next(p for n_, p in net.named_parameters() if n_ == "inlier_model.fc_label.6.bias").add_(0.5)          # what optimizer.step() does
    T5, ep5 = net({"points_src": src[:1], "points_ref": ref[:1]}, (3, True))
    assert not torch.equal(ep5["perm_matrices"][0], want["logits"][0, :1]), "the engines did not reload the changed weights"
    assert torch.allclose(ep5["perm_matrices"][0], want["logits"][0, :1] + 0.5, atol=1e-5)
    del opt
    eng.close()

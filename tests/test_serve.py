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


def test_graph_replay_guard_sees_a_gpu_touched_before_the_import():
    """a HIP call before ``import deepsir_amd`` (``torch.cuda.is_available()``, or any library calling hipGetDeviceCount)
    initialises the HIP runtime - the ROCclr flag that makes hipGraph replay right can no longer reach it - while
    ``torch.cuda.is_initialized()`` stays False.  The guard reads
    /dev/kfd descriptors instead of torch's state: such a process is NOT replay-safe, ``Engine.enable_graph`` refuses, and
    ``Network.forward`` keeps the eager engine - same bits as the served path of a process that imported the package first.
    Also: a request whose tensors are dropped right after ``submit`` is served from intact inputs (record_stream)."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    code = r'''
import sys, argparse, hashlib
order = sys.argv[1]
import torch
if order == "late":
    torch.cuda.is_available()
    import ctypes                        # "any other library": the HIP runtime initialised behind torch's back
    n = ctypes.c_int(0)
    assert ctypes.CDLL("libamdhip64.so.7").hipGetDeviceCount(ctypes.byref(n)) == 0 and n.value >= 1
    assert not torch.cuda.is_initialized()          # torch's own state does not show it
import deepsir_amd
safe = deepsir_amd.graph_replay_safe()
from deepsir_amd.arch import NetConfig
from deepsir_amd.engine import Engine, EngineError
from deepsir_amd.model import Network
from deepsir_amd.synth import make_batch
from deepsir_amd.weights import generate_state_dict, to_torch_state_dict
cfg = NetConfig(feat_len=3)
sd = generate_state_dict(cfg, 0)
refused = False
eng = Engine(cfg, 0, max_points=2048, max_pairs=1)
try:
    eng.enable_graph(True)
except EngineError:
    refused = True
eng.close()
args = argparse.Namespace(pipeline="align", num_sub=-1, num_knn=16, out_feat_dim=64, clip_weight_thresh=0.0, feat_len=3,
                          d_out=[16, 64, 128, 256], num_points=2048, sub_sampling_ratio=[4, 4, 4, 4], use_ppf=False, num_reg_iter=3)
net = Network(args); net.load_state_dict(to_torch_state_dict(sd)); net = net.cuda().eval()
b = make_batch(2048, [901, 902], 3)
h = hashlib.sha256()
for i in range(2):
    for rep in range(3):                 # replays with the host waiting in between: what the broken path gets wrong from the third on
        T, ep = net({"points_src": torch.from_numpy(b["points_src"][i:i + 1]).cuda(), "points_ref": torch.from_numpy(b["points_ref"][i:i + 1]).cuda()}, (3, True))
        t = torch.stack(T, 1).cpu().numpy().tobytes()
        torch.cuda.synchronize()
    h.update(t)
print("RESULT", int(safe), int(refused), int(net._server is not None), h.hexdigest())
'''
    env = {k: v for k, v in os.environ.items() if k != "DEBUG_CLR_GRAPH_PACKET_CAPTURE"}
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    got = {}
    for order in ("early", "late"):
        r = subprocess.run([sys.executable, "-c", code, order], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        got[order] = [l for l in r.stdout.splitlines() if l.startswith("RESULT")][-1].split()[1:]
    assert got["early"][:3] == ["1", "0", "1"], got        # safe, graph accepted, forward used the server
    assert got["late"][:3] == ["0", "1", "0"], got         # not safe, graph refused, forward stayed eager
    assert got["early"][3] == got["late"][3], "eager and served registrations differ"


def test_request_tensors_may_be_dropped_right_after_submit():
    """serve.py marks request tensors as used by the slot's stream: memory of a request freed right after ``submit`` is not handed
    to the caller's next allocation before the queued copy has read it."""
    from deepsir_amd.engine import Engine
    from deepsir_amd.serve import PairServer
    cfg, sd, src, ref, _ = _setup(6, 2048, 1200)
    eng = Engine(cfg, 0, max_points=2048, max_pairs=1)
    eng.load_state_dict(sd)
    want = [eng.register(src[i:i + 1], ref[i:i + 1], 2, want_aux=False)["transforms"][0].clone() for i in range(6)]
    srv = PairServer(cfg, sd, 0, max_points=2048, max_in_flight=2, engines=1, n_iter=2, want_aux=False)
    futs = []
    for i in range(6):
        s, r = src[i].clone(), ref[i].clone()
        futs.append(srv.submit(s, r))
        del s, r
        junk = [torch.full((2048, 3), float("nan"), device="cuda") for _ in range(4)]      # would land in the freed blocks
        del junk
    for i, f in enumerate(futs):
        assert torch.equal(f.result()["transforms"], want[i]), i
    srv.close(); eng.close()

"""The deep-level walker (csrc/walk.hip; off by default, `Engine.enable_walk`): levels 2 / 3, mlp_mid and the first two decoder blocks
of a RandLA pass as ONE persistent launch whose phases are the former launches' tiles.  Same code on the same operands, statistics in
exact atomics: the results must equal the separate launches' bit for bit - alone, in ragged batches, replayed from a graph, and with
two engines walking at once."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(n_pairs, n_src, n_ref=None, seed0=300, feat_len=3):
    from deepsir_amd.arch import NetConfig
    from deepsir_amd.synth import make_batch
    from deepsir_amd.weights import generate_state_dict
    cfg = NetConfig(feat_len=feat_len)
    sd = generate_state_dict(cfg, 0)
    b = make_batch(max(n_src, n_ref or n_src), [seed0 + i for i in range(n_pairs)], feat_len)
    src = torch.from_numpy(b["points_src"][:, :n_src]).cuda().contiguous()
    ref = torch.from_numpy(b["points_ref"][:, :(n_ref or n_src)]).cuda().contiguous()
    return cfg, sd, src, ref


def _same(a, b, keys=("transforms", "idx", "logits", "pt_ref_new", "invalid")):
    for k in keys:
        assert torch.equal(a[k], b[k]), f"{k} differs"


@pytest.mark.parametrize("pairs,n_src,n_ref", [(1, 5000, 5000), (3, 2048, 1800), (8, 5000, 5000), (5, 1357, 1357)])
def test_walker_equals_separate_launches(pairs, n_src, n_ref):
    """Eager calls, walker on against off: joint src / ref batches (2 P clouds in the extractor pass), unequal clouds (two extractor
    passes), ragged level sizes (1357: 339 / 84 / 21 / 5)."""
    from deepsir_amd.engine import Engine
    cfg, sd, src, ref = _setup(pairs, n_src, n_ref)
    eng = Engine(cfg, 0, max_points=max(n_src, n_ref), max_pairs=pairs)
    eng.load_state_dict(sd)
    eng.enable_walk(False)
    want = {k: v.clone() for k, v in eng.register(src, ref, 5).items() if k != "_keep"}
    eng.enable_walk(True)
    for rep in range(3):
        got = eng.register(src, ref, 5)
        _same(got, want)
    eng.close()


def test_walker_in_a_replayed_graph_and_the_launch_census():
    """One pair replayed from its hipGraph (the batch-1 path of bench.py and of Network.forward): same bits as the eager separate
    launches on every replay, host waits in between; the graph's launch census (dsir_graph_stats) meets the round-5 targets."""
    from deepsir_amd.engine import Engine
    cfg, sd, src, ref = _setup(1, 5000)
    eng = Engine(cfg, 0, max_points=5000, max_pairs=1)
    eng.load_state_dict(sd)
    eng.enable_walk(False)
    want = {k: v.clone() for k, v in eng.register(src, ref, 5).items() if k != "_keep"}
    out = {k: torch.empty_like(v) for k, v in want.items()}
    eng.enable_graph(True)
    eng.register(src, ref, 5, out=out)
    off = eng.graph_stats()
    _same(out, want)
    eng.enable_walk(True)                      # drops the graph: the next call captures the walker's launches
    for rep in range(6):
        for k, v in out.items():
            if k != "_keep":
                v.fill_(-1) if v.dtype != torch.float32 else v.fill_(float("nan"))
        eng.register(src, ref, 5, out=out)
        torch.cuda.synchronize()
        _same(out, want)
    on = eng.graph_stats()
    print(f"[launch census] separate launches {off}  walker {on}")
    assert on["memsets"] + on["memcpys"] <= 8, on
    assert on["kernels"] <= 230, on
    assert on["kernels"] <= off["kernels"] - 90, (on, off)
    eng.close()


def test_randla_forward_operator_with_the_walker():
    """The operator entry point (dsir_randla_forward, eager, statistics zeroed by its own memset): feature extractor on 4 clouds."""
    from deepsir_amd.engine import Engine
    cfg, sd, src, ref = _setup(2, 4096)
    eng = Engine(cfg, 0, max_points=4096, max_pairs=2)
    eng.load_state_dict(sd)
    pts = torch.cat([src, ref], 0)
    xyz, neigh, sub, interp = eng.knn_pyramid(pts)
    eng.enable_walk(False)
    f0, l0 = eng.randla_forward("feat_extractor", pts, xyz, neigh, sub, interp)
    eng.enable_walk(True)
    for rep in range(3):
        f1, l1 = eng.randla_forward("feat_extractor", pts, xyz, neigh, sub, interp)
        assert torch.equal(f0, f1) and torch.equal(l0, l1)
    eng.close()


def test_two_engines_walking_at_once():
    """Uneven load: two engines on their own streams register different batches at the same time, again and again - every result
    equals the quiet separate-launch run (the hand-off between workgroups is agent-scope release / acquire; a stale read would show
    here, with consumers whose caches hold the previous repetition's lines)."""
    from deepsir_amd.engine import Engine
    cfg, sd, src, ref = _setup(6, 3000, 3000, seed0=700)
    engs = [Engine(cfg, 0, max_points=3000, max_pairs=4) for _ in range(2)]
    for e in engs:
        e.load_state_dict(sd)
    parts = [(src[:4], ref[:4]), (src[4:], ref[4:])]
    want = []
    for e, (s, r) in zip(engs, parts):
        e.enable_walk(False)
        want.append({k: v.clone() for k, v in e.register(s, r, 5).items() if k != "_keep"})
        e.enable_walk(True)
    for rep in range(25):
        outs = [e.register(s, r, 5, sync=False) for e, (s, r) in zip(engs, parts)]
        for e in engs:
            e.sync()
        for o, w in zip(outs, want):
            _same(o, w)
    for e in engs:
        e.close()


# ------------------------------------------------------------------------------------------- independent branches on auxiliary streams
@pytest.mark.parametrize("pairs,n_src,n_ref", [(1, 5000, 5000), (3, 2048, 1800), (8, 5000, 5000), (5, 1357, 1357)])
def test_forked_schedule_equals_the_single_stream(pairs, n_src, n_ref):
    """`Engine.enable_fork`: launches of up to 16 clouds run the KNN searches of the four levels, every block's position-encoding branch and
    mlp_skip, and the loop-invariant halves of the aggregation on auxiliary streams beside the main chain (csrc/engine.hip, fork_on).
    Same kernels, same operands: the results equal the single-stream schedule's bit for bit, eager and replayed from a graph."""
    from deepsir_amd.engine import Engine
    cfg, sd, src, ref = _setup(pairs, n_src, n_ref, seed0=1300)
    eng = Engine(cfg, 0, max_points=max(n_src, n_ref), max_pairs=pairs)
    eng.load_state_dict(sd)
    eng.enable_fork(False)
    want = {k: v.clone() for k, v in eng.register(src, ref, 5).items() if k != "_keep"}
    pyr0 = [t.clone() for t in eng.knn_pyramid(torch.cat([src[:, :min(n_src, n_ref)], ref[:, :min(n_src, n_ref)]], 0))]
    eng.enable_fork(True)
    for rep in range(3):
        _same(eng.register(src, ref, 5), want)
    pyr1 = eng.knn_pyramid(torch.cat([src[:, :min(n_src, n_ref)], ref[:, :min(n_src, n_ref)]], 0))
    assert all(torch.equal(a, b) for a, b in zip(pyr0, pyr1))
    out = {k: torch.empty_like(v) for k, v in want.items()}
    eng.enable_graph(True)
    for rep in range(5):
        for k, v in out.items():
            if k != "_keep":
                v.fill_(-1) if v.dtype != torch.float32 else v.fill_(float("nan"))
        eng.register(src, ref, 5, out=out)
        torch.cuda.synchronize()
        _same(out, want)
    eng.close()


def test_forked_schedules_of_two_engines_at_once():
    """Two engines with forked schedules (six streams in all) registering at the same time, again and again: every result equals the
    quiet single-stream run."""
    from deepsir_amd.engine import Engine
    cfg, sd, src, ref = _setup(6, 3000, 3000, seed0=1700)
    engs = [Engine(cfg, 0, max_points=3000, max_pairs=4) for _ in range(2)]
    parts = [(src[:4], ref[:4]), (src[4:], ref[4:])]
    want = []
    for e, (s, r) in zip(engs, parts):
        e.load_state_dict(sd)
        e.enable_fork(False)
        want.append({k: v.clone() for k, v in e.register(s, r, 5).items() if k != "_keep"})
        e.enable_fork(True)
    for rep in range(25):
        outs = [e.register(s, r, 5, sync=False) for e, (s, r) in zip(engs, parts)]
        for e in engs:
            e.sync()
        for o, w in zip(outs, want):
            _same(o, w)
    for e in engs:
        e.close()

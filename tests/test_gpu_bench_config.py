"""GPU parity on the configuration bench.py measures: many 5000-point pairs per call through
``EnginePool`` on 4 HIP streams, large enough per engine (P*J*K >= 2e8, csrc/engine.hip) for the fp16-screened
arg-min (csrc/nn_screen.hip) to run - the path every golden / oracle test of tests/test_gpu_parity.py misses
because one or two pairs take the exhaustive kernel.  Checked directly against the CPU oracle, not against
another HIP kernel.  Robustness of the same path to non-finite input, bad caller indices and weight reloads
under hipGraph replay lives here too.
"""
import numpy as np
import pytest
import torch

from test_gpu_parity import assert_pose_close, cu, _dev

pytestmark = pytest.mark.gpu


def _oracle_check(cfg, sd, raw_pair, idx_engine, T_engine, logits_engine, n_iter, tag):
    """Teacher-force the oracle with the engine's correspondences of ONE pair: poses within 1e-4 rad / 1e-4 m at every
    iteration; and wherever the oracle's own descriptors show a clear fp64 top-2 gap the engine's arg-min must be the
    oracle's (the rows without a clear gap are the fp32 near-ties no implementation reproduces)."""
    from oracle.knn import add_pyramids
    from oracle.network import OracleNet, to_torch
    data = to_torch(add_pyramids(raw_pair, cfg.num_knn, cfg.sub_sampling_ratio))
    taps = {}
    T_forced, ep = OracleNet(cfg, sd).register(data, n_iter, forced_idx=[idx_engine[i][None].long() for i in range(n_iter)], taps=taps)
    assert_pose_close(T_engine, np.stack([t.numpy()[0] for t in T_forced]), 1e-4, 1e-4, tag)
    np.testing.assert_allclose(logits_engine, np.stack([l.numpy()[0] for l in ep["perm_matrices"]]), rtol=2e-3, atol=2e-3)
    clear_frac = []
    for i in range(n_iter):
        best, second, arg = OracleNet.nn_gap(taps["desc_src"][i], taps["desc_ref"][i])
        clear = ((second - best) > 1e-4 * (1.0 + best.abs()))[0].numpy()
        clear_frac.append(float(clear.mean()))
        assert np.array_equal(idx_engine[i].numpy()[clear], arg[0].numpy()[clear]), f"{tag}: iteration {i}"
    print(f"[bench-config] {tag}: rows with a clear fp64 gap per iteration {['%.3f' % c for c in clear_frac]} - all equal to the oracle's arg-min")
    assert clear_frac[0] > 0.3


def test_pool_screened_path_vs_oracle():
    """32 pairs x 5000 points on EnginePool(streams=4): 8 pairs per engine = 2e8 (row, column) pairs per search, the
    threshold from which dsir_register screens (bench.py default: 64 per engine).  The engine's correspondences of two
    pairs (first engine, last engine) are forced into the oracle."""
    from deepsir_amd.arch import NetConfig
    from deepsir_amd.engine import EnginePool
    from deepsir_amd.synth import make_batch, make_pair
    from deepsir_amd.weights import generate_state_dict
    torch.set_num_threads(16)
    cfg = NetConfig(feat_len=3)
    sd = generate_state_dict(cfg, 0)
    P, N, n_iter = 32, 5000, 5
    seeds = [10_000 + i for i in range(P)]                       # bench.py's seeds of rank 0
    b = make_batch(N, seeds, 3)
    pool = EnginePool(cfg, 0, max_points=N, max_pairs=P, streams=4)
    pool.load_state_dict(sd)
    pool.screen_stats(reset=True)
    out = pool.register(cu(b["points_src"]), cu(b["points_ref"]), n_iter)
    st = pool.screen_stats()
    print(f"[bench-config] screening: {st}")
    assert st["screened_searches"] == 4 * n_iter and st["exhaustive_searches"] == 0, "the screened path did not run"
    assert st["rows_searched"] == P * N * n_iter
    assert not bool(out["invalid"].any())
    idx, T, lg = out["idx"].cpu(), out["transforms"].cpu().numpy(), out["logits"].cpu().numpy()
    assert int(idx.min()) >= 0 and int(idx.max()) < N
    for p in (0, P - 1):
        raw = make_pair(N, seeds[p], 3)
        assert np.array_equal(raw["points_src"][0], b["points_src"][p])
        _oracle_check(cfg, sd, raw, idx[:, p], T[p], lg[:, p], n_iter, f"pair {p} of {P} (4 streams, screened)")
    # same call without the aux outputs (what bench.py times) returns the same transforms
    q = pool.register(cu(b["points_src"]), cu(b["points_ref"]), n_iter, want_aux=False)
    assert torch.equal(q["transforms"], out["transforms"])
    pool.close()


@pytest.mark.parametrize("where,value", [("src", float("nan")), ("ref", float("nan")), ("src", float("inf")), ("src", -float("inf"))])
def test_non_finite_point_poisons_only_its_pair(where, value):
    """One non-finite coordinate in one cloud (reference: the whole forward ends in the SVD's except branch, identity +
    invalid_gradient, model.py:45-64).  Here: that pair alone returns the identity with invalid bit 0, the other pairs of
    the batch are bitwise what they are without it, every index stays in range, nothing faults."""
    from deepsir_amd.arch import NetConfig
    from deepsir_amd.engine import Engine
    from deepsir_amd.synth import make_batch
    from deepsir_amd.weights import generate_state_dict
    cfg = NetConfig(feat_len=3)
    P, N = 3, 2048
    eng = Engine(cfg, 0, max_points=N, max_pairs=P)
    eng.load_state_dict(generate_state_dict(cfg, 0))
    b = make_batch(N, [61, 62, 63], 3)
    src, ref = cu(b["points_src"]), cu(b["points_ref"])
    clean = {k: v.clone() for k, v in eng.register(src, ref, 3).items() if isinstance(v, torch.Tensor)}
    assert not bool(clean["invalid"].any())
    (src if where == "src" else ref)[1, 100, 1] = value
    out = eng.register(src, ref, 3)
    if where == "src":
        assert [int(v) for v in out["invalid"]] == [0, 1, 0]
        eye = torch.eye(3, 4, device=src.device)[None].expand(3, 3, 4)
        assert torch.equal(out["transforms"][1], eye)
    else:
        # a poisoned REF cloud: every ref descriptor is NaN, every distance of the pair is NaN, and the arg-min of an
        # all-NaN row is column 0 here as in torch.min (the first NaN).  xyz_ref[0] is finite, so the Kabsch step sees
        # finite input and - in the reference too - nothing is flagged: a finite, meaningless pose for that pair.
        assert [int(v) for v in out["invalid"]] == [0, 0, 0]
        assert int(out["idx"][:, 1].abs().max()) == 0
        assert torch.isfinite(out["transforms"]).all()
    assert int(out["idx"].min()) >= 0 and int(out["idx"].max()) < N
    for p in (0, 2):
        for k in ("transforms", "logits", "pt_ref_new"):
            assert torch.equal(out[k][p] if k != "logits" else out[k][:, p], clean[k][p] if k != "logits" else clean[k][:, p]), (k, p)
        assert torch.equal(out["idx"][:, p], clean["idx"][:, p])
    # the stage entry point stays in range as well
    xyz, neigh, sub, interp = eng.knn_pyramid(src)
    for t, hi in ((neigh, N), (sub, N), (interp, N)):
        assert int(t.min()) >= 0 and int(t.max()) < hi
    eng.close()


def test_non_finite_point_in_a_screened_batch():
    """The same with the fp16-screened arg-min engaged (8 pairs x 5000 on one engine): the rows of the poisoned pair go
    through the exhaustive fallback, come back as index 0 instead of -1, and the other pairs keep their bits."""
    from deepsir_amd.arch import NetConfig
    from deepsir_amd.engine import Engine
    from deepsir_amd.synth import make_batch
    from deepsir_amd.weights import generate_state_dict
    cfg = NetConfig(feat_len=3)
    P, N = 8, 5000
    eng = Engine(cfg, 0, max_points=N, max_pairs=P)
    eng.load_state_dict(generate_state_dict(cfg, 0))
    b = make_batch(N, list(range(71, 71 + P)), 3)
    src, ref = cu(b["points_src"]), cu(b["points_ref"])
    clean = {k: v.clone() for k, v in eng.register(src, ref, 2).items() if isinstance(v, torch.Tensor)}
    eng.screen_stats(reset=True)
    src[5, 4321, 0] = float("nan")
    out = eng.register(src, ref, 2)
    assert eng.screen_stats()["screened_searches"] == 2
    assert [int(v) for v in out["invalid"]] == [0, 0, 0, 0, 0, 1, 0, 0]
    assert int(out["idx"].min()) >= 0 and int(out["idx"].max()) < N
    keep = [p for p in range(P) if p != 5]
    assert torch.equal(out["transforms"][keep], clean["transforms"][keep])
    assert torch.equal(out["idx"][:, keep], clean["idx"][:, keep])
    assert torch.equal(out["transforms"][5], torch.eye(3, 4, device=src.device)[None].expand(2, 3, 4))
    eng.close()


def test_bad_caller_indices_are_clamped_and_flagged():
    """forced_idx / caller pyramids with out-of-range entries: clamped on device (no fault), reported as invalid bit 1."""
    from deepsir_amd.arch import NetConfig
    from deepsir_amd.engine import Engine
    from deepsir_amd.synth import make_batch
    from deepsir_amd.weights import generate_state_dict
    cfg = NetConfig(feat_len=3)
    P, N = 2, 2048
    eng = Engine(cfg, 0, max_points=N, max_pairs=P)
    eng.load_state_dict(generate_state_dict(cfg, 0))
    b = make_batch(N, [81, 82], 3)
    src, ref = cu(b["points_src"]), cu(b["points_ref"])
    good = eng.register(src, ref, 2)
    forced = good["idx"].clone()
    ok = eng.register(src, ref, 2, forced_idx=forced)
    assert torch.equal(ok["transforms"], good["transforms"]) and not bool(ok["invalid"].any())
    forced[1, 1, 7] = -5
    forced[0, 1, 9] = 1 << 30
    bad = eng.register(src, ref, 2, forced_idx=forced)
    assert [int(v) for v in bad["invalid"]] == [0, 2]
    assert torch.equal(bad["transforms"][0], good["transforms"][0])
    assert torch.isfinite(bad["transforms"]).all()
    xyz, neigh, sub, interp = eng.knn_pyramid(torch.cat([src, ref], 0))
    pyr = {"points_src_xyz": xyz[:P], "points_ref_xyz": xyz[P:], "points_src_neigh_idx": neigh[:P].clone(), "points_ref_neigh_idx": neigh[P:],
           "points_src_sub_idx": sub[:P], "points_ref_sub_idx": sub[P:].clone(), "points_src_interp_idx": interp[:P].clone(),
           "points_ref_interp_idx": interp[P:]}
    sup = eng.register(src, ref, 2, pyramids=pyr)
    assert torch.equal(sup["transforms"], good["transforms"]) and not bool(sup["invalid"].any())
    pyr["points_src_neigh_idx"][0, N + 3, 2] = N            # level 1 has N/4 points: far out of range
    pyr["points_ref_sub_idx"][1, 0, 0] = -1
    pyr["points_src_interp_idx"][0, 5, 0] = N // 4          # level-0 interp indexes the N/4 points of level 1
    sup = eng.register(src, ref, 2, pyramids=pyr)
    assert [int(v) & 2 for v in sup["invalid"]] == [2, 2]
    assert torch.isfinite(sup["transforms"]).all()
    eng.close()


def test_graph_replay_after_weight_reload():
    """dsir_finalize_weights drops a captured hipGraph (it holds the old weight blob's addresses): graph on, register,
    load other weights, register again into the same buffers == a fresh engine with those weights."""
    from deepsir_amd.arch import NetConfig
    from deepsir_amd.engine import Engine
    from deepsir_amd.synth import make_batch
    from deepsir_amd.weights import generate_state_dict
    cfg = NetConfig(feat_len=3)
    sd0, sd1 = generate_state_dict(cfg, 0), generate_state_dict(cfg, 5)
    b = make_batch(2048, [91], 3)
    src, ref = cu(b["points_src"]), cu(b["points_ref"])
    eng = Engine(cfg, 0, max_points=2048, max_pairs=1)
    eng.load_state_dict(sd0)
    eng.enable_graph(True)
    first = eng.register(src, ref, 3)
    keep = {k: v for k, v in first.items() if isinstance(v, torch.Tensor)}
    T0 = keep["transforms"].clone()
    eng.load_state_dict(sd1)
    second = eng.register(src, ref, 3, out=keep)
    fresh = Engine(cfg, 0, max_points=2048, max_pairs=1)
    fresh.load_state_dict(sd1)
    want = fresh.register(src, ref, 3)
    for k in ("transforms", "idx", "logits"):
        assert torch.equal(second[k], want[k]), k
    assert not torch.equal(T0, want["transforms"])
    eng.close(); fresh.close()


def test_hoisted_loop_invariants_are_bit_identical(tmp_path):
    """The loop invariants hoisted out of the registration iterations - the inlier model's position-encoding layers and the
    enc half of its attention scores (EncCache, csrc/engine.hip) - change nothing: recomputing them every iteration
    (DSIR_NO_HOIST), or only the score halves (DSIR_NO_S2), gives the same bits.  Switches are read once per process."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for name, extra in (("hoisted", {}), ("no_s2", {"DSIR_TUNING": "1", "DSIR_NO_S2": "1"}), ("no_hoist", {"DSIR_TUNING": "1", "DSIR_NO_HOIST": "1"})):
        out = str(tmp_path / f"{name}.npz")
        env = {k: v for k, v in os.environ.items() if k not in ("DSIR_NO_S2", "DSIR_NO_HOIST")}
        env.update(extra)
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "register_dump.py"), out, "3", "5000", "4"], env=env,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[name] = np.load(out)
    for other in ("no_s2", "no_hoist"):
        for k in ("idx", "logits", "transforms"):
            assert np.array_equal(outs["hoisted"][k], outs[other][k]), f"{k} differs between hoisted and {other}"


def test_round4_launch_restructurings_are_bit_identical(tmp_path):
    """Round 4 changed HOW several layers are launched without touching what they compute: mlp1 + mlp_skip of the deep levels share a
    launch (GemmArgs::c_split; DSIR_NO_PAIR: two launches), GroupNorm layers of pw_stream.hip play their virtual workgroups on fewer
    physical ones in chip-filling launches (DSIR_STREAM_PHYS_BLOCKS=0: one per virtual workgroup), the KNN grid of a level is built by
    one launch (DSIR_GRID_NO_BUILD: five).  64-pair registrations (every large-launch choice active) under each switch: same bits."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    switches = {"default": {}, "no_pair": {"DSIR_NO_PAIR": "1"}, "no_fold": {"DSIR_STREAM_PHYS_BLOCKS": "0"}, "no_build": {"DSIR_GRID_NO_BUILD": "1"}}
    outs = {}
    for name, extra in switches.items():
        out = str(tmp_path / f"{name}.npz")
        env = {k: v for k, v in os.environ.items() if not k.startswith("DSIR_")}
        if extra:
            env.update(extra)
            env["DSIR_TUNING"] = "1"
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "ab_outputs.py"), out, "64"], env=env, capture_output=True, text=True,
                           timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[name] = np.load(out)
    for other in ("no_pair", "no_fold", "no_build"):
        for k in outs["default"].files:
            assert np.array_equal(outs["default"][k], outs[other][k]), f"{k} differs between default and {other}"


def test_engine_on_the_callers_stream_gives_the_same_bits():
    """dsir_set_stream (include/dsir.h): the engine's launches ordered on torch's current stream instead of the context's own
    - no host synchronisation around a call - must not change a bit, on the default stream and on a side stream, and the
    context goes back to its own stream afterwards."""
    from deepsir_amd.arch import NetConfig
    from deepsir_amd.engine import Engine
    from deepsir_amd.synth import make_batch
    from deepsir_amd.weights import generate_state_dict
    cfg = NetConfig(feat_len=3)
    eng = Engine(cfg, 0, max_points=2048, max_pairs=2)
    eng.load_state_dict(generate_state_dict(cfg, 4))
    b = make_batch(2048, [31, 32], 3)
    src, ref = torch.from_numpy(b["points_src"]).cuda(), torch.from_numpy(b["points_ref"]).cuda()
    want = eng.register(src, ref, 3)
    eng.use_torch_stream(True)
    got = eng.register(src, ref, 3)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        got2 = eng.register(src, ref, 3)
        desc = eng.aggregate(got2["pt_ref_new"], torch.zeros(2, 2048, 64, device="cuda"), torch.ones(2, 2048, device="cuda"))
    side.synchronize()
    torch.cuda.synchronize()
    eng.use_torch_stream(False)
    again = eng.register(src, ref, 3)
    for k in ("transforms", "idx", "logits"):
        assert torch.equal(want[k], got[k]) and torch.equal(want[k], got2[k]) and torch.equal(want[k], again[k]), k
    assert torch.isfinite(desc).all()
    eng.close()


def test_alternating_torch_streams_share_the_workspace_safely():
    """A context re-uses ONE workspace arena in every call.  With ``use_torch_stream`` the wrapper re-binds the context to
    torch's current stream whenever it changed: the new stream must then wait (on the device) for the launches still running
    on the old one (dsir_set_stream, include/dsir.h) - back-to-back calls on two alternating streams, no host synchronisation
    in between, return the bits of the serial run."""
    from deepsir_amd.arch import NetConfig
    from deepsir_amd.engine import Engine
    from deepsir_amd.synth import make_batch
    from deepsir_amd.weights import generate_state_dict
    cfg = NetConfig(feat_len=3)
    eng = Engine(cfg, 0, max_points=5000, max_pairs=4)
    eng.load_state_dict(generate_state_dict(cfg, 4))
    sets = []
    for k in range(2):
        b = make_batch(5000, [41 + 4 * k + i for i in range(4)], 3)
        sets.append((torch.from_numpy(b["points_src"]).cuda(), torch.from_numpy(b["points_ref"]).cuda()))
    want = [eng.register(s, r, 3) for s, r in sets]
    eng.use_torch_stream(True)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for st in streams:
        st.wait_stream(torch.cuda.current_stream())
    got = []
    for rep in range(6):                      # 12 calls, alternating streams and inputs, nothing synchronised in between
        for k in range(2):
            with torch.cuda.stream(streams[k]):
                got.append((k, eng.register(sets[k][0], sets[k][1], 3, sync=False)))
    torch.cuda.synchronize()
    eng.use_torch_stream(False)
    for k, o in got:
        for key in ("transforms", "idx", "logits"):
            assert torch.equal(want[k][key], o[key]), (k, key)
    eng.close()


def test_groupnorm_statistics_do_not_depend_on_arrival_order():
    """GroupNorm statistics of a layer meet across workgroups in EXACT atomics - integer-valued fp64 limbs added with the hardware fp64 atomic
    add, exact below 2^53 (csrc/device_utils.h, gn_block_commit): the totals
    - hence every bit downstream - cannot depend on the order in which workgroups arrive.  One 32-pair registration repeated 200
    times on two concurrently running engines (EnginePool, 2 HIP streams, kernels of both interleaving on the CUs): identical
    correspondences, logits and poses every time; the two halves of the batch also equal a single-engine run."""
    from deepsir_amd.arch import NetConfig
    from deepsir_amd.engine import Engine, EnginePool
    from deepsir_amd.synth import make_batch
    from deepsir_amd.weights import generate_state_dict
    cfg = NetConfig(feat_len=3)
    sd = generate_state_dict(cfg, 0)
    P, N, n_iter = 32, 5000, 3
    b = make_batch(N, [20_000 + i for i in range(P)], 3)
    src, ref = cu(b["points_src"]), cu(b["points_ref"])
    pool = EnginePool(cfg, 0, max_points=N, max_pairs=P, streams=2)
    pool.load_state_dict(sd)
    first = pool.register(src, ref, n_iter)
    keep = {k: first[k].clone() for k in ("transforms", "idx", "logits")}
    differing = 0
    for rep in range(200):
        out = pool.register(src, ref, n_iter)
        differing += int(any(not torch.equal(out[k], keep[k]) for k in keep))
    pool.close()
    assert differing == 0, f"{differing} of 200 repeats differ"
    eng = Engine(cfg, 0, max_points=N, max_pairs=P)
    eng.load_state_dict(sd)
    one = eng.register(src, ref, n_iter)
    eng.close()
    for k in keep:
        assert torch.equal(one[k], keep[k]), k


def test_a_pairs_bits_do_not_depend_on_the_launch_size():
    """Several kernels are chosen, and several grids sized, by how much work a launch holds (round 4: one 8-wave workgroup per row range
    of the d = 128 attentive pooling from 8192 units on, GroupNorm layers of pw_stream.hip folded onto the natural grid in chip-filling
    launches, four lanes per query in the grid 16-NN of small launches, one- or eight-tile waves in the epilogues without a reduction,
    the screened arg-min from four pairs on).  None of that may change a pair's result: 64 pairs registered in ONE call - every
    large-launch choice active - equal the same pairs registered two at a time, bit for bit."""
    from deepsir_amd.arch import NetConfig
    from deepsir_amd.engine import Engine
    from deepsir_amd.synth import make_batch
    from deepsir_amd.weights import generate_state_dict
    cfg = NetConfig(feat_len=3)
    sd = generate_state_dict(cfg, 3)
    P, N, n_iter = 64, 5000, 3
    b = make_batch(N, [30_000 + i for i in range(P)], 3)
    src, ref = cu(b["points_src"]), cu(b["points_ref"])
    eng = Engine(cfg, 0, max_points=N, max_pairs=P)
    eng.load_state_dict(sd)
    big = eng.register(src, ref, n_iter)
    big = {k: big[k].clone() for k in ("transforms", "idx", "logits")}
    for a in range(0, P, 2):
        small = eng.register(src[a:a + 2], ref[a:a + 2], n_iter)
        assert torch.equal(small["transforms"], big["transforms"][a:a + 2]), a
        assert torch.equal(small["idx"], big["idx"][:, a:a + 2]), a
        assert torch.equal(small["logits"], big["logits"][:, a:a + 2]), a
    eng.close()


def test_config4_shard_walked_in_ragged_calls():
    """BASELINE configs[3] (the 1623-pair 3DMatch test set, pair-sharded): one rank's shard of a 1623-pair set at world size 8 is
    203 pairs; here 300 pairs of 5000 points are walked the way bench.py --total-pairs walks a shard - engine calls of at most
    128 pairs on the 2-stream pool, the last one ragged (128 + 128 + 44), results written into one [pairs, n_iter, 3, 4] buffer
    and gathered with the padded all_gather path - and once more in calls of 100 on a single engine: a pair's pose does not
    depend on how the shard is cut (bitwise), every pose is a rigid transform."""
    from deepsir_amd.arch import NetConfig
    from deepsir_amd.dist import gather_results, shard_range, shard_sizes
    from deepsir_amd.engine import Engine, EnginePool
    from deepsir_amd.synth import make_batch
    from deepsir_amd.weights import generate_state_dict
    cfg = NetConfig(feat_len=3)
    sd = generate_state_dict(cfg, 0)
    T, N, n_iter = 300, 5000, 5
    assert shard_sizes(1623, 8) == [203] * 7 + [202] and list(shard_range(T, 0, 1)) == list(range(T))
    b = make_batch(N, [50_000 + i for i in range(T)], 3)
    src, ref = cu(b["points_src"]), cu(b["points_ref"])
    out = torch.empty((T, n_iter, 3, 4), dtype=torch.float32, device=src.device)
    pool = EnginePool(cfg, 0, max_points=N, max_pairs=128, streams=2)
    pool.load_state_dict(sd)
    for c0 in range(0, T, 128):
        c1 = min(c0 + 128, T)
        pool.register(src[c0:c1], ref[c0:c1], n_iter, want_aux=False, sync=False, out={"transforms": out[c0:c1]})
    pool.sync()
    got = gather_results(out, None, sizes=[T])
    pool.close()
    assert got.shape == (T, n_iter, 3, 4) and bool(torch.isfinite(got).all())
    R = got[:, -1, :, :3].double()
    assert float((R @ R.transpose(1, 2) - torch.eye(3, dtype=torch.float64, device=R.device)).abs().max()) < 1e-5
    assert float((torch.linalg.det(R) - 1).abs().max()) < 1e-5
    eng = Engine(cfg, 0, max_points=N, max_pairs=100)
    eng.load_state_dict(sd)
    again = torch.empty_like(out)
    for c0 in range(0, T, 100):
        eng.register(src[c0:c0 + 100], ref[c0:c0 + 100], n_iter, want_aux=False, sync=False, out={"transforms": again[c0:c0 + 100]})
    eng.sync()
    eng.close()
    assert torch.equal(got, again), "a pair's pose depends on how the shard is cut into engine calls"

"""The training slice (backward half): the inlier model's forward-in-training-mode, backward and
Adam step.  CPU: the oracle restatement (oracle/train.py) against the vectors the imported reference's autograd produced
(tests/golden/train_cases.npz, oracle/gen_golden_train.py).  GPU: the HIP operators one by one against torch autograd, then
the whole pass and the optimiser against the same reference vectors, through the C ABI of include/dsir_train.h."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import ROOT
from deepsir_amd.arch import NetConfig
from deepsir_amd.weights import generate_state_dict
from oracle.gen_golden_train import FULL_MAX, case_inputs, sample_index
from oracle.network import OracleNet, to_torch
from oracle import train as otrain

GOLD = np.load(os.path.join(ROOT, "tests", "golden", "train_cases.npz"))
CFG = NetConfig(feat_len=3)


def _case(c):
    meta = json.loads(str(GOLD[f"c{c}_meta"]))
    d = case_inputs(CFG, meta["n"], meta["seed"])
    sd = generate_state_dict(CFG, meta["wseed"], meta["variant"])
    keep = np.unpackbits(GOLD[f"c{c}_keep"])[: 2 * 64 * meta["n"]].reshape(2, 64, meta["n"]).astype(bool)
    return meta, d, sd, keep


ZERO_BY_CONSTRUCTION = ("fc_label.0.bias", "fc_label.3.bias")   # a bias in front of BatchNorm: the batch mean removes it


def _check_grads(c, grads, seed, rtol, atol=1e-7):
    """grads: name -> flat numpy; against the reference's (whole tensors, or samples + sum + norm).  Gradients that are
    zero by construction are rounding noise on both sides: required to be noise (1e-4 of the layer's weight gradient)."""
    checked = 0
    for name, g in grads.items():
        g = np.asarray(g, np.float64).reshape(-1)
        key = f"c{c}_g_{name}"
        if name.endswith(ZERO_BY_CONSTRUCTION):
            wscale = np.abs(GOLD[f"c{c}_g_{name[:-4]}weight"]).max()
            assert np.abs(g).max() <= 1e-4 * wscale and np.abs(GOLD[key]).max() <= 1e-4 * wscale, name
            continue
        if key in GOLD:
            ref = GOLD[key].astype(np.float64)
        else:
            ref = GOLD[key + "_samples"].astype(np.float64)
            s, nrm = GOLD[key + "_sum_norm"]
            assert abs(np.sqrt((g ** 2).sum()) - nrm) <= rtol * nrm + atol * np.sqrt(g.size), name
            g = g[sample_index(name[len("inlier_model."):], g.size, seed)]
        scale = np.abs(ref).max()
        assert np.abs(g - ref).max() <= rtol * scale + atol, f"{name}: {np.abs(g - ref).max():.3e} vs scale {scale:.3e}"
        checked += 1
    assert checked > 100


@pytest.mark.parametrize("c", range(int(GOLD["n_cases"])))
def test_oracle_training_pass_matches_reference_autograd(c):
    meta, d, sd, keep = _case(c)
    net = OracleNet(CFG, sd)
    params = otrain.trainable(net)
    t = to_torch(d)
    logits = otrain.randla_train(net, "inlier_model", t["cat"], t["points_src_xyz"], t["points_src_neigh_idx"], t["points_src_sub_idx"],
                                 t["points_src_interp_idx"], torch.from_numpy(keep))
    assert np.abs(logits.detach().numpy() - GOLD[f"c{c}_logits"]).max() < 2e-5
    (logits * t["G"]).sum().backward()
    _check_grads(c, {k: v.grad.numpy() for k, v in params.items()}, meta["seed"], 2e-4)
    for k in GOLD.files:
        if k.startswith(f"c{c}_buf_"):
            assert np.allclose(net.p[k[len(f"c{c}_buf_"):]].numpy(), GOLD[k], rtol=1e-5, atol=1e-6), k


def _adam_targets():
    return {k[len("adam_"):]: GOLD[k] for k in GOLD.files if k.startswith("adam_inlier_model.")}


def test_oracle_adam_steps_match_reference():
    meta, d, sd, _ = _case(1)
    net = OracleNet(CFG, sd)
    params = otrain.trainable(net)
    opt = otrain.adam_reference(params, 1e-3)
    t = to_torch(d)
    n = meta["n"]
    for s in range(3):
        keep = np.unpackbits(GOLD["adam_keep"][s])[: 2 * 64 * n].reshape(2, 64, n).astype(bool)
        opt.zero_grad()
        lg = otrain.randla_train(net, "inlier_model", t["cat"], t["points_src_xyz"], t["points_src_neigh_idx"], t["points_src_sub_idx"],
                                 t["points_src_interp_idx"], torch.from_numpy(keep))
        (lg * t["G"]).sum().backward()
        opt.step()
    # the reference ran case 1's single pass first: its running statistics had one more update (checked on the device test)
    for name, ref in _adam_targets().items():
        if name.endswith("running_var"):
            continue
        v = net.p[name].detach().numpy().reshape(-1)
        v = v if v.size <= FULL_MAX else v[sample_index(name[len("inlier_model."):], v.size, meta["seed"])]
        assert np.abs(v - ref).max() <= 2e-4 * np.abs(ref).max() + 1e-6, name


def test_semantic_class_weights_are_the_references():
    from deepsir_amd.arch import semantic_class_weights
    assert np.allclose(semantic_class_weights(), GOLD["semantic_class_weights"], rtol=1e-12)
    lg = torch.randn(2, 19, 50, generator=torch.Generator().manual_seed(0))
    lb = torch.zeros(2, 50, dtype=torch.long)
    lb[0, :7] = torch.arange(1, 8)
    w = torch.tensor(semantic_class_weights())
    want = sum(-w[c] * torch.log_softmax(lg[0, :, c], 0)[c] for c in range(7)) / w[:7].sum()     # class = label - 1, label 0 ignored
    assert abs(float(otrain.semantic_loss(lg, lb, semantic_class_weights())) - float(want)) < 1e-5


def test_training_has_no_cpu_path():
    """The training operators are HIP kernels: building a trainer on the CPU fails loudly (no autograd fallback)."""
    from deepsir_amd.train import AggregationTrainer, RandlaTrainer
    sd = generate_state_dict(CFG, 1, "plain")
    with pytest.raises(RuntimeError, match="needs a GPU"):
        RandlaTrainer(CFG, sd, "inlier_model", 6, 1, "cpu")
    with pytest.raises(RuntimeError, match="needs a GPU"):
        AggregationTrainer(CFG, sd, "cpu")
    lib = __import__("deepsir_amd._lib", fromlist=["load"]).load()
    assert lib.dsir_t_gemm(None, None, 0, None, 0, 0, None, None, 0, 0, 0, 0, 0.0) != 0          # argument validation, no launch


# ------------------------------------------------------------------------------------------------- GPU
def _dev():
    return torch.device("cuda:0")


@pytest.mark.gpu
def test_ops_against_torch_autograd():
    """Every operator of include/dsir_train.h on small ragged shapes (tile edges, odd channel counts)."""
    import torch.nn.functional as F
    from deepsir_amd.train import _Ops
    o = _Ops(_dev())
    g = torch.Generator().manual_seed(5)
    rnd = lambda *s: torch.randn(*s, generator=g)
    # 1x1 convolution: forward, d input (with accumulation), d weight / d bias
    for rows, cin, cout in ((1000, 10, 8), (333, 96, 160), (77, 6, 1), (4100, 64, 64)):
        x, w, b, dy = rnd(rows, cin).requires_grad_(), rnd(cout, cin).requires_grad_(), rnd(cout).requires_grad_(), rnd(rows, cout)
        y = F.linear(x, w, b)
        y.backward(dy)
        xd, wd, bd, dyd = x.detach().to(_dev()), w.detach().to(_dev()), b.detach().to(_dev()), dy.to(_dev())
        assert torch.allclose(o.conv(xd, wd, bd).cpu(), y.detach(), rtol=1e-4, atol=1e-4)
        assert torch.allclose(o.conv_dx(dyd, wd).cpu(), x.grad, rtol=1e-4, atol=1e-4)
        base = torch.ones(rows, cin, device=_dev())
        assert torch.allclose(o.conv_dx(dyd, wd, into=base).cpu(), x.grad + 1.0, rtol=1e-4, atol=1e-4)
        dw, db = torch.zeros(cout, cin, device=_dev()), torch.zeros(cout, device=_dev())
        o.conv_dw(dyd, xd, dw, db)
        o.conv_dw(dyd, xd, dw, db)                                        # accumulates
        assert torch.allclose(dw.cpu(), 2 * w.grad, rtol=1e-4, atol=2e-3)
        assert torch.allclose(db.cpu(), 2 * b.grad, rtol=1e-4, atol=2e-3)
    # GroupNorm (+ LeakyReLU) per cloud; BatchNorm1d in training mode = one cloud, one group per channel
    for clouds, M, C_, groups, act in ((3, 500, 32, 4, True), (2, 77, 128, 8, False), (1, 900, 64, 64, True)):
        y = rnd(clouds, C_, M).requires_grad_()
        ga, be, dout = (torch.rand(C_, generator=g) + 0.5).requires_grad_(), rnd(C_).requires_grad_(), rnd(clouds, C_, M)
        ref = F.group_norm(y, groups, ga, be, 1e-5) if groups != C_ else F.batch_norm(y, None, None, ga, be, True, 0.1, 1e-5)
        ref = F.leaky_relu(ref, 0.2) if act else ref
        ref.backward(dout)
        pm = lambda t: t.detach().permute(0, 2, 1).reshape(clouds * M, C_).contiguous().to(_dev())
        out, stats = o.gn_fwd(pm(y), clouds, groups, ga.detach().to(_dev()), be.detach().to(_dev()), act)
        assert torch.allclose(out.cpu(), pm(ref).cpu(), rtol=1e-4, atol=1e-4)
        dga, dbe = torch.zeros(C_, device=_dev()), torch.zeros(C_, device=_dev())
        dy = o.gn_bwd(pm(dout), pm(y), stats, clouds, groups, ga.detach().to(_dev()), be.detach().to(_dev()), act, dga, dbe)
        assert torch.allclose(dy.cpu(), pm(y.grad).cpu(), rtol=1e-3, atol=1e-4)
        assert torch.allclose(dga.cpu(), ga.grad, rtol=1e-4, atol=1e-3) and torch.allclose(dbe.cpu(), be.grad, rtol=1e-4, atol=1e-3)
    # gather / scatter-add, max-pool, attentive pooling, add + LeakyReLU
    clouds, n, m, C_ = 2, 300, 75, 24
    x = rnd(clouds, n, C_).requires_grad_()
    idx = torch.randint(0, n, (clouds, n * 16), generator=g)
    cat = torch.zeros(clouds * n * 16, C_ + 5, device=_dev())
    o.gather(x.detach().to(_dev()), idx.int().to(_dev()), cat, 5)
    ref = torch.gather(x, 1, idx[:, :, None].expand(-1, -1, C_))
    assert torch.equal(cat[:, 5:].cpu(), ref.detach().reshape(-1, C_))
    dy = rnd(clouds * n * 16, C_ + 5)
    ref.backward(dy[:, 5:].reshape(clouds, n * 16, C_))
    idx_d = idx.int().to(_dev())
    got = o.scatter_add(dy.to(_dev()), 5, C_, idx_d, n)
    assert torch.allclose(got.cpu(), x.grad, rtol=1e-4, atol=1e-4)
    # the sum over a row's sources runs in ascending source order through the index's inverse (dsir_t_scatter_plan): no float
    # atomics, every run the same bits - and exactly the sequential fp32 sum in that order
    for _ in range(3):
        o.new_step()
        assert torch.equal(o.scatter_add(dy.to(_dev()), 5, C_, idx_d, n), got)
    seq = torch.zeros(n, C_)
    for j in range(n * 16):
        seq[idx[0, j]] += dy[j, 5:]
    assert torch.equal(got[0].cpu(), seq)
    x.grad = None
    pool = torch.randint(0, n, (clouds, m, 16), generator=g)
    ref = torch.gather(x, 1, pool.reshape(clouds, m * 16, 1).expand(-1, -1, C_)).reshape(clouds, m, 16, C_).max(dim=2)[0]
    out, arg = o.maxpool_fwd(x.detach().to(_dev()), pool.int().to(_dev()))
    assert torch.equal(out.cpu(), ref.detach())
    dp = rnd(clouds, m, C_)
    ref.backward(dp)
    pool_d = pool.int().to(_dev())
    gp = o.maxpool_bwd(dp.to(_dev()), arg, pool_d, n)
    assert torch.allclose(gp.cpu(), x.grad, rtol=1e-5, atol=1e-5)
    o.new_step()
    assert torch.equal(o.maxpool_bwd(dp.to(_dev()), arg, pool_d, n), gp)
    pts = 130
    cat = rnd(pts, 16, C_).requires_grad_()
    sc = rnd(pts, 16, C_).requires_grad_()
    a = torch.softmax(sc, dim=1)
    ref = (cat * a).sum(1)
    dout = rnd(pts, C_)
    ref.backward(dout)
    sd = sc.detach().reshape(pts * 16, C_).clone().to(_dev())
    cd = cat.detach().reshape(pts * 16, C_).to(_dev())
    out = o.attpool_fwd(cd, sd, pts)
    assert torch.allclose(out.cpu(), ref.detach(), rtol=1e-4, atol=1e-5) and torch.allclose(sd.cpu().reshape(pts, 16, C_), a.detach(), atol=1e-6)
    dcat, ds = o.attpool_bwd(dout.to(_dev()), cd, sd, pts)
    assert torch.allclose(dcat.cpu().reshape(pts, 16, C_), cat.grad, rtol=1e-4, atol=1e-5)
    assert torch.allclose(ds.cpu().reshape(pts, 16, C_), sc.grad, rtol=1e-4, atol=1e-5)
    a_, b_ = rnd(1000).requires_grad_(), rnd(1000)
    ref = F.leaky_relu(a_ + b_, 0.2)
    ref.backward(torch.ones(1000))
    out = o.add_leaky_fwd(a_.detach().to(_dev()), b_.to(_dev()))
    assert torch.allclose(out.cpu(), ref.detach())
    assert torch.allclose(o.add_leaky_bwd(torch.ones(1000, device=_dev()), out).cpu(), a_.grad)
    # relative position encoding and the inlier model's input
    xyz = rnd(clouds, n, 3)
    nb = torch.randint(0, n, (clouds, n, 16), generator=g)
    enc = o.relpos(xyz.to(_dev()), nb.int().to(_dev())).cpu().reshape(clouds, n, 16, 10)
    pj = torch.gather(xyz, 1, nb.reshape(clouds, n * 16, 1).expand(-1, -1, 3)).reshape(clouds, n, 16, 3)
    pi = xyz[:, :, None, :].expand_as(pj)
    assert torch.allclose(enc, torch.cat([(pj - pi).norm(dim=3, keepdim=True), pj - pi, pi, pj], 3), atol=1e-6)
    T = rnd(clouds, 4, 3, 4)
    ii = torch.randint(0, n, (clouds, n), generator=g)
    got = o.inlier_input(xyz.to(_dev()), xyz.flip(1).contiguous().to(_dev()), ii.int().to(_dev()), T.to(_dev())[:, 2]).cpu()
    want = torch.cat([xyz @ T[:, 2, :, :3].transpose(1, 2) + T[:, 2, :, 3][:, None], torch.gather(xyz.flip(1), 1, ii[:, :, None].expand(-1, -1, 3))], 2)
    assert torch.allclose(got, want, atol=1e-5)
    torch.cuda.synchronize()


def _trainer(sd):
    from deepsir_amd.train import RandlaTrainer
    return RandlaTrainer(CFG, sd, "inlier_model", 6, 1, _dev())


def _dev_inputs(d):
    f = lambda k, dt: torch.from_numpy(np.ascontiguousarray(d[k])).to(dt).to(_dev())
    return (f("cat", torch.float32), f("points_src_xyz", torch.float32), f("points_src_neigh_idx", torch.int32),
            f("points_src_sub_idx", torch.int32), f("points_src_interp_idx", torch.int32))


@pytest.mark.gpu
@pytest.mark.parametrize("c", range(int(GOLD["n_cases"])))
def test_device_training_pass_matches_reference_autograd(c):
    """Forward in training mode (logits, running statistics) and every parameter gradient of the inlier model, HIP path
    vs the imported reference's autograd."""
    meta, d, sd, keep = _case(c)
    tr = _trainer(sd)
    mask = torch.from_numpy(np.ascontiguousarray(keep.transpose(0, 2, 1))).to(torch.uint8).to(_dev())    # [clouds][N][64]
    logits, tape = tr.forward(*_dev_inputs(d), dropout_mask=mask)
    ref = GOLD[f"c{c}_logits"].transpose(0, 2, 1)
    assert np.abs(logits.cpu().numpy() - ref).max() < 5e-4
    G = torch.from_numpy(d["G"].transpose(0, 2, 1).copy()).to(_dev())
    tr.backward(tape, G)
    torch.cuda.synchronize()
    _check_grads(c, {k: v.cpu().numpy() for k, v in tr.grads.items()}, meta["seed"], 2e-3, 1e-6)
    for k in GOLD.files:
        if k.startswith(f"c{c}_buf_"):
            assert np.allclose(tr.buffers[k[len(f"c{c}_buf_"):]].cpu().numpy(), GOLD[k], rtol=1e-4, atol=1e-5), k
    # gradients of a second pass add up, as autograd's do
    logits, tape = tr.forward(*_dev_inputs(d), dropout_mask=mask, update_running_stats=False)
    tr.backward(tape, G)
    g2 = tr.grads["inlier_model.fc_label.6.weight"].cpu().numpy().reshape(-1)
    assert np.allclose(g2, 2 * GOLD[f"c{c}_g_inlier_model.fc_label.6.weight"], rtol=2e-3, atol=1e-5)


@pytest.mark.gpu
def test_device_adam_steps_match_reference():
    """Three optimiser steps (forward in training mode, backward, torch.optim.Adam's rule) vs the imported reference."""
    meta, d, sd, keep1 = _case(1)
    tr = _trainer(sd)
    n = meta["n"]
    inp = _dev_inputs(d)
    G = torch.from_numpy(d["G"].transpose(0, 2, 1).copy()).to(_dev())
    mk = lambda bits: torch.from_numpy(np.ascontiguousarray(np.unpackbits(bits)[: 2 * 64 * n].reshape(2, 64, n).transpose(0, 2, 1))).to(_dev())
    tr.forward(*inp, dropout_mask=mk(GOLD["c1_keep"]))                    # the reference's single pass of case 1 came first
    for s in range(3):
        tr.zero_grad()
        lg, tape = tr.forward(*inp, dropout_mask=mk(GOLD["adam_keep"][s]))
        tr.backward(tape, G)
        tr.adam_step(1e-3)
    torch.cuda.synchronize()
    # Adam divides by sqrt(v): a parameter whose gradient is at rounding-noise level still moves by ~lr per step, in the
    # direction the noise happens to have - on the reference's CPU as here.  So: every entry within the 3 lr any entry can
    # have moved apart (+ margin), the bulk (clear gradients) tight, the logits close.
    lr = 1e-3
    assert np.abs(lg.cpu().numpy() - GOLD["adam_logits_after"].transpose(0, 2, 1)).max() < 0.25
    assert np.median(np.abs(lg.cpu().numpy() - GOLD["adam_logits_after"].transpose(0, 2, 1))) < 0.02
    state = tr.state_dict()
    for name, ref in _adam_targets().items():
        v = state[name].reshape(-1)
        v = v if v.size <= FULL_MAX else v[sample_index(name[len("inlier_model."):], v.size, meta["seed"])]
        err = np.abs(v - ref)
        if name.endswith("running_var"):
            assert err.max() <= 2e-3 * np.abs(ref).max(), name
            continue
        assert err.max() <= 2 * 3 * lr * 1.05, name
        assert np.median(err) <= 2e-4, (name, float(np.median(err)))


@pytest.mark.gpu
def test_adam_kernel_matches_torch_optim():
    """dsir_t_adam against torch.optim.Adam on given gradients, five steps (bias corrections included)."""
    from deepsir_amd.train import _Ops, _ptr
    o = _Ops(_dev())
    g = torch.Generator().manual_seed(11)
    p = torch.randn(5000, generator=g)
    ref = p.clone().requires_grad_()
    opt = torch.optim.Adam([ref], lr=3e-3)
    pd, m, v = p.to(_dev()), torch.zeros(5000, device=_dev()), torch.zeros(5000, device=_dev())
    for step in range(1, 6):
        grad = torch.randn(5000, generator=g) * 10.0 ** float(torch.randint(-6, 2, (1,), generator=g))
        ref.grad = grad.clone()
        opt.step()
        assert o.lib.dsir_t_adam(o.stream, _ptr(pd), _ptr(grad.to(_dev())), _ptr(m), _ptr(v), 5000, 3e-3, 0.9, 0.999, 1e-8, step) == 0
    assert torch.allclose(pd.cpu(), ref.detach(), rtol=1e-5, atol=2e-6)


@pytest.mark.gpu
def test_train_step_align_lowers_the_loss():
    """The whole step of the `align` pipeline on the device: inference engine (no_grad half) -> inlier model in training
    mode per iteration -> ScanAlignmentLoss + gradient -> backward -> Adam.  A few steps on one batch must lower the loss,
    and the updated weights, loaded back into the engine, must change its poses accordingly."""
    from deepsir_amd.engine import Engine
    from deepsir_amd.synth import make_pair
    from deepsir_amd.train import train_step_align
    n, P, n_iter = 1024, 2, 3
    sd = generate_state_dict(CFG, 3, "plain")
    eng = Engine(CFG, max_points=n, max_pairs=P)
    eng.load_state_dict(sd)
    raws = [make_pair(n, 100 + b, 3) for b in range(P)]
    src = torch.from_numpy(np.concatenate([r["points_src"] for r in raws])).to(_dev())
    ref = torch.from_numpy(np.concatenate([r["points_ref"] for r in raws])).to(_dev())
    gt = torch.from_numpy(np.concatenate([r["transform_gt"] for r in raws]).astype(np.float32)).to(_dev())
    sx, sn, ss, si = eng.knn_pyramid(src)
    res = eng.register(src, ref, n_iter=n_iter)
    batch = {"points_src": src, "points_ref": ref, "src_xyz": sx, "src_neigh": sn, "src_sub": ss, "src_interp": si}
    tr = _trainer(sd)
    losses = []
    for step in range(6):
        out = train_step_align(eng, tr, batch, res, gt, labels=None, lr=2e-3, dropout_seed=None)
        assert not out["skipped"]
        losses.append(out["losses"]["total"])
    assert np.isfinite(losses).all() and losses[-1] < losses[0], losses


@pytest.mark.gpu
def test_train_step_align_gradients_match_the_oracle():
    """The composed step (inlier input per iteration, training-mode forward, ScanAlignmentLoss incl. the confidence term,
    gradients of all iterations accumulated) against the same composition under torch autograd on the CPU oracle."""
    from deepsir_amd.engine import Engine
    from deepsir_amd.synth import make_pair
    from deepsir_amd.train import train_step_align
    from oracle import align_loss as oal
    n, P, n_iter = 1024, 2, 3
    sd = generate_state_dict(CFG, 4, "plain")
    eng = Engine(CFG, max_points=n, max_pairs=P)
    eng.load_state_dict(sd)
    raws = [make_pair(n, 200 + b, 3) for b in range(P)]
    src_np = np.concatenate([r["points_src"] for r in raws])
    ref_np = np.concatenate([r["points_ref"] for r in raws])
    gt_np = np.concatenate([r["transform_gt"] for r in raws]).astype(np.float32)
    src, ref, gt = (torch.from_numpy(a).to(_dev()) for a in (src_np, ref_np, gt_np))
    sx, sn, ss, si = eng.knn_pyramid(src)
    res = eng.register(src, ref, n_iter=n_iter)
    rng = np.random.Generator(np.random.Philox(key=8))
    labels_np = (rng.random((n_iter, P, n)) < 0.6).astype(np.float32)
    batch = {"points_src": src, "points_ref": ref, "src_xyz": sx, "src_neigh": sn, "src_sub": ss, "src_interp": si}
    tr = _trainer(sd)
    out = train_step_align(eng, tr, batch, res, gt, labels=torch.from_numpy(labels_np).to(_dev()), apply=False)
    torch.cuda.synchronize()
    # ---- the same composition on the oracle
    net = OracleNet(CFG, sd)
    params = otrain.trainable(net)
    idx = [res["idx"][i].cpu().long() for i in range(n_iter)]
    T = res["transforms"].cpu()
    ps, pr = torch.from_numpy(src_np[:, :, :3]), torch.from_numpy(ref_np[:, :, :3])
    pyr = [t.cpu() for t in (sx, sn.long(), ss.long(), si.long())]
    logits = []
    for i in range(n_iter):
        cur = ps if i == 0 else OracleNet.se3_apply(T[:, i - 1], ps)
        cat = torch.cat([cur, torch.gather(pr, 1, idx[i][:, :, None].expand(-1, -1, 3))], 2)
        logits.append(otrain.randla_train(net, "inlier_model", cat, *pyr, None).squeeze(1))
    poses = oal.replay(ps, pr, idx, logits)
    d = oal.scan_alignment_loss(ps, poses, torch.from_numpy(gt_np), logits, [torch.from_numpy(l) for l in labels_np])
    d["total"].backward()
    assert abs(out["losses"]["total"] - float(d["total"])) < 1e-4 * max(1.0, abs(float(d["total"])))
    assert np.abs(out["logits"].cpu().numpy() - torch.stack(logits).detach().numpy()).max() < 1e-3
    worst = 0.0
    for k, p in params.items():
        g, r = tr.grads[k].cpu().numpy().reshape(-1), p.grad.numpy().reshape(-1)
        if k.endswith(ZERO_BY_CONSTRUCTION):
            continue
        err = np.abs(g - r).max() / (np.abs(r).max() + 1e-12)
        worst = max(worst, err)
        assert err < 2e-2, (k, err)
    assert worst > 0.0


@pytest.mark.gpu
def test_graph_replayed_step_equals_the_eager_step():
    """AlignTrainStep (forward and backward halves replayed from hipGraphs) against train_step_align, three steps each from
    the same start: same losses and, up to the order of the fp32 atomics in the scatter-adds, the same weights."""
    from deepsir_amd.engine import Engine
    from deepsir_amd.synth import make_pair
    from deepsir_amd.train import AlignTrainStep, train_step_align
    n, P, n_iter = 1024, 2, 3
    sd = generate_state_dict(CFG, 6, "plain")
    eng = Engine(CFG, max_points=n, max_pairs=P)
    eng.load_state_dict(sd)
    raws = [make_pair(n, 300 + b, 3) for b in range(P)]
    src = torch.from_numpy(np.concatenate([r["points_src"] for r in raws])).to(_dev())
    ref = torch.from_numpy(np.concatenate([r["points_ref"] for r in raws])).to(_dev())
    gt = torch.from_numpy(np.concatenate([r["transform_gt"] for r in raws]).astype(np.float32)).to(_dev())
    sx, sn, ss, si = eng.knn_pyramid(src)
    res = eng.register(src, ref, n_iter=n_iter)
    batch = {"points_src": src, "points_ref": ref, "src_xyz": sx, "src_neigh": sn, "src_sub": ss, "src_interp": si}
    a, b = _trainer(sd), _trainer(sd)
    stepper = AlignTrainStep(eng, b, P, n, n, n_iter, dropout=False)
    la, lb = [], []
    for s in range(4):
        la.append(train_step_align(eng, a, batch, res, gt, lr=1e-3)["losses"]["total"])
        lb.append(stepper.step(batch, res, gt, lr=1e-3)["losses"]["total"])
    assert stepper.gf is not None and stepper.gb is not None
    assert np.allclose(la, lb, rtol=2e-3), (la, lb)
    assert la[-1] < la[0]
    torch.cuda.synchronize()
    d = (a.flat_p - b.flat_p).abs()
    assert float(d.max()) <= 2 * 4 * 1e-3 * 1.05 and float(d.median()) < 1e-4     # see test_device_adam_steps_match_reference


@pytest.mark.gpu
def test_label_pipeline_training_step_matches_the_oracle():
    """`label` pipeline (semantic head of the feature extractor): weighted cross entropy over the labelled points (op vs
    F.cross_entropy), then the whole step - two training-mode forwards (src, ref: separate BatchNorm statistics),
    loss_src + loss_ref, backward - against the same composition under torch autograd on the oracle; then Adam lowers it."""
    import torch.nn.functional as F
    from deepsir_amd.arch import semantic_class_weights
    from deepsir_amd.engine import Engine
    from deepsir_amd.synth import make_pair
    from deepsir_amd.train import RandlaTrainer, _Ops, train_step_label
    o = _Ops(_dev())
    g = torch.Generator().manual_seed(3)
    cw = torch.tensor(semantic_class_weights())
    lg = torch.randn(3000, 19, generator=g).requires_grad_()
    lb = torch.randint(0, 20, (3000,), generator=g)
    keep = lb != 0
    ref = F.cross_entropy(lg[keep], lb[keep] - 1, weight=cw)
    ref.backward()
    d, out = o.weighted_ce(lg.detach().to(_dev()), lb.int().to(_dev()), cw.to(_dev()))
    out = out.cpu().numpy()
    assert abs(out[0] - float(ref)) < 1e-5 and out[3] == int(keep.sum())
    assert out[2] == int((lg.detach()[keep].argmax(1) == lb[keep] - 1).sum())
    assert torch.allclose(d.cpu(), lg.grad, rtol=1e-4, atol=1e-8)
    # ---- the step
    n, P = 1024, 2
    sd = generate_state_dict(CFG, 7, "plain")
    eng = Engine(CFG, max_points=n, max_pairs=P)
    eng.load_state_dict(sd)
    raws = [make_pair(n, 400 + b, 3) for b in range(P)]
    pts = {s: torch.from_numpy(np.concatenate([r[f"points_{s}"] for r in raws])).to(_dev()) for s in ("src", "ref")}
    batch = {"points_src": pts["src"], "points_ref": pts["ref"]}
    for s in ("src", "ref"):
        batch[f"{s}_xyz"], batch[f"{s}_neigh"], batch[f"{s}_sub"], batch[f"{s}_interp"] = eng.knn_pyramid(pts[s])
    labels = {s: torch.randint(0, 20, (P, n), generator=g) for s in ("src", "ref")}
    tr = RandlaTrainer(CFG, sd, "feat_extractor", CFG.feat_len, CFG.num_classes, _dev())
    res = train_step_label(tr, batch, labels["src"].int().to(_dev()), labels["ref"].int().to(_dev()), apply=False)
    torch.cuda.synchronize()
    net = OracleNet(CFG, sd)
    params = otrain.trainable(net, "feat_extractor")
    total = 0.0
    for s in ("src", "ref"):
        pyr = [batch[f"{s}_xyz"].cpu(), batch[f"{s}_neigh"].cpu().long(), batch[f"{s}_sub"].cpu().long(), batch[f"{s}_interp"].cpu().long()]
        lgt = otrain.randla_train(net, "feat_extractor", pts[s].cpu(), *pyr, None)
        assert np.abs(res[f"logits_{s}"].cpu().numpy() - lgt.detach().permute(0, 2, 1).numpy()).max() < 1e-3
        total = total + otrain.semantic_loss(lgt, labels[s], semantic_class_weights())
    total.backward()
    assert abs(res["loss"] - float(total)) < 1e-4 * float(total)
    for k, p in params.items():
        if k.endswith(ZERO_BY_CONSTRUCTION):
            continue
        gd, r = tr.grads[k].cpu().numpy().reshape(-1).astype(np.float64), p.grad.numpy().reshape(-1).astype(np.float64)
        # a gradient is piecewise in the forward values (arg-max of the pooling, LeakyReLU sign): a rounding-level difference of
        # the forward may re-route single points, so: tight in the L2 sense, bounded entry-wise
        assert np.linalg.norm(gd - r) <= 5e-3 * np.linalg.norm(r) + 1e-7, k
        assert np.abs(gd - r).max() <= 3e-2 * np.abs(r).max() + 1e-7, k
    losses = [train_step_label(tr, batch, labels["src"].int().to(_dev()), labels["ref"].int().to(_dev()), lr=2e-3, dropout_seed=i)["loss"]
              for i in range(6)]
    assert losses[-1] < losses[0], losses


# ------------------------------------------------------------------------------------------------- `feat` pipeline
def _feat_case():
    meta = json.loads(str(GOLD["feat_meta"]))
    cfg = NetConfig(feat_len=3, pipeline="feat", num_sub=meta["num_sub"])
    sd = generate_state_dict(cfg, meta["wseed"], meta["variant"])
    t = {k: torch.from_numpy(GOLD["feat_in_" + k]) for k in ("xyz_src", "xyz_ref", "feat_src", "feat_ref", "score_src", "score_ref")}
    return meta, cfg, sd, t


def _check_feat_grads(grads, rtol, atol=1e-7):
    n = 0
    for name, g in grads.items():
        g = np.asarray(g, np.float64).reshape(-1)
        key = "feat_g_" + name
        if name.endswith(".bias") and (name[:-4] + "weight") in grads and name.split(".")[0] in ("mlp_feat", "mlp_att") and \
                not name.endswith(("mlp_feat.6.bias", "mlp_att.12.bias")) and int(name.split(".")[1]) % 3 == 0:
            continue                                   # a Conv1d bias in front of BatchNorm: zero by construction (noise both sides)
        if key in GOLD:
            ref = GOLD[key].astype(np.float64)
        else:
            ref = GOLD[key + "_samples"].astype(np.float64)
            g = g[sample_index(name, g.size, 21)]
        assert np.abs(g - ref).max() <= rtol * np.abs(ref).max() + atol, (name, float(np.abs(g - ref).max()), float(np.abs(ref).max()))
        n += 1
    assert n >= 20


def test_oracle_feat_pipeline_matches_reference_autograd():
    """oracle/train.py's aggregation in training mode + DetDesLoss against the imported reference's own forward / backward
    (descriptors, loss, accuracy, the gradient of all 30 trainable tensors, BatchNorm running statistics)."""
    meta, cfg, sd, t = _feat_case()
    net = OracleNet(cfg, sd)
    params = {k: v.requires_grad_(True) for k, v in net.p.items()
              if k.startswith(("mlp_feat", "mlp_att", "mlp_proj")) and v.dtype == torch.float32 and not k.endswith(("running_mean", "running_var"))}
    # the reference's call order: mlp_feat(src), mlp_feat(ref), mlp_att(src), mlp_att(ref) - per module the src pass first
    d_src = otrain.aggregate_train(net, t["xyz_src"], t["feat_src"], t["score_src"])
    d_ref = otrain.aggregate_train(net, t["xyz_ref"], t["feat_ref"], t["score_ref"])
    assert np.abs(d_src.detach().numpy() - GOLD["feat_desc_src"]).max() < 1e-5
    assert np.abs(d_ref.detach().numpy() - GOLD["feat_desc_ref"]).max() < 1e-5
    loss, acc = otrain.det_des_loss(d_src, d_ref, t["xyz_src"], t["xyz_ref"], t["score_ref"], torch.from_numpy(GOLD["feat_transform_gt"]),
                                    meta["thres_radius"], meta["det_loss_weight"])
    assert abs(float(loss) - GOLD["feat_loss_acc"][0]) < 1e-5 and abs(float(acc) - GOLD["feat_loss_acc"][1]) < 1e-3
    loss.backward()
    _check_feat_grads({k: v.grad.numpy() for k, v in params.items()}, 1e-3)
    for k in GOLD.files:
        if k.startswith("feat_buf_"):
            assert np.allclose(net.p[k[len("feat_buf_"):]].numpy(), GOLD[k], rtol=1e-5, atol=1e-6), k


@pytest.mark.gpu
def test_device_feat_pipeline_step_matches_reference_autograd():
    """`feat` pipeline on the device: aggregation MLPs in training mode (descriptors, running statistics), DetDesLoss (total,
    accuracy) and the gradient of all 30 trainable tensors, against the imported reference's own forward / backward; the
    loss operator also against the oracle on a second problem with coincident points and ties; then Adam lowers the loss."""
    from deepsir_amd.train import AggregationTrainer, _Ops, train_step_feat
    meta, cfg, sd, t = _feat_case()
    tr = AggregationTrainer(cfg, sd, _dev())
    pm = lambda x: x.permute(0, 2, 1).contiguous().to(_dev())
    inp = {"xyz_src": pm(t["xyz_src"]), "xyz_ref": pm(t["xyz_ref"]), "feat_src": pm(t["feat_src"]), "feat_ref": pm(t["feat_ref"]),
           "score_src": t["score_src"].to(_dev()), "score_ref": t["score_ref"].to(_dev())}
    gt = torch.from_numpy(GOLD["feat_transform_gt"]).to(_dev())
    res = train_step_feat(tr, inp, gt, meta["thres_radius"], meta["det_loss_weight"], apply=False)
    torch.cuda.synchronize()
    assert np.abs(res["desc_src"].cpu().numpy() - GOLD["feat_desc_src"].transpose(0, 2, 1)).max() < 1e-4
    assert np.abs(res["desc_ref"].cpu().numpy() - GOLD["feat_desc_ref"].transpose(0, 2, 1)).max() < 1e-4
    assert abs(res["loss"] - GOLD["feat_loss_acc"][0]) < 2e-4 and abs(res["acc"] - GOLD["feat_loss_acc"][1]) < 1.0
    _check_feat_grads({k: v.cpu().numpy() for k, v in tr.grads.items()}, 5e-3, 1e-6)
    for k in GOLD.files:
        if k.startswith("feat_buf_"):
            assert np.allclose(tr.buffers[k[len("feat_buf_"):]].cpu().numpy(), GOLD[k], rtol=1e-4, atol=1e-5), k
    # the loss operator alone: exact coincidences (pos_mask true), duplicated descriptors, P = 3
    o = _Ops(_dev())
    g = torch.Generator().manual_seed(9)
    P, M = 3, 200
    fr = torch.nn.functional.normalize(torch.randn(P, 64, M, generator=g), dim=1).requires_grad_()
    fs = torch.nn.functional.normalize(torch.randn(P, 64, M, generator=g), dim=1).requires_grad_()
    ps = torch.rand(P, 3, M, generator=g) * 2
    T = torch.eye(3, 4)[None].repeat(P, 1, 1)
    T[:, :, 3] = torch.randn(P, 3, generator=g) * 0.1      # identity rotation: p + t rounds the same in any evaluation order
    pr = (ps + T[:, :, 3:]).clone()
    pr[:, :, M // 2:] += torch.randn(P, 3, M - M // 2, generator=g) * 0.3      # half the points coincide exactly after T_gt
    sc = torch.rand(P, M, generator=g) + 0.1
    loss, acc = otrain.det_des_loss(fs, fr, ps, pr, sc, T, 0.2, 0.7)
    loss.backward()
    out, d_ref, d_src = o.det_des_loss(pm(fr.detach()), pm(fs.detach()), pm(pr), pm(ps), sc.to(_dev()), T.to(_dev()), 0.2, 0.7)
    out = out.cpu().numpy()
    assert abs(out[0] - float(loss.detach())) < 1e-4 * max(1.0, abs(float(loss.detach()))) and abs(out[3] - float(acc)) < 1.0
    assert np.abs(d_ref.cpu().numpy() - fr.grad.permute(0, 2, 1).numpy()).max() <= 5e-3 * float(fr.grad.abs().max())
    assert np.abs(d_src.cpu().numpy() - fs.grad.permute(0, 2, 1).numpy()).max() <= 5e-3 * float(fs.grad.abs().max())
    losses = [train_step_feat(tr, inp, gt, meta["thres_radius"], meta["det_loss_weight"], lr=1e-3)["loss"] for _ in range(8)]
    assert losses[-1] < losses[0], losses


@pytest.mark.gpu
def test_feat_pipeline_step_from_the_inference_engine():
    """`feat_pipeline_inputs` (frozen half from an Engine built for the `feat` pipeline: key-point selection, raw features of the
    selected points) feeding `train_step_feat`: the selected features are the extractor's rows at `index`, and training lowers
    the loss on real engine outputs."""
    from deepsir_amd.engine import Engine
    from deepsir_amd.synth import make_pair
    from deepsir_amd.train import AggregationTrainer, feat_pipeline_inputs, train_step_feat
    n, P, M = 1024, 2, 256
    cfg = NetConfig(feat_len=3, pipeline="feat", num_sub=M)
    sd = generate_state_dict(cfg, 21, "separated")
    eng = Engine(cfg, max_points=n, max_pairs=P)
    eng.load_state_dict(sd)
    raws = [make_pair(n, 600 + b, 3) for b in range(P)]
    batch = {f"points_{s}": torch.from_numpy(np.concatenate([r[f"points_{s}"] for r in raws])).to(_dev()) for s in ("src", "ref")}
    gt = torch.from_numpy(np.concatenate([r["transform_gt"] for r in raws]).astype(np.float32)).to(_dev())
    gt[:, :, 3] += 2e-3                                                  # off the exact-coincidence knife edge
    inp = feat_pipeline_inputs(eng, batch, M)
    fp = eng.forward_pair(batch["points_src"], batch["points_ref"], num_sub=M)
    feat, _ = eng.randla_forward("feat_extractor", batch["points_src"], *eng.knn_pyramid(batch["points_src"]), want_logits=False)
    want = torch.gather(feat, 1, fp["src"]["index"].long()[:, :, None].expand(-1, -1, 64))
    assert torch.equal(inp["feat_src"], want) and torch.equal(inp["xyz_src"], fp["src"]["xyz"])
    tr = AggregationTrainer(cfg, sd, _dev())
    first = train_step_feat(tr, inp, gt, 0.15, 1.0, apply=False)
    assert np.allclose(np.linalg.norm(first["desc_src"].cpu().numpy(), axis=2), 1.0, atol=1e-5)    # unit descriptors (training-mode
    # BatchNorm uses batch statistics, so they differ from the engine's evaluation-mode descriptors by construction)
    losses = [train_step_feat(tr, inp, gt, 0.15, 1.0, lr=1e-3)["loss"] for _ in range(8)]
    assert np.isfinite(losses).all() and losses[-1] < losses[0], losses


@pytest.mark.gpu
def test_data_parallel_step_two_ranks_on_one_gpu():
    """Data-parallel training (`all_reduce_gradients`): two ranks with different pairs (both on the one visible GPU, gloo because
    RCCL refuses two ranks per device), one all_reduce of the flat gradient buffer per step: the reduced gradient is the mean of
    the local ones and both ranks end with bit-identical weights."""
    import socket, subprocess, sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["DSIR_BENCH_BACKEND"] = "gloo"
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "tools", "train_dp_check.py")], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert j["world_size"] == 2 and j["grad_is_mean"] and j["params_identical"], j


def test_find_correct_correspondence_is_the_references_rule():
    from deepsir_amd.train import find_correct_correspondence
    from oracle.align_loss import find_correct_correspondence as oracle_fcc      # pinned restatement (align_loss_cases.npz)
    rng = np.random.Generator(np.random.Philox(key=4))
    J, P, n_iter = 300, 3, 2
    matches = [np.stack([rng.permutation(J)[:200], rng.integers(0, J, 200)], 1) for _ in range(P)]
    idx = rng.integers(0, J, (n_iter, P, J))
    for p in range(P):
        idx[:, p, matches[p][:50, 0]] = matches[p][:50, 1]                         # some predictions hit their match
    got = find_correct_correspondence(matches, idx, J)
    for i in range(n_iter):
        for p in range(P):
            want = oracle_fcc(matches[p], np.stack([np.arange(J), idx[i, p]], 1), J)
            assert np.array_equal(got[i, p].astype(bool), want)
    assert got.sum() >= n_iter * P * 50


@pytest.mark.gpu
@pytest.mark.parametrize("pipeline", ["align", "label", "feat"])
def test_network_train_step_drop_in(pipeline):
    """`Network.train_step`: the reference loop's per-batch work behind the drop-in module - the loss falls over a few steps,
    the trained tensors land in `state_dict()` and the next `forward` uses them."""
    from types import SimpleNamespace
    from deepsir_amd.model import Network
    from deepsir_amd.synth import make_pair
    from deepsir_amd.weights import to_torch_state_dict
    n, P = 1024, 2
    args = SimpleNamespace(pipeline=pipeline, feat_len=3, num_sub=256 if pipeline == "feat" else -1, num_reg_iter=3)
    net = Network(args)
    sd = generate_state_dict(net.cfg, 8, "plain" if pipeline == "label" else "separated")
    net.load_state_dict(to_torch_state_dict(sd))
    net.cuda().eval()
    raws = [make_pair(n, 800 + b, 3) for b in range(P)]
    data = {k: torch.from_numpy(np.concatenate([r[k] for r in raws])).cuda() for k in ("points_src", "points_ref")}
    data["transform_gt"] = torch.from_numpy(np.concatenate([r["transform_gt"] for r in raws]).astype(np.float32)).cuda()
    if pipeline == "feat":
        data["transform_gt"][:, :, 3] += 2e-3
    if pipeline == "label":
        g = torch.Generator().manual_seed(2)
        data["labels_src"], data["labels_ref"] = (torch.randint(0, 20, (P, n), generator=g) for _ in range(2))
    if pipeline == "align":
        data["matches"] = [np.stack([np.arange(n), np.arange(n)], 1) for _ in range(P)]
    key = {"align": "inlier_model.mlp_out.weight", "label": "feat_extractor.mlp_out.weight", "feat": "mlp_proj.0.weight"}[pipeline]
    frozen = {"align": "feat_extractor.mlp_out.weight", "label": None, "feat": "feat_extractor.mlp_out.weight"}[pipeline]
    before = {k: v.clone() for k, v in net.state_dict().items()}
    losses = [net.train_step(data, (3, False), lr=2e-3, dropout_seed=None if pipeline == "align" else i)["loss"] for i in range(6)]
    assert np.isfinite(losses).all() and losses[-1] < losses[0], losses
    after = net.state_dict()
    assert not torch.equal(after[key], before[key])
    if frozen:
        assert torch.equal(after[frozen], before[frozen])
    if pipeline == "align":       # my_model.train(): the frozen sub-networks' BatchNorm running statistics move, their weights do not
        assert not torch.equal(after["mlp_att.1.running_mean"], before["mlp_att.1.running_mean"])
        assert torch.equal(after["mlp_att.0.weight"], before["mlp_att.0.weight"])
        l2 = [net.train_step(data, (3, False), lr=2e-3, dropout_seed=i, frozen_mode="eval")["loss"] for i in range(2)]
        assert np.isfinite(l2).all()
    out = net(data, (3, False))                                                     # inference with the trained weights
    assert out[1] is not None


# ------------------------------------------------------------------------------------------------- `align`, whole network in training mode
def _align_case():
    from oracle.knn import add_pyramids
    from deepsir_amd.synth import make_pair
    meta = json.loads(str(GOLD["align_meta"]))
    cfg = NetConfig(feat_len=3)
    sd = generate_state_dict(cfg, meta["wseed"], meta["variant"])
    raws = [add_pyramids(make_pair(meta["n"], s, 3), cfg.num_knn, cfg.sub_sampling_ratio) for s in meta["seeds"]]
    d = {k: np.concatenate([r[k] for r in raws], 0) for k in raws[0]}
    n, B = meta["n"], len(meta["seeds"])
    unpack = lambda bits: np.unpackbits(bits)[: B * 64 * n].reshape(B, 64, n).astype(bool)
    masks = {"fe_src": unpack(GOLD["align_keep_fe"][0]), "fe_ref": unpack(GOLD["align_keep_fe"][1]),
             "inlier": [unpack(b) for b in GOLD["align_keep_inl"]]}
    return meta, cfg, sd, d, masks


def _check_align_grads(grads, rtol, atol):
    n = 0
    for name, g in grads.items():
        if name.endswith(ZERO_BY_CONSTRUCTION):
            continue
        g = np.asarray(g, np.float64).reshape(-1)
        key = "align_g_" + name
        ref = GOLD[key].astype(np.float64) if key in GOLD else GOLD[key + "_samples"].astype(np.float64)
        if key not in GOLD:
            g = g[sample_index(name, g.size, 33)]
        # piecewise gradients (pooling arg-max, LeakyReLU sign): tight in the L2 sense, bounded entry-wise (3 rtol)
        assert np.linalg.norm(g - ref) <= rtol * np.linalg.norm(ref) + atol * np.sqrt(g.size), (name, float(np.linalg.norm(g - ref)), float(np.linalg.norm(ref)))
        assert np.abs(g - ref).max() <= 3 * rtol * np.abs(ref).max() + atol, (name, float(np.abs(g - ref).max()), float(np.abs(ref).max()))
        n += 1
    assert n > 100


def test_oracle_whole_network_training_forward_matches_reference():
    """forward_align_4 with EVERY sub-network in training mode (my_model.train(), train.py:379), ScanAlignmentLoss, backward:
    the oracle composition against the imported reference's own run - correspondences, logits, poses, loss, the gradient of all
    155 inlier tensors and the running statistics of every BatchNorm (frozen sub-networks included)."""
    from oracle import align_loss as oal
    meta, cfg, sd, d, masks = _align_case()
    net = OracleNet(cfg, sd)
    params = otrain.trainable(net)
    t = to_torch(d)
    tm = {"fe_src": torch.from_numpy(masks["fe_src"]), "fe_ref": torch.from_numpy(masks["fe_ref"]),
          "inlier": [torch.from_numpy(m) for m in masks["inlier"]]}
    T, idx, lg = otrain.register_train(net, t, meta["n_iter"], tm)
    assert np.array_equal(torch.stack(idx).numpy(), GOLD["align_idx"].astype(np.int64))
    assert np.abs(torch.stack(lg).detach().numpy() - GOLD["align_logits"]).max() < 5e-5
    assert np.abs(torch.stack(T, 1).detach().numpy() - GOLD["align_transforms"]).max() < 1e-5
    loss = oal.scan_alignment_loss(t["points_src"][:, :, :3], T, t["transform_gt"], lg, None)["total"]
    assert abs(float(loss.detach()) - float(GOLD["align_loss"])) < 1e-5
    loss.backward()
    _check_align_grads({k: v.grad.numpy() for k, v in params.items()}, 1e-3, 1e-8)
    bufs = [k for k in GOLD.files if k.startswith("align_buf_")]
    assert len(bufs) == 2 * (2 + 2 + 4 + 2)
    for k in bufs:
        assert np.allclose(net.p[k[len("align_buf_"):]].numpy(), GOLD[k], rtol=1e-4, atol=1e-6), k


@pytest.mark.gpu
def test_device_whole_network_training_step_matches_reference():
    """`train_step_align_full` (whole network in training mode on the device: extractor, score, aggregation per iteration,
    arg-min, inlier model, Kabsch, loss, backward) against the imported reference's own training forward / backward."""
    from deepsir_amd.engine import Engine
    from deepsir_amd.train import AggregationTrainer, RandlaTrainer, train_step_align_full
    meta, cfg, sd, d, masks = _align_case()
    n, P, n_iter = meta["n"], len(meta["seeds"]), meta["n_iter"]
    eng = Engine(cfg, max_points=n, max_pairs=P)
    eng.load_state_dict(sd)
    f = lambda k, dt: torch.from_numpy(np.ascontiguousarray(d[k])).to(dt).to(_dev())
    batch = {"points_src": f("points_src", torch.float32), "points_ref": f("points_ref", torch.float32)}
    for s in ("src", "ref"):
        batch[f"{s}_xyz"], batch[f"{s}_neigh"] = f(f"points_{s}_xyz", torch.float32), f(f"points_{s}_neigh_idx", torch.int32)
        batch[f"{s}_sub"], batch[f"{s}_interp"] = f(f"points_{s}_sub_idx", torch.int32), f(f"points_{s}_interp_idx", torch.int32)
    mk = lambda m: torch.from_numpy(np.ascontiguousarray(m.transpose(0, 2, 1))).to(torch.uint8).to(_dev())
    dm = {"fe_src": mk(masks["fe_src"]), "fe_ref": mk(masks["fe_ref"]), "inlier": torch.stack([mk(m) for m in masks["inlier"]])}
    inl = RandlaTrainer(cfg, sd, "inlier_model", 6, 1, _dev())
    fe = RandlaTrainer(cfg, sd, "feat_extractor", cfg.feat_len, cfg.num_classes, _dev())
    ag = AggregationTrainer(cfg, sd, _dev())
    out = train_step_align_full(eng, inl, fe, ag, batch, f("transform_gt", torch.float32), n_iter, None, masks=dm, apply=False)
    torch.cuda.synchronize()
    agree = (out["idx"].cpu().numpy() == GOLD["align_idx"].astype(np.int32)).mean()
    assert agree == 1.0, agree
    assert np.abs(out["logits"].cpu().numpy() - GOLD["align_logits"]).max() < 2e-3
    assert np.abs(out["transforms"].cpu().numpy() - GOLD["align_transforms"]).max() < 1e-4
    assert abs(out["losses"]["total"] - float(GOLD["align_loss"])) < 1e-4
    _check_align_grads({k: v.cpu().numpy() for k, v in inl.grads.items()}, 1e-2, 1e-7)
    for tr in (inl, fe, ag):
        for k, v in tr.buffers.items():
            assert np.allclose(v.cpu().numpy(), GOLD["align_buf_" + k], rtol=1e-3, atol=1e-5), k


@pytest.mark.gpu
def test_feat_pipeline_front_end_in_training_mode_matches_reference():
    """The frozen extractor as train.py leaves it (training mode: Dropout on, BatchNorm of the semantic head on batch
    statistics) -> key-point scores -> top-num_sub selection: the key points, their raw features and scores handed to the
    aggregation equal what the imported reference handed its own (the `feat_in_*` vectors of train_cases.npz)."""
    from oracle.knn import add_pyramids
    from deepsir_amd.engine import Engine
    from deepsir_amd.synth import make_pair
    from deepsir_amd.train import RandlaTrainer, feat_pipeline_inputs_train
    meta, cfg, sd, t = _feat_case()
    n, P, M = 1024, 2, meta["num_sub"]
    raws = [add_pyramids(make_pair(n, int(s), 3), cfg.num_knn, cfg.sub_sampling_ratio) for s in GOLD["feat_seeds"]]
    d = {k: np.concatenate([r[k] for r in raws], 0) for k in raws[0]}
    eng = Engine(cfg, max_points=n, max_pairs=P)
    eng.load_state_dict(sd)
    f = lambda k, dt: torch.from_numpy(np.ascontiguousarray(d[k])).to(dt).to(_dev())
    batch = {"points_src": f("points_src", torch.float32), "points_ref": f("points_ref", torch.float32)}
    for s in ("src", "ref"):
        batch[f"{s}_xyz"], batch[f"{s}_neigh"] = f(f"points_{s}_xyz", torch.float32), f(f"points_{s}_neigh_idx", torch.int32)
        batch[f"{s}_sub"], batch[f"{s}_interp"] = f(f"points_{s}_sub_idx", torch.int32), f(f"points_{s}_interp_idx", torch.int32)
    unpack = lambda bits: torch.from_numpy(np.ascontiguousarray(np.unpackbits(bits)[: P * 64 * n].reshape(P, 64, n).transpose(0, 2, 1))).to(_dev())
    masks = {"fe_src": unpack(GOLD["feat_keep_fe"][0]), "fe_ref": unpack(GOLD["feat_keep_fe"][1])}
    fe = RandlaTrainer(cfg, sd, "feat_extractor", cfg.feat_len, cfg.num_classes, _dev())
    inp = feat_pipeline_inputs_train(eng, fe, batch, M, masks)
    torch.cuda.synchronize()
    for s in ("src", "ref"):
        want_score = GOLD[f"feat_in_score_{s}"]
        got_score = inp[f"score_{s}"].cpu().numpy()
        assert np.abs(got_score - want_score).max() < 1e-4
        clear = np.ones_like(want_score, bool)                       # entries whose rank is not decided by a rounding-level score gap
        clear[:, 1:] &= (want_score[:, :-1] - want_score[:, 1:]) > 1e-5
        clear[:, :-1] &= (want_score[:, :-1] - want_score[:, 1:]) > 1e-5
        assert clear.mean() > 0.8
        got_xyz, want_xyz = inp[f"xyz_{s}"].cpu().numpy(), GOLD[f"feat_in_xyz_{s}"].transpose(0, 2, 1)
        assert np.abs(got_xyz - want_xyz)[clear].max() == 0.0        # the same key points in the same order
        got_f, want_f = inp[f"feat_{s}"].cpu().numpy(), GOLD[f"feat_in_feat_{s}"].transpose(0, 2, 1)
        assert np.abs(got_f - want_f)[clear].max() < 1e-3


@pytest.mark.gpu
def test_optimizer_state_round_trips_with_torch_adam():
    """The device optimiser's state in torch.optim.Adam's checkpoint format (what CheckPointManager saves): after the same
    three steps on the same gradients it equals torch's (indices = my_model.parameters() order), and a torch state loaded
    into the device trainer continues identically."""
    from types import SimpleNamespace
    from deepsir_amd.model import Network
    from deepsir_amd.weights import to_torch_state_dict
    net = Network(SimpleNamespace(pipeline="feat", feat_len=3, num_sub=256))
    sd = generate_state_dict(net.cfg, 5, "separated")
    net.load_state_dict(to_torch_state_dict(sd))
    net.cuda()
    net.prepare_training()
    tr = net._trainer
    order = net._param_order()
    ref_params = [torch.nn.Parameter(torch.from_numpy(np.array(sd[k])).clone()) for k in order]
    opt = torch.optim.Adam(ref_params, lr=2e-3)
    g = torch.Generator().manual_seed(1)
    trained = [i for i, k in enumerate(order) if k in tr.params]
    for step in range(3):
        tr.zero_grad()
        for i in trained:
            gr = torch.randn(ref_params[i].shape, generator=g)
            ref_params[i].grad = gr.clone()
            tr.grads[order[i]].copy_(gr.reshape(tr.grads[order[i]].shape).cuda())
        opt.step()
        tr.adam_step(2e-3)
    mine, theirs = net.optimizer_state_dict(2e-3), opt.state_dict()
    assert sorted(mine["state"]) == sorted(theirs["state"]) == trained
    for i in trained:
        for key in ("exp_avg", "exp_avg_sq"):
            assert torch.allclose(mine["state"][i][key], theirs["state"][i][key], rtol=1e-4, atol=1e-6), (order[i], key)   # lerp vs mul-add rounding
        assert float(mine["state"][i]["step"]) == float(theirs["state"][i]["step"]) == 3.0
        assert torch.allclose(tr.params[order[i]].cpu().reshape(ref_params[i].shape), ref_params[i].detach(), rtol=1e-5, atol=2e-6)
    # resume: a fresh trainer loaded with torch's state takes the same fourth step
    net2 = Network(SimpleNamespace(pipeline="feat", feat_len=3, num_sub=256))
    cur = {k: v for k, v in net.state_dict().items()}
    for i in trained:
        cur[order[i]] = ref_params[i].detach().clone()
    net2.load_state_dict(cur)
    net2.cuda()
    net2.prepare_training()
    net2.load_optimizer_state_dict(theirs)
    for i in trained:
        gr = torch.randn(ref_params[i].shape, generator=g)
        ref_params[i].grad = gr.clone()
        net2._trainer.grads[order[i]].copy_(gr.reshape(net2._trainer.grads[order[i]].shape).cuda())
    opt.step()
    net2._trainer.adam_step(2e-3)
    for i in trained:
        assert torch.allclose(net2._trainer.params[order[i]].cpu().reshape(ref_params[i].shape), ref_params[i].detach(), rtol=1e-5, atol=2e-6)


# --------------------------------------------------------------------------- dW scratch sizing
def test_dw_scratch_covers_both_plans_host_arithmetic():
    """dsir_t_gemm_dw plans its row split with or without the bias column; the sizing call must cover whichever needs more
    partial matrices (N = K = 128 at 319 488 rows, db = NULL, needs 33.5 MB - the round-2 sizing returned 22.4 MB).  Host
    arithmetic only: callable without a GPU."""
    from deepsir_amd import _lib
    lib = _lib.load()
    TB, TK = 64, 16

    def plan(rows, N, K, bias):
        Kp = K + (1 if bias else 0)
        tiles = ((N + TB - 1) // TB) * ((Kp + TB - 1) // TB)
        sp = min((rows + 511) // 512, 1 if tiles >= 2048 else 2048 // tiles, 1024)
        sp = max(sp, 1)
        rps = (((rows + sp - 1) // sp) + TK - 1) // TK * TK
        return max((rows + rps - 1) // rps, 1) * N * Kp * 4

    for rows, N, K in [(319488, 128, 128), (200000, 128, 128), (1_000_000, 256, 256), (80000, 8, 10), (5000, 64, 64), (1, 1, 1),
                       (294912, 64, 64), (640000, 128, 64)]:
        need = max(plan(rows, N, K, True), plan(rows, N, K, False))
        got = int(lib.dsir_t_gemm_dw_scratch(rows, N, K))
        assert got >= need, (rows, N, K, got, need)


@pytest.mark.gpu
def test_dw_without_bias_stays_inside_its_scratch():
    """conv_dw(db = None) at N = K = 128, 319 488 rows: the partial matrices must stay inside dsir_t_gemm_dw_scratch bytes
    (guard region behind the scratch untouched) and the result must equal dY^T X."""
    import ctypes as C
    from deepsir_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    rows, N, K = 319488, 128, 128
    g = torch.Generator(device="cpu").manual_seed(3)
    dy = torch.randn(rows, N, generator=g).to(dev)
    x = torch.randn(rows, K, generator=g).to(dev)
    nbytes = int(lib.dsir_t_gemm_dw_scratch(rows, N, K))
    guard = 1 << 20
    buf = torch.full(((nbytes + 3) // 4 + guard,), 12345.0, device=dev)
    dw = torch.zeros(N, K, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    rc = lib.dsir_t_gemm_dw(st, dy.data_ptr(), N, x.data_ptr(), K, rows, N, K, dw.data_ptr(), None, buf.data_ptr())
    torch.cuda.synchronize()
    assert rc == 0
    assert bool((buf[(nbytes + 3) // 4:] == 12345.0).all()), "dsir_t_gemm_dw wrote past dsir_t_gemm_dw_scratch bytes"
    ref = dy.double().t() @ x.double()
    err = float((dw.double() - ref).abs().max() / ref.abs().max())
    assert err < 1e-5, err

"""The C-ABI library must load on a CPU-only host and export every symbol
include/*.h declare (no compute calls here: there is no GPU)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _declared_symbols():
    txt = "".join(open(os.path.join(ROOT, "include", h)).read() for h in ("dsir.h", "dsir_train.h"))
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(dsir_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from deepsir_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build it first: python -c 'import __graft_entry__ as g; g.build()'"
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 19
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/*.h but not exported"
    assert set(declared) == set(_lib.SYMBOLS), "ctypes binding and header disagree"


def test_struct_layout_matches_header():
    from deepsir_amd import _lib
    assert ctypes.sizeof(_lib.dsir_cfg) == 4 * (3 + 4 + 4 + 4 + 1)
    assert ctypes.sizeof(_lib.dsir_pair_batch) == 16 + 11 * 8
    assert ctypes.sizeof(_lib.dsir_pair_result) == 7 * 8
    assert ctypes.sizeof(_lib.dsir_cloud_out) == 6 * 8


def test_create_fails_loudly_without_gpu_or_with_bad_cfg():
    import torch
    from deepsir_amd import _lib
    lib = _lib.load()
    cfg = _lib.dsir_cfg()
    h = ctypes.c_void_p()
    cfg.num_knn = 8   # unsupported
    assert lib.dsir_create(0, ctypes.byref(cfg), ctypes.byref(h)) != 0
    assert b"num_knn" in lib.dsir_last_error(None)
    if not torch.cuda.is_available():
        from deepsir_amd.arch import NetConfig
        from deepsir_amd.engine import Engine, EngineError
        with pytest.raises(EngineError):
            Engine(NetConfig(), 0)    # no device: must raise, never fall back to a CPU path


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "deepsir_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"


def test_fp16_operand_split_is_exact_to_2_pow_minus_22():
    """dsir_split_f16 (host, no GPU): the split the fp16-matrix-pipe layers rest on (csrc/agg_chain_h.hip, head_mlp_h.hip,
    pw_tile.hip, pw_stream.hip).  hi = fp16(x) and lo = fp16(x - hi), both round-to-nearest-even (numpy's conversion is the
    oracle here), and x = hi + lo + r with |r| <= max(2^-22 |x|, 2^-25) over the whole fp16 range, sub-normals included."""
    import numpy as np
    from deepsir_amd import _lib
    lib = _lib.load()
    rng = np.random.Generator(np.random.Philox(key=16))
    x = np.concatenate([rng.standard_normal(200000) * 10.0 ** rng.uniform(-9, 4.5, 200000),
                        [0.0, -0.0, 6.1e-5, -6.0e-5, 5.96e-8, 65504.0, -65504.0, 1.0, 1.0 + 2.0 ** -11, 1.0 + 2.0 ** -12]]).astype(np.float32)
    x = x[np.abs(x) <= 65504.0]
    hi = np.zeros(x.size, np.uint16)
    lo = np.zeros(x.size, np.uint16)
    lib.dsir_split_f16(x.ctypes.data, x.size, hi.ctypes.data, lo.ctypes.data)
    h = x.astype(np.float16)
    l = (x - h.astype(np.float32)).astype(np.float16)
    assert np.array_equal(hi, h.view(np.uint16)) and np.array_equal(lo, l.view(np.uint16))
    r = np.abs(x.astype(np.float64) - h.astype(np.float64) - l.astype(np.float64))
    assert (r <= np.maximum(2.0 ** -22 * np.abs(x.astype(np.float64)), 2.0 ** -25)).all()


def _strip_comments(src):
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return re.sub(r"//[^\n]*", "", src)


def test_environment_is_read_in_one_gated_function_only():
    """The DSIR_* measurement switches are environment variables read through ONE gate (csrc/engine.hip::tuning_env,
    include/dsir.h "The tuning gate"): no other getenv in the library, and the gate is closed unless DSIR_TUNING=1 /
    dsir_set_tuning(1)."""
    import subprocess
    import sys
    csrc = os.path.join(ROOT, "deepsir_amd", "csrc")
    sites = {}
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h", ".cpp")):
            n = len(re.findall(r"\bgetenv\s*\(", _strip_comments(open(os.path.join(csrc, f)).read())))
            if n:
                sites[f] = n
    assert sites == {"engine.hip": 2}, sites
    eng = _strip_comments(open(os.path.join(csrc, "engine.hip")).read())
    body = eng[eng.index("const char* dsir::tuning_env("):]
    body = body[:body.index("\n}\n") + 3]
    assert len(re.findall(r"\bgetenv\s*\(", body)) == 2, "both reads live in tuning_env()"
    code = ("from deepsir_amd import _lib; lib = _lib.load(); a = lib.dsir_tuning(); lib.dsir_set_tuning(1); b = lib.dsir_tuning(); "
            "lib.dsir_set_tuning(0); print(a, b, lib.dsir_tuning())")
    for env_val, want in ((None, "0 1 0"), ("1", "1 1 0"), ("yes", "0 1 0")):
        env = {k: v for k, v in os.environ.items() if k != "DSIR_TUNING"}
        if env_val is not None:
            env["DSIR_TUNING"] = env_val
        r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-1500:]
        assert r.stdout.split("\n")[-2].strip() == want, (env_val, r.stdout)


def test_groupnorm_statistics_meet_in_exact_atomics_only():
    """GroupNorm statistics meet across workgroups in ONE helper (csrc/device_utils.h, gn_block_commit): the point-wise GEMM family
    holds no atomic of its own, and the helper's only atomic adds integer-valued doubles (gn_stat_limb: rint(..) of both limbs) -
    sums that stay below 2^53 are exact in fp64, hence independent of the arrival order (the GPU stress test asserts the bits)."""
    csrc = os.path.join(ROOT, "deepsir_amd", "csrc")
    for f in ("pw_stream.hip", "pw_tile.hip", "pw_gemm.hip", "att_pool.hip"):
        assert "tomicAdd" not in _strip_comments(open(os.path.join(csrc, f)).read()), f
    util = _strip_comments(open(os.path.join(csrc, "device_utils.h")).read())
    helper = util[util.index("void gn_block_commit("):]
    helper = helper[:helper.index("\n}\n")]
    adds = re.findall(r"tomicAdd\(([^;]*)\);", helper)
    assert len(adds) == 1 and "gn_stat_limb(d, j)" in adds[0], adds
    limb = util[util.index("double gn_stat_limb("):]
    limb = limb[:limb.index("\n}\n")]
    assert limb.count("rint(") == 2 and "return limb ? rint(" in limb and "const double l1 = rint(" in limb          # both limbs are integers by construction
    assert "tomicAdd" not in util.replace(helper, "")                          # no other adding atomic among the shared helpers
    # the training path: the backward of the gathers sums through the index's inverse in a fixed order, the loss
    # terms are added in pair order - no adding atomic at all in these files
    for f in ("train_ops.hip", "align_loss.hip"):
        assert "tomicAdd" not in _strip_comments(open(os.path.join(csrc, f)).read()), f


def test_groupnorm_exactness_bound_limits_max_points():
    """The statistics' atomics are exact for at most 2^12 contributions per (cloud, group) statistic (csrc/device_utils.h: integer
    limbs |L0| <= 2^39, |L1| < 2^40, partial totals below 2^53).  The count is a function of the layer shapes alone - the launchers'
    virtual grids (csrc/kernels.h: pw_stream / pw_tile / lse_uv _gn_contributions) - and grows with the cloud: the largest is the
    level-0 k = 16 layers' n / 32 (one commit per 32 sixteen-row tiles).  dsir_create refuses a max_points whose largest layer
    would pass the bound (2^20 points, accepted until round 4, meant 32768 contributions)."""
    from deepsir_amd import _lib
    lib = _lib.load()
    cfg = _lib.dsir_cfg()
    cfg.feat_len, cfg.num_knn, cfg.num_layers, cfg.out_feat_dim, cfg.num_classes, cfg.max_pairs, cfg.pipeline = 3, 16, 4, 64, 19, 1, 0
    for i, (r, d) in enumerate(zip((4, 4, 4, 4), (16, 64, 128, 256))):
        cfg.sub_sampling_ratio[i], cfg.d_out[i] = r, d
    limit = lib.dsir_gn_contribution_limit()
    assert limit == 1 << 12
    # limbs: N contributions of |L0| <= 2^39 and |L1| < 2^40 stay below 2^53 for N <= 2^12 (with a factor 2 to spare on L1)
    assert limit * (1 << 40) < (1 << 53)
    for n, want in ((5000, 157), (65536, 2048), (1 << 17, 4096), (1 << 20, 32768)):
        assert lib.dsir_gn_contributions(ctypes.byref(cfg), n) == want == -(-n // 32)
    top = lib.dsir_max_points_limit(ctypes.byref(cfg))
    assert top == 1 << 17
    assert lib.dsir_gn_contributions(ctypes.byref(cfg), top) <= limit < lib.dsir_gn_contributions(ctypes.byref(cfg), top + 32)
    h = ctypes.c_void_p()
    cfg.max_points = top + 1
    assert lib.dsir_create(0, ctypes.byref(cfg), ctypes.byref(h)) != 0
    assert b"exactness bound" in lib.dsir_last_error(None)
    # the launchers use the same functions for their grids: the guard inside them names the same constant
    csrc = os.path.join(ROOT, "deepsir_amd", "csrc")
    assert "stream_blocks(a.M, gy, big)" in open(os.path.join(csrc, "pw_stream.hip")).read()
    assert "b.vgrid = lse_uv_gn_contributions(a.n, a.KH);" in open(os.path.join(csrc, "lse_uv.hip")).read()

"""The C ABI is usable without Python: examples/cabi_register.cpp (plain C++ host linking libdsir.so only) must give
the same transforms, bit for bit, as the Python host over the same library."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

BIN = os.path.join(ROOT, "examples", "cabi_register")


def test_example_is_built_and_links_only_the_c_abi():
    assert os.path.exists(BIN), "build it: python -c 'import __graft_entry__ as g; g.build()'"
    src = open(os.path.join(ROOT, "examples", "cabi_register.cpp")).read()
    incs = [l.strip() for l in src.splitlines() if l.strip().startswith("#include")]
    assert '#include "dsir.h"' in incs and not any("torch" in i or "Python" in i or "pybind" in i for i in incs)
    import re
    calls = set(re.findall(r"\b(dsir_[a-z_]+)\s*\(", src))
    assert {"dsir_create", "dsir_load_weight", "dsir_finalize_weights", "dsir_register", "dsir_sync", "dsir_destroy"} <= calls


@pytest.mark.gpu
def test_cpp_host_matches_python_host(tmp_path):
    import sys
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from export_cabi_inputs import write_pairs, write_weights
    from deepsir_amd.arch import NetConfig
    from deepsir_amd.engine import Engine
    from deepsir_amd.synth import make_batch
    from deepsir_amd.weights import generate_state_dict
    cfg = NetConfig()
    sd = generate_state_dict(cfg, 3)
    b = make_batch(2048, [31, 32], 3)
    w, p, o = (str(tmp_path / n) for n in ("weights.bin", "pairs.bin", "out.bin"))
    write_weights(w, sd)
    write_pairs(p, b["points_src"], b["points_ref"])
    r = subprocess.run([BIN, w, p, o, "5"], capture_output=True, text=True, timeout=300)   # a CHILD process, never an exec
    assert r.returncode == 0, r.stdout + r.stderr
    raw = np.fromfile(o, dtype=np.float32)
    T = raw[: 2 * 5 * 12].reshape(2, 5, 3, 4)
    inv = raw[2 * 5 * 12:].view(np.int32)
    eng = Engine(cfg, 0, max_points=2048, max_pairs=2)
    eng.load_state_dict(sd)
    out = eng.register(torch.from_numpy(b["points_src"]).cuda(), torch.from_numpy(b["points_ref"]).cuda(), 5)
    assert np.array_equal(T, out["transforms"].cpu().numpy())
    assert np.array_equal(inv, out["invalid"].cpu().numpy())
    eng.close()

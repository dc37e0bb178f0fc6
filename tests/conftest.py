import json
import os
import sys

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")   # before the GPU is touched: deepsir_amd/__init__.py

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    meta = json.loads(str(g["meta"])) if "meta" in g.files else {}
    return g, meta


@pytest.fixture(scope="session")
def golden_names():
    return ["stage_n1024_s1", "stage_n1024_s2_sep", "e2e_n2048_s3", "e2e_n2048_s4_sep", "e2e_n5000_s5",
            "e2e_n2048_s6_f4"]


_CASE_CACHE = {}


def build_case(name):
    """Re-create (cfg, state-dict, data dict with pyramids) of a golden case from its seeds."""
    if name in _CASE_CACHE:
        return _CASE_CACHE[name]
    from deepsir_amd.arch import NetConfig
    from deepsir_amd.synth import make_pair
    from deepsir_amd.weights import generate_state_dict
    from oracle.knn import add_pyramids

    g, m = load_golden(name)
    cfg = NetConfig(feat_len=m["feat_len"], pipeline=m.get("pipeline", "align"), num_sub=m.get("num_sub", -1))
    sd = generate_state_dict(cfg, m["wseed"], m.get("variant", "plain"))
    data = add_pyramids(make_pair(m["n"], m["seed"], m["feat_len"]), cfg.num_knn, cfg.sub_sampling_ratio)
    _CASE_CACHE[name] = (g, m, cfg, sd, data)
    return _CASE_CACHE[name]

"""The state-dict schema must equal the key/shape dump of the imported reference
(tests/golden/state_dict_keys.json, written by oracle/gen_golden.py)."""
import json
import os

import numpy as np

from conftest import GOLD
from deepsir_amd.arch import NetConfig, level_sizes, network_specs
from deepsir_amd.weights import generate_state_dict


def test_schema_matches_reference_dump():
    with open(os.path.join(GOLD, "state_dict_keys.json")) as f:
        ref = json.load(f)
    specs = network_specs(NetConfig(feat_len=ref["feat_len"]))
    assert [s.name for s in specs] == [k for k, _, _ in ref["keys"]]
    assert [list(s.shape) for s in specs] == [shp for _, shp, _ in ref["keys"]]
    assert len(specs) == 370


def test_generator_is_deterministic_and_complete():
    cfg = NetConfig(feat_len=3)
    a, b = generate_state_dict(cfg, 0), generate_state_dict(cfg, 0)
    c = generate_state_dict(cfg, 1)
    assert list(a) == [s.name for s in network_specs(cfg)]
    assert all(np.array_equal(a[k], b[k]) for k in a)
    assert any(not np.array_equal(a[k], c[k]) for k in a)
    n_float = sum(v.size for v in a.values() if v.dtype == np.float32)
    assert n_float == 2578412  # float entries of the reference state-dict (probe, gen_golden)


def test_level_sizes():
    assert level_sizes(5000, (4, 4, 4, 4)) == [5000, 1250, 312, 78, 19]
    assert level_sizes(2048, (4, 4, 4, 4)) == [2048, 512, 128, 32, 8]


def test_network_parameters_match_the_reference_list():
    """``optim.Adam(my_model.parameters(), lr)`` (reference train.py:323) must construct on the drop-in: trainable tensors are
    nn.Parameters under the reference's names, in its order, with its requires_grad flags (freeze_model / freeze_model_2,
    model.py:196-207) - captured from the imported reference by oracle/gen_golden.py for all three pipelines."""
    import types
    import torch
    from deepsir_amd.model import Network
    with open(os.path.join(GOLD, "state_dict_keys.json")) as f:
        ref = json.load(f)
    for pipe, want in ref["parameters"].items():
        args = types.SimpleNamespace(pipeline=pipe, feat_len=ref["feat_len"], num_sub=-1 if pipe == "align" else 512)
        net = Network(args)
        got = [[k, bool(v.requires_grad)] for k, v in net.named_parameters()]
        assert got == want, pipe
        assert len(list(net.parameters())) == len(want) == {"align": 340, "feat": 185, "label": 155}[pipe]
        opt = torch.optim.Adam(net.parameters(), lr=1e-3)           # "optimizer got an empty parameter list" before
        assert sum(len(g["params"]) for g in opt.param_groups) == len(want)
        # buffers = the BatchNorm running statistics only; parameters + buffers = the state-dict
        assert len(list(net.buffers())) + len(want) == len(net.state_dict())
    sd_keys = [k for k, _, _ in ref["keys"]]
    assert list(Network(types.SimpleNamespace(pipeline="align", feat_len=3, num_sub=-1)).state_dict().keys()) == sd_keys

"""State-dict schema of the registration network (names, shapes, dtypes).

The engine keeps the reference's checkpoint layout so that
``net.load_state_dict(torch.load(p)['state_dict'])`` (reference test.py:614)
works unchanged.  The tree below is derived from the constructors at
reference network/model.py:119-195 (``Network.__init__``) and
network/RandLANet.py:233-285 (``RandLA.__init__``), :58-107 (``MLP2D``),
:140-146 (``Att_pooling``), :160-171 (``Building_block``), :215-223
(``Dilated_res_block``), :34-55 (``MLP``).  It is written from the module
structure, not copied: ``tests/test_arch.py`` pins it against the key/shape
dump captured from the imported reference (tests/golden/state_dict_keys.json).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Sequence, Tuple

LABEL_WEIGHTS = (3, 1, 1, 3, 2, 0, 0, 0, 6, 5, 6, 4, 7, 7, 6, 8, 4, 9, 9)
"""Semantic score LUT, reference network/model.py:146-150 (not in the state-dict)."""


SEMANTIC_KITTI_POINTS_PER_CLASS = (55437630, 320797, 541736, 2578735, 3274484, 552662, 184064, 78858, 240942562, 17294618, 170599734,
                                   6369672, 230413074, 101130274, 476491114, 9833174, 129609852, 4506626, 1168181)
"""Training-set point counts of the 19 SemanticKITTI classes: the constants behind SemanticLoss's cross-entropy weights
(reference network/loss.py:906-909)."""


def semantic_class_weights():
    """SemanticLoss.get_class_weights('SemanticKITTI') (loss.py:896-912): 1 / (class frequency + 0.02) -> 19 floats."""
    total = float(sum(SEMANTIC_KITTI_POINTS_PER_CLASS))
    return [1.0 / (n / total + 0.02) for n in SEMANTIC_KITTI_POINTS_PER_CLASS]


@dataclass
class NetConfig:
    """The ``args`` fields the hot path reads (reference arguments.py:27-82,
    network/model.py:122-126, network/RandLANet.py:240-247)."""

    feat_len: int = 3
    num_knn: int = 16
    sub_sampling_ratio: Tuple[int, ...] = (4, 4, 4, 4)
    d_out: Tuple[int, ...] = (16, 64, 128, 256)
    out_feat_dim: int = 64
    num_classes: int = 19
    num_reg_iter: int = 5
    pipeline: str = "align"
    num_sub: int = -1
    use_ppf: bool = False

    @classmethod
    def from_args(cls, args) -> "NetConfig":
        g = lambda k, d: getattr(args, k, d)
        return cls(
            feat_len=int(g("feat_len", 3)),
            num_knn=int(g("num_knn", 16)),
            sub_sampling_ratio=tuple(int(x) for x in g("sub_sampling_ratio", (4, 4, 4, 4))),
            d_out=tuple(int(x) for x in g("d_out", (16, 64, 128, 256))),
            out_feat_dim=int(g("out_feat_dim", 64)),
            num_reg_iter=int(g("num_reg_iter", 5)),
            pipeline=str(g("pipeline", "align")),
            num_sub=int(g("num_sub", -1)),
            use_ppf=bool(g("use_ppf", False)),
        )


@dataclass
class ParamSpec:
    name: str
    shape: Tuple[int, ...]
    kind: str  # conv_w | conv1d_w | bias | gn_w | gn_b | bn_w | bn_b | bn_mean | bn_var | bn_count
    fan_in: int = 0


def _mlp2d(prefix: str, cin: int, cout: int) -> List[ParamSpec]:
    # RandLANet.py:58-107 : conv (1x1, bias) + GroupNorm (affine)
    return [
        ParamSpec(prefix + ".conv.weight", (cout, cin, 1, 1), "conv_w", cin),
        ParamSpec(prefix + ".conv.bias", (cout,), "bias", cin),
        ParamSpec(prefix + ".norm.weight", (cout,), "gn_w"),
        ParamSpec(prefix + ".norm.bias", (cout,), "gn_b"),
    ]


def _att_pooling(prefix: str, d_in: int, d_out: int) -> List[ParamSpec]:
    # RandLANet.py:140-146 : fc (no bias) + MLP2D
    return [ParamSpec(prefix + ".fc.weight", (d_in, d_in, 1, 1), "conv_w", d_in)] + _mlp2d(
        prefix + ".mlp", d_in, d_out
    )


def _mlp1d(prefix: str, channels: Sequence[int]) -> List[ParamSpec]:
    # RandLANet.py:34-55 : Conv1d [+ BatchNorm1d + LeakyReLU] ; Sequential indices 0,1,(2),3,4,(5),6...
    out: List[ParamSpec] = []
    n = len(channels)
    pos = 0
    for i in range(1, n):
        cin, cout = channels[i - 1], channels[i]
        out.append(ParamSpec(f"{prefix}.{pos}.weight", (cout, cin, 1), "conv1d_w", cin))
        out.append(ParamSpec(f"{prefix}.{pos}.bias", (cout,), "bias", cin))
        pos += 1
        if i < n - 1:
            out.append(ParamSpec(f"{prefix}.{pos}.weight", (cout,), "bn_w"))
            out.append(ParamSpec(f"{prefix}.{pos}.bias", (cout,), "bn_b"))
            out.append(ParamSpec(f"{prefix}.{pos}.running_mean", (cout,), "bn_mean"))
            out.append(ParamSpec(f"{prefix}.{pos}.running_var", (cout,), "bn_var"))
            out.append(ParamSpec(f"{prefix}.{pos}.num_batches_tracked", (), "bn_count"))
            pos += 2  # BN + activation slot
    return out


def randla_specs(prefix: str, feat_in: int, num_classes: int, cfg: NetConfig) -> List[ParamSpec]:
    """RandLANet.py:233-285."""
    out: List[ParamSpec] = []
    dim = 8
    out += _mlp2d(prefix + ".mlp_pre", feat_in, dim)
    for i, d in enumerate(cfg.d_out):
        p = f"{prefix}.dilated_res_blocks.{i}"
        out += _mlp2d(p + ".mlp1", dim, d // 2)
        out += _mlp2d(p + ".lfa.mlp1", 10, d // 2)
        out += _att_pooling(p + ".lfa.att_pooling_1", d, d // 2)
        out += _mlp2d(p + ".lfa.mlp2", d // 2, d // 2)
        out += _att_pooling(p + ".lfa.att_pooling_2", d, d)
        out += _mlp2d(p + ".mlp2", d, 2 * d)
        out += _mlp2d(p + ".mlp_skip", dim, 2 * d)
        dim = 2 * d
    d_mid = dim
    out += _mlp2d(prefix + ".mlp_mid", dim, d_mid)
    L = len(cfg.d_out)
    d_cur = d_mid
    for j in range(L):
        if j < L - 1:
            cin = d_cur + 2 * cfg.d_out[-j - 2]
            d_cur = 2 * cfg.d_out[-j - 2]
        else:
            cin = 4 * cfg.d_out[0]
            d_cur = 2 * cfg.d_out[0]
        out += _mlp2d(f"{prefix}.decoder_blocks.{j}", cin, d_cur)
    out.append(ParamSpec(prefix + ".mlp_out.weight", (cfg.out_feat_dim, d_cur, 1, 1), "conv_w", d_cur))
    out += _mlp1d(prefix + ".fc_label", [cfg.out_feat_dim, 64, 32, num_classes])
    return out


def network_specs(cfg: NetConfig) -> List[ParamSpec]:
    """Key order follows module registration order at model.py:133-193:
    feat_extractor, mlp_feat, mlp_att, mlp_proj, inlier_model."""
    D = cfg.out_feat_dim
    if cfg.pipeline not in PIPELINES:
        raise ValueError(f"pipeline must be one of {PIPELINES} (reference model.py:131)")
    out = randla_specs("feat_extractor", cfg.feat_len, cfg.num_classes, cfg)
    if cfg.pipeline != "label":          # model.py:135
        out += _mlp1d("mlp_feat", [D, D, 128, D])
        out += _mlp1d("mlp_att", [4, 32, 64, 128, 256, D])
        out += _mlp1d("mlp_proj", [D, D])
    if cfg.pipeline == "align":          # model.py:181-191
        out += randla_specs("inlier_model", 6, 1, cfg)
    return out


PIPELINES = ("align", "feat", "label")
"""args.pipeline values (reference model.py:131); the index is DSIR_PIPELINE_* of include/dsir.h."""


def level_sizes(n: int, ratios: Sequence[int]) -> List[int]:
    """n_l for l = 0..L (L+1 entries): prefix sub-sampling n_{l+1} = n_l // r_l
    (reference dataloader/data_base.py:166-168)."""
    out = [int(n)]
    for r in ratios:
        out.append(out[-1] // int(r))
    return out

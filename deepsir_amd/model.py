"""Drop-in for the reference's ``network.model.Network`` (align, feat and label pipelines: inference, and the optimisation
step of each through ``Network.train_step``).

Keeps the reference's host API verbatim (reference network/model.py:119-195,
:297-298, :520-607; test.py:609-614):

    net = Network(args); net.load_state_dict(torch.load(p)['state_dict'])
    net.to(device); net.eval()
    transforms, endpoints = net(data, (num_reg_iter, clip_weight))

but everything inside ``forward`` runs in libdsir.so (hand-written HIP for
gfx950) through the C ABI of include/dsir.h.  The module holds the
checkpoint tensors as buffers under the reference's key names so that
``state_dict()`` / ``load_state_dict()`` round-trip a reference checkpoint.
There is no CPU path: calling it without the built library or without a GPU
raises.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn as nn

from collections.abc import Sequence

from .arch import NetConfig, network_specs
from . import graph_replay_safe
from .engine import Engine, EngineError, EnginePool

_PYR_KEYS = ("xyz", "neigh_idx", "sub_idx", "interp_idx")


class _Node(nn.Module):
    """Name-space node of the checkpoint tree (no computation)."""


def _attach(root: nn.Module, dotted: str, tensor: torch.Tensor, trainable: bool, requires_grad: bool):
    """Trainable tensors are ``nn.Parameter``s under the reference's names and in its registration order, so that
    ``optim.Adam(my_model.parameters(), lr)`` (reference train.py:323) constructs and ``named_parameters()`` reads as the
    reference's; BatchNorm running statistics stay buffers."""
    parts = dotted.split(".")
    m = root
    for p in parts[:-1]:
        if p not in m._modules:
            m.add_module(p, _Node())
        m = m._modules[p]
    if trainable:
        m.register_parameter(parts[-1], nn.Parameter(tensor, requires_grad=requires_grad))
    else:
        m.register_buffer(parts[-1], tensor)


def _requires_grad(pipeline: str, name: str) -> bool:
    """freeze_model / freeze_model_2 (reference model.py:196-207): 'label' trains the extractor; 'feat' freezes it; 'align'
    also freezes the aggregation layers and trains the inlier model alone."""
    top = name.split(".", 1)[0]
    if pipeline == "label":
        return True
    if pipeline == "feat":
        return top != "feat_extractor"
    return top == "inlier_model"


class _LazyPredPairs(Sequence):
    """``endpoints['pred_pairs']``: per iteration an int32 CPU tensor [B, J, 2] of (src row, matched ref row) as the reference
    builds it (model.py:603-606).  The device-to-host copy - a device synchronisation - is deferred until an element is READ:
    a caller that only wants the poses never waits for it."""

    def __init__(self, idx_dev: torch.Tensor):
        self._idx, self._items = idx_dev, None          # [n_iter, B, J] int32 on the device

    def _materialise(self):
        if self._items is None:
            idx = self._idx.cpu()
            n, B, J = idx.shape
            ar = torch.arange(J, dtype=torch.int32)[None, :, None].expand(B, J, 1)
            self._items = [torch.cat([ar, idx[i][:, :, None]], dim=2) for i in range(n)]
            self._idx = None
        return self._items

    def __len__(self):
        return len(self._items) if self._items is not None else int(self._idx.shape[0])

    def __getitem__(self, i):
        return self._materialise()[i]


class _LazyFlag:
    """``endpoints['invalid_gradient']`` (model.py:61-64): truth value read from the device only when asked for."""

    def __init__(self, invalid_dev: torch.Tensor):
        self._t, self._v = invalid_dev, None

    def __bool__(self):
        if self._v is None:
            self._v = bool((self._t & 1).any().item())   # bit 0 = SVD failure
            self._t = None
        return self._v

    def __eq__(self, other):
        return bool(self) == other

    def __repr__(self):
        return repr(bool(self))


class Network(nn.Module):
    POOL_MIN_PAIRS = 64
    """Batches of that many pairs and more run on an ``EnginePool`` (two HIP streams): same bits, higher throughput."""
    SERVE_MAX_PAIRS = 8
    """Batches of up to that many pairs (the reference evaluates ONE, test.py:56) go through a ``PairServer``: the launch
    sequence replayed from a captured hipGraph on one of two engines in turn, no host synchronisation in ``forward`` - the
    results are ordered on torch's current stream, consecutive calls overlap until the caller reads a result."""

    def __init__(self, args):
        super().__init__()
        self.cfg = NetConfig.from_args(args)
        self.pipeline = self.cfg.pipeline
        if self.pipeline not in ("align", "feat", "label"):
            raise AssertionError("pipeline must be 'align', 'feat' or 'label' (reference model.py:131)")
        if self.pipeline == "align" and self.cfg.num_sub > 0:
            # the reference's inlier model runs on the FULL src pyramid (model.py:575): it cannot follow a top-k selection
            raise NotImplementedError("num_sub > 0 with pipeline='align' is not runnable in the reference either")
        self.num_sub, self.num_knn, self.d_out = self.cfg.num_sub, self.cfg.num_knn, self.cfg.out_feat_dim
        self.clip_weight_thresh = getattr(args, "clip_weight_thresh", 0.0)
        for spec in network_specs(self.cfg):
            dtype = torch.int64 if spec.kind == "bn_count" else torch.float32
            trainable = spec.kind not in ("bn_mean", "bn_var", "bn_count")
            _attach(self, spec.name, torch.zeros(spec.shape, dtype=dtype), trainable, _requires_grad(self.pipeline, spec.name))
        self._engine: Optional[Engine] = None
        self._pool: Optional[EnginePool] = None
        self._dirty = True
        self._pool_dirty = True
        self._max_points = 0
        self._max_pairs = 0
        self._pool_points = 0
        self._pool_pairs = 0
        self._seen_version = -1
        self._server = None
        self._server_dirty = True
        # the loss modules the reference's training loop calls (model.py:173-195; train.py:409, :421, :426): call contract restated
        # over the HIP loss operators (deepsir_amd/autograd.py)
        from .autograd import DetDesLoss, ScanAlignmentLoss, SemanticLoss
        if self.pipeline == "align":
            self.loss_align_fun = ScanAlignmentLoss(self, args)
        elif self.pipeline == "feat":
            self.loss_feat_fun = DetDesLoss(self, args)
        else:
            self.loss_label_fun = SemanticLoss(self, args)
        self._tstate = None               # trainers of the training-mode forward (their storage IS the module's parameters)
        self.dropout_masks = None         # test aid: {'fe_src', 'fe_ref', 'inlier'} keep flags instead of random Dropout draws

    # ---- checkpoint plumbing
    def load_state_dict(self, state_dict, strict: bool = True):
        r = super().load_state_dict(state_dict, strict=strict)
        self._dirty = self._pool_dirty = self._server_dirty = True
        self._trainer = self._frozen_trainers = self._stepper = self._stepper_key = None     # they hold the previous weights
        return r

    def _device_index(self) -> int:
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            raise EngineError("Network is on the CPU: this engine has no CPU path; call .to('cuda') / .cuda() first")
        return dev.index or 0

    def _check_weights_touched(self):
        """An optimiser (or any in-place write) that changed a parameter bumps its version counter: the engines reload."""
        v = sum(p._version for p in self.parameters()) + sum(b._version for b in self.buffers())
        if v != self._seen_version:
            self._dirty = self._pool_dirty = self._server_dirty = True
            self._seen_version = v

    def _ensure_pool(self, n_points: int, pairs: int) -> EnginePool:
        dev = self._device_index()
        self._check_weights_touched()
        if self._pool is None or n_points > self._pool_points or pairs > self._pool_pairs:
            if self._pool is not None:
                self._pool.close()
            self._pool_points = max(self._pool_points, n_points, 1024)
            self._pool_pairs = max(self._pool_pairs, pairs)
            self._pool = EnginePool(self.cfg, dev, self._pool_points, self._pool_pairs, streams=2)
            self._pool_dirty = True
        if self._pool_dirty:
            self._pool.load_state_dict({k: v for k, v in self.state_dict().items()})
            self._pool_dirty = False
        return self._pool

    def _ensure_server(self, n_points: int, n_iter: int):
        from .serve import PairServer
        dev = self._device_index()
        self._check_weights_touched()
        srv = getattr(self, "_server", None)
        if srv is None or n_points > srv.max_points or srv.n_iter != n_iter:
            if srv is not None:
                srv.close()
            self._server = srv = PairServer(self.cfg, {k: v for k, v in self.state_dict().items()}, dev, max(n_points, 1024),
                                            2 * self.SERVE_MAX_PAIRS, 2, n_iter, True)
            self._server_dirty = False
        if self._server_dirty:       # new weights (load_state_dict, an optimiser step): reloaded into the running engines
            srv.load_state_dict({k: v for k, v in self.state_dict().items()})
            self._server_dirty = False
        return srv

    def serve(self, max_points: int = 5000, max_in_flight: int = 8, engines: Optional[int] = None, n_iter: Optional[int] = None, want_aux: bool = True):
        """A ``deepsir_amd.serve.PairServer`` on this network's weights: K single-pair registrations in flight
        (the reference's batch-1 evaluation mode, test.py:56, fed ahead)."""
        from .serve import PairServer
        return PairServer(self.cfg, {k: v for k, v in self.state_dict().items()}, self._device_index(), max_points, max_in_flight,
                          engines, self.cfg.num_reg_iter if n_iter is None else n_iter, want_aux)

    def _ensure_engine(self, n_points: int, pairs: int) -> Engine:
        dev = self._device_index()
        self._check_weights_touched()
        if self._engine is None or n_points > self._max_points or pairs > self._max_pairs:
            if self._engine is not None:
                self._engine.close()
            self._max_points = max(self._max_points, n_points, 1024)
            self._max_pairs = max(self._max_pairs, pairs)
            self._engine = Engine(self.cfg, dev, self._max_points, self._max_pairs)
            self._dirty = True
        if self._dirty:
            self._engine.load_state_dict({k: v for k, v in self.state_dict().items()})
            self._dirty = False
        return self._engine

    # ---- forward = forward_pair for 'feat' / 'label' (model.py:173-179, :609-666)
    def _forward_pair(self, eng: Engine, src, ref, pyr):
        out = eng.forward_pair(src.float(), ref.float(), self.cfg.num_sub, pyramids=pyr)
        endpoints = {}
        for side in ("src", "ref"):
            o = out[side]
            endpoints[f"pt_{side}"] = o["xyz"].permute(0, 2, 1).contiguous()          # [B, 3, M]
            endpoints[f"feat_{side}"] = o["feat"].permute(0, 2, 1).contiguous()       # [B, C, M]
            endpoints[f"logits_{side}"] = o["logits"].permute(0, 2, 1).contiguous()   # [B, num_class, N]
            if self.pipeline != "label":
                endpoints[f"score_{side}"] = o["score"]                               # [B, M]
        return None, endpoints

    # ---- forward in TRAINING mode (train.py:379 my_model.train(); :401): outputs with a grad_fn that leads into the parameters
    def _training_state(self, dev: torch.device):
        """The pipeline's trainers (deepsir_amd/train.py) over THIS module's tensors: after the first training forward a trainable
        ``nn.Parameter`` (and every BatchNorm running statistic) is a view of its trainer's flat device buffer, so what
        ``optimizer.step()`` writes is what the next training forward computes with - no copies either way."""
        from . import train as T
        st = self._tstate
        named = dict(self.named_parameters())
        named.update(dict(self.named_buffers()))
        if st is not None:
            k0, v0 = next(iter(st["main"].params.items()))
            if st["dev"] == dev and named[k0].data_ptr() == v0.data_ptr():
                return st
        sd = self.state_dict()
        if self.pipeline == "align":
            main = T.RandlaTrainer(self.cfg, sd, "inlier_model", 6, 1, dev)
            frozen = (T.RandlaTrainer(self.cfg, sd, "feat_extractor", self.cfg.feat_len, self.cfg.num_classes, dev), T.AggregationTrainer(self.cfg, sd, dev))
        elif self.pipeline == "label":
            main, frozen = T.RandlaTrainer(self.cfg, sd, "feat_extractor", self.cfg.feat_len, self.cfg.num_classes, dev), ()
        else:
            main = T.AggregationTrainer(self.cfg, sd, dev)
            frozen = (T.RandlaTrainer(self.cfg, sd, "feat_extractor", self.cfg.feat_len, self.cfg.num_classes, dev),)
        with torch.no_grad():
            for tr in (main,) + tuple(frozen):
                for k, v in list(tr.params.items()) + list(tr.buffers.items()):
                    named[k].data = v.view(named[k].shape)
        eng = Engine(self.cfg, dev.index or 0, 1024, 1)          # the weight-free operators of the training forward (score, arg-min, Kabsch, loss)
        eng.load_state_dict({k: v for k, v in sd.items()})
        self._tstate = st = {"dev": dev, "main": main, "frozen": frozen, "engine": eng, "names": [k for k in main.params if named[k].requires_grad]}
        st["params"] = [named[k] for k in st["names"]]
        return st

    def _train_engine(self, st, n_points: int, pairs: int) -> Engine:
        eng = st["engine"]
        if n_points > eng.max_points or pairs > eng.max_pairs:
            eng.close()
            eng = Engine(self.cfg, st["dev"].index or 0, max(n_points, eng.max_points), max(pairs, eng.max_pairs))
            eng.load_state_dict({k: v for k, v in self.state_dict().items()})
            st["engine"] = eng
        return eng

    def _pyramids(self, eng: Engine, data, src, ref) -> dict:
        batch = {"points_src": src, "points_ref": ref}
        for s_, pts in (("src", src), ("ref", ref)):
            if all(f"points_{s_}_{k}" in data for k in _PYR_KEYS):
                pyr = [data[f"points_{s_}_xyz"].float()] + [data[f"points_{s_}_{k}"].to(torch.int32) for k in _PYR_KEYS[1:]]
            else:
                pyr = eng.knn_pyramid(pts)
            batch[f"{s_}_xyz"], batch[f"{s_}_neigh"], batch[f"{s_}_sub"], batch[f"{s_}_interp"] = [t.contiguous() for t in pyr]
        return batch

    def _draw_masks(self, shapes: Dict[str, tuple], dev) -> Optional[dict]:
        """Dropout(0.5) keep flags of one training forward (RandLANet.py:363-367): drawn from torch's global generator like
        nn.Dropout's, unless the test aid ``self.dropout_masks`` supplies them."""
        from . import train as T
        if self.dropout_masks is not None:
            return self.dropout_masks
        seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        return {k: T.dropout_keep_masks(seed + 7919 * i, shp, dev) for i, (k, shp) in enumerate(shapes.items())}

    def _forward_train(self, data: Dict[str, torch.Tensor], opt=None):
        from . import se3
        from . import train as T
        from .autograd import run_taped
        src, ref = data["points_src"].float().contiguous(), data["points_ref"].float().contiguous()
        if not src.is_cuda:
            raise EngineError("Network is on the CPU: this engine has no CPU path; call .to('cuda') / .cuda() first")
        dev = src.device
        B, J, _ = src.shape
        K = ref.shape[1]
        st = self._training_state(dev)
        eng = self._train_engine(st, max(J, K), B)
        main, frozen, params, names = st["main"], st["frozen"], st["params"], st["names"]
        batch = self._pyramids(eng, data, src, ref)
        self._dirty = self._pool_dirty = self._server_dirty = True      # running statistics move now, the weights at optimizer.step()

        def param_grads(tr):
            g = tr.flat_g.clone()              # ONE copy: autograd may keep what it is handed, the trainer's buffer is zeroed by the next forward
            out = []
            for k, p in zip(names, params):
                v = tr.grads[k]
                off = v.data_ptr() - tr.flat_g.data_ptr()
                out.append(g[off // 4: off // 4 + v.numel()].view(p.shape))
            return out

        if self.pipeline == "align":
            n_iter = int(opt[0]) if opt is not None else self.cfg.num_reg_iter
            fe, ag = frozen
            masks = self._draw_masks({"fe_src": (B, J, 64), "fe_ref": (B, K, 64), "inlier": (n_iter, B, J, 64)}, dev)
            box = {}

            def run():
                main.zero_grad()
                fw = T.forward_align_train(eng, main, fe, ag, batch, n_iter, masks)
                box["fw"] = fw

                def back(grads):
                    g = grads[0]
                    for it in range(n_iter):
                        main.backward(fw["tapes"][it], g[it].contiguous(), shared=fw["shared"])
                    main.backward_shared(fw["shared"])
                    return param_grads(main)
                return (fw["logits"],), back

            (logits,) = run_taped(params, run, 1)
            fw = box["fw"]
            transforms, cum = [], None
            for it in range(n_iter):                                   # se3_torch.concatenate(R_t, transforms[-1]), model.py:595
                cum = fw["T"][it] if cum is None else se3.concatenate(fw["T"][it], cum)
                transforms.append(cum.detach())
            invalid = fw["invalid"][0]
            for f_ in fw["invalid"][1:]:
                invalid = invalid | f_
            endpoints = {"pt_src": src[:, :, :3].contiguous(), "pt_ref": ref[:, :, :3].contiguous(),
                         "perm_matrices": [logits[i] for i in range(n_iter)], "pred_pairs": _LazyPredPairs(fw["idx"]),
                         "invalid_gradient": _LazyFlag(invalid), "pt_ref_new": fw["pt_ref_new"],
                         "_train": {"engine": eng, "idx": fw["idx"], "logits": logits}}
            return transforms, endpoints

        if self.pipeline == "label":
            masks = self._draw_masks({"fe_src": (B, J, 64), "fe_ref": (B, K, 64)}, dev)
            box = {}

            def run():
                main.zero_grad()
                outs, tapes = [], []
                for s_ in ("src", "ref"):
                    lg, tape = main.forward(batch[f"points_{s_}"], batch[f"{s_}_xyz"], batch[f"{s_}_neigh"], batch[f"{s_}_sub"], batch[f"{s_}_interp"],
                                            masks.get(f"fe_{s_}"))
                    outs.append(lg); tapes.append(tape)
                box["tapes"] = tapes

                def back(grads):
                    for tape, g, lg in zip(tapes, grads, outs):
                        if g is not None:
                            main.backward(tape, g.reshape(-1, lg.shape[-1]).contiguous())
                    return param_grads(main)
                return tuple(outs), back

            lg_s, lg_r = run_taped(params, run, 2)
            endpoints = {}
            for s_, lg, tape, pts in (("src", lg_s, box["tapes"][0], src), ("ref", lg_r, box["tapes"][1], ref)):
                endpoints[f"pt_{s_}"] = pts[:, :, :3].permute(0, 2, 1).contiguous()
                endpoints[f"feat_{s_}"] = torch.nn.functional.normalize(tape.misc["feat"].detach(), dim=2).permute(0, 2, 1).contiguous()
                endpoints[f"logits_{s_}"] = lg.permute(0, 2, 1)
            return None, endpoints

        # feat: the frozen extractor in training mode picks the key points, the aggregation layers are taped
        if self.cfg.num_sub <= 0:
            raise EngineError("pipeline='feat' trains on the top-num_sub key points: set args.num_sub > 0")
        masks = self._draw_masks({"fe_src": (B, J, 64), "fe_ref": (B, K, 64)}, dev)
        inp = T.feat_pipeline_inputs_train(eng, frozen[0], batch, self.cfg.num_sub, masks)

        def run():
            main.zero_grad()
            d_s, tape_s = main.forward(inp["xyz_src"], inp["feat_src"], inp["score_src"])
            d_r, tape_r = main.forward(inp["xyz_ref"], inp["feat_ref"], inp["score_ref"])

            def back(grads):
                if grads[0] is not None:
                    main.backward(tape_s, grads[0])
                if grads[1] is not None:
                    main.backward(tape_r, grads[1])
                return param_grads(main)
            return (d_s, d_r), back

        d_s, d_r = run_taped(params, run, 2)
        endpoints = {}
        for s_, d in (("src", d_s), ("ref", d_r)):
            endpoints[f"pt_{s_}"] = inp[f"xyz_{s_}"].permute(0, 2, 1).contiguous()
            endpoints[f"feat_{s_}"] = d.permute(0, 2, 1)
            endpoints[f"score_{s_}"] = inp[f"score_{s_}"]
        return None, endpoints

    # ---- forward = forward_align_4 for 'align'
    def forward(self, data: Dict[str, torch.Tensor], opt=None):
        """Evaluation mode (or under ``torch.no_grad()``): the inference engine.  Training mode with gradients enabled - the state the
        reference's loop calls it in (train.py:379, :401) - the training forward on the device, outputs carrying a grad_fn."""
        if self.training and torch.is_grad_enabled():
            return self._forward_train(data, opt)
        with torch.no_grad():
            return self._forward_eval(data, opt)

    def _forward_eval(self, data: Dict[str, torch.Tensor], opt=None):
        src, ref = data["points_src"], data["points_ref"]
        B, J, _ = src.shape
        K = ref.shape[1]
        have = all(f"points_{s}_{k}" in data for s in ("src", "ref") for k in _PYR_KEYS)
        pyr = {f"points_{s}_{k}": data[f"points_{s}_{k}"] for s in ("src", "ref") for k in _PYR_KEYS} if have else None
        if self.pipeline != "align":
            return self._forward_pair(self._ensure_engine(max(J, K), B), src, ref, pyr)
        num_reg_iter, _clip_weight = opt  # clip_weight is ignored by the reference too (model.py:581-582)
        # the served path replays captured hipGraphs: only where replay is known to be right in this process (deepsir_amd/__init__.py);
        # a caller that had touched the GPU before importing the package keeps the eager engine below - same bits, host-synchronised
        if B <= self.SERVE_MAX_PAIRS and pyr is None and src.is_cuda and graph_replay_safe():
            out = self._ensure_server(max(J, K), int(num_reg_iter)).submit_batch(src.float(), ref.float()).result(wait="stream")
            return self._align_outputs(out, src, ref, int(num_reg_iter))
        # large batches: two engines on two HIP streams (EnginePool) - same bits, the throughput configuration of bench.py
        eng = self._ensure_pool(max(J, K), B) if B >= self.POOL_MIN_PAIRS else self._ensure_engine(max(J, K), B)
        if pyr is not None and isinstance(eng, EnginePool):
            pyr = {k: (v if v.dtype != torch.int64 else v.to(torch.int32)) for k, v in pyr.items()}
        out = eng.register(src.float(), ref.float(), int(num_reg_iter), pyramids=pyr)
        return self._align_outputs(out, src, ref, int(num_reg_iter))

    @staticmethod
    def _align_outputs(out, src, ref, num_reg_iter: int):
        transforms: List[torch.Tensor] = [out["transforms"][:, i].contiguous() for i in range(num_reg_iter)]
        endpoints = {
            "pt_src": src[:, :, :3].contiguous(),
            "pt_ref": ref[:, :, :3].contiguous(),
            "perm_matrices": [out["logits"][i] for i in range(num_reg_iter)],
            "pred_pairs": _LazyPredPairs(out["idx"]),          # CPU tensors like the reference's, copied when first read
            "invalid_gradient": _LazyFlag(out["invalid"]),     # model.py:61-64; read from the device when tested
            "pt_ref_new": out["pt_ref_new"],
        }
        return transforms, endpoints

    # ---- one optimisation step of the pipeline: train.py:396-448 without autograd (deepsir_amd/train.py)
    def train_step(self, data: Dict[str, torch.Tensor], opt=None, lr: float = 1e-3, dropout_seed: Optional[int] = None,
                   thres_radius: float = 0.1, det_loss_weight: float = 1.0, loss_kwargs: Optional[dict] = None, dist=None,
                   frozen_mode: str = "train") -> dict:
        """What the reference's loop does per batch - ``my_model(train_data, opt)``, ``loss_*_fun``, ``loss.backward()``,
        ``optimizer.step()`` (train.py:396-448) - for this network's pipeline, on the device:
          align: trains ``inlier_model`` (the only sub-network ScanAlignmentLoss reaches; data: transform_gt [B,3,4] and,
                 for the confidence term, ``matches`` = per pair an int [n',2] array as the reference's data loader gives).
                 frozen_mode 'train' (default): the whole network in training mode as ``my_model.train()`` leaves it - the
                 frozen sub-networks' BatchNorm on batch statistics, their running statistics moving, Dropout on
                 (``train_step_align_full``); 'eval': the frozen half from ONE inference pass of the engine (faster; the
                 correspondences are those of the evaluation-mode network);
          label: trains ``feat_extractor`` through SemanticLoss (data: labels_src / labels_ref [B,N] in 0..19);
          feat:  trains ``mlp_feat`` / ``mlp_att`` / ``mlp_proj`` through DetDesLoss (data: transform_gt; needs num_sub > 0); frozen_mode
                 as for align: 'train' runs the frozen extractor in training mode, 'eval' takes the key points from the engine.
        The updated tensors are written back into this module's buffers (``state_dict()`` is the trained checkpoint) and
        serve the next ``forward``.  Adam state lives in the trainer kept on the module.  Returns the step's dict (loss ...)."""
        from . import train as T
        src, ref = data["points_src"].float(), data["points_ref"].float()
        B, J, _ = src.shape
        eng = self._ensure_engine(max(J, ref.shape[1]), B)
        dev = src.device
        sd = self.state_dict()
        if getattr(self, "_trainer", None) is None:
            if self.pipeline == "align":
                self._trainer = T.RandlaTrainer(self.cfg, sd, "inlier_model", 6, 1, dev)
            elif self.pipeline == "label":
                self._trainer = T.RandlaTrainer(self.cfg, sd, "feat_extractor", self.cfg.feat_len, self.cfg.num_classes, dev)
            else:
                self._trainer = T.AggregationTrainer(self.cfg, sd, dev)
        tr = self._trainer
        batch = {"points_src": src, "points_ref": ref}
        for s_, pts in (("src", src), ("ref", ref)):
            if all(f"points_{s_}_{k}" in data for k in _PYR_KEYS):
                pyr = [data[f"points_{s_}_xyz"].float()] + [data[f"points_{s_}_{k}"].to(torch.int32) for k in _PYR_KEYS[1:]]
            else:
                pyr = eng.knn_pyramid(pts)
            batch[f"{s_}_xyz"], batch[f"{s_}_neigh"], batch[f"{s_}_sub"], batch[f"{s_}_interp"] = [t.contiguous() for t in pyr]
        if self.pipeline == "align" and frozen_mode == "train":
            n_iter = int(opt[0]) if opt is not None else self.cfg.num_reg_iter
            if getattr(self, "_frozen_trainers", None) is None:
                self._frozen_trainers = (T.RandlaTrainer(self.cfg, sd, "feat_extractor", self.cfg.feat_len, self.cfg.num_classes, dev),
                                         T.AggregationTrainer(self.cfg, sd, dev))
            fe, ag = self._frozen_trainers
            masks = None
            if dropout_seed is not None:     # the inlier model's masks are those of the 'eval' variant (train.dropout_keep_masks)
                sd_ = int(dropout_seed)
                masks = {"inlier": T.dropout_keep_masks(sd_, (n_iter, B, J, 64), dev),
                         "fe_src": T.dropout_keep_masks(3 * sd_ + 1, (B, J, 64), dev),
                         "fe_ref": T.dropout_keep_masks(3 * sd_ + 2, (B, ref.shape[1], 64), dev)}
            fn = None
            if "matches" in data:
                fn = lambda idx: torch.from_numpy(T.find_correct_correspondence(data["matches"], idx, J)).to(dev)
            out = T.train_step_align_full(eng, tr, fe, ag, batch, data["transform_gt"].float().to(dev), n_iter, fn, lr, masks, loss_kwargs,
                                          dist=dist)
            out["loss"] = out["losses"]["total"]
        elif self.pipeline == "align":
            n_iter = int(opt[0]) if opt is not None else self.cfg.num_reg_iter
            res = eng.register(src, ref, n_iter)
            labels = None
            if "matches" in data:
                labels = torch.from_numpy(T.find_correct_correspondence(data["matches"], res["idx"], J)).to(dev)
            key = (id(eng), B, J, ref.shape[1], n_iter)
            if getattr(self, "_stepper_key", None) != key:                      # hipGraph-replayed halves, fixed batch geometry
                self._stepper, self._stepper_key = T.AlignTrainStep(eng, tr, B, J, ref.shape[1], n_iter), key
            out = self._stepper.step(batch, res, data["transform_gt"].float().to(dev), labels, lr, dropout_seed, loss_kwargs, dist=dist)
            out["loss"] = out["losses"]["total"]
        elif self.pipeline == "label":
            out = T.train_step_label(tr, batch, data["labels_src"].to(torch.int32).to(dev), data["labels_ref"].to(torch.int32).to(dev), lr,
                                     dropout_seed, dist=dist)
        else:
            if self.cfg.num_sub <= 0:
                raise EngineError("pipeline='feat' trains on the top-num_sub key points: set args.num_sub > 0")
            if frozen_mode == "train":      # the frozen extractor in training mode, as my_model.train() leaves it
                if getattr(self, "_frozen_trainers", None) is None:
                    self._frozen_trainers = (T.RandlaTrainer(self.cfg, sd, "feat_extractor", self.cfg.feat_len, self.cfg.num_classes, dev),)
                masks = None
                if dropout_seed is not None:
                    masks = {f"fe_{s_}": T.dropout_keep_masks(3 * int(dropout_seed) + o_, (B, n_, 64), dev)
                             for s_, n_, o_ in (("src", J, 1), ("ref", ref.shape[1], 2))}
                inp = T.feat_pipeline_inputs_train(eng, self._frozen_trainers[0], batch, self.cfg.num_sub, masks)
            else:
                inp = T.feat_pipeline_inputs(eng, batch, self.cfg.num_sub)
            out = T.train_step_feat(tr, inp, data["transform_gt"].float().to(dev), thres_radius, det_loss_weight, lr, dist=dist)
        # written back on skipped steps too: the parameters are then unchanged, but the BatchNorm running statistics moved in
        # the forward pass whatever optimizer.step() did afterwards (train.py:401 runs before :437-446)
        new = tr.state_dict()
        for ft in (getattr(self, "_frozen_trainers", None) or ()):          # frozen weights, moving running statistics
            new.update({k: v.detach().cpu().numpy().reshape(ft._shapes[k]) for k, v in ft.buffers.items()})
        with torch.no_grad():
            own = dict(self.named_buffers())
            own.update(dict(self.named_parameters()))
            for k, v in new.items():
                own[k].copy_(torch.from_numpy(np.ascontiguousarray(v)).to(own[k].device))
        self._dirty = True
        return out

    # ---- the optimiser's checkpoint entry (CheckPointManager saves optimizer.state_dict(), common/torch_utils.py:62-67)
    def _param_order(self) -> List[str]:
        """``my_model.parameters()`` order = the state-dict order without the buffers (train.py:323 hands it to Adam)."""
        return [sp.name for sp in network_specs(self.cfg) if sp.kind not in ("bn_mean", "bn_var", "bn_count")]

    def optimizer_state_dict(self, lr: float = 1e-3) -> dict:
        """What ``torch.optim.Adam(my_model.parameters(), lr).state_dict()`` would hold after the steps taken through
        ``train_step``: state entries (by parameter index) for the tensors that received gradients, one param group."""
        order = self._param_order()
        tr = getattr(self, "_trainer", None)
        named = tr.adam_state() if tr is not None and tr.step_count > 0 else {}
        state = {i: named[k] for i, k in enumerate(order) if k in named}
        group = {"lr": lr, "betas": (0.9, 0.999), "eps": 1e-8, "weight_decay": 0, "amsgrad": False, "maximize": False, "foreach": None,
                 "capturable": False, "differentiable": False, "fused": None, "params": list(range(len(order)))}
        return {"state": state, "param_groups": [group]}

    def load_optimizer_state_dict(self, sd: dict) -> None:
        """Resume the device optimiser from a reference checkpoint's 'optimizer' entry (call after the first ``train_step`` built
        the trainer, or after ``prepare_training``)."""
        tr = getattr(self, "_trainer", None)
        if tr is None:
            raise EngineError("no trainer yet: call prepare_training(device) or train_step first")
        order = self._param_order()
        tr.load_adam_state({order[int(i)]: st for i, st in sd["state"].items()})

    def prepare_training(self) -> None:
        """Builds the pipeline's trainer from the current weights (train_step does it lazily)."""
        from . import train as T
        dev = next(self.buffers()).device
        sd = self.state_dict()
        if self.pipeline == "align":
            self._trainer = T.RandlaTrainer(self.cfg, sd, "inlier_model", 6, 1, dev)
        elif self.pipeline == "label":
            self._trainer = T.RandlaTrainer(self.cfg, sd, "feat_extractor", self.cfg.feat_len, self.cfg.num_classes, dev)
        else:
            self._trainer = T.AggregationTrainer(self.cfg, sd, dev)

"""Training step of the inlier model on the device (the backward half of the `align` pipeline).

The reference trains with torch autograd (train.py:379-448): ``Network.forward`` -> ``ScanAlignmentLoss`` ->
``loss.backward()`` -> ``optim.Adam.step()``.  In ``forward_align_4`` the matching runs under ``no_grad`` and the src cloud is
moved by ``R_t.detach()`` (network/model.py:556-588), so the alignment loss reaches exactly one sub-network: the inlier
``RandLA`` (network/RandLANet.py:233-372), through its logits.  This module is that sub-network's training pass without
autograd: ``RandlaTrainer.forward`` runs the layers in the reference's module order and keeps what the backward needs,
``backward`` walks them in reverse from d loss / d logits (``Engine.align_loss_backward``) to every parameter, and
``adam_step`` applies ``torch.optim.Adam``'s update rule.  Every tensor operation is a HIP kernel behind the C ABI of
``include/dsir_train.h``; torch is used for device memory (allocation, views, copies into concatenation buffers) only.

Layout: point-major ``[clouds][points][C]`` fp32, indices int32 - as everywhere in this package.  Parameters and
gradients are keyed by the reference's state-dict names (deepsir_amd/arch.py).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .arch import NetConfig, level_sizes, network_specs, randla_specs, semantic_class_weights

K_NN = 16


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def find_correct_correspondence(matches: Sequence, idx: "np.ndarray | torch.Tensor", hash_seed: int) -> np.ndarray:
    """ScanAlignmentLoss.find_correct_correspondence with ``_hash`` (network/loss.py:280-294, :723-749; host work in the
    reference too): is the predicted pair (j, idx[j]) among the ground-truth matches of its cloud pair?
    matches: per pair an int array [n', 2] of (src, ref) indices; idx [n_iter][P][J] (``pred_pairs[i][..., 1]``);
    hash_seed: the reference passes points_src.shape[1] (:819).  -> float32 [n_iter][P][J] of 0 / 1 (the BCE targets)."""
    ix = idx.detach().cpu().numpy() if isinstance(idx, torch.Tensor) else np.asarray(idx)
    n_iter, P, J = ix.shape
    out = np.zeros((n_iter, P, J), np.float32)
    ar = np.arange(J, dtype=np.int64)
    for p in range(P):
        m = matches[p].detach().cpu().numpy() if isinstance(matches[p], torch.Tensor) else np.asarray(matches[p])
        keys = m[:, 0].astype(np.int64) + m[:, 1].astype(np.int64) * int(hash_seed)
        for i in range(n_iter):
            out[i, p] = np.isin(ar + ix[i, p].astype(np.int64) * int(hash_seed), keys)
    return out


class _Ops:
    """The C ABI of include/dsir_train.h on torch device tensors (memory only) and torch's current HIP stream."""

    def __init__(self, device: torch.device):
        if device.type != "cuda":
            raise RuntimeError("deepsir_amd.train needs a GPU: the training operators are HIP kernels, there is no CPU path")
        self.lib = _lib.load()
        self.device = device
        self._scratch: Optional[torch.Tensor] = None
        self._plans: dict = {}
        self.stream = 0
        self.begin()

    def begin(self) -> None:
        """Binds the operators to torch's CURRENT stream (called at the start of every forward / backward / optimiser step:
        looking the stream up per launch costs more host time than most of these kernels run)."""
        self.stream = torch.cuda.current_stream(self.device).cuda_stream

    def _ok(self, rc: int, what: str) -> None:
        if rc != 0:
            raise RuntimeError(f"{what} failed (hipError {rc})")

    def scratch(self, nbytes: int) -> torch.Tensor:
        if self._scratch is None or self._scratch.numel() * 4 < nbytes:
            self._scratch = torch.empty((nbytes + 3) // 4 + 1024, dtype=torch.float32, device=self.device)
        return self._scratch

    def empty(self, *shape, dtype=torch.float32) -> torch.Tensor:
        return torch.empty(*shape, dtype=dtype, device=self.device)

    # ---- 1x1 convolutions
    def conv(self, x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
        """x [rows][Cin] (contiguous), w [Cout][Cin...] -> [rows][Cout]."""
        rows, cin = x.shape[0], x.shape[1]
        cout = w.shape[0]
        y = self.empty(rows, cout)
        self._ok(self.lib.dsir_t_gemm(self.stream, _ptr(x), cin, _ptr(w), cin, 1, _ptr(bias), _ptr(y), cout, rows, cin, cout, 0.0),
                 "dsir_t_gemm")
        return y

    def conv_dx(self, dy: torch.Tensor, w: torch.Tensor, into: Optional[torch.Tensor] = None) -> torch.Tensor:
        """dX = dY W; `into` (contiguous [rows][Cin]): accumulate."""
        rows, cout = dy.shape
        cin = w.numel() // cout
        dx = into if into is not None else self.empty(rows, cin)
        self._ok(self.lib.dsir_t_gemm(self.stream, _ptr(dy), cout, _ptr(w), 1, cin, None, _ptr(dx), cin, rows, cout, cin,
                                      1.0 if into is not None else 0.0), "dsir_t_gemm (dX)")
        return dx

    def conv_dw(self, dy: torch.Tensor, x: torch.Tensor, dw: torch.Tensor, db: Optional[torch.Tensor]) -> None:
        rows, cout = dy.shape
        cin = x.shape[1]
        sc = self.scratch(self.lib.dsir_t_gemm_dw_scratch(rows, cout, cin))
        self._ok(self.lib.dsir_t_gemm_dw(self.stream, _ptr(dy), cout, _ptr(x), cin, rows, cout, cin, _ptr(dw), _ptr(db), _ptr(sc)),
                 "dsir_t_gemm_dw")

    # ---- normalisation
    def gn_fwd(self, y: torch.Tensor, clouds: int, groups: int, gamma, beta, act: bool):
        C_ = y.shape[1]
        M = y.shape[0] // clouds
        out = self.empty(y.shape[0], C_)
        stats = self.empty(clouds, groups, 2)
        sc = self.scratch(self.lib.dsir_t_gn_scratch(clouds, M, C_))
        self._ok(self.lib.dsir_t_gn_fwd(self.stream, _ptr(y), clouds, M, C_, groups, _ptr(gamma), _ptr(beta), int(act), _ptr(out),
                                        _ptr(stats), _ptr(sc)), "dsir_t_gn_fwd")
        return out, stats

    def gn_bwd(self, dout, y, stats, clouds, groups, gamma, beta, act, dgamma, dbeta) -> torch.Tensor:
        C_ = y.shape[1]
        M = y.shape[0] // clouds
        dy = self.empty(y.shape[0], C_)
        sc = self.scratch(self.lib.dsir_t_gn_scratch(clouds, M, C_))
        self._ok(self.lib.dsir_t_gn_bwd(self.stream, _ptr(dout), _ptr(y), _ptr(stats), clouds, M, C_, groups, _ptr(gamma), _ptr(beta),
                                        int(act), _ptr(dy), _ptr(dgamma), _ptr(dbeta), _ptr(sc)), "dsir_t_gn_bwd")
        return dy

    # ---- gathers
    def gather(self, x: torch.Tensor, idx: torch.Tensor, out: torch.Tensor, col_off: int) -> None:
        """x [clouds][n][C], idx [clouds][m] -> out[clouds * m][col_off : col_off + C] (out's row length = its ld)."""
        clouds, n, C_ = x.shape
        m = idx.shape[1]
        self._ok(self.lib.dsir_t_gather(self.stream, _ptr(x), n, C_, _ptr(idx), m, clouds, _ptr(out), out.shape[-1], col_off),
                 "dsir_t_gather")

    def plan(self, idx: torch.Tensor, n: int):
        """The inverse of a gather index idx [clouds][m] into [clouds][n] (include/dsir_train.h, dsir_t_scatter_plan): (order, offsets).
        Cached per index tensor for the duration of one step (``new_step`` drops the cache): the neighbour index of a level serves
        three scatter-adds per pass, and every registration iteration of an `align` step runs on the same pyramid.  The cache holds
        the index tensor itself, so its storage cannot be handed to another tensor while the entry lives."""
        clouds, m = idx.shape
        key = (idx.data_ptr(), clouds, m, n)
        hit = self._plans.get(key)
        if hit is not None:
            return hit[1], hit[2]
        order = torch.empty(clouds * m, dtype=torch.int32, device=self.device)
        offsets = torch.empty(clouds * n + 1, dtype=torch.int32, device=self.device)
        nb = int(self.lib.dsir_t_scatter_plan_scratch(clouds * m))
        scratch = torch.empty(nb, dtype=torch.uint8, device=self.device)
        self._ok(self.lib.dsir_t_scatter_plan(self.stream, _ptr(idx), m, clouds, n, _ptr(order), _ptr(offsets), _ptr(scratch)),
                 "dsir_t_scatter_plan")
        self._plans[key] = (idx, order, offsets)
        return order, offsets

    def new_step(self) -> None:
        """Start of an optimisation step: index tensors may have new contents, plans are rebuilt on demand."""
        self._plans = {}

    def scatter_add(self, dy: torch.Tensor, col_off: int, C_: int, idx: torch.Tensor, n: int) -> torch.Tensor:
        clouds, m = idx.shape
        order, offsets = self.plan(idx, n)
        dx = torch.empty(clouds, n, C_, dtype=torch.float32, device=self.device)
        self._ok(self.lib.dsir_t_scatter_add(self.stream, _ptr(dy), dy.shape[-1], col_off, _ptr(order), _ptr(offsets), clouds, _ptr(dx), n, C_),
                 "dsir_t_scatter_add")
        return dx

    def relpos(self, xyz: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
        clouds, n, k = idx.shape
        out = self.empty(clouds * n * k, 10)
        self._ok(self.lib.dsir_t_relpos(self.stream, _ptr(xyz), _ptr(idx), n, k, clouds, _ptr(out)), "dsir_t_relpos")
        return out

    def attpool_fwd(self, cat: torch.Tensor, scores: torch.Tensor, points: int) -> torch.Tensor:
        C_ = cat.shape[1]
        out = self.empty(points, C_)
        self._ok(self.lib.dsir_t_attpool_fwd(self.stream, _ptr(cat), _ptr(scores), points, K_NN, C_, _ptr(out)), "dsir_t_attpool_fwd")
        return out

    def attpool_bwd(self, dout, cat, att, points):
        C_ = cat.shape[1]
        dcat, ds = self.empty(cat.shape[0], C_), self.empty(cat.shape[0], C_)
        self._ok(self.lib.dsir_t_attpool_bwd(self.stream, _ptr(dout), _ptr(cat), _ptr(att), points, K_NN, C_, _ptr(dcat), _ptr(ds)),
                 "dsir_t_attpool_bwd")
        return dcat, ds

    def maxpool_fwd(self, x: torch.Tensor, pool: torch.Tensor):
        clouds, n, C_ = x.shape
        m, k = pool.shape[1], pool.shape[2]
        out = self.empty(clouds, m, C_)
        arg = self.empty(clouds, m, C_, dtype=torch.int32)
        self._ok(self.lib.dsir_t_maxpool_fwd(self.stream, _ptr(x), n, C_, _ptr(pool), m, k, clouds, _ptr(out), _ptr(arg)),
                 "dsir_t_maxpool_fwd")
        return out, arg

    def maxpool_bwd(self, dout: torch.Tensor, arg: torch.Tensor, pool: torch.Tensor, n: int) -> torch.Tensor:
        """pool [clouds][m][k]: the index the forward pooled with (its inverse orders the sum)."""
        clouds, m, C_ = arg.shape
        k = pool.shape[2]
        order, offsets = self.plan(pool.reshape(clouds, m * k), n)
        dx = torch.empty(clouds, n, C_, dtype=torch.float32, device=self.device)
        self._ok(self.lib.dsir_t_maxpool_bwd(self.stream, _ptr(dout), _ptr(arg), _ptr(order), _ptr(offsets), m, k, C_, clouds, _ptr(dx), n),
                 "dsir_t_maxpool_bwd")
        return dx

    def add_leaky_fwd(self, a, b):
        out = torch.empty_like(a)
        self._ok(self.lib.dsir_t_add_leaky_fwd(self.stream, _ptr(a), _ptr(b), a.numel(), _ptr(out)), "dsir_t_add_leaky_fwd")
        return out

    def add_leaky_bwd(self, dout, out):
        d = torch.empty_like(out)
        self._ok(self.lib.dsir_t_add_leaky_bwd(self.stream, _ptr(dout), _ptr(out), out.numel(), _ptr(d)), "dsir_t_add_leaky_bwd")
        return d

    def mul_mask(self, x, mask, scale):
        y = torch.empty_like(x)
        self._ok(self.lib.dsir_t_mul_mask(self.stream, _ptr(x), _ptr(mask), scale, x.numel(), _ptr(y)), "dsir_t_mul_mask")
        return y

    def axpy(self, a: float, x: torch.Tensor, y: torch.Tensor) -> None:
        self._ok(self.lib.dsir_t_axpy(self.stream, a, _ptr(x), x.numel(), _ptr(y)), "dsir_t_axpy")

    def weighted_ce(self, logits: torch.Tensor, labels: torch.Tensor, class_weights: torch.Tensor, grad_scale: float = 1.0):
        """SemanticLoss.compute_loss: logits [rows][C], labels [rows] int32 (0 = ignored, class = label - 1)
        -> (d loss / d logits [rows][C], device float64 [4] = {loss, sum of weights, correct, valid})."""
        rows, C_ = logits.shape
        d = self.empty(rows, C_)
        out = self.empty(4, dtype=torch.float64)
        sc = self.scratch(self.lib.dsir_t_weighted_ce_scratch(rows))
        self._ok(self.lib.dsir_t_weighted_ce(self.stream, _ptr(logits), _ptr(labels), _ptr(class_weights), rows, C_, grad_scale, _ptr(d),
                                             _ptr(out), _ptr(sc)), "dsir_t_weighted_ce")
        return d, out

    def l2norm_fwd(self, x: torch.Tensor):
        y, nrm = torch.empty_like(x), self.empty(x.shape[0])
        self._ok(self.lib.dsir_t_l2norm_fwd(self.stream, _ptr(x), x.shape[0], x.shape[1], _ptr(y), _ptr(nrm)), "dsir_t_l2norm_fwd")
        return y, nrm

    def l2norm_bwd(self, dy: torch.Tensor, y: torch.Tensor, nrm: torch.Tensor) -> torch.Tensor:
        dx = torch.empty_like(y)
        self._ok(self.lib.dsir_t_l2norm_bwd(self.stream, _ptr(dy), _ptr(y), _ptr(nrm), y.shape[0], y.shape[1], _ptr(dx)), "dsir_t_l2norm_bwd")
        return dx

    def det_des_loss(self, feat_ref, feat_src, pt_ref, pt_src, score_ref, transform_gt, thres_radius: float, det_loss_weight: float = 1.0):
        """DetDesLoss + its backward: feat_* [P][M][C], pt_* [P][M][3], score_ref [P][M], transform_gt [P][3][4]
        -> (device float64 [4] = {total, loss_feat, loss_det, accuracy}, d_feat_ref, d_feat_src)."""
        P, M, C_ = feat_ref.shape
        out = self.empty(4, dtype=torch.float64)
        d_ref, d_src = torch.empty_like(feat_ref), torch.empty_like(feat_src)
        sc = self.scratch(self.lib.dsir_t_det_des_loss_scratch(P, M))
        self._ok(self.lib.dsir_t_det_des_loss(self.stream, _ptr(feat_ref), _ptr(feat_src), _ptr(pt_ref), _ptr(pt_src), _ptr(score_ref),
                                              _ptr(transform_gt), P, M, C_, float(thres_radius), float(det_loss_weight), _ptr(out), _ptr(d_ref),
                                              _ptr(d_src), _ptr(sc)), "dsir_t_det_des_loss")
        return out, d_ref, d_src

    def topk(self, score: torch.Tensor, k: int):
        """score [clouds][n] -> (idx [clouds][k] int32, score [clouds][k]): torch.topk's selection (model.py:692)."""
        clouds, n = score.shape
        idx, out = self.empty(clouds, k, dtype=torch.int32), self.empty(clouds, k)
        sc = self.scratch(self.lib.dsir_t_topk_scratch(clouds, n))
        self._ok(self.lib.dsir_t_topk(self.stream, _ptr(score), clouds, n, k, _ptr(idx), _ptr(out), _ptr(sc)), "dsir_t_topk")
        return idx, out

    def sigmoid(self, x: torch.Tensor) -> torch.Tensor:
        y = torch.empty_like(x)
        self._ok(self.lib.dsir_t_sigmoid(self.stream, _ptr(x), x.numel(), _ptr(y)), "dsir_t_sigmoid")
        return y

    def acc(self, dst: Optional[torch.Tensor], src: torch.Tensor) -> torch.Tensor:
        """dst += src (dst None: a private copy of src)."""
        src = src.contiguous()
        if dst is None:
            return src.clone()
        self.axpy(1.0, src, dst)
        return dst

    def inlier_input(self, xyz_src, xyz_ref, idx, T: Optional[torch.Tensor]) -> torch.Tensor:
        """[P][J][6] = [T x_src ; x_ref[idx]]; T [P][..][3][4] view of one iteration (row stride taken from it) or None."""
        P, J, _ = xyz_src.shape
        out = self.empty(P, J, 6)
        self._ok(self.lib.dsir_t_inlier_input(self.stream, _ptr(xyz_src), _ptr(xyz_ref), _ptr(idx), _ptr(T), 0 if T is None else T.stride(0),
                                              P, J, xyz_ref.shape[1], _ptr(out)), "dsir_t_inlier_input")
        return out


class _ParamStore:
    """Parameters, gradients and Adam moments of a set of layers: ONE flat device buffer each (every tensor a 256-byte
    aligned view), so that zero_grad is one fill and the optimiser step ONE launch; BatchNorm running statistics apart."""

    def _build_store(self, specs, state_dict) -> None:
        host: Dict[str, torch.Tensor] = {}
        self.buffers: Dict[str, torch.Tensor] = {}
        self._shapes = {sp.name: sp.shape for sp in specs}
        offsets: Dict[str, int] = {}
        total = 0
        for spec in specs:
            if spec.kind == "bn_count":
                continue
            if spec.name not in state_dict:
                raise KeyError(f"state_dict lacks {spec.name}")
            v = state_dict[spec.name]
            t = (v.detach().cpu() if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v))).float()
            if tuple(t.shape) != tuple(spec.shape):
                raise ValueError(f"{spec.name}: shape {tuple(t.shape)} != {tuple(spec.shape)}")
            t = t.reshape(t.shape[0], -1).contiguous() if t.dim() > 1 else t.contiguous()
            if spec.kind in ("bn_mean", "bn_var"):
                self.buffers[spec.name] = t.to(self.device)
            else:
                host[spec.name] = t
                offsets[spec.name] = total
                total += (t.numel() + 63) // 64 * 64
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=self.device)
        self.flat_g, self.flat_m, self.flat_v = (torch.zeros_like(self.flat_p) for _ in range(3))
        view = lambda flat, k: flat[offsets[k]:offsets[k] + host[k].numel()].view(host[k].shape)
        self.params: Dict[str, torch.Tensor] = {k: view(self.flat_p, k) for k in host}
        self.grads: Dict[str, torch.Tensor] = {k: view(self.flat_g, k) for k in host}
        self._moments = {k: (view(self.flat_m, k), view(self.flat_v, k)) for k in host}
        for k, t in host.items():
            self.params[k].copy_(t)
        self.step_count = 0

    def adam_state(self) -> Dict[str, dict]:
        """The optimiser state in torch.optim.Adam's terms, per parameter name: {'step', 'exp_avg', 'exp_avg_sq'} (host tensors
        in the reference's shapes) - what ``optimizer.state_dict()['state']`` holds after the same steps (train.py:478)."""
        out = {}
        for k, (m, v) in self._moments.items():
            shape = self._shapes[k]
            out[k] = {"step": torch.tensor(float(self.step_count)), "exp_avg": m.detach().cpu().reshape(shape).clone(),
                      "exp_avg_sq": v.detach().cpu().reshape(shape).clone()}
        return out

    def load_adam_state(self, state: Dict[str, dict]) -> None:
        """Resume from torch.optim.Adam state entries keyed by parameter name (all entries must carry the same step)."""
        steps = set()
        for k, st in state.items():
            if k not in self._moments:
                continue
            m, v = self._moments[k]
            m.copy_(torch.as_tensor(st["exp_avg"]).reshape(m.shape))
            v.copy_(torch.as_tensor(st["exp_avg_sq"]).reshape(v.shape))
            steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError(f"parameters at different Adam steps {sorted(steps)}: one launch updates them all with one bias correction")
        if steps:
            self.step_count = steps.pop()

    def zero_grad(self) -> None:
        self.flat_g.zero_()
        self.ops.new_step()

    def grads_have_nan(self) -> bool:
        """train.py:437-441 ("Gradients include NaN values. Parameters will not be updated"): one kernel, one 4-byte read."""
        o = self.ops
        o.begin()
        flag = torch.empty(1, dtype=torch.int32, device=self.device)
        o._ok(o.lib.dsir_t_any_nan(o.stream, _ptr(self.flat_g), self.flat_g.numel(), _ptr(flag)), "dsir_t_any_nan")
        return bool(flag.item())

    def state_dict(self) -> Dict[str, np.ndarray]:
        """Parameters and BatchNorm running statistics in the reference's shapes (host)."""
        return {k: v.detach().cpu().numpy().reshape(self._shapes[k]) for k, v in list(self.params.items()) + list(self.buffers.items())}

    def grad_dict(self) -> Dict[str, np.ndarray]:
        return {k: v.detach().cpu().numpy().reshape(self._shapes[k]) for k, v in self.grads.items()}

    def adam_step(self, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8) -> None:
        """torch.optim.Adam.step (train.py:323, :446) over the flat buffer (the alignment padding stays 0)."""
        self.step_count += 1
        o = self.ops
        o.begin()
        o._ok(o.lib.dsir_t_adam(o.stream, _ptr(self.flat_p), _ptr(self.flat_g), _ptr(self.flat_m), _ptr(self.flat_v), self.flat_p.numel(),
                                lr, betas[0], betas[1], eps, self.step_count), "dsir_t_adam")

    # ---- MLP (RandLANet.py:34-55) in training mode: Conv1d + BatchNorm1d (batch statistics) + LeakyReLU, last layer bare
    def _mlp1d(self, saved: list, prefix: str, x: torch.Tensor, n_layers: int, update_running: bool = True) -> torch.Tensor:
        o = self.ops
        pos = 0
        for i in range(n_layers):
            w, bias = self.params[f"{prefix}.{pos}.weight"], self.params[f"{prefix}.{pos}.bias"]
            y = o.conv(x, w, bias)
            if i < n_layers - 1:
                g, be = self.params[f"{prefix}.{pos + 1}.weight"], self.params[f"{prefix}.{pos + 1}.bias"]
                out, stats = o.gn_fwd(y, 1, w.shape[0], g, be, True)
                if update_running:
                    o._ok(o.lib.dsir_t_bn_running(o.stream, _ptr(stats), w.shape[0], y.shape[0], 0.1,
                                                  _ptr(self.buffers[f"{prefix}.{pos + 1}.running_mean"]),
                                                  _ptr(self.buffers[f"{prefix}.{pos + 1}.running_var"])), "dsir_t_bn_running")
                saved.append((prefix, pos, x, y, stats))
                x = out
                pos += 3
            else:
                saved.append((prefix, pos, x, None, None))
                x = y
        return x

    def _mlp1d_bwd(self, saved: list, d: torch.Tensor, need_dx: bool = True) -> Optional[torch.Tensor]:
        """``saved``: the entries one _mlp1d call appended, walked backwards."""
        o = self.ops
        for k, (prefix, pos, xin, y, stats) in enumerate(reversed(saved)):
            w = self.params[f"{prefix}.{pos}.weight"]
            if y is not None:
                g, be = self.params[f"{prefix}.{pos + 1}.weight"], self.params[f"{prefix}.{pos + 1}.bias"]
                d = o.gn_bwd(d, y, stats, 1, w.shape[0], g, be, True, self.grads[f"{prefix}.{pos + 1}.weight"],
                             self.grads[f"{prefix}.{pos + 1}.bias"])
            o.conv_dw(d, xin, self.grads[f"{prefix}.{pos}.weight"], self.grads[f"{prefix}.{pos}.bias"])
            if k == len(saved) - 1 and not need_dx:
                return None
            d = o.conv_dx(d, w)
        return d


class _Layer:
    """What the backward of one conv (+ norm) needs."""
    __slots__ = ("name", "x", "y", "stats", "clouds", "groups", "act", "norm")


class RandlaTape:
    def __init__(self):
        self.layers: Dict[str, _Layer] = {}
        self.blocks: List[dict] = []
        self.misc: dict = {}


class RandlaTrainer(_ParamStore):
    """One ``RandLA`` (network/RandLANet.py:233-372) with its parameters, gradients and Adam state on the device.

    ``state_dict``: the reference's keys under ``prefix`` (others are ignored).  ``feat_in`` / ``num_classes``: 6 / 1 for the
    inlier model (network/model.py:181-191)."""

    def __init__(self, cfg: NetConfig, state_dict: Dict[str, "np.ndarray | torch.Tensor"], prefix: str = "inlier_model", feat_in: int = 6,
                 num_classes: int = 1, device: "str | torch.device" = "cuda:0"):
        self.cfg, self.prefix, self.feat_in, self.num_classes = cfg, prefix, feat_in, num_classes
        self.device = torch.device(device)
        self.ops = _Ops(self.device)
        self._specs = randla_specs(prefix, feat_in, num_classes, cfg)
        self._build_store(self._specs, state_dict)

    # ------------------------------------------------------------------ layers
    def _mlp2d(self, tape: RandlaTape, name: str, x: torch.Tensor, clouds: int, act: bool = True) -> torch.Tensor:
        """MLP2D (RandLANet.py:58-107): conv 1x1 + GroupNorm(4 | 8) [+ LeakyReLU]; x [clouds * M][Cin]."""
        w = self.params[name + ".conv.weight"]
        L = _Layer()
        L.name, L.x, L.clouds, L.act, L.norm = name, x, clouds, act, "gn"
        L.groups = 8 if w.shape[0] >= 64 else 4
        L.y = self.ops.conv(x, w, self.params[name + ".conv.bias"])
        out, L.stats = self.ops.gn_fwd(L.y, clouds, L.groups, self.params[name + ".norm.weight"], self.params[name + ".norm.bias"], act)
        tape.layers[name] = L
        return out

    def _mlp2d_bwd(self, tape: RandlaTape, name: str, dout: torch.Tensor, need_dx: bool = True) -> Optional[torch.Tensor]:
        L = tape.layers[name]
        w = self.params[name + ".conv.weight"]
        dy = self.ops.gn_bwd(dout, L.y, L.stats, L.clouds, L.groups, self.params[name + ".norm.weight"], self.params[name + ".norm.bias"],
                             L.act, self.grads[name + ".norm.weight"], self.grads[name + ".norm.bias"])
        self.ops.conv_dw(dy, L.x, self.grads[name + ".conv.weight"], self.grads[name + ".conv.bias"])
        return self.ops.conv_dx(dy, w) if need_dx else None

    def _att(self, tape: RandlaTape, name: str, cat: torch.Tensor, clouds: int, n: int) -> torch.Tensor:
        """Att_pooling (RandLANet.py:140-157); cat [clouds * n * 16][d]."""
        scores = self.ops.conv(cat, self.params[name + ".fc.weight"], None)
        pooled = self.ops.attpool_fwd(cat, scores, clouds * n)           # scores become the softmax in place
        tape.misc[name] = (cat, scores)
        return self._mlp2d(tape, name + ".mlp", pooled, clouds)

    def _att_bwd(self, tape: RandlaTape, name: str, dout: torch.Tensor, clouds: int, n: int) -> torch.Tensor:
        cat, att = tape.misc[name]
        dpooled = self._mlp2d_bwd(tape, name + ".mlp", dout)
        dcat, ds = self.ops.attpool_bwd(dpooled, cat, att, clouds * n)
        self.ops.conv_dw(ds, cat, self.grads[name + ".fc.weight"], None)
        return self.ops.conv_dx(ds, self.params[name + ".fc.weight"], into=dcat)

    def _res_block(self, tape: RandlaTape, p: str, feat: torch.Tensor, xyz: torch.Tensor, idx: torch.Tensor,
                   shared: Optional[dict] = None) -> torch.Tensor:
        """Dilated_res_block + Building_block (RandLANet.py:160-230); feat [clouds][n][Cin] -> [clouds][n][2 d].
        ``shared`` (see ``forward``): the position-encoding branch (lfa.mlp1 on the relative position code, lfa.mlp2 on top)
        computed by the first pass that carries this dict and re-used by the later ones."""
        clouds, n, cin = feat.shape
        d = self.params[p + ".mlp2.conv.weight"].shape[1]
        h = d // 2
        flat_idx = idx.reshape(clouds, n * K_NN)
        f = self._mlp2d(tape, p + ".mlp1", feat.reshape(clouds * n, cin), clouds)                    # [clouds n][h]
        sh = None if shared is None else shared.get(p)
        if sh is None:
            stape = tape if shared is None else RandlaTape()
            enc = self._mlp2d(stape, p + ".lfa.mlp1", self.ops.relpos(xyz, idx), clouds)            # [clouds n 16][h]
            enc2 = self._mlp2d(stape, p + ".lfa.mlp2", enc, clouds)
            if shared is not None:
                shared[p] = {"tape": stape, "enc": enc, "enc2": enc2, "denc": None, "denc2": None, "clouds": clouds}
        else:
            enc, enc2 = sh["enc"], sh["enc2"]
        cat1 = self.ops.empty(clouds * n * K_NN, d)
        self.ops.gather(f.reshape(clouds, n, h), flat_idx, cat1, 0)
        cat1[:, h:] = enc
        agg1 = self._att(tape, p + ".lfa.att_pooling_1", cat1, clouds, n)                           # [clouds n][h]
        cat2 = self.ops.empty(clouds * n * K_NN, d)
        self.ops.gather(agg1.reshape(clouds, n, h), flat_idx, cat2, 0)
        cat2[:, h:] = enc2
        agg2 = self._att(tape, p + ".lfa.att_pooling_2", cat2, clouds, n)                           # [clouds n][d]
        main = self._mlp2d(tape, p + ".mlp2", agg2, clouds, act=False)
        skip = self._mlp2d(tape, p + ".mlp_skip", feat.reshape(clouds * n, cin), clouds, act=False)
        out = self.ops.add_leaky_fwd(main, skip)
        tape.misc[p] = (out, flat_idx, clouds, n, h, d, cin)
        return out.reshape(clouds, n, 2 * d)

    def _res_block_bwd(self, tape: RandlaTape, p: str, dout: torch.Tensor, need_dx: bool = True,
                       shared: Optional[dict] = None) -> Optional[torch.Tensor]:
        out, flat_idx, clouds, n, h, d, cin = tape.misc[p]
        o = self.ops
        dsum = o.add_leaky_bwd(dout.reshape(clouds * n, 2 * d), out)
        dfeat = self._mlp2d_bwd(tape, p + ".mlp_skip", dsum, need_dx)
        dagg2 = self._mlp2d_bwd(tape, p + ".mlp2", dsum)
        dcat2 = self._att_bwd(tape, p + ".lfa.att_pooling_2", dagg2, clouds, n)
        dagg1 = o.scatter_add(dcat2, 0, h, flat_idx, n)                                               # [clouds][n][h]
        if shared is None:
            denc = self._mlp2d_bwd(tape, p + ".lfa.mlp2", dcat2[:, h:].contiguous())                  # w.r.t. enc
            dcat1 = self._att_bwd(tape, p + ".lfa.att_pooling_1", dagg1.reshape(clouds * n, h), clouds, n)
            df = o.scatter_add(dcat1, 0, h, flat_idx, n)
            o.axpy(1.0, dcat1[:, h:].contiguous(), denc)
            self._mlp2d_bwd(tape, p + ".lfa.mlp1", denc, need_dx=False)                               # its input is data
        else:
            # shared position-encoding branch: only the upstream gradients are collected here (d enc2 from the second pooling,
            # d enc from the first); the branch itself is walked back once, by backward_shared
            sh = shared[p]
            sh["denc2"] = o.acc(sh["denc2"], dcat2[:, h:])
            dcat1 = self._att_bwd(tape, p + ".lfa.att_pooling_1", dagg1.reshape(clouds * n, h), clouds, n)
            df = o.scatter_add(dcat1, 0, h, flat_idx, n)
            sh["denc"] = o.acc(sh["denc"], dcat1[:, h:])
        d1 = self._mlp2d_bwd(tape, p + ".mlp1", df.reshape(clouds * n, h), need_dx)
        if not need_dx:
            return None
        o.axpy(1.0, d1, dfeat)
        return dfeat.reshape(clouds, n, cin)

    def backward_shared(self, shared: dict) -> None:
        """The backward of the position-encoding branches ``shared`` holds, ONCE for all the passes that used them: the
        upstream gradients of those passes were summed by ``backward`` (every operator on the way back is linear in the
        incoming gradient - convolution, GroupNorm for fixed forward values, LeakyReLU - so the sum of the passes' gradients
        is the gradient of the summed upstream, in real arithmetic; the reference's autograd walks the branch once per
        registration iteration and adds the results)."""
        o = self.ops
        o.begin()
        for p, sh in shared.items():
            if p == "_pyr" or (sh["denc2"] is None and sh["denc"] is None):    # "_pyr": the cached pyramid slices of ``forward``
                continue
            denc = self._mlp2d_bwd(sh["tape"], p + ".lfa.mlp2", sh["denc2"])
            o.axpy(1.0, sh["denc"], denc)
            self._mlp2d_bwd(sh["tape"], p + ".lfa.mlp1", denc, need_dx=False)                         # its input is data
            sh["denc"] = sh["denc2"] = None

    # ------------------------------------------------------------------ the network
    def forward(self, features: torch.Tensor, xyz_multi: torch.Tensor, neigh_idx: torch.Tensor, sub_idx: torch.Tensor,
                interp_idx: torch.Tensor, dropout_mask: Optional[torch.Tensor] = None, update_running_stats: bool = True,
                shared: Optional[dict] = None):
        """RandLA.forward in TRAINING mode (RandLANet.py:311-372; train.py:379 ``my_model.train()``): GroupNorm as always,
        the two BatchNorm1d of ``fc_label`` on batch statistics (running statistics updated with momentum 0.1), Dropout(0.5)
        with ``dropout_mask`` ([clouds][N][64] uint8 keep flags; None = keep everything, i.e. dropout off).
        features [clouds][N][feat_in]; pyramids as ``Engine.knn_pyramid`` returns them (int32).
        shared: a dict the caller keeps across SEVERAL forward passes on the SAME pyramid with the SAME weights (the
        registration iterations of one `align` step, model.py:575: the inlier model always runs on the src pyramid): the
        position-encoding branch of every level - lfa.mlp1 on the relative position code and lfa.mlp2 on top, the two
        heaviest layers of a block (n x 16 rows) - depends on nothing else, so the first pass computes it and the later
        ones re-use its outputs; hand the same dict to every ``backward`` and finish with ``backward_shared``.
        -> logits [clouds][N][num_classes], tape."""
        o = self.ops
        o.begin()
        pf = self.prefix
        clouds, N, cin = features.shape
        L = len(self.cfg.d_out)
        n = level_sizes(N, self.cfg.sub_sampling_ratio)
        off = np.concatenate([[0], np.cumsum(n[:L])]).astype(int)
        soff = np.concatenate([[0], np.cumsum(n[1:L + 1])]).astype(int)
        tape = RandlaTape()
        # the per-level slices of the pyramid (strided views -> contiguous copies) are the same in every pass that shares ``shared``
        # (the registration iterations of one step run on ONE pyramid): cut once, 16 copy kernels per pass less
        pyr = None if shared is None else shared.get("_pyr")
        if pyr is None:
            pyr = {"xyz": [xyz_multi[:, off[l]:off[l + 1]].contiguous() for l in range(L)],
                   "neigh": [neigh_idx[:, off[l]:off[l + 1]].contiguous() for l in range(L)],
                   "sub": [sub_idx[:, soff[l]:soff[l + 1]].contiguous() for l in range(L)],
                   "interp": [interp_idx[:, off[l]:off[l + 1], 0].contiguous() for l in range(L)]}
            if shared is not None:
                shared["_pyr"] = pyr
        x = self._mlp2d(tape, pf + ".mlp_pre", features.reshape(clouds * N, cin).contiguous(), clouds).reshape(clouds, N, -1)
        skips: List[torch.Tensor] = []
        args: List[torch.Tensor] = []
        for l in range(L):
            a, b = off[l], off[l + 1]
            enc = self._res_block(tape, f"{pf}.dilated_res_blocks.{l}", x, pyr["xyz"][l], pyr["neigh"][l], shared)
            x, arg = o.maxpool_fwd(enc, pyr["sub"][l])                                                 # random_sample (:374-391)
            args.append(arg)
            if l == 0:
                skips.append(enc)
            skips.append(x)
        m = n[L]
        x = self._mlp2d(tape, pf + ".mlp_mid", skips[-1].reshape(clouds * m, -1), clouds).reshape(clouds, m, -1)
        dec = []
        for j in range(L):
            a, b = off[L - j - 1], off[L - j]
            ii = pyr["interp"][L - j - 1]
            sk = skips[-j - 2]
            cs, cu = sk.shape[2], x.shape[2]
            cat = o.empty(clouds * (b - a), cs + cu)
            cat[:, :cs] = sk.reshape(clouds * (b - a), cs)
            o.gather(x, ii, cat, cs)                                                                    # nearest_interpolation (:393-408)
            dec.append((ii, cs, cu, x.shape[1]))
            x = self._mlp2d(tape, f"{pf}.decoder_blocks.{j}", cat, clouds).reshape(clouds, b - a, -1)
        xf = x.reshape(clouds * N, -1)
        feat = o.conv(xf, self.params[pf + ".mlp_out.weight"], None)
        h = o.mul_mask(feat, dropout_mask.reshape(-1).contiguous(), 2.0) if dropout_mask is not None else feat
        # fc_label: Conv1d + BatchNorm1d (batch statistics) + LeakyReLU, twice, then Conv1d (RandLANet.py:34-55, :272-273)
        fc = []
        pos = 0
        for i in range(3):
            w, bias = self.params[f"{pf}.fc_label.{pos}.weight"], self.params[f"{pf}.fc_label.{pos}.bias"]
            y = o.conv(h, w, bias)
            if i < 2:
                g, be = self.params[f"{pf}.fc_label.{pos + 1}.weight"], self.params[f"{pf}.fc_label.{pos + 1}.bias"]
                out, stats = o.gn_fwd(y, 1, w.shape[0], g, be, True)
                if update_running_stats:
                    o._ok(o.lib.dsir_t_bn_running(o.stream, _ptr(stats), w.shape[0], y.shape[0], 0.1,
                                                  _ptr(self.buffers[f"{pf}.fc_label.{pos + 1}.running_mean"]),
                                                  _ptr(self.buffers[f"{pf}.fc_label.{pos + 1}.running_var"])), "dsir_t_bn_running")
                fc.append((pos, h, y, stats))
                h = out
                pos += 3
            else:
                fc.append((pos, h, None, None))
                h = y
        tape.misc["net"] = dict(clouds=clouds, N=N, n=n, skips_shapes=[s.shape for s in skips], args=args, pools=pyr["sub"], dec=dec, xf=xf, fc=fc,
                                mask=dropout_mask, L=L)
        tape.misc["feat"] = feat.reshape(clouds, N, -1)            # RandLA.forward's first output (before the dropout)
        return h.reshape(clouds, N, self.num_classes), tape

    def backward(self, tape: RandlaTape, dlogits: torch.Tensor, dfeat: Optional[torch.Tensor] = None, shared: Optional[dict] = None) -> None:
        """Accumulates d loss / d parameter into ``self.grads`` (call ``zero_grad`` between steps, not between the
        registration iterations of one step: their gradients add up, as autograd's do).  dlogits [clouds][N][num_classes];
        dfeat [clouds][N][out_feat_dim] (optional): gradient w.r.t. the feature output ``tape.misc['feat']``."""
        o = self.ops
        o.begin()
        pf = self.prefix
        net = tape.misc["net"]
        clouds, N, n, L = net["clouds"], net["N"], net["n"], net["L"]
        d = dlogits.reshape(clouds * N, self.num_classes).contiguous().float()
        for pos, hin, y, stats in reversed(net["fc"]):
            w = self.params[f"{pf}.fc_label.{pos}.weight"]
            if y is not None:
                g, be = self.params[f"{pf}.fc_label.{pos + 1}.weight"], self.params[f"{pf}.fc_label.{pos + 1}.bias"]
                d = o.gn_bwd(d, y, stats, 1, w.shape[0], g, be, True, self.grads[f"{pf}.fc_label.{pos + 1}.weight"],
                             self.grads[f"{pf}.fc_label.{pos + 1}.bias"])
            o.conv_dw(d, hin, self.grads[f"{pf}.fc_label.{pos}.weight"], self.grads[f"{pf}.fc_label.{pos}.bias"])
            d = o.conv_dx(d, w)
        if net["mask"] is not None:
            d = o.mul_mask(d, net["mask"].reshape(-1).contiguous(), 2.0)
        if dfeat is not None:
            o.axpy(1.0, dfeat.reshape(d.shape).contiguous().float(), d)
        o.conv_dw(d, net["xf"], self.grads[pf + ".mlp_out.weight"], None)
        dx = o.conv_dx(d, self.params[pf + ".mlp_out.weight"])                                        # [clouds N][C]
        shapes = net["skips_shapes"]
        dskips: List[Optional[torch.Tensor]] = [None] * len(shapes)
        for j in reversed(range(L)):
            ii, cs, cu, n_prev = net["dec"][j]
            dcat = self._mlp2d_bwd(tape, f"{pf}.decoder_blocks.{j}", dx)
            k = len(shapes) - j - 2
            dskips[k] = o.acc(dskips[k], dcat[:, :cs]).reshape(shapes[k])
            dx = o.scatter_add(dcat, cs, cu, ii, n_prev).reshape(clouds * n_prev, cu)
        dmid = self._mlp2d_bwd(tape, pf + ".mlp_mid", dx).reshape(shapes[-1])
        dskips[-1] = o.acc(dskips[-1], dmid).reshape(shapes[-1])
        for l in reversed(range(L)):
            denc = o.maxpool_bwd(dskips[l + 1].contiguous(), net["args"][l], net["pools"][l], n[l])  # [clouds][n_l][2 d_l]
            if l == 0 and dskips[0] is not None:
                o.axpy(1.0, dskips[0].contiguous(), denc)
            dfeat = self._res_block_bwd(tape, f"{pf}.dilated_res_blocks.{l}", denc, need_dx=True, shared=shared)
            if l > 0:
                dskips[l] = o.acc(dskips[l], dfeat).reshape(shapes[l])
            else:
                self._mlp2d_bwd(tape, pf + ".mlp_pre", dfeat.reshape(clouds * N, -1), need_dx=False)


def dropout_keep_masks(seed: Optional[int], shape, device) -> Optional[torch.Tensor]:
    """Dropout(0.5) keep flags (RandLANet.py:363-367) of one optimisation step, uint8 ``shape`` on ``device``: ONE draw from a
    device generator seeded with ``seed``.  Every entry point that takes ``dropout_seed`` (``train_step_align``,
    ``AlignTrainStep.step``, ``train_step_label``, ``Network.train_step``) draws through this function, so the same seed gives
    the same masks - hence the same gradients up to the order of the fp32 atomics - on the eager and the hipGraph path.
    None: dropout off."""
    if seed is None:
        return None
    g = torch.Generator(device=device).manual_seed(int(seed))
    return (torch.rand(tuple(shape), generator=g, device=device) >= 0.5).to(torch.uint8)


def any_pose_invalid(flags, dist=None) -> bool:
    """The reference skips ``optimizer.step()`` when a gradient is NaN OR ``endpoints['invalid_gradient']`` is set (a
    degenerate weighted-Kabsch covariance, model.py:61-64; train.py:437-446).  flags: the ``invalid`` outputs of
    ``Engine.kabsch`` / ``Engine.register`` (bit 0 = degenerate pose); with ``dist`` the verdict is shared by all ranks
    (they must take the same decision, or their parameters diverge)."""
    bad = False
    for f in flags:
        if f is not None and bool((f.to(torch.int32) & 1).any()):
            bad = True
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dev = next((f.device for f in flags if f is not None), torch.device("cpu"))
        if dist.get_backend() != "nccl":
            dev = torch.device("cpu")
        t = torch.tensor([1 if bad else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        bad = bool(t.item())
    return bad


def forward_align_train(engine, inlier: RandlaTrainer, extractor: RandlaTrainer, aggregation: "AggregationTrainer", batch: dict,
                        n_iter: int, masks: Optional[dict] = None) -> dict:
    """``forward_align_4`` (network/model.py:520-607) as the reference's training loop runs it: ``my_model.train()`` (train.py:379)
    puts EVERY BatchNorm on batch statistics (and keeps updating its running statistics) and every Dropout on - also in the
    sub-networks the `align` pipeline freezes.  So the frozen half is run here in training mode too: the feature extractor by
    ``extractor.forward`` (src, then ref), the key-point score by the engine's score operator, ``aggregation.forward`` for
    src and ref in EVERY iteration (model.py:552), the arg-min and the weighted Kabsch step by the engine's operators, the
    inlier model by ``inlier.forward`` with its tape kept.  masks: {'fe_src', 'fe_ref': [P][N][64] uint8, 'inlier': [n_iter][P][N][64]}
    Dropout keep flags (None: dropout off).
    -> idx [n_iter][P][J] i32, logits [n_iter][P][J], tapes, xyz_src / xyz_ref [P][.][3] (for the loss)."""
    o = inlier.ops
    masks = masks or {}
    side = {}
    for s_ in ("src", "ref"):
        pts = batch[f"points_{s_}"].contiguous()
        logits, tape = extractor.forward(pts, batch[f"{s_}_xyz"], batch[f"{s_}_neigh"], batch[f"{s_}_sub"], batch[f"{s_}_interp"],
                                         masks.get(f"fe_{s_}"))
        feat = tape.misc["feat"].contiguous()
        score, _label = engine.score(feat, logits.contiguous(), batch[f"{s_}_xyz"], batch[f"{s_}_neigh"])
        side[s_] = (pts[:, :, :3].contiguous(), feat, score)
    xyz0, f_s, sc_s = side["src"]
    xyz_r, f_r, sc_r = side["ref"]
    P, J, _ = xyz0.shape
    idxs, logits, tapes, invalid, Ts = [], [], [], [], []
    pt_ref_new = None
    shared: dict = {}                       # the inlier model's position-encoding branch: once per step (RandlaTrainer.forward)
    cur = xyz0
    for it in range(n_iter):
        d_s, _ = aggregation.forward(cur, f_s, sc_s, second_normalize=False)
        d_r, _ = aggregation.forward(xyz_r, f_r, sc_r, second_normalize=False)
        idx = engine.nn_match(d_s.contiguous(), d_r.contiguous())
        cat = o.inlier_input(cur, xyz_r, idx, None)
        m = masks.get("inlier")
        lg, tape = inlier.forward(cat, batch["src_xyz"], batch["src_neigh"], batch["src_sub"], batch["src_interp"], None if m is None else m[it],
                                  shared=shared)
        lg = lg.reshape(P, J)
        T_it, bad_it = engine.kabsch(cur, cat[:, :, 3:].contiguous(), o.sigmoid(lg.contiguous()))
        invalid.append(bad_it)
        Ts.append(T_it)
        pt_ref_new = cat[:, :, 3:].contiguous()
        cur = o.inlier_input(cur, xyz_r, idx, T_it)[:, :, :3].contiguous()        # xyz_src <- R_t.detach() xyz_src (model.py:587)
        idxs.append(idx); logits.append(lg); tapes.append(tape)
    return {"idx": torch.stack(idxs).contiguous(), "logits": torch.stack(logits).contiguous(), "tapes": tapes, "xyz_src": xyz0,
            "xyz_ref": xyz_r, "invalid": invalid, "shared": shared, "T": Ts, "pt_ref_new": pt_ref_new}


def train_step_align_full(engine, inlier: RandlaTrainer, extractor: RandlaTrainer, aggregation: "AggregationTrainer", batch: dict,
                          transform_gt, n_iter: int, labels_fn=None, lr: float = 1e-3, masks: Optional[dict] = None,
                          loss_kwargs: Optional[dict] = None, apply: bool = True, dist=None) -> dict:
    """One optimisation step of the `align` pipeline with the WHOLE network in training mode, as train.py:379-448 runs it
    (``forward_align_train``): only the inlier model receives gradients and is updated (model.py:136, :182), but the frozen
    sub-networks' BatchNorm running statistics move, as they do in the reference.  labels_fn(idx) -> [n_iter][P][J] float 0/1
    (``find_correct_correspondence``) or None: no confidence term."""
    inlier.zero_grad()
    fw = forward_align_train(engine, inlier, extractor, aggregation, batch, n_iter, masks)
    labels = None if labels_fn is None else labels_fn(fw["idx"])
    out = engine.align_loss_backward(fw["xyz_src"], fw["xyz_ref"], fw["idx"], fw["logits"], labels, transform_gt, **(loss_kwargs or {}))
    for it in range(n_iter):
        inlier.backward(fw["tapes"][it], out["grad_logits"][it], shared=fw["shared"])
    inlier.backward_shared(fw["shared"])
    all_reduce_gradients(inlier, dist)
    bad = inlier.grads_have_nan() or any_pose_invalid(fw["invalid"], dist)      # train.py:437-446: NaN gradient OR invalid_gradient
    if apply and not bad:
        inlier.adam_step(lr)
    out.update(logits=fw["logits"], idx=fw["idx"], skipped=bad)
    return out


def train_step_align(engine, trainer: RandlaTrainer, batch: dict, result: dict, transform_gt: np.ndarray,
                     labels: Optional[np.ndarray] = None, lr: float = 1e-3, dropout_seed: Optional[int] = None,
                     loss_kwargs: Optional[dict] = None, apply: bool = True, dist=None) -> dict:
    """One optimisation step of the `align` pipeline on the inlier model (train.py:396-448).

    ``batch``: the device tensors of one ``Engine.register`` call (points_src [P][N][C], the src pyramid
    src_xyz / src_neigh / src_sub / src_interp, points_ref); ``result``: what that call returned (idx [n_iter][P][N],
    transforms [P][n_iter][3][4]) - the no_grad half of ``forward_align_4`` (matching, ``R_t.detach()``), taken from the
    inference engine.  Per iteration the inlier model's input cat(xyz_src_i, xyz_ref[idx_i]) (model.py:571-573) is rebuilt
    from them, run forward in training mode, the alignment loss and its gradient w.r.t. the logits come from
    ``Engine.align_loss_backward``, and the gradients of all iterations are accumulated before ONE Adam step - skipped,
    like the reference's (train.py:437-446), when a gradient is NaN or a pose was degenerate."""
    dev = trainer.device
    xyz_s = batch["points_src"][:, :, :3].contiguous()
    xyz_r = batch["points_ref"][:, :, :3].contiguous()
    idx = result["idx"]
    T = result["transforms"]
    n_iter, P, N = idx.shape
    trainer.zero_grad()
    logits, tapes = [], []
    masks = dropout_keep_masks(dropout_seed, (n_iter, P, N, trainer.cfg.out_feat_dim), dev)
    shared: dict = {}                       # position-encoding branch of the inlier model: once per step (RandlaTrainer.forward)
    trainer.ops.begin()
    for it in range(n_iter):
        # the src cloud moved by the previous cumulative pose (model.py:587; R_t.detach()) next to its correspondences
        cat = trainer.ops.inlier_input(xyz_s, xyz_r, idx[it], None if it == 0 else T[:, it - 1])
        mask = None if masks is None else masks[it]
        lg, tape = trainer.forward(cat, batch["src_xyz"], batch["src_neigh"], batch["src_sub"], batch["src_interp"], mask, shared=shared)
        logits.append(lg.reshape(P, N))
        tapes.append(tape)
    lg_all = torch.stack(logits).contiguous()
    out = engine.align_loss_backward(xyz_s, xyz_r, idx, lg_all, labels, transform_gt, **(loss_kwargs or {}))
    g = out["grad_logits"]
    for it in range(n_iter):
        trainer.backward(tapes[it], g[it], shared=shared)
    trainer.backward_shared(shared)
    all_reduce_gradients(trainer, dist)
    bad = trainer.grads_have_nan() or any_pose_invalid([result.get("invalid")], dist)   # train.py:437-446
    if apply and not bad:
        trainer.adam_step(lr)
    out["logits"] = lg_all
    out["skipped"] = bad
    return out


def all_reduce_gradients(trainer: "_ParamStore", dist) -> None:
    """Data-parallel training over the GPUs of a node (one process per GPU): every rank runs the step on its own pairs, then ONE
    all_reduce of the flat gradient buffer - the whole model is a single bucket, so there is nothing to overlap or schedule -
    and the mean, before the (identical) optimiser step.  ``dist``: torch.distributed (backend "nccl" = RCCL over xGMI), or
    None / world size 1: no-op.  GroupNorm is per cloud and needs nothing; BatchNorm statistics stay per rank, as they would
    under the reference's optimiser with torch's DistributedDataParallel and no SyncBatchNorm."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return
    w = dist.get_world_size()
    dist.all_reduce(trainer.flat_g)
    trainer.ops.begin()
    trainer.ops.axpy(1.0 / w - 1.0, trainer.flat_g, trainer.flat_g)        # g += (1/w - 1) g, element-wise in place


class AggregationTrainer(_ParamStore):
    """``mlp_feat`` / ``mlp_att`` / ``mlp_proj`` (network/model.py:160-176) in training mode: the 30 tensors the `feat`
    pipeline trains - its feature extractor is frozen (model.py:136, :196-198)."""

    def __init__(self, cfg: NetConfig, state_dict, device: "str | torch.device" = "cuda:0"):
        self.cfg = cfg
        self.device = torch.device(device)
        self.ops = _Ops(self.device)
        fcfg = NetConfig(**{**cfg.__dict__, "pipeline": "feat"})
        self._build_store([sp for sp in network_specs(fcfg) if sp.name.startswith(("mlp_feat.", "mlp_att.", "mlp_proj."))], state_dict)

    def forward(self, xyz: torch.Tensor, feat0: torch.Tensor, score: torch.Tensor, update_running_stats: bool = True,
                second_normalize: bool = True):
        """One side of ``Network.aggregation`` (model.py:209-235) + forward_pair's second F.normalize (:651-652; not in the
        `align` pipeline, which matches on aggregation's own output: ``second_normalize=False``), all clouds of
        the batch in one call (BatchNorm statistics over batch x points, as the reference's).  xyz [P][M][3], feat0 [P][M][64]
        (the frozen extractor's selected features), score [P][M] -> descriptors [P][M][64], tape."""
        o = self.ops
        o.begin()
        P, M, C_ = feat0.shape
        tape = {"f": [], "a": [], "p": [], "shape": (P, M, C_)}
        a = self._mlp1d(tape["f"], "mlp_feat", feat0.reshape(P * M, C_).contiguous(), 3, update_running_stats)
        g = o.empty(P * M, 4)
        g[:, :3] = xyz.reshape(P * M, 3)
        g[:, 3] = score.reshape(P * M)
        b = self._mlp1d(tape["a"], "mlp_att", g, 5, update_running_stats)
        o.axpy(1.0, b, a)                                                  # feat + xyz_g (model.py:226-227)
        e = self._mlp1d(tape["p"], "mlp_proj", a, 1, update_running_stats)
        n1, r1 = o.l2norm_fwd(e)
        if not second_normalize:
            tape["norm"] = (n1, r1, None, None)
            return n1.reshape(P, M, C_), tape
        n2, r2 = o.l2norm_fwd(n1)
        tape["norm"] = (n1, r1, n2, r2)
        return n2.reshape(P, M, C_), tape

    def backward(self, tape: dict, ddesc: torch.Tensor) -> None:
        o = self.ops
        o.begin()
        P, M, C_ = tape["shape"]
        n1, r1, n2, r2 = tape["norm"]
        d = ddesc.reshape(P * M, C_).contiguous()
        if n2 is not None:
            d = o.l2norm_bwd(d, n2, r2)
        d = o.l2norm_bwd(d, n1, r1)
        d = self._mlp1d_bwd(tape["p"], d)
        self._mlp1d_bwd(tape["f"], d, need_dx=False)                        # inputs come from the frozen extractor
        self._mlp1d_bwd(tape["a"], d, need_dx=False)


def feat_pipeline_inputs(engine, batch: dict, num_sub: int) -> dict:
    """What the frozen half of the `feat` pipeline hands the aggregation (model.py:629-648), from the inference engine
    (an Engine built with pipeline='feat'): top-``num_sub`` key points per cloud with their raw 64-d features and scores."""
    fp = engine.forward_pair(batch["points_src"], batch["points_ref"], num_sub=num_sub)
    ops = _Ops(engine.device)
    out = {}
    for side in ("src", "ref"):
        pts = batch[f"points_{side}"]
        pyr = engine.knn_pyramid(pts)
        feat, _ = engine.randla_forward("feat_extractor", pts, *pyr, want_logits=False)
        idx = fp[side]["index"]
        P, M = idx.shape
        sel = ops.empty(P * M, feat.shape[2])
        ops.gather(feat, idx, sel, 0)
        out[f"xyz_{side}"], out[f"score_{side}"], out[f"feat_{side}"] = fp[side]["xyz"], fp[side]["score"], sel.reshape(P, M, -1)
    return out


def feat_pipeline_inputs_train(engine, extractor: RandlaTrainer, batch: dict, num_sub: int, masks: Optional[dict] = None) -> dict:
    """The same hand-over with the frozen extractor in TRAINING mode, as train.py:379 leaves it (BatchNorm of its semantic head
    on batch statistics, Dropout on: the labels behind the key-point scores are those of the training-mode head):
    ``extractor.forward`` per side, the engine's score operator, top-``num_sub`` selection, gathers.  masks: {'fe_src', 'fe_ref'}
    [P][N][64] uint8 keep flags or None."""
    o = extractor.ops
    masks = masks or {}
    out = {}
    for side in ("src", "ref"):
        pts = batch[f"points_{side}"].contiguous()
        logits, tape = extractor.forward(pts, batch[f"{side}_xyz"], batch[f"{side}_neigh"], batch[f"{side}_sub"], batch[f"{side}_interp"],
                                         masks.get(f"fe_{side}"))
        feat = tape.misc["feat"].contiguous()
        score, _ = engine.score(feat, logits.contiguous(), batch[f"{side}_xyz"], batch[f"{side}_neigh"])
        o.begin()
        idx, sel_score = o.topk(score, num_sub)
        P, M = idx.shape
        sel_feat, sel_xyz = o.empty(P * M, feat.shape[2]), o.empty(P * M, 3)
        o.gather(feat, idx, sel_feat, 0)
        o.gather(pts[:, :, :3].contiguous(), idx, sel_xyz, 0)
        out[f"xyz_{side}"], out[f"score_{side}"], out[f"feat_{side}"] = sel_xyz.reshape(P, M, 3), sel_score, sel_feat.reshape(P, M, -1)
        out[f"index_{side}"] = idx
    return out


def train_step_feat(trainer: AggregationTrainer, inp: dict, transform_gt: torch.Tensor, thres_radius: float, det_loss_weight: float = 1.0,
                    lr: float = 1e-3, apply: bool = True, dist=None) -> dict:
    """One optimisation step of the `feat` pipeline (train.py:407-410, :448): DetDesLoss on the descriptors of the selected key
    points, backward through the aggregation MLPs in training mode, Adam.  ``inp``: xyz_{src,ref} [P][M][3],
    feat_{src,ref} [P][M][64], score_{src,ref} [P][M] (``feat_pipeline_inputs``)."""
    trainer.zero_grad()
    d_src, tape_s = trainer.forward(inp["xyz_src"], inp["feat_src"], inp["score_src"])      # per module the src pass first, as
    d_ref, tape_r = trainer.forward(inp["xyz_ref"], inp["feat_ref"], inp["score_ref"])      # in aggregation (model.py:217-224)
    out, g_ref, g_src = trainer.ops.det_des_loss(d_ref.contiguous(), d_src.contiguous(), inp["xyz_ref"].contiguous(),
                                                 inp["xyz_src"].contiguous(), inp["score_ref"].contiguous(), transform_gt.contiguous(),
                                                 thres_radius, det_loss_weight)
    trainer.backward(tape_s, g_src)
    trainer.backward(tape_r, g_ref)
    vals = out.cpu().numpy()
    all_reduce_gradients(trainer, dist)
    bad = trainer.grads_have_nan()
    if apply and not bad:
        trainer.adam_step(lr)
    return {"loss": float(vals[0]), "loss_feat": float(vals[1]), "loss_det": float(vals[2]), "acc": float(vals[3]),
            "desc_src": d_src, "desc_ref": d_ref, "skipped": bad}


def train_step_label(trainer: RandlaTrainer, batch: dict, labels_src: torch.Tensor, labels_ref: torch.Tensor, lr: float = 1e-3,
                     dropout_seed: Optional[int] = None, apply: bool = True, dist=None) -> dict:
    """One optimisation step of the `label` pipeline (train.py:412-415, :448): the semantic head of the feature extractor.
    ``forward_pair`` runs ``feat_extractor`` on the src and on the ref clouds SEPARATELY (model.py:629-632: BatchNorm batch
    statistics per call), ``SemanticLoss.forward`` adds the two weighted cross entropies (loss.py:991-995).
    ``trainer``: a RandlaTrainer over prefix 'feat_extractor' (feat_in = cfg.feat_len, num_classes 19); ``batch``: points_src /
    points_ref [P][N][feat_len] and both pyramids ({src,ref}_{xyz,neigh,sub,interp}); labels_* [P][N] int32 in 0..19
    (0 = unlabeled, ignored)."""
    o = trainer.ops
    o.begin()
    dev = trainer.device
    cw = torch.tensor(semantic_class_weights(), dtype=torch.float32, device=dev)
    trainer.zero_grad()
    res = {}
    outs = []
    for si, (side, labels) in enumerate((("src", labels_src), ("ref", labels_ref))):
        pts = batch[f"points_{side}"].contiguous()
        P, N, _ = pts.shape
        mask = dropout_keep_masks(None if dropout_seed is None else 2 * int(dropout_seed) + si, (P, N, trainer.cfg.out_feat_dim), dev)
        logits, tape = trainer.forward(pts, batch[f"{side}_xyz"], batch[f"{side}_neigh"], batch[f"{side}_sub"], batch[f"{side}_interp"], mask)
        d, out = o.weighted_ce(logits.reshape(P * N, trainer.num_classes), labels.reshape(-1).contiguous(), cw)
        trainer.backward(tape, d)
        outs.append(out)
        res[f"logits_{side}"] = logits
    vals = torch.stack(outs).cpu().numpy()                      # one host read for both sides
    res["loss"] = float(vals[0, 0] + vals[1, 0])
    res["acc"] = float(vals[0, 2] / max(vals[0, 3], 1.0) + vals[1, 2] / max(vals[1, 3], 1.0))    # acc_src + acc_ref (loss.py:994)
    all_reduce_gradients(trainer, dist)
    bad = trainer.grads_have_nan()
    if apply and not bad:
        trainer.adam_step(lr)
    res["skipped"] = bad
    return res


class AlignTrainStep:
    """``train_step_align`` for a fixed batch geometry with its two launch-bound halves replayed from hipGraphs.

    A step is ~6 000 small launches (20 operators x ~60 layers x n_iter, forward and backward); issued one by one from the
    host they cost more wall time than they run (32 pairs x 5000 points: 370 ms per step for 115 ms of kernels).  The
    operators are launched on torch's current stream, so ``torch.cuda.graph`` captures them like torch's own kernels: graph F
    = the n_iter training-mode forwards (inlier input, layers, BatchNorm running statistics), graph B = zero_grad + the n_iter
    backwards.  Between them the loss and its gradient (``Engine.align_loss_backward``, its own launch) and after them ONE
    Adam launch over the flat parameter buffer, whose bias corrections change per step and therefore stay outside.
    The first call runs eagerly (it IS a training step), the second captures, later ones replay.  Inputs are copied into
    static device buffers; with the same ``dropout_seed`` the masks are those of ``train_step_align`` (``dropout_keep_masks``) and the
    results equal its results except for the order of the fp32 atomics (tests/test_train.py::test_graph_replayed_step_equals_the_eager_step
    runs the comparison with dropout on)."""

    def __init__(self, engine, trainer: RandlaTrainer, pairs: int, n_src: int, n_ref: int, n_iter: int, dropout: bool = True,
                 use_graph: bool = True):
        self.engine, self.tr = engine, trainer
        self.P, self.N, self.K, self.n_iter, self.dropout, self.use_graph = pairs, n_src, n_ref, n_iter, dropout, use_graph
        dev = trainer.device
        S = sum(level_sizes(n_src, trainer.cfg.sub_sampling_ratio)[:len(trainer.cfg.d_out)])
        S1 = sum(level_sizes(n_src, trainer.cfg.sub_sampling_ratio)[1:len(trainer.cfg.d_out) + 1])
        z = lambda shape, dt=torch.float32: torch.zeros(shape, dtype=dt, device=dev)
        self.xyz_s, self.xyz_r = z((pairs, n_src, 3)), z((pairs, n_ref, 3))
        self.src_xyz, self.neigh = z((pairs, S, 3)), z((pairs, S, K_NN), torch.int32)
        self.sub, self.interp = z((pairs, S1, K_NN), torch.int32), z((pairs, S, 1), torch.int32)
        self.idx, self.T = z((n_iter, pairs, n_src), torch.int32), z((pairs, n_iter, 3, 4))
        self.masks = z((n_iter, pairs, n_src, trainer.cfg.out_feat_dim), torch.uint8) if dropout else None
        self.logits, self.grad = z((n_iter, pairs, n_src)), z((n_iter, pairs, n_src))
        self.tapes: List[RandlaTape] = []
        self.gf = self.gb = None
        self.calls = 0
        self.gen = torch.Generator(device=dev)

    def _forward_all(self) -> None:
        tr = self.tr
        tr.ops.begin()
        self.tapes = []
        self.shared = {}                    # position-encoding branch: computed by iteration 0, re-used by the others
        for it in range(self.n_iter):
            cat = tr.ops.inlier_input(self.xyz_s, self.xyz_r, self.idx[it], None if it == 0 else self.T[:, it - 1])
            lg, tape = tr.forward(cat, self.src_xyz, self.neigh, self.sub, self.interp, None if self.masks is None else self.masks[it],
                                  shared=self.shared)
            self.logits[it].copy_(lg.reshape(self.P, self.N))
            self.tapes.append(tape)

    def _backward_all(self) -> None:
        self.tr.zero_grad()
        for it in range(self.n_iter):
            self.tr.backward(self.tapes[it], self.grad[it], shared=self.shared)
        self.tr.backward_shared(self.shared)

    def step(self, batch: dict, result: dict, transform_gt, labels=None, lr: float = 1e-3, dropout_seed: Optional[int] = None,
             loss_kwargs: Optional[dict] = None, apply: bool = True, dist=None) -> dict:
        tr = self.tr
        self.xyz_s.copy_(batch["points_src"][:, :, :3]); self.xyz_r.copy_(batch["points_ref"][:, :, :3])
        self.src_xyz.copy_(batch["src_xyz"]); self.neigh.copy_(batch["src_neigh"])
        self.sub.copy_(batch["src_sub"]); self.interp.copy_(batch["src_interp"])
        self.idx.copy_(result["idx"]); self.T.copy_(result["transforms"])
        if self.masks is not None:
            # the same draw as train_step_align's (dropout_keep_masks); without a seed: the stepper's own running generator
            m = dropout_keep_masks(dropout_seed, self.masks.shape, tr.device)
            self.masks.copy_(m if m is not None else torch.rand(self.masks.shape, generator=self.gen, device=tr.device) >= 0.5)
        self.calls += 1
        graphs = self.use_graph and self.calls >= 2
        if graphs and self.gf is None:
            tr.ops._scratch = None                                     # the captured launches get scratch from the graph's pool
            torch.cuda.synchronize()
            self.gf = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.gf):
                self._forward_all()
        if graphs:
            self.gf.replay()
        else:
            self._forward_all()
        out = self.engine.align_loss_backward(self.xyz_s, self.xyz_r, self.idx, self.logits, labels, transform_gt, **(loss_kwargs or {}))
        self.grad.copy_(out["grad_logits"])
        if graphs and self.gb is None:
            torch.cuda.synchronize()
            self.gb = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.gb, pool=self.gf.pool()):
                self._backward_all()
            tr.ops._scratch = None                                     # eager callers allocate their own again
        if graphs:
            self.gb.replay()
        else:
            self._backward_all()
        all_reduce_gradients(tr, dist)
        bad = tr.grads_have_nan() or any_pose_invalid([result.get("invalid")], dist)   # train.py:437-446
        if apply and not bad:
            tr.adam_step(lr)
        out["logits"] = self.logits
        out["skipped"] = bad
        return out

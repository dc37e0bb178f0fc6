"""Batch-1 SERVING: K independent single-pair registrations in flight.

The reference evaluates one pair per ``model(data, opt)`` call (test.py:56 ``BATCH_SIZE = 1``, the loop at
test.py:386-450, fed by DataLoader workers).  One pair alone leaves the GPU almost empty - its registration is ~290
dependent kernel launches of a few microseconds each - so a caller that can keep K requests outstanding is served by a
small scheduler instead:

* ``submit(src, ref)`` queues ONE pair and returns a future at once (no host synchronisation);
* requests of the same shape are coalesced, up to ``max_batch = K / engines`` of them, into one ``dsir_register`` call that
  is replayed from a captured hipGraph (one graph per batch size, ``include/dsir.h`` ``dsir_enable_graph``);
* ``engines`` contexts on their own HIP streams take the batches in turn, so the launch chain of one batch fills the gaps of
  the other's.  Measured (bench.py `concurrent_single_pairs`, tools/k8_sweep.py; 5000-point pairs, round 5): up to 8 requests in flight
  ONE engine with the whole window in its batch is as fast as two (K = 8: 1720 - 1790 pairs/s either way - a batch of 8 costs little more
  than a batch of 4), from 16 on two engines win (K = 16: 2530 - 2710 against 2310 - 2510); more than two never do - every replayed graph
  holds the host ~1.4 ms, so E engines need E x 1.4 ms per round (profiles/r05_serving_queues.txt).  ``engines=None`` picks by that rule;
* ``future.result()`` waits for that request's batch only.

A pair's result does not depend on what shares its batch (every kernel's tiling is a function of the per-cloud shape alone,
GroupNorm statistics meet in exact, order-independent atomics on integer-valued fp64 limbs): served results are bit-identical to the batched path's -
``tests/test_serve.py`` asserts it.  Nothing here computes: all work is ``Engine.register`` (libdsir.so).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch

from .arch import NetConfig
from .engine import Engine, EngineError

_OUT_KEYS = ("transforms", "idx", "logits", "pt_ref_new", "invalid")


class PairFuture:
    """Result handle of one submitted pair."""

    __slots__ = ("_server", "_batch", "_j", "_res")

    def __init__(self, server: "PairServer"):
        self._server, self._batch, self._j, self._res = server, None, -1, None

    def done(self) -> bool:
        return self._batch is not None and self._batch["event"].query()

    def result(self, wait: str = "host") -> Dict[str, torch.Tensor]:
        """dict(transforms [n_iter,3,4], idx [n_iter,J] i32, logits [n_iter,J], pt_ref_new [J,3], invalid [] i32) on the device
        (a request submitted as a whole batch, ``submit_batch``: the batched shapes of ``Engine.register``).
        wait='host': blocks the host until this request's batch has finished; wait='stream': torch's CURRENT stream waits for it
        on the device instead and the host returns at once - the tensors are then stream-ordered like any torch result.
        A batch still queued is dispatched first."""
        if self._batch is None:
            self._server.flush()
        b = self._batch
        cur = torch.cuda.current_stream(self._server.device)
        if wait == "host":
            b["event"].synchronize()
        else:
            cur.wait_event(b["event"])
        if cur.cuda_stream not in b["recorded"]:
            for v in b["out"].values():
                v.record_stream(cur)          # allocated on the engine's stream, consumed on the caller's
            b["recorded"].add(cur.cuda_stream)
        if self._res is None:
            o, j = b["out"], self._j
            if j < 0:
                self._res = dict(o)
            else:
                self._res = {"transforms": o["transforms"][j]}
                if "idx" in o:
                    self._res.update(idx=o["idx"][:, j], logits=o["logits"][:, j], pt_ref_new=o["pt_ref_new"][j], invalid=o["invalid"][j])
        return self._res


class _Slot:
    """One engine with its stream and the static buffers its captured graphs are bound to."""

    def __init__(self, cfg: NetConfig, sd, device: int, max_points: int, max_batch: int, n_iter: int, want_aux: bool):
        self.eng = Engine(cfg, device, max_points, max_batch)
        self.eng.load_state_dict(sd)
        self.stream = torch.cuda.Stream(device=self.eng.device)
        with torch.cuda.stream(self.stream):
            self.eng.use_torch_stream(True)
        self.eng.enable_graph(True)
        dev, C = self.eng.device, cfg.feat_len
        self.src = torch.empty(max_batch * max_points * C, dtype=torch.float32, device=dev)
        self.ref = torch.empty(max_batch * max_points * C, dtype=torch.float32, device=dev)
        self.T = torch.empty(max_batch * n_iter * 12, dtype=torch.float32, device=dev)
        self.aux = None
        if want_aux:
            self.aux = {"idx": torch.empty(n_iter * max_batch * max_points, dtype=torch.int32, device=dev),
                        "logits": torch.empty(n_iter * max_batch * max_points, dtype=torch.float32, device=dev),
                        "pt_ref_new": torch.empty(max_batch * max_points * 3, dtype=torch.float32, device=dev),
                        "invalid": torch.empty(max_batch, dtype=torch.int32, device=dev)}


class PairServer:
    def __init__(self, cfg: NetConfig, state_dict, device: int = 0, max_points: int = 5000, max_in_flight: int = 8,
                 engines: Optional[int] = None, n_iter: int = 5, want_aux: bool = True):
        if cfg.pipeline != "align":
            raise EngineError("PairServer serves the align pipeline (dsir_register)")
        self.cfg, self.n_iter, self.want_aux = cfg, int(n_iter), bool(want_aux)
        if engines is None:
            engines = 1 if int(max_in_flight) <= 8 else 2
        self.engines = max(1, min(int(engines), int(max_in_flight)))
        self.max_batch = max(1, int(max_in_flight) // self.engines)
        self.max_in_flight = self.max_batch * self.engines
        self.max_points = int(max_points)
        self.slots = [_Slot(cfg, state_dict, device, self.max_points, self.max_batch, self.n_iter, self.want_aux) for _ in range(self.engines)]
        self.device = self.slots[0].eng.device
        self._next = 0
        self._pending: Dict[Tuple[int, int], List[Tuple[torch.Tensor, torch.Tensor, PairFuture]]] = {}
        self.batches_dispatched = 0
        self.pairs_dispatched = 0

    def close(self):
        for s in self.slots:
            s.stream.synchronize()
            s.eng.close()
        self.slots = []

    def load_state_dict(self, state_dict):
        """New weights into the running engines (``dsir_finalize_weights`` drops their captured graphs; the next batch of every
        signature is captured again): no engine, workspace or static buffer is rebuilt."""
        for s in self.slots:
            s.stream.synchronize()          # a batch still replaying reads the old blob
            s.eng.load_state_dict(state_dict)

    # ------------------------------------------------------------------ requests
    def submit(self, points_src: torch.Tensor, points_ref: torch.Tensor) -> PairFuture:
        """One pair: points_src [N_src, C] (or [1, N_src, C]), points_ref likewise; CUDA or host tensors (host tensors are copied
        asynchronously when the batch is dispatched)."""
        src = points_src[0] if points_src.dim() == 3 else points_src
        ref = points_ref[0] if points_ref.dim() == 3 else points_ref
        J, K = int(src.shape[0]), int(ref.shape[0])
        if max(J, K) > self.max_points:
            raise EngineError(f"cloud of {max(J, K)} points exceeds the server's max_points={self.max_points}")
        if src.shape[1] != self.cfg.feat_len or ref.shape[1] != self.cfg.feat_len:
            raise EngineError(f"points have {src.shape[1]} channels, the server was built for feat_len={self.cfg.feat_len}")
        fut = PairFuture(self)
        q = self._pending.setdefault((J, K), [])
        q.append((src, ref, fut))
        if len(q) >= self.max_batch:
            self._dispatch((J, K))
        return fut

    def submit_batch(self, points_src: torch.Tensor, points_ref: torch.Tensor) -> PairFuture:
        """B <= max_batch pairs that already come as one batch ([B, N, C] each): dispatched at once as ONE engine call; the
        future's result has the batched shapes (transforms [B,n_iter,3,4], idx / logits [n_iter,B,J], pt_ref_new [B,J,3],
        invalid [B])."""
        B, J, K = int(points_src.shape[0]), int(points_src.shape[1]), int(points_ref.shape[1])
        if B > self.max_batch or max(J, K) > self.max_points:
            raise EngineError(f"batch of {B} x {max(J, K)} points exceeds the server's max_batch={self.max_batch} / max_points={self.max_points}")
        fut = PairFuture(self)
        self._dispatch((J, K), [(points_src, points_ref, fut)], whole=B)
        return fut

    def flush(self):
        """Dispatch whatever is queued (partial batches included)."""
        for shape in [s for s, q in self._pending.items() if q]:
            self._dispatch(shape)

    def _dispatch(self, shape, q=None, whole: int = 0):
        if q is None:
            q = self._pending.pop(shape)
        J, K = shape
        b, C, n = (whole or len(q)), self.cfg.feat_len, self.n_iter
        slot = self.slots[self._next]
        self._next = (self._next + 1) % self.engines
        src = slot.src[: b * J * C].view(b, J, C)
        ref = slot.ref[: b * K * C].view(b, K, C)
        out = {"transforms": slot.T[: b * n * 12].view(b, n, 3, 4)}
        if slot.aux is not None:
            out.update(idx=slot.aux["idx"][: n * b * J].view(n, b, J), logits=slot.aux["logits"][: n * b * J].view(n, b, J),
                       pt_ref_new=slot.aux["pt_ref_new"][: b * J * 3].view(b, J, 3), invalid=slot.aux["invalid"][:b])
        caller = torch.cuda.current_stream(self.device)
        with torch.cuda.stream(slot.stream):
            slot.stream.wait_stream(caller)                    # the request tensors may still be being produced there
            if whole:
                src.copy_(q[0][0], non_blocking=True)
                ref.copy_(q[0][1], non_blocking=True)
            elif len(q) > 1 and all(s.is_cuda and r.is_cuda and s.dtype == src.dtype and r.dtype == ref.dtype for s, r, _ in q):
                # device-resident requests: ONE gather launch per side instead of one copy per request (16 launches of ~5 us in front
                # of a batch of 8: 2 % of the batch's time)
                torch.stack([s for s, _, _ in q], out=src)
                torch.stack([r for _, r, _ in q], out=ref)
            else:
                for j, (s, r, _) in enumerate(q):
                    src[j].copy_(s, non_blocking=True)
                    ref[j].copy_(r, non_blocking=True)
            # the copies above are queued on THIS slot's stream behind its previous batch, but the request tensors were allocated on
            # the caller's stream and the pending queue drops the last reference to them when this function returns: without the
            # marks below torch's caching allocator may hand their memory to the caller's next allocation before the copies have run
            for s, r, _ in q:
                for t in (s, r):
                    if t.is_cuda:
                        t.record_stream(slot.stream)
            slot.eng.register(src, ref, n, want_aux=self.want_aux, sync=False, out=out)   # hipGraph replay, one graph per (b, J, K)
            res = {k: v.clone() for k, v in out.items() if k in _OUT_KEYS}                  # the static buffers are re-used by the next batch
            ev = torch.cuda.Event()
            ev.record(slot.stream)
        batch = {"out": res, "event": ev, "recorded": set()}
        for j, (_, _, fut) in enumerate(q):
            fut._batch, fut._j = batch, (-1 if whole else j)
        self.batches_dispatched += 1
        self.pairs_dispatched += b

    # ------------------------------------------------------------------ closed-loop driver (bench / harness)
    def run_closed_loop(self, pairs, in_flight: Optional[int] = None):
        """Registers ``pairs`` (an iterable of (src, ref)) keeping ``in_flight`` requests outstanding: the next pair is submitted
        as soon as the oldest result has been collected.  Returns the results in submission order."""
        K = self.max_in_flight if in_flight is None else max(1, min(int(in_flight), self.max_in_flight))
        futs: List[PairFuture] = []
        results = []
        head = 0
        for s, r in pairs:
            if len(futs) - head >= K:
                results.append(futs[head].result())
                head += 1
            futs.append(self.submit(s, r))
        self.flush()
        while head < len(futs):
            results.append(futs[head].result())
            head += 1
        return results

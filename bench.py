#!/usr/bin/env python3
"""Headline benchmark: registered pairs/sec on 3DMatch-shaped 5000-point pairs (BASELINE.json configs[1]/[3], "C2").

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the whole hot path (KNN pyramid -> 2x feature RandLA + score -> 5 x {aggregation, nearest
descriptor, inlier RandLA, weighted Kabsch}) over one batch of P synthetic pairs per GPU, through the C ABI
(dsir_register).  Inputs (raw [P,5000,3] clouds) are resident in HBM before the timed region; outputs (R,t per
iteration) stay in HBM.  Pairs shard across ranks with no data-path collective; one RCCL all_gather of the (R,t)
results closes the timed region.

Launched WITHOUT a torchrun environment and with --gpus N > 1, the script re-launches itself as N ranks through
``python -m torch.distributed.run`` (before anything touches the GPU) and forwards rank 0's JSON line.
``--dry-run`` rehearses the launch / shard / gather / n_gpus logic on CPU with gloo (no engine, no number).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import os

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")   # before the GPU is touched: deepsir_amd/__init__.py
import json
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F16_MFMA_TFLOPS = 2500.0   # dense fp16/bf16 MFMA, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_F32_MFMA_TFLOPS = 157.3    # same guide: v_mfma_f32_16x16x4_f32 dense peak (= the fp32 vector peak)
PEAK_HBM_GBPS = 8000.0          # HBM3E spec (6.3 TB/s measured achievable)
SCREEN_MIN_WORK = 100_000_000   # csrc/engine.hip: P*J*K from which dsir_register screens the arg-min


def match_flops(P, J, K):
    """Algorithmic FLOPs of one nearest-descriptor search (128 J K + 3 J K per pair)."""
    return float(P) * (128.0 * J * K + 3.0 * J * K)


def screen_flops(P, J, K):
    """MFMA FLOPs screen_kernel executes: three fp16 products (ah.bh, ah.bl, al.bh) x 2 x 64 per (row, column)."""
    return float(P) * 384.0 * J * K


def path_flops(n, n_iter):
    """Algorithmic FLOPs of one pair through the whole path: F = 2 R(N) + n_iter [2 A(N) + 131 N^2 + R(N)
    + 40 N] + KNN(N), R(N) = 327.4 kFLOP/pt (RandLA pass), A(N) = 2 N (20480 + 59520 + 4096) (aggregation of one cloud),
    KNN(N) = 20 sum_l n_l^2 (SURVEY counts 0.53 G at N = 5000).  Hoisted loop invariants are NOT subtracted."""
    R = 327.4e3 * n
    A = 2.0 * n * (20480 + 59520 + 4096)
    lv, knn = n, 0.0
    for _ in range(4):
        knn += 20.0 * lv * lv
        lv //= 4
    return 2 * R + n_iter * (2 * A + 131.0 * n * n + R + 40.0 * n) + knn


class _StdoutToStderr:
    """RCCL prints a version banner on STDOUT when its first communicator comes up; the driver wants ONE JSON line there."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=int(os.environ.get("DSIR_BENCH_PAIRS", "256")), help="pairs per step per GPU")
    ap.add_argument("--points", type=int, default=5000)
    ap.add_argument("--streams", type=int, default=int(os.environ.get("DSIR_BENCH_STREAMS", "2")),
                    help="engine streams per GPU; the batch is split across them and registered concurrently")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--feat-len", type=int, default=3, help="3 = xyz (3DMatch), 4 = xyz + reflectance (KITTI)")
    ap.add_argument("--shape", default="3dmatch", choices=["3dmatch", "kitti"], help="extent of the synthetic clouds")
    ap.add_argument("--partial-overlap", action="store_true", help="config 5: 50 %% overlap crops + jitter")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU oracle leg (baseline + parity check)")
    ap.add_argument("--no-latency", action="store_true", help="skip the batch-1 latency phase (profiling runs)")
    ap.add_argument("--no-companion", action="store_true", help="skip the exhaustive-arg-min companion run (profiling runs)")
    ap.add_argument("--timed-only", action="store_true",
                    help="nothing but the allocation call, the warm-up and the timed steps (PMC passes: every dispatch then belongs to a step)")
    ap.add_argument("--total-pairs", type=int, default=0,
                    help="STRONG scaling (BASELINE configs[3]: the 3DMatch test set has 1623 pairs): a FIXED set of this many pairs, "
                         "block-sharded over the ranks (unequal shards), each rank registering its shard in calls of --pairs, one padded "
                         "all_gather of the results per pass; a step = one pass over the whole set.  0 = weak scaling (default)")
    ap.add_argument("--weights", default="plain", choices=["plain", "separated"],
                    help="seeded weight variant (deepsir_amd/weights.py); companion runs only, the headline is quoted on 'plain'")
    ap.add_argument("--cluster-descriptors", type=float, default=0.0, metavar="S",
                    help="0 < S < 1: descriptor head scaled down by S under a fixed bias (weights variant clustered:S): descriptors of all "
                         "points within a cap of angular radius ~S - the regime the arg-min screening cannot thin out")
    ap.add_argument("--dry-run", action="store_true",
                    help="CPU rehearsal of the launcher, the sharding and the result gather (gloo); no engine, no number")
    return ap.parse_args(argv)


def relaunch_as_ranks(a, argv):
    """--gpus N > 1 without a torchrun environment: become N ranks.  Runs before torch.cuda is touched in this process
    (a process that has initialised the GPU must not exec or fork GPU children), as a child process whose exit code is
    passed on."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def dry_run(a, rank, local_rank, world):
    """The N > 1 plumbing on CPU: rendezvous, world-size check, n_gpus by all_reduce, block sharding of a pair list,
    padded all_gather of per-pair results (deepsir_amd/dist.py) - everything of the multi-GPU path but the engine."""
    import torch
    import torch.distributed as dist
    from deepsir_amd.dist import gather_results, shard_range, shard_sizes
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    assert world == a.gpus, f"launched as {world} ranks but --gpus {a.gpus}"
    ones = torch.ones(1, dtype=torch.int64)
    if world > 1:
        dist.all_reduce(ones)
    total = a.total_pairs if a.total_pairs > 0 else a.pairs * world + 3      # unequal shards on purpose
    mine = shard_range(total, rank, world)
    local = torch.stack([torch.full((a.iters, 3, 4), float(i)) for i in mine]) if len(mine) else torch.zeros(0, a.iters, 3, 4)
    out = gather_results(local, dist if world > 1 else None, sizes=shard_sizes(total, world))
    ok = out.shape[0] == total and out[:, 0, 0, 0].tolist() == [float(i) for i in range(total)]
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": int(ones.item()), "world_size": world, "pairs_total": total,
                          "shard_sizes": shard_sizes(total, world), "calls_per_rank": [-(-n // a.pairs) for n in shard_sizes(total, world)],
                          "scaling": "strong" if a.total_pairs > 0 else "weak",
                          "gather_ok": bool(ok), "backend": "gloo", "value": None}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if ok else 1


def cpu_leg(cfg, sd, n_points, n_iter, checks, budget_s=15.0):
    """The CPU oracle leg (rank 0, N = 1 only): (1) the oracle - the PyTorch-CPU port of the reference path, bit-identical
    to the imported reference on the golden fixtures - timed on this box's host cores over a bounded sample, in both
    windows: model-only (the reference's own, test.py:399-402, KNN pyramid pre-built) and with the CPU KNN pyramid;
    (2) the oracle as CHECKER of what was just measured: for each sampled pair of the benchmarked batch the engine's
    correspondences are forced into the oracle and the poses compared at every iteration."""
    import numpy as np
    import torch
    from deepsir_amd.synth import make_pair
    from oracle.knn import add_pyramids
    from oracle.network import OracleNet, to_torch

    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    # the GPU box gives one GPU a share of 16 host cores; never oversubscribe (affinity may list far more)
    threads = max(1, min(avail, int(os.environ.get("DSIR_CPU_THREADS", "16"))))
    torch.set_num_threads(threads)
    net = OracleNet(cfg, sd)
    raw = make_pair(n_points, 1001, cfg.feat_len)
    t0 = time.time()
    data_np = add_pyramids(raw, cfg.num_knn, cfg.sub_sampling_ratio)
    knn_s = [time.time() - t0]
    data = to_torch(data_np)
    t0 = time.time()
    net.register(data, n_iter)  # warm-up (timed only to bound the sample)
    warm = time.time() - t0
    times = []
    t_end = time.time() + budget_s
    while (len(times) < 3 and warm < budget_s / 3) or (time.time() < t_end and len(times) < 60):
        t0 = time.time()
        net.register(data, n_iter)
        times.append(time.time() - t0)
    if not times:
        times = [warm]
    for _ in range(2 if knn_s[0] < 3.0 else 0):
        t0 = time.time()
        add_pyramids(raw, cfg.num_knn, cfg.sub_sampling_ratio)
        knn_s.append(time.time() - t0)
    med, knn_med = float(np.median(times)), float(np.median(knn_s))
    base = {"value": round(1.0 / med, 4), "unit": "pairs/s", "cores": torch.get_num_threads(), "kind": "port",
            "window": "model-only (KNN pyramid pre-built, the reference's own timing window test.py:399-402)",
            "with_cpu_knn": {"value": round(1.0 / (med + knn_med), 4), "unit": "pairs/s", "knn_s_per_pair": round(knn_med, 4),
                             "note": "plus the oracle's exact numpy KNN pyramid of both clouds (1 thread; the reference runs "
                                     "nanoflann in DataLoader workers outside its timing window) - compare with `value` of this line"},
            "sample": f"{len(times)} timed runs of 1 pair x {n_points} pts x {n_iter} iters after 1 warm-up, median; "
                      f"oracle = PyTorch-CPU restatement, bit-identical to the imported reference on the golden fixtures"}
    parity = None
    if checks:
        worst_r = worst_t = worst_excess = 0.0
        near_ok, differ, rows = True, 0, 0
        for raw_p, idx_e, T_e, desc_s, desc_r in checks:
            d = to_torch(add_pyramids(raw_p, cfg.num_knn, cfg.sub_sampling_ratio))
            T_o, _ = net.register(d, n_iter, forced_idx=[idx_e[i][None].long() for i in range(n_iter)])
            for i in range(n_iter):
                A, B = T_e[i].astype(np.float64), T_o[i][0].numpy().astype(np.float64)
                D = A[:, :3].T @ B[:, :3]
                v = 0.5 * np.array([D[2, 1] - D[1, 2], D[0, 2] - D[2, 0], D[1, 0] - D[0, 1]])
                worst_r = max(worst_r, float(np.arctan2(np.linalg.norm(v), 0.5 * (np.trace(D) - 1.0))))
                worst_t = max(worst_t, float(np.linalg.norm(A[:, 3] - B[:, 3])))
                # the arg-min on the ENGINE'S OWN descriptors of this iteration, in fp64 on the CPU: the picked column must be
                # within 2e-6 (1 + |d|) of the row minimum on EVERY row (fp32 evaluation noise of the reference formula)
                a64, b64 = desc_s[i].double(), desc_r.double()
                sb = (b64 * b64).sum(1)
                pick = idx_e[i].long()
                for c0 in range(0, a64.shape[0], 1024):
                    x = a64[c0:c0 + 1024]
                    dd = (x * x).sum(1)[:, None] + sb[None, :] - 2.0 * (x @ b64.t())
                    dmin, amin = dd.min(1)
                    dp = dd.gather(1, pick[c0:c0 + 1024, None])[:, 0]
                    worst_excess = max(worst_excess, float(((dp - dmin) / (2e-6 * (1.0 + dmin.abs()))).max()))
                    differ += int((pick[c0:c0 + 1024] != amin).sum())
                rows += int(a64.shape[0])
        near_ok = worst_excess <= 1.0
        parity = {"pairs_checked": len(checks), "iterations": n_iter,
                  "max_rot_err_rad": float(f"{worst_r:.3e}"), "max_trans_err_m": float(f"{worst_t:.3e}"),
                  "tolerance": "1e-4 rad / 1e-4 m (BASELINE north_star)", "argmin_rows": rows,
                  "argmin_rows_not_the_fp64_argmin": differ, "argmin_worst_excess_over_fp64_min": float(f"{worst_excess * 2e-6:.3e}"),
                  "argmin_every_row_within_2e-6": near_ok,
                  "ok": bool(worst_r < 1e-4 and worst_t < 1e-4 and near_ok and differ <= 5e-3 * rows),
                  "method": "pairs sampled from the benchmarked batch, registered by the benchmarked configuration; the engine's "
                            "correspondences forced into the CPU oracle, poses compared at every iteration; every arg-min compared in "
                            "fp64 (CPU) on the engine's own descriptors of that iteration: the picked column within 2e-6 (1 + |d|) of "
                            "the row minimum on EVERY row (none excused), equal to the fp64 arg-min on all but <= 0.5 % (fp32-level ties)"}
    return base, parity


def main():
    argv = sys.argv[1:]
    a = parse_args(argv)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(relaunch_as_ranks(a, argv))
    if a.dry_run:
        sys.exit(dry_run(a, rank, local_rank, world))

    import numpy as np
    import torch
    assert torch.cuda.is_available(), "bench.py needs a GPU (use --dry-run to rehearse the multi-rank plumbing on CPU)"
    assert world == a.gpus, f"launched as {world} ranks but --gpus {a.gpus}"
    ndev = torch.cuda.device_count()           # counting devices does not initialise the GPU
    backend = os.environ.get("DSIR_BENCH_BACKEND", "nccl")   # "nccl" is RCCL on ROCm; "gloo" only for rehearsals
    if world > 1 and backend == "nccl":
        # RCCL wants one device per rank: fail before any GPU call instead of hanging in the communicator set-up
        assert ndev >= world, (f"--gpus {world} over RCCL needs {world} visible devices, this node shows {ndev} "
                               f"(rehearse on fewer GPUs with DSIR_BENCH_BACKEND=gloo)")
    dev_index = local_rank % max(ndev, 1)      # rehearsing N ranks on fewer GPUs maps ranks round-robin
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    # The result gather is an RCCL all_gather at EVERY world size, N = 1 included (one rank gathering from itself runs the same
    # library call path), so the communicator is created before any other GPU work - also when no launcher set the rendezvous up.
    import torch.distributed as dist  # noqa: F811
    dist_error = None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
    n_gpus = 1
    with _StdoutToStderr():
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
        except Exception as e:
            if world > 1:
                raise
            dist_error = f"{type(e).__name__}: {e}"[:300]     # N = 1 can run without a communicator; the line says so
            dist = None
        # number of distinct GPUs in the job, from a collective (not from the environment) - also the communicator's first use
        if dist is not None:
            on = dev if dist.get_backend() == "nccl" else "cpu"
            ids = [torch.zeros(1, dtype=torch.int64, device=on) for _ in range(world)]
            dist.all_gather(ids, torch.tensor([dev_index], dtype=torch.int64, device=on))
            ones = torch.ones(1, dtype=torch.int64, device=on)
            dist.all_reduce(ones)
            assert int(ones.item()) == world
            n_gpus = len({int(t.item()) for t in ids})
            torch.cuda.synchronize()

    from deepsir_amd.arch import NetConfig
    from deepsir_amd.dist import gather_results, shard_range, shard_sizes
    from deepsir_amd.engine import Engine, EnginePool
    from deepsir_amd.synth import make_batch, make_pair
    from deepsir_amd.weights import generate_state_dict

    cfg = NetConfig(feat_len=a.feat_len)
    variant = f"clustered:{a.cluster_descriptors}" if a.cluster_descriptors > 0 else a.weights
    sd = generate_state_dict(cfg, 0, variant)
    P, N, n_iter = a.pairs, a.points, a.iters
    strong = a.total_pairs > 0
    if strong:
        # STRONG scaling: a fixed set of pairs (BASELINE configs[3]), contiguous block shards (the first total % world ranks
        # hold one pair more), every rank walks its shard in engine calls of at most P pairs
        mine = shard_range(a.total_pairs, rank, world)
        sizes = shard_sizes(a.total_pairs, world)
        seeds = [10_000 + i for i in mine]
        P = max(1, min(P, max(sizes)))
    else:
        # WEAK scaling: every rank registers its own P pairs per step, seeds partitioned by rank
        sizes = None
        seeds = [10_000 + rank * P + i for i in range(P)]
    L = len(seeds)                             # pairs this rank registers per step
    S = max(1, min(a.streams, P))
    eng = EnginePool(cfg, dev_index, max_points=N, max_pairs=P, streams=S) if S > 1 else Engine(cfg, dev_index, max_points=N, max_pairs=P)
    P_launch = (min(P, max(L, 1)) + S - 1) // S   # pairs per arg-min search (of a full call)
    eng.load_state_dict(sd)
    if L:
        batch = make_batch(N, seeds, cfg.feat_len, a.shape, a.partial_overlap)
    else:                                      # more ranks than pairs: this rank only takes part in the gather
        batch = {"points_src": np.zeros((0, N, cfg.feat_len), np.float32), "points_ref": np.zeros((0, N, cfg.feat_len), np.float32),
                 "transform_gt": np.zeros((0, 3, 4), np.float32)}
    src = torch.from_numpy(batch["points_src"]).to(dev)
    ref = torch.from_numpy(batch["points_ref"]).to(dev)
    calls = [(c0, min(c0 + P, L)) for c0 in range(0, L, P)]
    out_all = torch.empty((L, n_iter, 3, 4), dtype=torch.float32, device=dev)      # (R,t) of every pair of the shard, in HBM
    out_buf = {"transforms": out_all}

    def step():
        for c0, c1 in calls:
            eng.register(src[c0:c1], ref[c0:c1], n_iter, want_aux=False, sync=False, out={"transforms": out_all[c0:c1]})

    def fence():
        eng.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    for _ in range(a.warmup):
        step()
    if strong and a.warmup:
        eng.sync()
        gather_results(out_all, dist, sizes=sizes)   # the communicator's first collective is not a steady-state one
    fence()
    eng.enable_match_timer(True)
    eng.match_timer2(reset=True)
    eng.screen_stats(reset=True)
    fence()
    gather_s = 0.0
    t0 = time.perf_counter()
    if strong:
        # one pass over the fixed set per step; its results meet in ONE padded all_gather per pass (the job's only exchange)
        for _ in range(a.steps):
            step()
            eng.sync()
            tg = time.perf_counter()
            results = gather_results(out_all, dist, sizes=sizes)
            torch.cuda.synchronize()
            gather_s += time.perf_counter() - tg
        t_local = time.perf_counter() - t0
    else:
        for _ in range(a.steps):
            step()
        eng.sync()
        t_local = time.perf_counter() - t0
        tg = time.perf_counter()
        results = gather_results(out_all, dist)   # RCCL all_gather of (R,t) - the only collective
        torch.cuda.synchronize()
        gather_s = time.perf_counter() - tg
    fence()
    dt = time.perf_counter() - t0
    c_op_ms, c_k_ms, c_n = eng.match_timer2(reset=True)
    sstats = eng.screen_stats(reset=True)
    eng.enable_match_timer(False)
    per_rank = [[float(L), t_local * 1e3 / a.steps, gather_s * 1e3 / (a.steps if strong else 1)]]
    if dist is not None:
        on = dev if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([dt], dtype=torch.float64, device=on)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        rows = [torch.zeros(3, dtype=torch.float64, device=on) for _ in range(world)]
        dist.all_gather(rows, torch.tensor(per_rank[0], dtype=torch.float64, device=on))
        per_rank = [r.tolist() for r in rows]
    pairs_per_step = a.total_pairs if strong else world * P
    assert results.shape[0] == pairs_per_step and torch.isfinite(results).all()
    screened = sstats["screened_searches"] > 0

    # ------------------------------------------------------------------ rank-0 extras (outside the timed region)
    model_only = single = single_ex = companion = latency = training = line_stress = concurrent_single = throughput_curve = config4 = large_cfgs = None
    checks = []
    if a.timed_only:
        a.no_cpu_baseline = a.no_latency = a.no_companion = True
    if rank == 0 and not a.timed_only:
        e0 = eng.engines[0] if hasattr(eng, "engines") else eng
        # Roofline pass: the dominant kernel timed with HIP events on its stream while ONE engine runs the hot path over
        # its share of the batch (same shapes and launches as above).  With several engines in flight a bracket also
        # contains the time the launch shares the CUs with other engines' kernels (roofline.concurrent); on one stream it
        # is the kernel's own duration - what rocprofv3's kernel trace of this command averages to.
        P_launch = min(P_launch, L)
        s0, r0 = src[:P_launch].contiguous(), ref[:P_launch].contiguous()
        o0 = e0.register(s0, r0, n_iter, want_aux=False)

        def one_engine_pass():
            e0.enable_match_timer(True)
            e0.match_timer2(reset=True)
            e0.prune_stats(reset=True)
            for _ in range(3):
                e0.register(s0, r0, n_iter, want_aux=False, sync=False, out={"transforms": o0["transforms"]})
            e0.sync()
            op, k, n = e0.match_timer2(reset=True)
            kept, unpruned = e0.prune_stats(reset=True)      # (row block, column tile) products of the pruned launches (csrc/nn_prune.hip)
            e0.enable_match_timer(False)
            return (op / n, k / n, int(n), kept, unpruned) if n else None

        single = one_engine_pass()
        if screened and not a.no_companion:
            e0.enable_screen(False)
            e0.register(s0, r0, n_iter, want_aux=False, out={"transforms": o0["transforms"]})
            single_ex = one_engine_pass()
            e0.enable_screen(True)
        del s0, r0

    if world == 1 and not a.timed_only and strong:
        # companions below are written for one engine call per step: run them on the first call's pairs
        src, ref, L = src[:P].contiguous(), ref[:P].contiguous(), min(L, P)
        out_buf = {"transforms": out_all[:L]}
        calls = [(0, L)]
        P = L
    if world == 1 and not a.timed_only:
        # model-only rate = the reference's own timing window (test.py:399-402): KNN pyramids pre-built and passed in
        e0 = eng.engines[0] if hasattr(eng, "engines") else eng
        half = (P + 1) // 2
        pyr = {}
        for side, pts in (("src", src), ("ref", ref)):
            parts = [e0.knn_pyramid(pts[b:b + half]) for b in range(0, P, half)]   # 2 calls: stays inside the workspace
            for k, j in (("xyz", 0), ("neigh_idx", 1), ("sub_idx", 2), ("interp_idx", 3)):
                pyr[f"points_{side}_{k}"] = torch.cat([q[j] for q in parts], 0)
        for _ in range(2):
            eng.register(src, ref, n_iter, want_aux=False, sync=False, out={"transforms": out_buf["transforms"]}, pyramids=pyr)
        eng.sync()
        reps = max(3, min(a.steps, 8))
        t1 = time.perf_counter()
        for _ in range(reps):
            eng.register(src, ref, n_iter, want_aux=False, sync=False, out={"transforms": out_buf["transforms"]}, pyramids=pyr)
        eng.sync()
        model_only = {"value": round(P * reps / (time.perf_counter() - t1), 3), "unit": "pairs/s",
                      "note": "KNN pyramids supplied by the caller (reference timing window, test.py:399-402)"}
        del pyr

        # companion: the same step with the exhaustive exact-fp32 arg-min kernel throughout (same bits).  How much of
        # `value` survives a descriptor distribution the screening cannot thin out (its floor).
        if screened and not a.no_companion:
            eng.enable_screen(False)
            step(); eng.sync()
            reps = max(2, min(a.steps, 4))
            t1 = time.perf_counter()
            for _ in range(reps):
                step()
            eng.sync()
            companion = {"value": round(P * reps / (time.perf_counter() - t1), 3), "unit": "pairs/s", "steps": reps,
                         "note": "dsir_enable_screen(0): every arg-min by the exhaustive exact-fp32 MFMA kernel (csrc/nn_match.hip); "
                                 "same results bit for bit"}
            eng.enable_screen(True)

        # the screening under descriptor distributions it cannot thin out as well: the same step with other seeded weight
        # variants (companions, never `value`): 'separated' (descriptor head scaled up) and 'clustered:S' (descriptor head scaled
        # down by S under a fixed bias: all descriptors crowd around one direction, top-2 gaps shrink ~S^2 - 2048-point pairs on
        # the CPU oracle: median gap 9e-3 plain, 1.5e-3 at S = 0.03, 1.9e-5 at S = 0.003, where 97 % of the rows fall inside
        # the screening's bound width).  exhaustive_argmin is the floor.
        if screened and not a.no_companion and variant == "plain":
            stress = {}
            for var in ("separated", "clustered:0.03", "clustered:0.01", "clustered:0.003"):
                eng.load_state_dict(generate_state_dict(cfg, 0, var))
                step(); eng.sync()
                eng.screen_stats(reset=True)
                reps = max(2, min(a.steps, 3))
                t1 = time.perf_counter()
                for _ in range(reps):
                    step()
                eng.sync()
                t1 = time.perf_counter() - t1
                st_ = eng.screen_stats(reset=True)
                stress[var] = {"value": round(P * reps / t1, 1), "unit": "pairs/s",
                               "undecided_row_rate": round(st_["rows_undecided"] / max(st_["rows_searched"], 1), 5),
                               "pairs_searched_exhaustively": st_["pairs_exhaustive"], "pair_searches": st_["screened_searches"] * P_launch}
            eng.load_state_dict(sd)
            step(); eng.sync()          # out_buf holds the headline configuration's results again (parity_check compares with them)
            stress["note"] = ("same step, other seeded weight variants (deepsir_amd/weights.py): how far `value` falls towards "
                              "`exhaustive_argmin` as descriptors cluster; results stay exact in every case (undecided rows go to the "
                              "exact fp32 kernel)")
            line_stress = stress
        else:
            line_stress = None

        # batch-1 latency (the reference's own evaluation mode, test.py:56 BATCH_SIZE = 1): one pair in flight,
        # launch sequence replayed from a hipGraph.  Reported beside the throughput number, not as `value`.
        if not a.no_latency:
            s1, r1 = src[:1].contiguous(), ref[:1].contiguous()
            eng.enable_graph(True)
            o1 = eng.register(s1, r1, n_iter, want_aux=False)
            for _ in range(3):
                eng.register(s1, r1, n_iter, want_aux=False, sync=False, out={"transforms": o1["transforms"]})
            eng.sync()
            reps = 20
            t1 = time.perf_counter()
            for _ in range(reps):
                eng.register(s1, r1, n_iter, want_aux=False, sync=False, out={"transforms": o1["transforms"]})
            eng.sync()
            ms = (time.perf_counter() - t1) / reps * 1e3
            e0 = eng.engines[0] if hasattr(eng, "engines") else eng
            census = e0.graph_stats()
            # the same registration with the deep-level walker (csrc/walk.hip: levels 2 / 3 + mlp_mid + two decoder blocks of every pass as
            # one persistent launch; off by default): fewer launches, same bits, and what it costs
            e0.enable_walk(True)
            ow = {"transforms": torch.empty_like(o1["transforms"])}
            for _ in range(3):
                eng.register(s1, r1, n_iter, want_aux=False, sync=False, out=ow)
            eng.sync()
            t1 = time.perf_counter()
            for _ in range(reps):
                eng.register(s1, r1, n_iter, want_aux=False, sync=False, out=ow)
            eng.sync()
            ms_w = (time.perf_counter() - t1) / reps * 1e3
            census_w = e0.graph_stats()
            same_w = bool(torch.equal(ow["transforms"], o1["transforms"]))
            e0.enable_walk(False)
            eng.enable_graph(False)
            latency = {"pairs_in_flight": 1, "ms_per_pair": round(ms, 4), "pairs_per_s": round(1e3 / ms, 2), "hipgraph": True,
                       "launches_per_registration": census,
                       "deep_level_walker": {"ms_per_pair": round(ms_w, 4), "launches_per_registration": census_w, "equal_bits": same_w,
                                             "note": "dsir_enable_walk(1): 114 launches fewer, slower - a launch boundary costs ~1.5 us, a phase "
                                                     "hand-off ~2 us, and the tiles' own latency chains are what the time is made of (csrc/walk.hip)"},
                       "note": "the reference's own evaluation mode (test.py:56 BATCH_SIZE = 1, BASELINE configs[1] 'batch=1'); "
                               "launches_per_registration: node census of the captured graph (dsir_graph_stats)"}

        # K single-pair registrations in flight (deepsir_amd/serve.py): what a test.py-style caller that feeds one pair per call
        # AHEAD of the results gets - requests coalesced into hipGraph-replayed batches of K / engines on the engines in turn
        if not a.no_latency and N <= 16384:
            from deepsir_amd.serve import PairServer
            serving = {}
            for K_, E_ in ((2, 1), (4, 1), (8, 1), (8, 2), (16, 2)):
                srv = PairServer(cfg, sd, dev_index, max_points=N, max_in_flight=K_, engines=E_, n_iter=n_iter, want_aux=False)
                nreq = 128
                reqs = [(src[i % L], ref[i % L]) for i in range(nreq)]
                srv.run_closed_loop(reqs[: 2 * K_], K_)               # captures the graphs
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                res_ = srv.run_closed_loop(reqs, K_)
                torch.cuda.synchronize()
                t1 = time.perf_counter() - t1
                same = all(torch.equal(res_[i]["transforms"], out_buf["transforms"][i % L]) for i in range(nreq))
                serving[f"K{K_}_engines{E_}"] = {"pairs_in_flight": K_, "engines": E_, "pairs_per_batch": srv.max_batch, "pairs_per_s": round(nreq / t1, 1), "requests": nreq,
                                     "batches": srv.batches_dispatched, "equal_to_batched_results": bool(same)}
                srv.close()
            best = max((v for k, v in serving.items() if v["pairs_in_flight"] == 8), key=lambda v: v["pairs_per_s"])
            serving["K8"] = {"pairs_per_s": best["pairs_per_s"], "engines": best["engines"]}
            serving["note"] = ("closed loop: K single-pair requests outstanding, the next one submitted when the oldest result is collected; "
                               "same bits as the batched path (equal_to_batched_results compares every pose with the timed run's)")
            concurrent_single = serving
            # pairs/s against pairs in flight: the same pool, P pairs per step (hipGraph replay up to 16 pairs per engine)
            curve = {}
            for p_ in (1, 2, 4, 8, 16, 32, 64, 128, 256):
                if p_ > P:
                    break
                eng.enable_graph(p_ <= 32)
                o_ = {"transforms": out_all[:p_]}
                for _ in range(2):
                    eng.register(src[:p_], ref[:p_], n_iter, want_aux=False, sync=False, out=o_)
                eng.sync()
                reps = 20 if p_ <= 8 else (8 if p_ <= 64 else 4)
                t1 = time.perf_counter()
                for _ in range(reps):
                    eng.register(src[:p_], ref[:p_], n_iter, want_aux=False, sync=False, out=o_)
                eng.sync()
                curve[str(p_)] = round(p_ * reps / (time.perf_counter() - t1), 1)
            eng.enable_graph(False)
            throughput_curve = {"pairs_per_s_by_pairs_in_flight": curve,
                                "note": f"one engine call of P pairs per step on the benchmarked pool ({S} streams: ceil(P / {S}) pairs per engine), "
                                        "hipGraph replay up to 32 pairs per step; where a caller with B pairs per call lands between batch1_latency and `value`"}

        # BASELINE configs[3] at N = 1: the 3DMatch test set's 1623 pairs as ONE rank's shard, walked in ragged engine calls
        # (6 x 256 + 87), one padded all_gather per pass - the strong-scaling pass of --total-pairs on a single GPU
        if not a.no_companion and not strong and N == 5000 and a.shape == "3dmatch" and not a.partial_overlap and P >= 64:
            T4 = 1623
            b4 = make_batch(N, [50_000 + i for i in range(T4)], cfg.feat_len, a.shape, False)
            s4, r4 = torch.from_numpy(b4["points_src"]).to(dev), torch.from_numpy(b4["points_ref"]).to(dev)
            o4 = torch.empty((T4, n_iter, 3, 4), dtype=torch.float32, device=dev)
            calls4 = [(c0, min(c0 + P, T4)) for c0 in range(0, T4, P)]

            def pass4():
                for c0, c1 in calls4:
                    eng.register(s4[c0:c1], r4[c0:c1], n_iter, want_aux=False, sync=False, out={"transforms": o4[c0:c1]})
                eng.sync()
                g4 = gather_results(o4, dist, sizes=[T4])
                torch.cuda.synchronize()
                return g4
            pass4()
            t1 = time.perf_counter()
            for _ in range(2):
                g4 = pass4()
            t1 = (time.perf_counter() - t1) / 2
            assert g4.shape[0] == T4 and torch.isfinite(g4).all()
            config4 = {"total_pairs": T4, "engine_calls_per_pass": len(calls4), "pairs_per_s": round(T4 / t1, 1), "ms_per_pass": round(t1 * 1e3, 2),
                       "note": "BASELINE configs[3] on one GPU: the 1623-pair set walked in engine calls of at most "
                               f"{P} pairs (the last one ragged), results gathered once per pass; the 8-GPU form is bench.py --gpus 8 --total-pairs 1623"}
            del s4, r4, o4, b4

        # BASELINE configs[2] / configs[4] at N = 1 (KITTI-shaped 16384-point pairs; 65536-point partial-overlap pairs): a few steps each on
        # engines of their own, so that the large configurations are measured by whoever runs this file, not only by the builder
        # (full runs: `bench.py --points 16384 --feat-len 4 --shape kitti --pairs 128 --streams 4`, `--points 65536 --partial-overlap --pairs 16`)
        if not a.no_companion and not strong and world == 1 and N == 5000 and a.shape == "3dmatch" and not a.partial_overlap and P >= 64:
            large_cfgs = {}
            for key, (n_, fl_, shape_, part_, p_, s_, reps_) in {"c3_16384": (16384, 4, "kitti", False, 64, 2, 3),
                                                                 "c5_65536": (65536, 3, "3dmatch", True, 8, 2, 2)}.items():
                t_set = time.perf_counter()
                cfg_ = NetConfig(feat_len=fl_)
                eng_ = EnginePool(cfg_, dev_index, max_points=n_, max_pairs=p_, streams=s_)
                eng_.load_state_dict(generate_state_dict(cfg_, 0, variant))
                b_ = make_batch(n_, [70_000 + i for i in range(min(p_, 8))], fl_, shape_, part_)        # 8 distinct pairs, tiled: generation is host time
                reps_p = (p_ + b_["points_src"].shape[0] - 1) // b_["points_src"].shape[0]
                s_l = torch.from_numpy(np.concatenate([b_["points_src"]] * reps_p)[:p_]).to(dev)
                r_l = torch.from_numpy(np.concatenate([b_["points_ref"]] * reps_p)[:p_]).to(dev)
                o_l = torch.empty((p_, n_iter, 3, 4), dtype=torch.float32, device=dev)
                eng_.register(s_l, r_l, n_iter, want_aux=False, sync=False, out={"transforms": o_l}); eng_.sync()
                t1 = time.perf_counter()
                for _ in range(reps_):
                    eng_.register(s_l, r_l, n_iter, want_aux=False, sync=False, out={"transforms": o_l})
                eng_.sync()
                t1 = time.perf_counter() - t1
                gt_ = torch.from_numpy(np.concatenate([b_["transform_gt"]] * reps_p)[:p_]).to(dev)
                err_ = float((o_l[:, -1] - gt_).abs().max())
                large_cfgs[key] = {"points": n_, "pairs_per_step": p_, "streams": s_, "steps": reps_, "pairs_per_s": round(p_ * reps_ / t1, 1),
                                   "ms_per_step": round(t1 / reps_ * 1e3, 2), "finite": bool(torch.isfinite(o_l).all()),
                                   "max_abs_pose_minus_gt": round(err_, 4), "setup_s": round(time.perf_counter() - t_set - t1, 1)}
                eng_.close()
                del s_l, r_l, o_l, eng_
            large_cfgs["note"] = ("BASELINE configs[2] (KITTI-shaped, feat_len 4) and configs[4] (50 % overlap crops + jitter) on one GPU, short runs "
                                  "beside the headline (never `value`); parity of these shapes: tests/test_gpu_large_configs.py; seeded random weights "
                                  "do not register (max_abs_pose_minus_gt is informational)")

        # the training step of the same pipeline, reported beside the headline, never as `value`
        if world == 1 and not a.no_latency and not a.no_companion and N <= 16384:
            try:
                from deepsir_amd.train import AlignTrainStep, RandlaTrainer
                tp = min(8, P)
                teng = Engine(cfg, dev_index, max_points=N, max_pairs=tp)
                teng.load_state_dict(sd)
                ts, trf = src[:tp].contiguous(), ref[:tp].contiguous()
                tgt = torch.from_numpy(np.ascontiguousarray(batch["transform_gt"][:tp], dtype=np.float32)).to(dev)
                sx, sn, ss, si = teng.knn_pyramid(ts)
                tb = {"points_src": ts, "points_ref": trf, "src_xyz": sx, "src_neigh": sn, "src_sub": ss, "src_interp": si}
                trn = RandlaTrainer(cfg, sd, "inlier_model", 6, 1, dev)
                stepper = AlignTrainStep(teng, trn, tp, N, N, n_iter)
                tres = teng.register(ts, trf, n_iter)
                tl = []
                for s_ in range(5):
                    if s_ == 2:
                        torch.cuda.synchronize(); t1 = time.perf_counter()
                    tl.append(stepper.step(tb, tres, tgt, lr=1e-3, dropout_seed=s_)["losses"]["total"])
                torch.cuda.synchronize()
                tms = (time.perf_counter() - t1) / 3 * 1e3
                training = {"pipeline": "align", "pairs": tp, "ms_per_step": round(tms, 2), "pairs_per_s": round(tp / tms * 1e3, 1),
                            "loss_first_last": [round(tl[0], 5), round(tl[-1], 5)],
                            "note": "one optimisation step of the inlier model without autograd: n_iter training-mode forwards, "
                                    "ScanAlignmentLoss + gradient, n_iter backwards, Adam (deepsir_amd/train.py; hipGraph replay); "
                                    "frozen half from one inference pass"}
                teng.close()
            except Exception as e:  # the headline must not depend on the companion
                training = {"error": f"{type(e).__name__}: {e}"[:300]}

        # parity of what was just measured: the benchmarked configuration once more with the aux outputs, two pairs
        # (first stream's first, last stream's last) handed to the CPU oracle in cpu_leg
        if not a.no_cpu_baseline and N <= 16384:
            aux = eng.register(src, ref, n_iter, want_aux=True, want_desc=True)
            assert torch.equal(aux["transforms"], out_buf["transforms"]), "aux and timed runs disagree"
            for p in sorted({0, P - 1}):
                raw_p = make_pair(N, seeds[p], cfg.feat_len, a.shape, a.partial_overlap)
                assert np.array_equal(raw_p["points_src"][0], batch["points_src"][p])
                checks.append((raw_p, aux["idx"][:, p].cpu(), aux["transforms"][p].cpu().numpy(), aux["desc_src"][:, p].cpu(),
                               aux["desc_ref"][p].cpu()))
            del aux

    if rank == 0:
        total_pairs = pairs_per_step * a.steps
        wl = {"3dmatch": "3DMatch-shaped pairs, uniform [0,3]^3 m clouds", "kitti": "KITTI-shaped pairs, uniform [-50,50]^2 x [-3,3] m clouds"}[a.shape]
        cname = "C2" if (N == 5000 and a.shape == "3dmatch") else ("C3" if a.shape == "kitti" else ("C5" if a.partial_overlap else "C1" if N == 2048 else "custom"))
        line = {
            "metric": "registered pairs/sec (5k-pt 3DMatch-shaped synthetic pairs, 5 registration iterations, KNN pyramid included)",
            "value": round(total_pairs / dt, 3), "unit": "pairs/s", "n_gpus": n_gpus, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "dtype": "f32 (GEMMs as fp16-pair MFMA, fp32 accumulate; index decisions exact fp32)", "data": "synthetic",
            "config": {"workload": (f"{cname}: {wl}{', 50 % overlap crops + jitter' if a.partial_overlap else ''}, random SO(3)+t, raw clouds "
                                    f"resident in HBM -> (R,t) in HBM; THROUGHPUT mode: {P} pairs in flight per GPU per step on {S} HIP "
                                    f"streams ({P_launch} pairs per engine call) - the reference evaluates one pair at a time "
                                    f"(test.py:56), see batch1_latency for that mode"),
                       "points_per_cloud": N, "pairs_per_step_per_gpu": int(per_rank[0][0]), "pairs_in_flight_per_gpu": P, "streams_per_gpu": S,
                       "num_reg_iter": n_iter, "knn": 16, "world_size": world,
                       "weights": f"seeded random state-dict, variant '{variant}' (checkpoint not available)",
                       "parallelism": (f"pair-sharded x{world}, NO collective executed (process group not initialised: {dist_error})" if dist is None else
                                       f"pair-sharded x{world}, {'RCCL' if dist.get_backend() == 'nccl' else dist.get_backend() + ' (rehearsal)'} "
                                       f"all_gather of results (executed through a {world}-rank communicator; ranks[].gather_ms is its latency)")},
        }
        if strong:
            line["config"].update({"total_pairs": a.total_pairs, "pairs_per_engine_call": a.pairs,
                                   "sharding": "contiguous blocks (deepsir_amd/dist.py::shard_range), one padded all_gather of the results per pass; "
                                               "a step = one pass over the whole fixed set (BASELINE configs[3]: the 3DMatch test set has 1623 pairs)"})
            line["config"]["workload"] = line["config"]["workload"].replace("THROUGHPUT mode:", f"STRONG scaling over a fixed set of {a.total_pairs} pairs:")
        line["ranks"] = [{"rank": r, "pairs_per_step": int(v[0]), "ms_per_step": round(v[1], 3), "gather_ms": round(v[2], 4)}
                         for r, v in enumerate(per_rank)]
        # ---- roofline of the dominant kernel
        roof = {"bound": "mfma", "unit": "TFLOP/s"}
        if single is not None:
            op_ms, k_ms, nl, kept, unpruned = single
            if screened:
                ex = screen_flops(P_launch, N, N)
                pruned = None
                if unpruned > 0:
                    # long ref ranges: the launches visit only the listed (row block, column tile) products -
                    # EXECUTED flops per launch = the dense figure x the share of the products the timed launches visited
                    n_pruned = nl                                  # every iteration's launch walks tile lists
                    share = kept / unpruned
                    pruned = {"launches_pruned": n_pruned, "of_launches": nl, "tile_products_visited": kept, "tile_products_unpruned": unpruned,
                              "executed_share_of_dense_flops": round(share, 4), "dense_flops_per_launch": ex,
                              "note": "pruned search (csrc/nn_prune.hip): products that cannot hold a row's arg-min are skipped; same bits as the "
                                      "unpruned search (tests/test_gpu_large_configs.py)"}
                    ex = ex * share
                roof.update({
                    "kernel": "screen_kernel<4,8> (csrc/nn_screen.hip): fp16-split MFMA screening of the 64-channel descriptor arg-min "
                              "under a rigorous bound; the exact fp32 decision among the survivors follows in exact_pick_kernel / nn_match_kernel",
                    "dtype": "f16 in, f32 accumulate (v_mfma_f32_16x16x32_f16)",
                    # SURVEY 8d's rule: `achieved` / `frac` price the ALGORITHMIC flops of the operation the kernel serves (the reference
                    # formulation's 131 J K per pair) over the kernel's duration; what the kernel EXECUTES (three fp16 products per
                    # fp32 product) is reported beside it as achieved_executed / frac_executed
                    "achieved": round(match_flops(P_launch, N, N) / (k_ms / 1e3) / 1e12, 3), "peak": PEAK_F16_MFMA_TFLOPS,
                    "frac": round(match_flops(P_launch, N, N) / (k_ms / 1e3) / 1e12 / PEAK_F16_MFMA_TFLOPS, 4),
                    "flops_per_launch": match_flops(P_launch, N, N),
                    "flops_note": "ALGORITHMIC flops (SURVEY 8d): the reference formulation's 131 J K per pair of the search this kernel serves",
                    "achieved_executed": round(ex / (k_ms / 1e3) / 1e12, 3),
                    "frac_executed": round(ex / (k_ms / 1e3) / 1e12 / PEAK_F16_MFMA_TFLOPS, 4),
                    "flops_executed_per_launch": ex,
                    "flops_executed_note": "EXECUTED MFMA flops: (ah.bh + ah.bl + al.bh) x 2 x 64 per (row, column) = 384 J K per pair"
                                           + (" x the share of the products the launches visit (`pruned`)" if pruned else "")
                                           + ": two thirds of the MFMA work the kernel issues is the price of carrying fp32 operands as fp16 pairs; "
                                             "coarser screenings that would issue less were measured and are not selective enough "
                                             "(profiles/README.md, tools/survivor_stats.py)",
                    "sustained_note": "the bare chain of this kernel's MFMAs (ablation build, no ranking / LDS reads / staging) sustains 1.35 PFLOP/s "
                                      "= 0.54 of `peak` on the same descriptor data: the chip lowers its clock under a dense matrix stream "
                                      "(profiles/README.md); `frac` is quoted against the spec peak all the same"})
            else:
                ex = match_flops(P_launch, N, N)
                roof.update({
                    "kernel": "nn_match_kernel (csrc/nn_match.hip): exhaustive exact-fp32 MFMA distance GEMM + row arg-min",
                    "dtype": "f32 (v_mfma_f32_16x16x4_f32)",
                    "achieved": round(ex / (k_ms / 1e3) / 1e12, 3), "peak": PEAK_F32_MFMA_TFLOPS,
                    "frac": round(ex / (k_ms / 1e3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                    "flops_per_launch": ex, "flops_note": "algorithmic = executed: 131 J K per pair (SURVEY 8d)"})
            if screened and pruned:
                roof["pruned"] = pruned
            traffic = None
            pmc = os.path.join(ROOT, "profiles", "nn_match_pmc.json")
            if os.path.exists(pmc):
                try:
                    with open(pmc) as f:
                        j = json.load(f)
                    if j.get("pairs") == P_launch and j.get("points") == N and screened:
                        traffic = j.get("hbm_bytes_per_launch")
                except Exception:
                    traffic = None
            alg = match_flops(P_launch, N, N)
            roof.update({
                "traffic": traffic, "launches": nl, "avg_launch_ms": round(k_ms, 5), "pairs_per_launch": P_launch,
                "operation": {"avg_ms": round(op_ms, 5), "algorithmic_flops_per_launch": alg,
                              "algorithmic_tflops": round(alg / (op_ms / 1e3) / 1e12, 3),
                              "algorithmic_speedup_vs_fp32_peak": round(alg / (op_ms / 1e3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                              "note": "every kernel of one nearest-descriptor search (split, screening, exact pick, fallback, unpack); "
                                      "algorithmic = the exhaustive fp32 formulation's 131 J K per pair (SURVEY 8d) / that time, against "
                                      "the fp32 MFMA peak it would be bound by - a speed-up over that formulation's ceiling, NOT a roofline fraction"},
                "concurrent": {"streams": S, "launches": int(c_n), "avg_launch_ms": round(c_k_ms / max(c_n, 1), 5),
                               "avg_operation_ms": round(c_op_ms / max(c_n, 1), 5),
                               "note": "the same brackets inside the timed region, where `streams` engines share the GPU"},
                "note": "achieved = ALGORITHMIC flops of the dominant kernel's operation per launch (SURVEY 8d) / the kernel's average duration "
                        "(HIP events on the engine's stream around that kernel alone, ONE engine registering its share of the batch; rocprofv3 "
                        "--kernel-trace of this command agrees, profiles/), against the dense MFMA peak of the dtype it issues; *_executed: "
                        "the flops the kernel actually issues over the same duration"})
            if single_ex is not None:
                op2, k2, n2 = single_ex[:3]
                roof["exhaustive_kernel"] = {
                    "kernel": "nn_match_kernel<2> (csrc/nn_match.hip), dsir_enable_screen(0)", "dtype": "f32", "avg_launch_ms": round(k2, 5),
                    "achieved": round(alg / (k2 / 1e3) / 1e12, 3), "peak": PEAK_F32_MFMA_TFLOPS,
                    "frac": round(alg / (k2 / 1e3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4), "launches": n2}
        roof["whole_path"] = {"flops_per_pair": path_flops(N, n_iter),
                              "achieved": round(path_flops(N, n_iter) * total_pairs / dt / 1e12, 3),
                              "algorithmic_speedup_vs_fp32_peak": round(path_flops(N, n_iter) * total_pairs / dt / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                              "note": "algorithmic FLOPs of the entire job (SURVEY 8d formula, reference formulation, nothing subtracted for "
                                      "hoisting / screening) / wall time; part of the work runs on fp16 MFMA, so this is not a fraction of one peak"}
        # whole-step HBM traffic from the PMC passes of the same command (profiles/README.md), divided by the live step time
        hb = os.path.join(ROOT, "profiles", "step_hbm.json")
        if os.path.exists(hb):
            try:
                with open(hb) as f:
                    j = json.load(f)
                if (j.get("pairs"), j.get("points"), j.get("streams"), j.get("iters")) == (P, N, S, n_iter):
                    gbps = j["hbm_bytes_per_step"] / (dt / a.steps) / 1e9
                    roof["hbm_gbps"] = {"value": round(gbps, 1), "peak": PEAK_HBM_GBPS, "frac": round(gbps / PEAK_HBM_GBPS, 4),
                                        "bytes_per_step": j["hbm_bytes_per_step"], "source": j.get("method")}
            except Exception:
                pass
        # the timed step against BOTH rooflines: its algorithmic flops (the whole path) over the live step time against the fp16
        # matrix peak most of them run on, and its HBM-side bytes (PMC passes of the same command) over the same time against 8 TB/s
        ws = {"algorithmic_tflops": round(path_flops(N, n_iter) * total_pairs / dt / 1e12, 2),
              "algorithmic_frac_of_fp16_mfma_peak": round(path_flops(N, n_iter) * total_pairs / dt / 1e12 / PEAK_F16_MFMA_TFLOPS, 4),
              "hbm_tbps": None, "hbm_frac_of_8tbps": None,
              "note": "the step as a whole sits far below either roofline: it is a chain of ~300 dependent launches per engine whose kernels "
                      "run 2 - 3 waves per SIMD at the latency of their own load chains (profiles/README.md, round 5)"}
        if "hbm_gbps" in roof:
            ws["hbm_tbps"] = round(roof["hbm_gbps"]["value"] / 1e3, 3)
            ws["hbm_frac_of_8tbps"] = roof["hbm_gbps"]["frac"]
            ws["hbm_bytes_per_pair"] = round(roof["hbm_gbps"]["bytes_per_step"] / max(P, 1))
        roof["whole_step"] = ws
        # the step's top kernels with their own rooflines (tools/kernel_table.py over the committed rocprofv3 summaries of this build)
        kt = os.path.join(ROOT, "profiles", "kernel_table.json")
        if os.path.exists(kt) and (P_launch, N, n_iter) == (128, 5000, 5):
            try:
                with open(kt) as f:
                    j = json.load(f)
                roof["kernels"] = j["kernels"]
                roof["kernels_note"] = j.get("note")
            except Exception:
                pass
        line["roofline"] = roof
        if sstats["rows_searched"]:
            line["screening"] = {"searches": sstats["screened_searches"], "undecided_row_rate": round(sstats["rows_undecided"] / sstats["rows_searched"], 5),
                                 "pairs_searched_exhaustively": sstats["pairs_exhaustive"],
                                 "pair_searches": sstats["screened_searches"] * P_launch,
                                 "note": "rows the fp16 screening could not decide (searched by the exact fp32 kernel) / rows searched, inside "
                                         "the timed region; workload dependent (random weights here, checkpoint absent)"}
        if companion is not None:
            line["exhaustive_argmin"] = companion
        if line_stress is not None:
            line["screening_stress"] = line_stress
        if model_only is not None:
            line["model_only"] = model_only
        if latency is not None:
            line["batch1_latency"] = latency
        if concurrent_single is not None:
            line["concurrent_single_pairs"] = concurrent_single
        if throughput_curve is not None:
            line["throughput_curve"] = throughput_curve
        if config4 is not None:
            line["config4_1623_pairs_n1"] = config4
        if training is not None:
            line["training_step"] = training
        if large_cfgs is not None:
            line["large_configs_n1"] = large_cfgs
        if world == 1 and not a.no_cpu_baseline:
            base, parity = cpu_leg(cfg, sd, N, n_iter, checks)
            line["cpu_baseline"] = base
            if parity is not None:
                line["parity_check"] = parity
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()

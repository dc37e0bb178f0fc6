"""The N>1 path on CPU: world_size-2 gloo processes shard a pair list and gather the results."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from deepsir_amd.dist import gather_results, shard_range, shard_sizes


def test_shard_partition():
    for n in (0, 1, 7, 8, 1623):
        for w in (1, 2, 3, 8):
            parts = [list(shard_range(n, r, w)) for r in range(w)]
            assert sum(parts, []) == list(range(n))
            assert max(map(len, parts)) - min(map(len, parts)) <= 1
            assert shard_sizes(n, w) == [len(p) for p in parts]


def test_single_process_gather_is_identity():
    x = torch.arange(24.0).reshape(2, 12)
    assert torch.equal(gather_results(x), x)


def _worker(rank, world, port, n_pairs, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard_range(n_pairs, rank, world)
    # stand-in for the per-pair (R,t) results of this rank's shard: pair id encoded in the payload
    local = torch.stack([torch.full((5, 3, 4), float(i)) for i in mine]) if len(mine) else torch.zeros(0, 5, 3, 4)
    out = gather_results(local, dist, sizes=shard_sizes(n_pairs, world))
    eq = gather_results(torch.full((3, 2), float(rank)), dist)
    if rank == 0:
        q.put((out[:, 0, 0, 0].tolist(), eq[:, 0].tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_shard_and_gather():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    n_pairs = 7   # unequal shards: 4 + 3
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_pairs, q)) for r in range(2)]
    for p in procs:
        p.start()
    ids, eq = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ids == [float(i) for i in range(n_pairs)]
    assert eq == [0.0] * 3 + [1.0] * 3


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_match_single_process(tmp_path):
    """examples/eval_multi_gpu.py with 2 ranks (gloo: both on the one GPU of the test box) and alone: the gathered
    per-pair transforms are identical — sharding changes which process registers a pair, never its result."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = os.path.join(root, "examples", "eval_multi_gpu.py")
    one, two = str(tmp_path / "one.npz"), str(tmp_path / "two.npz")
    common = ["--pairs", "6", "--points", "2048", "--batch", "2", "--iters", "3"]
    r1 = subprocess.run([sys.executable, script] + common + ["--out", one], capture_output=True, text=True, timeout=600)
    assert r1.returncode == 0, r1.stdout + r1.stderr
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                         "127.0.0.1", "--master-port", "29541", script] + common + ["--backend", "gloo", "--out", two],
                        capture_output=True, text=True, timeout=600)
    assert r2.returncode == 0, r2.stdout + r2.stderr
    a, b = np.load(one), np.load(two)
    assert a["pred"].shape == b["pred"].shape == (6, 4, 3, 4)
    assert np.array_equal(a["pred"], b["pred"])
    assert np.array_equal(a["stats"][:, :3], b["stats"][:, :3])


def _world1_worker(backend, port, q):
    """One rank gathering from itself through an initialised process group: the collective itself must run."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if backend == "nccl":
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)   # RCCL, before any other GPU work
    else:
        dev = torch.device("cpu")
        dist.init_process_group("gloo", rank=0, world_size=1)
    calls = []
    real = dist.all_gather
    dist.all_gather = lambda out, x, *a, **k: (calls.append(tuple(x.shape)), real(out, x, *a, **k))[1]
    local = torch.arange(7 * 5 * 12, dtype=torch.float32, device=dev).reshape(7, 5, 3, 4)      # [pairs, n_iter, 3, 4]
    same = gather_results(local, dist)
    padded = gather_results(local[:5], dist, sizes=[5])
    if dev.type == "cuda":
        torch.cuda.synchronize()
    ok = bool(torch.equal(same, local)) and bool(torch.equal(padded, local[:5])) and same.device == local.device
    q.put((ok, calls, dist.get_backend()))
    dist.destroy_process_group()


def _run_world1(backend):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_world1_worker, args=(backend, port, q))
    p.start()
    ok, calls, name = q.get(timeout=300)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert ok and name == backend
    assert calls == [(7, 5, 3, 4), (5, 5, 3, 4)], calls      # both gathers went through the collective


def test_world_size_one_still_runs_the_collective():
    _run_world1("gloo")


@pytest.mark.gpu
def test_rccl_world_size_one_gather():
    """backend "nccl" IS RCCL on ROCm: the padded all_gather of the [pairs, n_iter, 3, 4] results through a one-rank RCCL
    communicator on the test box's GPU returns its input (the library is loaded and the collective executes)."""
    _run_world1("nccl")

"""bench.py's multi-rank plumbing on CPU: `--gpus N` without a torchrun environment re-launches the script as
N ranks, the ranks agree on the world size through a collective, pairs shard in contiguous blocks and the per-pair results
come back through one padded all_gather.  `--dry-run` swaps RCCL for gloo and leaves the engine out."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=timeout)


def _json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_gpus_2_spawns_two_ranks_and_counts_them():
    r = _run(["--gpus", "2", "--dry-run", "--pairs", "5"])
    assert r.returncode == 0, r.stderr[-2000:]
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 2 and j["world_size"] == 2 and j["gather_ok"] and j["pairs_total"] == 13


def test_single_rank_dry_run():
    j = _json_line(_run(["--dry-run", "--pairs", "4"]).stdout)
    assert j["n_gpus"] == 1 and j["gather_ok"]


def test_world_size_mismatch_is_an_error():
    """A rank started by someone else's launcher with a different world size than --gpus says must not print a number."""
    r = _run(["--gpus", "4", "--dry-run"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "launched as 1 ranks" in r.stderr


def test_strong_scaling_dry_run_8_ranks_1623_pairs():
    """BASELINE configs[3]: the 3DMatch test set (1623 pairs) sharded over 8 ranks - unequal shards (7 x 203 + 202), one padded
    all_gather, every pair back in its place.  CPU rehearsal (gloo) of exactly what `bench.py --gpus 8 --total-pairs 1623` does
    around the engine."""
    r = _run(["--gpus", "8", "--dry-run", "--total-pairs", "1623"], timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 8 and j["world_size"] == 8 and j["gather_ok"] and j["pairs_total"] == 1623 and j["scaling"] == "strong"
    assert j["shard_sizes"] == [203] * 7 + [202] and j["calls_per_rank"] == [1] * 8


def test_strong_scaling_dry_run_more_ranks_than_pairs():
    j = _json_line(_run(["--gpus", "3", "--dry-run", "--total-pairs", "2"]).stdout)
    assert j["gather_ok"] and j["shard_sizes"] == [1, 1, 0]


import pytest


@pytest.mark.gpu
def test_strong_scaling_two_ranks_with_the_engine_on_one_gpu():
    """`bench.py --gpus 2 --total-pairs 21 --pairs 8`: a fixed set in unequal shards (11 + 10), each walked in engine calls of
    at most 8 pairs (8 + 3 / 8 + 2: ragged last calls), one padded all_gather per pass, "scaling": "strong", per-rank rows."""
    r = _run(["--gpus", "2", "--total-pairs", "21", "--pairs", "8", "--points", "2048", "--steps", "2", "--warmup", "1", "--streams", "2"],
             {"DSIR_BENCH_BACKEND": "gloo"}, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _json_line(r.stdout)
    assert j["scaling"] == "strong" and j["config"]["total_pairs"] == 21 and j["config"]["world_size"] == 2
    assert [x["pairs_per_step"] for x in j["ranks"]] == [11, 10]
    assert abs(j["value"] * j["ms_per_step"] / 1e3 - 21) < 0.05          # value = the whole set / the slowest rank's pass time


@pytest.mark.gpu
def test_rccl_needs_one_device_per_rank():
    """Two ranks over RCCL on a one-GPU box: refused before any GPU call, with a message (never a hang)."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a single-GPU box")
    r = _run(["--gpus", "2", "--pairs", "2", "--points", "2048", "--steps", "1", "--warmup", "0"], timeout=300)
    assert r.returncode != 0 and "visible devices" in r.stderr


@pytest.mark.gpu
def test_two_ranks_with_the_engine_on_one_gpu():
    """The real multi-rank path on the GPU box: `bench.py --gpus 2` launches itself as two ranks, each with its own
    engine (both on the one visible device, collectives over gloo because RCCL refuses two ranks on one GPU), disjoint
    pair shards, result all_gather, MAX-over-ranks timing.  n_gpus counts distinct devices: 1."""
    r = _run(["--gpus", "2", "--pairs", "16", "--points", "2048", "--steps", "3", "--warmup", "1", "--streams", "1"],
             {"DSIR_BENCH_BACKEND": "gloo"}, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 1 and j["config"]["world_size"] == 2 and j["value"] > 0 and j["scaling"] == "weak"
    assert j["config"]["pairs_per_step_per_gpu"] == 16 and "rehearsal" in j["config"]["parallelism"]

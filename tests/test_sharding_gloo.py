"""world_size-2 test of the N>1 path on CPU (gloo): contiguous sharding of the batch, independent solves, result gather.
The per-shard solver here is the host lane emulator (tests/emu) standing in for the GPU; what is under test is the
sharding / gather plumbing of path_optimizer_2_amd/shard.py that bench.py --gpus N uses with the nccl (RCCL) backend."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, total, n, ret):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import emu_util as E
    from path_optimizer_2_amd.shard import gather_paths, reduce_stats, shard_range
    from path_optimizer_2_amd.synth import make_batch
    first, count = shard_range(total, world, rank)
    b = make_batch(count, n, first_qp=first)                       # counter-based RNG: a shard is regenerated anywhere
    prm = E.production()
    r = E.solve(prm, b["ref"], b["bounds"], b["scal"], passes=1)
    full = gather_paths(torch.from_numpy(r["out"]), total)
    mx, failed = reduce_stats(torch.from_numpy(r["iters"]), int((r["status"] != 1).sum()))
    if rank == 0:
        ret["full"] = full.numpy(); ret["max_iters"] = mx; ret["failed"] = failed
    dist.destroy_process_group()


def test_two_rank_shard_and_gather_equals_single_process():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import emu_util as E
    from path_optimizer_2_amd.shard import shard_range
    from path_optimizer_2_amd.synth import make_batch
    total, n, world = 7, 24, 2          # odd total: ragged shards
    assert [shard_range(total, world, r) for r in range(world)] == [(0, 4), (4, 3)]
    assert [shard_range(5, 8, r)[1] for r in range(8)] == [1, 1, 1, 1, 1, 0, 0, 0]
    E.load()                            # build the emulator before forking
    mgr = mp.Manager(); ret = mgr.dict()
    mp.spawn(_worker, args=(world, 29517, total, n, ret), nprocs=world, join=True)
    b = make_batch(total, n)
    prm = E.production()
    single = E.solve(prm, b["ref"], b["bounds"], b["scal"], passes=1)
    np.testing.assert_array_equal(ret["full"], single["out"])      # bit-identical: sharding changes nothing per QP
    assert ret["failed"] == 0 and ret["max_iters"] == int(single["iters"].max())

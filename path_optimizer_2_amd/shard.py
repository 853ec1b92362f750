"""Multi-GPU plumbing for the batched path QP: the batch of independent QPs is split contiguously
over the ranks (one process per GPU), every rank solves its shard with no data-path collective, and the only
collective is the gather of the result slabs (RCCL over xGMI on GPUs: torch.distributed backend "nccl";
"gloo" in the CPU tests)."""
import torch
import torch.distributed as dist


def shard_range(total, world, rank):
    """Contiguous split: rank g gets QPs [first, first + count).  The first (total % world) ranks get one more."""
    base, extra = divmod(total, world)
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


def gather_paths(out_local, total, group=None):
    """All-gather the per-rank result slabs [count_r][n][7] into [total][n][7] (same on every rank).
    Slabs may differ by one row between ranks; they are padded to the largest for the collective."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n, w = out_local.shape[1], out_local.shape[2]
    counts = [shard_range(total, world, r)[1] for r in range(world)]
    cmax = max(counts)
    pad = torch.zeros((cmax, n, w), dtype=out_local.dtype, device=out_local.device)
    pad[:counts[rank]] = out_local
    slabs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(slabs, pad, group=group)
    return torch.cat([slabs[r][:counts[r]] for r in range(world)], dim=0)


def reduce_stats(iters_local, failed_local, group=None):
    """Scalar statistics over all ranks: (max iterations, number of failed QPs)."""
    dev = iters_local.device
    mx = torch.tensor([int(iters_local.max().item()) if iters_local.numel() else 0], dtype=torch.int64, device=dev)
    fl = torch.tensor([int(failed_local)], dtype=torch.int64, device=dev)
    dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=group)
    dist.all_reduce(fl, op=dist.ReduceOp.SUM, group=group)
    return int(mx.item()), int(fl.item())

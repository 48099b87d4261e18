"""Sharding of a batch of independent (source, sample) pairs over ranks.

Pairs never talk to each other, so the data path has NO collective: rank r owns the block
[start, start+count) of the batch, its own plan, inputs and workspaces.  The only exchange is the
gather of the per-pair results (lag int64, coefficient float64, ret int32 = 20 bytes per pair),
done with torch.distributed (backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in the CPU tests).
"""
import torch
import torch.distributed as dist


def shard_range(total, rank, world):
    """block partition: the first (total % world) ranks get one extra pair -> (start, count)"""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, extra = divmod(int(total), int(world))
    count = base + (1 if rank < extra else 0)
    start = rank * base + min(rank, extra)
    return start, count


def gather_results(lag, coef, ret, total, group=None):
    """all-gather the shard results into full-batch tensors, in pair order.

    lag/coef/ret: this rank's shard (count entries each; count may differ by one across ranks).
    total: batch size over all ranks.  Returns (lag[total], coef[total], ret[total]) on every rank."""
    if not dist.is_available() or not dist.is_initialized():
        return lag, coef, ret
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    base, extra = divmod(int(total), world)
    width = base + (1 if extra else 0)            # every rank sends `width` entries (padded)

    start, count = shard_range(total, rank, world)
    assert lag.numel() == count and coef.numel() == count and ret.numel() == count
    # ONE collective per batch: the three results travel as three float64 columns (a lag is an
    # integer far below 2^53, ret is 0/-1: both exact in float64)
    packed = torch.full((width, 3), float("nan"), dtype=torch.float64, device=coef.device)
    packed[:count, 0] = lag.to(torch.float64)
    packed[:count, 1] = coef
    packed[:count, 2] = ret.to(torch.float64)
    parts = [torch.empty_like(packed) for _ in range(world)]
    dist.all_gather(parts, packed, group=group)
    full = torch.cat([parts[r][: shard_range(total, r, world)[1]] for r in range(world)])
    return full[:, 0].to(torch.int64), full[:, 1].contiguous(), full[:, 2].to(torch.int32)

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
    packed = torch.stack((lag.to(torch.float64), coef, ret.to(torch.float64)), dim=1)
    if count < width:  # uneven split: pad the short shards
        packed = torch.cat((packed, torch.full((width - count, 3), float("nan"), dtype=torch.float64, device=coef.device)))
    if dist.get_backend(group) == "nccl":
        # one flat output buffer: no per-rank output tensors, no copies out of a staging buffer
        flat = torch.empty((world * width, 3), dtype=torch.float64, device=coef.device)
        dist.all_gather_into_tensor(flat, packed, group=group)
        parts = flat.view(world, width, 3)
    else:
        parts = [torch.empty_like(packed) for _ in range(world)]
        dist.all_gather(parts, packed, group=group)
    if extra == 0:
        full = parts.reshape(world * width, 3) if torch.is_tensor(parts) else torch.cat(parts)
    else:
        full = torch.cat([parts[r][: shard_range(total, r, world)[1]] for r in range(world)])
    return full[:, 0].to(torch.int64), full[:, 1].contiguous(), full[:, 2].to(torch.int32)


# ---- zero-copy variant for equal shards -----------------------------------------------------------

def result_buffer(count, device):
    """One byte buffer holding a shard's results back to back -- lag int64[count] | coef float64[count] |
    ret int32[count] -- and the three typed views of it.  Hand the views' pointers to the library;
    gather_result_buffers() then moves the whole shard with ONE collective and no packing kernels."""
    count = int(count)
    buf = torch.zeros(result_bytes(count), dtype=torch.uint8, device=device)
    return buf, _views(buf, count)


def result_bytes(count):
    """size of a result_buffer: 20 bytes per pair, rounded up to 8 so that every rank's slice of the
    gathered buffer can be viewed as int64 / float64"""
    return (20 * int(count) + 7) // 8 * 8


def _views(buf, count):
    lag = buf[: 8 * count].view(torch.int64)
    coef = buf[8 * count: 16 * count].view(torch.float64)
    ret = buf[16 * count: 20 * count].view(torch.int32)
    return lag, coef, ret


def gather_result_buffers(buf, count, out=None, group=None):
    """all-gather result_buffer()s of `count` pairs per rank (equal shards).  Returns (out, [views of
    rank 0, views of rank 1, ...]); `out` (uint8 [world, result_bytes(count)]) may be passed in to avoid the
    allocation.  Asynchronous like any collective on the current stream."""
    if not dist.is_available() or not dist.is_initialized():
        return buf.view(1, -1), [_views(buf, count)]
    world = dist.get_world_size(group)
    if out is None:
        out = torch.empty((world, buf.numel()), dtype=torch.uint8, device=buf.device)
    if dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(out.view(-1), buf, group=group)
    else:
        dist.all_gather([out[r] for r in range(world)], buf, group=group)
    return out, [_views(out[r], count) for r in range(world)]

"""The N>1 path on CPU: world-size-2 (and 3) gloo runs of the shard/gather helpers the
multi-GPU benchmark uses.  The oracle plays the per-rank worker here (test infrastructure)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from util import graft


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_shard_range_partitions_exactly():
    sharding = __import__("importlib").import_module("audiosync_amd.sharding") if graft.load() else None
    for total in (0, 1, 7, 8, 8192, 1000003):
        for world in (1, 2, 3, 8):
            seen = 0
            for r in range(world):
                start, count = sharding.shard_range(total, r, world)
                assert start == seen
                seen += count
            assert seen == total
    with pytest.raises(ValueError):
        sharding.shard_range(10, 2, 2)


def _worker(rank, world, port, total, n, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    graft.load()
    from audiosync_amd import sharding
    start, count = sharding.shard_range(total, rank, world)
    lag = torch.zeros(count, dtype=torch.int64)
    coef = torch.zeros(count, dtype=torch.float64)
    ret = torch.zeros(count, dtype=torch.int32)
    for i in range(count):
        src, smp, _ = oracle.synth_pair(5, start + i, n, 1)
        r, l, c = oracle.cross_correlation(src, smp)
        lag[i], coef[i], ret[i] = l, c, r
    g_lag, g_coef, g_ret = sharding.gather_results(lag, coef, ret, total)
    np.save(os.path.join(out_dir, "lag%d.npy" % rank), g_lag.numpy())
    np.save(os.path.join(out_dir, "coef%d.npy" % rank), g_coef.numpy())
    np.save(os.path.join(out_dir, "ret%d.npy" % rank), g_ret.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("world,total", [(2, 6), (2, 7), (3, 8)])
def test_gather_over_gloo(tmp_path, world, total):
    n = 2000
    port = free_port()
    mp.spawn(_worker, args=(world, port, total, n, str(tmp_path)), nprocs=world, join=True)
    expect = [oracle.cross_correlation(*oracle.synth_pair(5, p, n, 1)[:2]) for p in range(total)]
    for r in range(world):
        lag = np.load(tmp_path / ("lag%d.npy" % r))
        coef = np.load(tmp_path / ("coef%d.npy" % r))
        ret = np.load(tmp_path / ("ret%d.npy" % r))
        assert lag.tolist() == [e[1] for e in expect]          # every rank holds the full batch, in order
        assert ret.tolist() == [e[0] for e in expect]
        assert np.allclose(coef, [e[2] for e in expect], rtol=0, atol=0)


def _worker_buffers(rank, world, port, count, n, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    graft.load()
    from audiosync_amd import sharding
    buf, (lag, coef, ret) = sharding.result_buffer(count, "cpu")     # what bench.py hands to the library
    assert buf.numel() == sharding.result_bytes(count) >= 20 * count and lag.dtype == torch.int64 and coef.dtype == torch.float64 and ret.dtype == torch.int32
    for i in range(count):
        src, smp, _ = oracle.synth_pair(9, rank * count + i, n, 0)
        r, l, c = oracle.cross_correlation(src, smp)
        lag[i], coef[i], ret[i] = l, c, r                            # written through the views = into the buffer
    out, views = sharding.gather_result_buffers(buf, count)
    assert out.shape == (world, sharding.result_bytes(count)) and len(views) == world
    np.save(os.path.join(out_dir, "blag%d.npy" % rank), torch.cat([v[0] for v in views]).numpy())
    np.save(os.path.join(out_dir, "bcoef%d.npy" % rank), torch.cat([v[1] for v in views]).numpy())
    np.save(os.path.join(out_dir, "bret%d.npy" % rank), torch.cat([v[2] for v in views]).numpy())
    dist.destroy_process_group()


def test_byte_buffer_gather_over_gloo(tmp_path):
    """bench.py's gather for equal shards: results of a shard back to back in one byte buffer, one collective."""
    world, count, n = 2, 3, 2000
    port = free_port()
    mp.spawn(_worker_buffers, args=(world, port, count, n, str(tmp_path)), nprocs=world, join=True)
    expect = [oracle.cross_correlation(*oracle.synth_pair(9, p, n, 0)[:2]) for p in range(world * count)]
    for r in range(world):
        assert np.load(tmp_path / ("blag%d.npy" % r)).tolist() == [e[1] for e in expect]
        assert np.load(tmp_path / ("bret%d.npy" % r)).tolist() == [e[0] for e in expect]
        assert np.array_equal(np.load(tmp_path / ("bcoef%d.npy" % r)), np.array([e[2] for e in expect]))


def test_c_abi_partition_matches_the_python_one():
    """asx_shard_range (the block partition asx_xcorr_batch_multi_dev's callers use, include/audiosync/xcorr_hip.h) and
    sharding.shard_range (bench.py, one process per GPU) are the same rule; the shards tile the batch in order."""
    from util import asx
    mod = asx()
    from audiosync_amd import sharding
    for total in (0, 1, 7, 8, 124, 8192, 8193):
        for world in (1, 2, 3, 8):
            nxt = 0
            for r in range(world):
                start, count = mod.shard_range(total, world, r)
                assert (start, count) == sharding.shard_range(total, r, world)
                assert start == nxt
                nxt += count
            assert nxt == total
    assert mod.result_bytes(5) == 104 and mod.result_bytes(6) == 120      # 20 bytes per pair, rounded up to 8
    assert all(mod.result_bytes(w) == sharding.result_bytes(w) for w in range(1, 40))
    with pytest.raises(mod.AsxError):
        mod.shard_range(10, 0, 0)

"""Multi-rank logic on CPU with the gloo backend (world_size 2 and 3): shard indices and the
gather order that the 8-GPU RCCL path relies on."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from decnet_amd.dist import GradBuckets, gather_disparity, shard_batch, shard_range


def test_shard_range_partitions_exactly():
    for n in (1, 7, 8, 32, 33):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [e - s for s, e in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _worker(rank, world, n_pairs, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        H, W = 3, 5
        full = torch.arange(n_pairs * H * W, dtype=torch.float32).reshape(n_pairs, H, W)
        masks = [torch.ones(n_pairs, 2, 2) * torch.arange(n_pairs).view(-1, 1, 1)]
        local, lmasks = shard_batch((full, masks))
        s, e = shard_range(n_pairs, rank, world)
        assert torch.equal(local, full[s:e]) and torch.equal(lmasks[0], masks[0][s:e])
        pred = local * 2 + 1                        # stands in for the per-rank hot path
        got = gather_disparity(pred, n_pairs=n_pairs)
        assert torch.equal(got, full * 2 + 1), "gathered maps are not in pair order"
        got2 = gather_disparity(pred)               # sizes discovered with a small all_gather
        assert torch.equal(got2, full * 2 + 1)
        # bench.py's N > 1 pattern: the gather of step k is waited for only when its buffers are
        # needed again (step k + 2) -- double-buffered sources and receive buffers, async_op=True
        if n_pairs % world == 0:
            srcs = [torch.empty_like(local), torch.empty_like(local)]
            gbuf, pending, seen = [None, None], [None, None], {}
            for k in range(5):
                par = k & 1
                if pending[par] is not None:
                    pending[par].wait()
                    assert torch.equal(gbuf[par], full * (k - 2) + k - 2), "gather %d" % (k - 2)
                    seen[k - 2] = True
                srcs[par].copy_(local * k + k)      # "kernel" of step k
                gbuf[par], pending[par] = gather_disparity(srcs[par], n_pairs=n_pairs, out=gbuf[par],
                                                           async_op=True)
            for par in (0, 1):                      # drain
                pending[par].wait()
            assert torch.equal(gbuf[0], full * 4 + 4) and torch.equal(gbuf[1], full * 3 + 3)
            assert sorted(seen) == [0, 1, 2]
        else:                                       # uneven shards: async falls back to a finished gather
            g3, w3 = gather_disparity(pred, n_pairs=n_pairs, async_op=True)
            assert w3 is None and torch.equal(g3, full * 2 + 1)
        # config 5: bucketed gradient all-reduce, buckets launched one by one between "backward kernels"
        gb = GradBuckets(1000, n_buckets=3)
        assert len(gb) == 3 and gb.bounds[0][0] == 0 and gb.bounds[-1][1] == 1000
        assert all(a[1] == b[0] for a, b in zip(gb.bounds, gb.bounds[1:]))
        for step in range(2):
            for i in range(len(gb)):
                gb.bucket(i).fill_(float(rank + 1 + i + step))      # this rank's gradients of bucket i
                gb.reduce_async(i)
            gb.wait()
            for i in range(len(gb)):
                want = sum(r + 1 + i + step for r in range(world)) / world
                assert torch.allclose(gb.bucket(i), torch.full_like(gb.bucket(i), want)), (i, step)
        ret[rank] = 1
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_pairs,port", [(2, 8, 29611), (2, 5, 29612), (3, 4, 29613)])
def test_shard_and_gather_gloo(world, n_pairs, port):
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, world, n_pairs, port, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert sorted(ret.keys()) == list(range(world))

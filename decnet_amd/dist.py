"""Multi-GPU sharding of independent stereo pairs (one process per GPU, RCCL over xGMI).

The reference's only parallel strategy is batch-split `torch.nn.DataParallel` inside one
process (eval.py:145-146): scatter inputs along B, replicate the 52.7 MB of weights on EVERY
forward, gather `pred [B,H,W]`.  Every op of the hot path is per-sample (the batch index only
offsets pointers, SM_kernel.cu:35), so here rank r simply owns pairs [start, end) of the global
batch, weights are replicated once at start-up, and the single collective per step is one
all-gather of the per-rank disparity maps (7.5 MB per rank for the KITTI config: latency-bound,
far below the xGMI link budget, so no bucketing or ring tuning is needed).
"""
import os

import torch
import torch.distributed as dist


def force_collective():
    """DECNET_FORCE_COLLECTIVE=1 (bench.py / eval --force-collective): run the collectives even in a world of one
    rank, so that the RCCL branch (init_process_group("nccl"), all_gather_into_tensor, the bucketed all_reduce)
    really executes on a single-GPU box; results are the same as without it."""
    return os.environ.get("DECNET_FORCE_COLLECTIVE", "0") == "1"


def shard_range(n_pairs, rank, world):
    """Contiguous, balanced split: the first (n_pairs % world) ranks own one extra pair."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world of %d" % (rank, world))
    base, extra = divmod(int(n_pairs), int(world))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_batch(tensors, rank=None, world=None):
    """Slice this rank's pairs out of global-batch tensors (dim 0).  Lists (mask lists) recurse."""
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0

    def cut(t):
        if isinstance(t, (list, tuple)):
            return type(t)(cut(u) for u in t)
        s, e = shard_range(t.shape[0], rank, world)
        return t[s:e].contiguous()
    return cut(tensors)


def gather_disparity(local, n_pairs=None, group=None, out=None, async_op=False):
    """All-gather per-rank disparity maps [b_r,H,W] into the global [B,H,W] on every rank
    (rank order == pair order).  Uneven shards are padded to the largest for the collective.

    ``out``: optional preallocated [world * b_max, H, W] receive buffer (reused across steps).
    ``async_op=True`` (RCCL, even shards): returns ``(gathered, work)`` without making the current
    stream wait -- the collective then overlaps the next batch's kernels; call ``work.wait()``
    before reading ``gathered`` or reusing ``local`` / ``out``."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force_collective()):
        return (local, None) if async_op else local
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if n_pairs is None:
        counts = [torch.zeros(1, dtype=torch.int64, device=local.device) for _ in range(world)]
        dist.all_gather(counts, torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device),
                        group=group)
        sizes = [int(c.item()) for c in counts]
    else:
        sizes = [shard_range(n_pairs, r, world)[1] - shard_range(n_pairs, r, world)[0]
                 for r in range(world)]
    assert sizes[rank] == local.shape[0], "local shard has %d pairs, expected %d" % (local.shape[0], sizes[rank])
    bmax = max(sizes)
    send = local.contiguous()
    if send.shape[0] < bmax:
        pad = torch.zeros((bmax - send.shape[0],) + tuple(send.shape[1:]), dtype=send.dtype, device=send.device)
        send = torch.cat([send, pad], 0)
    shape = (world * bmax,) + tuple(send.shape[1:])
    if out is None or tuple(out.shape) != shape or out.dtype != send.dtype or out.device != send.device:
        out = torch.empty(shape, dtype=send.dtype, device=send.device)
    even = all(s == bmax for s in sizes)
    if dist.get_backend(group) == "nccl":                    # RCCL
        work = dist.all_gather_into_tensor(out, send, group=group, async_op=bool(async_op and even))
    else:                                                    # gloo (CPU tests)
        parts = list(out.chunk(world, 0))
        work = dist.all_gather(parts, send, group=group, async_op=bool(async_op and even))
    if even:
        return (out, work) if async_op else out
    res = torch.cat([out[r * bmax:r * bmax + sizes[r]] for r in range(world)], 0)
    return (res, None) if async_op else res


class GradBuckets:
    """Bucketed gradient all-reduce for the data-parallel training configuration (BASELINE config 5;
    the reference's analogue is DataParallel's reduce of replica gradients onto GPU 0, eval.py:145-146 /
    sync_batchnorm/batchnorm.py:110-131 for the BN statistics).

    One flat fp32 buffer of ``numel`` gradients (13.19 M parameters = 52.7 MB for the shipped network)
    is cut into ``n_buckets`` contiguous buckets.  ``reduce_async(i)`` enqueues the all-reduce of bucket i
    behind whatever the current stream holds at that moment and returns at once -- RCCL runs it on its own
    stream, so it overlaps the backward kernels enqueued afterwards (xGMI ring: 2(N-1)/N x bucket bytes per
    link, ~13 MB buckets keep each collective in the bandwidth regime without delaying the first one).
    ``wait()`` makes the current stream wait for all of them and turns the sums into means.
    """

    def __init__(self, numel, n_buckets=4, device="cpu", group=None, dtype=torch.float32):
        if n_buckets < 1 or numel < n_buckets:
            raise ValueError("need 1 <= n_buckets <= numel")
        self.group = group
        self.flat = torch.zeros(int(numel), dtype=dtype, device=device)
        per = (int(numel) + n_buckets - 1) // n_buckets
        per = (per + 63) // 64 * 64                         # 256-byte aligned bucket starts
        self.bounds = [(i * per, min(int(numel), (i + 1) * per)) for i in range(n_buckets)
                       if i * per < int(numel)]
        self.pending = {}

    def __len__(self):
        return len(self.bounds)

    def bucket(self, i):
        s, e = self.bounds[i]
        return self.flat[s:e]

    def reduce_async(self, i):
        if i in self.pending:
            raise RuntimeError("bucket %d is already being reduced" % i)
        if not dist.is_initialized() or (dist.get_world_size(self.group) == 1 and not force_collective()):
            self.pending[i] = None
            return
        self.pending[i] = dist.all_reduce(self.bucket(i), op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def wait(self):
        """Wait for every bucket in flight; gradients become the mean over ranks."""
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        for i in sorted(self.pending):
            w = self.pending[i]
            if w is not None:
                w.wait()
            if world > 1:
                self.bucket(i).mul_(1.0 / world)
        self.pending = {}

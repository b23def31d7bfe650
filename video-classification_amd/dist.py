"""Data-parallel gradient exchange: one process per GPU, clip batches sharded across ranks, one all-reduce
(mean) of the flat fp32 gradient arena per step -- issued in arena SEGMENTS as the backward pass finishes
them, on a side stream, so the exchange over xGMI overlaps the rest of backward.

The reference is single-GPU (no torch.distributed anywhere); this is the new K15 row.
Backend: torch.distributed "nccl" (= RCCL on ROCm) on the GPU box, "gloo" in the CPU tests.
BatchNorm statistics stay per-rank (the reference has no SyncBN).
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist
import torch.utils.data

Range = Tuple[int, int]  # (offset, numel) in the arena


def init_process_group_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from torchrun's environment; no-op for a single process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def merge_ranges(ranges: Sequence[Range]) -> List[Range]:
    out: List[Range] = []
    for off, n in sorted(r for r in ranges if r[1] > 0):
        if out and out[-1][0] + out[-1][1] == off:
            out[-1] = (out[-1][0], out[-1][1] + n)
        else:
            assert not out or out[-1][0] + out[-1][1] < off, "overlapping gradient ranges"
            out.append((off, n))
    return out


def split_ranges(ranges: Sequence[Range], max_numel: int) -> List[Range]:
    """Cap every range at max_numel elements (bucket size): xGMI is point-to-point, so a handful of large
    messages per step beats many small ones, but a bucket must not delay the overlap by a whole backward."""
    out: List[Range] = []
    for off, n in ranges:
        while n > max_numel:
            out.append((off, max_numel))
            off, n = off + max_numel, n - max_numel
        if n:
            out.append((off, n))
    return out


class GradReducer:
    """All-reduces ranges of a flat gradient tensor while backward is still running.  `reduce(ranges, ...)` is called once per
    finished backward segment; `finish()` makes the caller's stream wait for all of them.

    No relay stream: a process gets FOUR hardware queues (GPU_MAX_HW_QUEUES) and streams beyond that share one and serialise
    (measured on one MI355X: a fifth active stream costs 2 %, eight queues 28 %).  ProcessGroupNCCL
    (= RCCL) runs every collective on ITS OWN internal stream whatever stream issued it, so that stream is the fourth queue:
    the step keeps three compute lanes when world > 1 (trunk, fast pathway, ONE filter-gradient lane -- `Engine.wgrad_one_lane`,
    measured neutral single-rank) and the buckets are issued with ``async_op=True`` from the filter-gradient lane, which
    first waits (events) for the other producers of the segment.  The lane itself never waits for the collective."""

    def __init__(self, flat_grad: torch.Tensor, group=None, bucket_mb: float = 32.0):
        self.g = flat_grad
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.bucket_numel = max(1, int(bucket_mb * (1 << 20) / flat_grad.element_size()))
        self.cuda = flat_grad.is_cuda
        self.reduced: List[Range] = []
        self.handles: list = []

    @property
    def active(self) -> bool:
        """does the step run its segmented backward (cuts, waits, buckets) for this reducer?"""
        return self.world > 1

    @property
    def grad_scale(self) -> float:
        """what Adam multiplies the exchanged gradients by (SUM -> mean)"""
        return 1.0 / self.world

    def begin(self):
        self.reduced = []
        self.handles = []

    def _order_after(self, issue, producers):
        """`issue` waits for everything issued so far on the producer streams"""
        for s in producers:
            if s is not issue:
                ev = torch.cuda.Event()
                ev.record(s)
                issue.wait_event(ev)

    def reduce(self, ranges: Sequence[Range], producers: Optional[Sequence["torch.cuda.Stream"]] = None, issue_on=None):
        """producers: the streams that wrote these gradient ranges (default: the current stream).  issue_on: the lane the
        buckets are enqueued from (default: the current stream); it waits for the producers, the collective's own stream
        waits for it, and nobody waits for the collective until finish()."""
        if not self.active:
            return
        ranges = split_ranges(merge_ranges(ranges), self.bucket_numel)
        self.reduced += ranges
        if self.cuda:
            cur = torch.cuda.current_stream(self.g.device)
            issue = issue_on if issue_on is not None else cur
            self._order_after(issue, producers or [cur])
            with torch.cuda.stream(issue):
                for off, n in ranges:
                    self.handles.append(dist.all_reduce(self.g[off:off + n], op=dist.ReduceOp.SUM, group=self.group,
                                                        async_op=True))
        else:
            for off, n in ranges:
                dist.all_reduce(self.g[off:off + n], op=dist.ReduceOp.SUM, group=self.group)

    def finish(self) -> float:
        """The caller's stream waits for every bucket.  Returns grad_scale (1/world: SUM -> mean)."""
        for h in self.handles:
            h.wait()                     # NCCL: the CURRENT stream waits for the collective's stream (no host block)
        self.handles = []
        return self.grad_scale


class LoopbackReducer(GradReducer):
    """Single-process REHEARSAL of the overlapped gradient exchange (bench.py --rehearse-comm): the segment cuts, event waits,
    issue lane and bucket sizes of GradReducer at `rehearse_world` ranks, with each bucket's all-reduce replaced by a local
    memory-bound stand-in on ONE extra stream that plays ProcessGroupNCCL's internal stream (a ring all-reduce moves
    2 (world-1)/world of the bucket in and out of HBM: here one read-modify-write of the bucket against a scratch copy).  It
    prices the fourth queue on ONE GPU -- queue sharing, CU and HBM contention -- not the xGMI transfer.  Gradients keep their
    values (x + 0) and `world` stays 1, so Adam's gradient scale is 1 and the rehearsed step trains exactly as the plain one."""

    def __init__(self, flat_grad: torch.Tensor, world: int = 8, bucket_mb: float = 32.0):
        super().__init__(flat_grad, None, bucket_mb)
        self.world = 1
        self.rehearse_world = world
        self.comm_stream = torch.cuda.Stream(flat_grad.device)
        self.scratch = torch.zeros(self.bucket_numel, dtype=flat_grad.dtype, device=flat_grad.device)

    @property
    def active(self) -> bool:
        return True

    def reduce(self, ranges, producers=None, issue_on=None):
        ranges = split_ranges(merge_ranges(ranges), self.bucket_numel)
        self.reduced += ranges
        cur = torch.cuda.current_stream(self.g.device)
        issue = issue_on if issue_on is not None else cur
        self._order_after(issue, producers or [cur])
        ev = torch.cuda.Event()
        ev.record(issue)
        self.comm_stream.wait_event(ev)            # the collective's stream waits for the issuing lane, as RCCL's does
        with torch.cuda.stream(self.comm_stream):
            for off, n in ranges:
                self.g[off:off + n].add_(self.scratch[:n])

    def finish(self) -> float:
        torch.cuda.current_stream(self.g.device).wait_stream(self.comm_stream)
        return self.grad_scale


def shard_indices(num_items: int, rank: int, world: int, epoch_seed: int, shuffle: bool = True,
                  drop_last: bool = True) -> List[int]:
    """Disjoint per-rank index subsets of one shuffled epoch (DistributedSampler semantics on top of the
    reference's DataLoader(shuffle=True, drop_last=True), train.py:164)."""
    g = torch.Generator().manual_seed(epoch_seed)
    order = torch.randperm(num_items, generator=g).tolist() if shuffle else list(range(num_items))
    per = num_items // world if drop_last else (num_items + world - 1) // world
    order = order[: per * world] if drop_last else (order + order[: per * world - num_items])
    return order[rank::world]


class EpochShardSampler(torch.utils.data.Sampler):
    """The train loader's sampler when world > 1: rank r draws shard_indices(len, r, world, seed + epoch) -- ONE
    permutation of the epoch shared by all ranks (same seed), cut into disjoint per-rank subsets of equal length, so with
    the loader's drop_last=True every rank runs the same number of steps (reference train.py:164: shuffle=True,
    drop_last=True on a single process).  Call set_epoch(e) before every epoch."""

    def __init__(self, num_items: int, rank: int, world: int, seed: int = 0, shuffle: bool = True, drop_last: bool = True):
        self.num_items, self.rank, self.world, self.seed = num_items, rank, world, seed
        self.shuffle, self.drop_last = shuffle, drop_last
        self.epoch = 0

    def set_epoch(self, epoch: int):
        self.epoch = int(epoch)

    def indices(self) -> List[int]:
        return shard_indices(self.num_items, self.rank, self.world, self.seed + self.epoch, self.shuffle, self.drop_last)

    def __iter__(self):
        return iter(self.indices())

    def __len__(self):
        per = self.num_items // self.world if self.drop_last else (self.num_items + self.world - 1) // self.world
        return per


class VideoShardSampler(torch.utils.data.Sampler):
    """The test loader's sampler when world > 1: rank r evaluates videos r, r + world, r + 2 world, ... in order (no
    padding, no duplicates: the ranks' counts may differ by one -- there is no collective inside the eval loop)."""

    def __init__(self, num_items: int, rank: int, world: int):
        self.num_items, self.rank, self.world = num_items, rank, world

    def __iter__(self):
        return iter(range(self.rank, self.num_items, self.world))

    def __len__(self):
        return len(range(self.rank, self.num_items, self.world))


def gather_objects(obj, group=None) -> list:
    """[obj of rank 0, obj of rank 1, ...] on every rank (small host objects: eval scores, labels)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return [obj]
    out = [None] * dist.get_world_size(group)
    dist.all_gather_object(out, obj, group=group)
    return out

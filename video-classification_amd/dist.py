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
    """All-reduces ranges of a flat gradient tensor.  `reduce(ranges)` may be called several times per step
    (once per finished backward segment); `finish()` makes the caller's stream wait for all of them."""

    def __init__(self, flat_grad: torch.Tensor, group=None, bucket_mb: float = 32.0):
        self.g = flat_grad
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.bucket_numel = max(1, int(bucket_mb * (1 << 20) / flat_grad.element_size()))
        self.cuda = flat_grad.is_cuda
        self.comm_stream = torch.cuda.Stream(flat_grad.device) if self.cuda else None
        self.reduced: List[Range] = []

    def begin(self):
        self.reduced = []

    def reduce(self, ranges: Sequence[Range], producers: Optional[Sequence["torch.cuda.Stream"]] = None):
        """producers: the streams that wrote these gradient ranges (default: the current stream); the exchange
        starts once everything issued on them so far has finished."""
        if self.world == 1:
            return
        ranges = split_ranges(merge_ranges(ranges), self.bucket_numel)
        self.reduced += ranges
        if self.cuda:
            evs = []
            for s in (producers or [torch.cuda.current_stream(self.g.device)]):
                ev = torch.cuda.Event()
                ev.record(s)
                evs.append(ev)
            with torch.cuda.stream(self.comm_stream):
                for ev in evs:
                    self.comm_stream.wait_event(ev)
                for off, n in ranges:
                    dist.all_reduce(self.g[off:off + n], op=dist.ReduceOp.SUM, group=self.group)
        else:
            for off, n in ranges:
                dist.all_reduce(self.g[off:off + n], op=dist.ReduceOp.SUM, group=self.group)

    def finish(self) -> float:
        """Returns the factor the optimiser must scale gradients by (1/world: SUM -> mean)."""
        if self.world > 1 and self.cuda:
            torch.cuda.current_stream(self.g.device).wait_stream(self.comm_stream)
        return 1.0 / self.world


class LoopbackReducer(GradReducer):
    """Single-process REHEARSAL of the overlapped gradient exchange (bench.py --rehearse-comm): the same segment cuts, event
    waits, comm stream and bucket sizes as GradReducer with world ranks, but each bucket's all-reduce is replaced by a local
    memory-bound stand-in on the comm stream (a ring all-reduce moves 2 (world-1)/world of the bucket in and out of HBM: here
    one read-modify-write of the bucket against a scratch copy).  It measures what the fifth stream costs the step on ONE GPU
    -- queue sharing, CU and HBM contention -- not the xGMI transfer time; gradients are left unchanged in value (x + 0)."""

    def __init__(self, flat_grad: torch.Tensor, world: int = 8, bucket_mb: float = 32.0):
        super().__init__(flat_grad, None, bucket_mb)
        self.world = world
        self.scratch = torch.zeros(self.bucket_numel, dtype=flat_grad.dtype, device=flat_grad.device)

    def reduce(self, ranges, producers=None):
        ranges = split_ranges(merge_ranges(ranges), self.bucket_numel)
        self.reduced += ranges
        evs = []
        for s in (producers or [torch.cuda.current_stream(self.g.device)]):
            ev = torch.cuda.Event()
            ev.record(s)
            evs.append(ev)
        with torch.cuda.stream(self.comm_stream):
            for ev in evs:
                self.comm_stream.wait_event(ev)
            for off, n in ranges:
                self.g[off:off + n].add_(self.scratch[:n])

    def finish(self) -> float:
        torch.cuda.current_stream(self.g.device).wait_stream(self.comm_stream)
        return 1.0


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

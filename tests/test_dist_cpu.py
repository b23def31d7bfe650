"""Multi-rank path on CPU: gloo, world_size 2.  Covers the gradient exchange (segments issued as backward finishes
them, SUM -> mean through Adam's grad_scale), rank sharding of the clip indices, and that two ranks that start from
the same weights and see different clips end the step with IDENTICAL weights equal to the large-batch gradient step."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    from emu_backend import EmuBackend
    from video_classification_amd import arch, dist as sdist
    from video_classification_amd.slowfast import SlowFast
    from video_classification_amd.train import TrainStep
    r, w, _ = sdist.init_process_group_from_env("gloo")
    assert (r, w) == (rank, world)
    spec = arch.ref_spec(num_class=5, depth=18, head_pool_kernels=((2, 1, 1), (2, 1, 1)))
    m = SlowFast(spec, dtype=torch.float32, device="cpu", backend=EmuBackend(), seed=3)   # same seed: same weights
    eng = m.engine
    g = torch.Generator().manual_seed(100)
    clips = torch.randn(world * 2, 4, 21, 32, 32, generator=g)
    labels = torch.randint(0, 5, (world * 2,), generator=g)
    idx = sdist.shard_indices(world * 2, rank, world, epoch_seed=0, shuffle=False)
    x = clips[idx].permute(0, 2, 1, 3, 4)
    red = sdist.GradReducer(eng.G, bucket_mb=0.25)
    step = TrainStep(eng, lr=1e-2, use_graph=False, reducer=red, overlap_segments=4)
    m.train()
    step(x[:, 0:5], x[:, 5:20], labels[idx])
    # every arena element was reduced exactly once
    cover = torch.zeros(eng.arena_numel, dtype=torch.int32)
    for off, n in red.reduced:
        cover[off:off + n] += 1
    assert int(cover.min()) == 1 and int(cover.max()) == 1
    torch.save({"P": eng.P.data.clone(), "G": eng.G.clone(), "idx": idx}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_step_gloo(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    a = torch.load(tmp_path / "rank0.pt")
    b = torch.load(tmp_path / "rank1.pt")
    assert sorted(a["idx"] + b["idx"]) == [0, 1, 2, 3] and not set(a["idx"]) & set(b["idx"])
    assert torch.equal(a["G"], b["G"])          # summed gradients are identical on both ranks
    assert torch.equal(a["P"], b["P"])          # so are the updated weights


def test_range_helpers():
    from video_classification_amd.dist import merge_ranges, shard_indices, split_ranges
    assert merge_ranges([(8, 8), (0, 8), (32, 4)]) == [(0, 16), (32, 4)]
    assert split_ranges([(0, 10)], 4) == [(0, 4), (4, 4), (8, 2)]
    parts = [shard_indices(10, r, 3, epoch_seed=1) for r in range(3)]
    assert all(len(p) == 3 for p in parts) and len(set(sum(parts, []))) == 9

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
    m.train()
    # this rank's own gradient first (no exchange, lr 0), with the dropout seed and BN statistics put back afterwards
    seed0 = eng.drop_seed.clone()
    sd0 = {k: v.clone() for k, v in m.state_dict().items()}
    TrainStep(eng, lr=0.0, use_graph=False)(x[:, 0:5], x[:, 5:20], labels[idx])
    g_local = eng.G.clone()
    eng.drop_seed.copy_(seed0)
    m.load_state_dict(sd0)
    eng.adam_m = eng.adam_v = eng.adam_step = None
    red = sdist.GradReducer(eng.G, bucket_mb=0.25)
    step = TrainStep(eng, lr=1e-2, use_graph=False, reducer=red, overlap_segments=4)
    assert step.world == world
    step(x[:, 0:5], x[:, 5:20], labels[idx])
    # every arena element was reduced exactly once
    cover = torch.zeros(eng.arena_numel, dtype=torch.int32)
    for off, n in red.reduced:
        cover[off:off + n] += 1
    assert int(cover.min()) == 1 and int(cover.max()) == 1
    torch.save({"P": eng.P.data.clone(), "G": eng.G.clone(), "idx": idx, "g_local": g_local, "P0": sd0},
               os.path.join(out_dir, f"rank{rank}.pt"))
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
    # ... and the exchanged gradient is the SUM of what each rank computed alone (Adam then applies 1/world)
    assert float((a["G"] - (a["g_local"] + b["g_local"])).abs().max()) <= 1e-6 * float(a["G"].abs().max())


def test_range_helpers():
    from video_classification_amd.dist import merge_ranges, shard_indices, split_ranges
    assert merge_ranges([(8, 8), (0, 8), (32, 4)]) == [(0, 16), (32, 4)]
    assert split_ranges([(0, 10)], 4) == [(0, 4), (4, 4), (8, 2)]
    parts = [shard_indices(10, r, 3, epoch_seed=1) for r in range(3)]
    assert all(len(p) == 3 for p in parts) and len(set(sum(parts, []))) == 9


# ------------------------------------------------------------------ the TRAINER on two ranks (not just TrainStep)
def _trainer_cfg(root):
    from video_classification_amd.config import get_cfg
    cfg = get_cfg()
    cfg.CHALEARN.ROOT = str(root)
    cfg.CHALEARN.BATCH_SIZE = 2
    cfg.CHALEARN.CLIP_LEN = 4
    cfg.CHALEARN.NUM_CLASS = 5
    cfg.MODEL.NAME = "slowfast-LHand"
    cfg.MODEL.R3D_INPUT = "CropLHand"
    cfg.MODEL.DEPTH = 18
    cfg.MODEL.LR = 1e-3
    cfg.NUM_CPU = 0
    return cfg


def _trainer_sets(cfg):
    from video_classification_amd.train import SyntheticChalearn

    class Logged(SyntheticChalearn):
        seen = None

        def __getitem__(self, i):
            self.seen.append(int(i))
            return super().__getitem__(i)
    tr = Logged(cfg, "train", num_videos=9, seed=1)          # 9 clips, 2 ranks: one is dropped (drop_last semantics)
    tr.seen = []
    te = SyntheticChalearn(cfg, "test", num_videos=5, clips_per_video=(1, 3), seed=2)
    return tr, te


def _trainer_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from emu_backend import EmuBackend
    from video_classification_amd.train import Trainer
    cfg = _trainer_cfg(out_dir)
    tr, te = _trainer_sets(cfg)
    t = Trainer(cfg, train_set=tr, test_set=te, device="cpu", backend=EmuBackend(), dist_backend="gloo")
    assert (t.rank, t.world) == (rank, world)
    ev = t.run_eval()                                        # fresh weights: comparable with the 1-rank run of the parent
    epochs = []
    for e in range(2):
        t.epoch = e
        tr.seen = []
        t.train_epoch()
        epochs.append(list(tr.seen))
    t.save_ckpt(epoch=1, acc=0.5)                            # rank 0 only
    torch.save({"eval": ev, "epochs": epochs, "P": t.model.engine.P.data.clone(),
                "steps": int(t.model.engine.adam_step[0])}, os.path.join(out_dir, f"trainer_rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_trainer_shards_epoch_and_eval_over_two_ranks(tmp_path):
    """/root/reference/train.py:164 semantics (one shuffled epoch, drop_last) cut into disjoint per-rank shards, a new
    permutation per epoch; run_eval sharded by video and gathered: the same dict as the 1-rank run (train.py:287-370);
    one checkpoint, written by rank 0."""
    world, port = 2, _free_port()
    mp.spawn(_trainer_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    a = torch.load(tmp_path / "trainer_rank0.pt", weights_only=False)
    b = torch.load(tmp_path / "trainer_rank1.pt", weights_only=False)
    for e in range(2):
        ia, ib = a["epochs"][e], b["epochs"][e]
        assert len(ia) == len(ib) == 4                       # 9 // 2 clips per rank, batch 2: two steps on each rank
        assert not set(ia) & set(ib) and len(set(ia) | set(ib)) == 8 and set(ia) | set(ib) <= set(range(9))
    assert (a["epochs"][0], b["epochs"][0]) != (a["epochs"][1], b["epochs"][1])      # re-shuffled per epoch
    assert a["steps"] == b["steps"] == 4 and torch.equal(a["P"], b["P"])
    # eval: both ranks hold the full result, identical to a single process over the whole test set
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from emu_backend import EmuBackend
    from video_classification_amd.train import Trainer
    cfg = _trainer_cfg(tmp_path / "single")
    tr, te = _trainer_sets(cfg)
    one = Trainer(cfg, train_set=tr, test_set=te, device="cpu", backend=EmuBackend()).run_eval()
    for r in (a["eval"], b["eval"]):
        assert r["sv"] == one["sv"] == te.nclips and r["acc"] == one["acc"]
        assert (r["t"] == one["t"]).all() and abs(r["ps"] - one["ps"]).max() < 1e-6
    files = os.listdir(tmp_path / "logs" / "checkpoints" / "slowfast-LHand")
    assert files == ["acc0.500_e1.ckpt"]


def _tiny_eval_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from emu_backend import EmuBackend
    from video_classification_amd.train import SyntheticChalearn, Trainer
    cfg = _trainer_cfg(out_dir)
    tr = SyntheticChalearn(cfg, "train", num_videos=4, seed=1)
    te = SyntheticChalearn(cfg, "test", num_videos=1, clips_per_video=(2, 2), seed=2)     # fewer videos than ranks
    t = Trainer(cfg, train_set=tr, test_set=te, device="cpu", backend=EmuBackend(), dist_backend="gloo")
    ev = t.run_eval()                                        # rank 1's shard is EMPTY: it must still reach the gather
    torch.save(ev, os.path.join(out_dir, f"tiny_eval_rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_eval_with_fewer_videos_than_ranks_does_not_hang(tmp_path):
    """VideoShardSampler pads nothing: with one test video and two ranks, rank 1 evaluates no clip at all.  It used to raise in
    torch.cat([]) while rank 0 waited in all_gather_object; now it contributes zero rows and both ranks return the 1-rank dict."""
    world, port = 2, _free_port()
    mp.spawn(_tiny_eval_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    a = torch.load(tmp_path / "tiny_eval_rank0.pt", weights_only=False)
    b = torch.load(tmp_path / "tiny_eval_rank1.pt", weights_only=False)
    assert a["sv"] == b["sv"] == [2] and a["acc"] == b["acc"]
    assert a["ps"].shape == b["ps"].shape == (2, 5) and abs(a["ps"] - b["ps"]).max() == 0
    assert (a["t"] == b["t"]).all()


class _FakeStream:
    def __init__(self, name, log):
        self.name, self.log = name, log

    def wait_event(self, ev):
        self.log.append(("wait", self.name, ev.recorded_on))


def test_grad_reducer_async_handles_and_lane_order(monkeypatch):
    """The CUDA branch of GradReducer without a GPU: torch.cuda's streams / events and dist.all_reduce are stand-ins that log.
    reduce() must make the issue lane wait for EVERY other producer before the first bucket is enqueued, enqueue each bucket
    with async_op=True from the issue lane, and finish() must wait on every Work handle exactly once and then forget them."""
    import contextlib
    from video_classification_amd import dist as sdist
    log = []
    cur = _FakeStream("cur", log)
    active = [cur]

    class Ev:
        def record(self, s):
            self.recorded_on = s.name

    class Work:
        def __init__(self, n):
            self.n, self.waits = n, 0

        def wait(self):
            self.waits += 1
            log.append(("work.wait", self.n))

    works = []

    def fake_all_reduce(t, op=None, group=None, async_op=False):
        log.append(("all_reduce", active[-1].name, int(t.numel()), bool(async_op)))
        w = Work(int(t.numel()))
        works.append(w)
        return w

    @contextlib.contextmanager
    def fake_stream_ctx(s):
        active.append(s)
        try:
            yield
        finally:
            active.pop()

    monkeypatch.setattr(sdist.torch.cuda, "Event", Ev)
    monkeypatch.setattr(sdist.torch.cuda, "current_stream", lambda dev=None: cur)
    monkeypatch.setattr(sdist.torch.cuda, "stream", fake_stream_ctx)
    monkeypatch.setattr(sdist.dist, "all_reduce", fake_all_reduce)
    g = torch.zeros(1000)
    red = sdist.GradReducer(g, bucket_mb=400 * 4 / (1 << 20))           # 400-element buckets
    red.world, red.cuda = 2, True
    red.begin()
    trunk, fast, wg = (_FakeStream(n, log) for n in ("trunk", "fast", "wg"))
    red.reduce([(0, 300), (300, 500)], producers=[trunk, fast, wg], issue_on=wg)      # merged to (0, 800): two buckets
    first_ar = next(i for i, e in enumerate(log) if e[0] == "all_reduce")
    assert sorted(log[:first_ar]) == [("wait", "wg", "fast"), ("wait", "wg", "trunk")]   # not on itself; before any bucket
    assert log[first_ar:] == [("all_reduce", "wg", 400, True), ("all_reduce", "wg", 400, True)]
    assert red.reduced == [(0, 400), (400, 400)] and len(red.handles) == 2
    red.reduce([(800, 200)])                                            # defaults: produced and issued on the current stream
    assert log[-1] == ("all_reduce", "cur", 200, True) and not [e for e in log[first_ar + 2:] if e[0] == "wait"]
    assert all(w.waits == 0 for w in works)                             # nobody waits for a collective before finish()
    assert red.finish() == 0.5
    assert [w.waits for w in works] == [1, 1, 1] and red.handles == []
    assert red.finish() == 0.5 and [w.waits for w in works] == [1, 1, 1]      # a second finish() has nothing left to wait on


def test_gather_eval_skips_the_placeholder_rows_of_an_empty_shard(monkeypatch):
    """the res2d network scores 1000 classes (reference train.py:64-76) while an empty shard's zero-row placeholder is
    NUM_CLASS wide: the gather must not concatenate it (torch.cat refuses (0, 5) next to (n, 1000))."""
    from video_classification_amd import dist as sdist
    from video_classification_amd.train import Trainer
    full = (torch.arange(3000, dtype=torch.float32).reshape(3, 1000), torch.tensor([4, 4, 7]), [2, 1])
    empty = (torch.zeros(0, 5), torch.zeros(0, dtype=torch.int64), [])
    monkeypatch.setattr(sdist, "gather_objects", lambda obj: [full, (empty[0], empty[1], [0])])
    lg, lb, sv = Trainer._gather_eval(None, empty[0], empty[1], [0], (1, 2, 3))
    assert sv == [2, 0, 1] and lg.shape == (3, 1000) and torch.equal(lg, full[0]) and lb.tolist() == [4, 4, 7]

"""Training loop with the reference's surface (train.py): ``ModelManager(cfg)``, ``Trainer(cfg)``, ``.train()``,
``.train_epoch()``, ``.run_eval(loader) -> {'ps','t','acc','sv'}``, ``.save_ckpt/.load_ckpt``.

What differs from /root/reference/train.py, and why:
  * the step (forward, CrossEntropyLoss, zero_grad, backward, Adam -- train.py:225-231) is ONE flat schedule of
    libsfk kernels (``TrainStep``), optionally replayed as a hipGraph; loss and accuracy are accumulated on the
    device, so there is no ``.item()`` sync per step (train.py:236) -- they are read once per epoch;
  * N>1: one process per GPU; loaders the Trainer builds itself (from ``train_set`` / ``test_set`` or the reference's
    dataset) shard one shuffled epoch over the ranks (``dist.EpochShardSampler``) and the test videos round-robin
    (``dist.VideoShardSampler``, scores gathered to every rank); ``TrainStep`` all-reduces the gradient arena over RCCL
    while backward is still running (dist.py); rank 0 writes checkpoints.  Loaders that are INJECTED are used as they are
    (the caller shards them, or every rank sees everything).  The reference is single-GPU;
  * datasets are injected (``train_loader`` / ``test_loader``) or built from the reference's own
    ``ChalearnVideoDataset`` when that module is importable; items keep its contract: a dict
    {cfg.MODEL.R3D_INPUT: (T,21,S,S) float32, 'label': int} (lists of such dicts for the test set).
"""
from __future__ import annotations

import glob
import os
from pathlib import Path
from typing import Callable, List, Optional

import numpy as np
import torch
import torch.utils.data
from torch.utils.data.dataloader import default_collate

from . import dist as sdist
from .config import crop_resize_dict
from .engine import Engine
from .slowfast import init_my_slowfast


class TrainStep:
    """One optimisation step on one rank: forward -> mean cross-entropy -> backward -> (all-reduce) -> Adam.

    Default execution is eager on four HIP streams (engine.OpList lanes: slow pathway / trunk, fast pathway, one
    filter-gradient lane per pathway); the same schedule captured into ONE hipGraph replays its branches one after the
    other on this ROCm (round 1: 51.7 against 46.5 ms for two eager lanes), so ``use_graph=True`` is kept only for the
    lowest host load.  Every switch comes from ``engine.options`` (engine.EngineOptions); nothing here reads the environment."""

    def __init__(self, engine: Engine, lr: float, betas=(0.9, 0.999), eps: float = 1e-8, use_graph: bool = False,
                 reducer: Optional[sdist.GradReducer] = None, overlap_segments: int = 6):
        self.eng, self.lr, self.betas, self.eps = engine, lr, betas, eps
        self.reducer = reducer
        self.world = reducer.world if reducer is not None else 1
        self.segmented = reducer is not None and reducer.active     # cut backward into segments and exchange them as they finish
        if self.segmented:
            # three compute lanes + the collective's own stream = the four hardware queues a process gets (dist.GradReducer);
            # measured only with the single-GPU stand-in (dist.LoopbackReducer), never on RCCL: options.dist_wgrad_one_lane = False
            # keeps the four compute lanes for an A/B on a multi-GPU node
            engine.wgrad_one_lane = engine.options.dist_wgrad_one_lane or engine.wgrad_one_lane
        self.use_graph = use_graph and not self.segmented and engine.device.type == "cuda"
        self.overlap_segments = overlap_segments
        dev = engine.device
        self.loss = torch.zeros(1, device=dev)            # mean loss of the last step
        self.loss_sum = torch.zeros(1, device=dev)        # running sums since reset_meters()
        self.correct = torch.zeros(1, dtype=torch.int32, device=dev)
        self.steps = 0
        self._cache = {}
        # The trunk (lane 0: slow pathway, head, loss, optimiser) is the longest dependency chain of the step; it runs on a
        # HIGH-priority stream of its own so that its next kernel gets CUs ahead of the filter-gradient lanes' backlog
        # (measured +0.8 % clips/s).  The caller's stream waits for the step as before.
        self._trunk = None
        if dev.type == "cuda" and not self.use_graph and engine.options.trunk_priority:
            self._trunk = torch.cuda.Stream(dev, priority=-1)

    def reset_meters(self):
        self.loss_sum.zero_()
        self.correct.zero_()
        self.steps = 0

    def _build(self, pl, labels):
        eng = self.eng
        ops = dict(
            loss=eng.loss_ops(pl, labels, self.loss, self.loss_sum, self.correct),
            adam=eng.adam_ops(self.lr, self.betas, self.eps, self.reducer.grad_scale if self.reducer is not None else 1.0),
            zero_loss=eng.be.fill_zero(self.loss),
            zero_grad=eng.be.fill_zero(eng.G),
        )
        # the optimiser beside the last kernel of the step (engine.Plan.tail_cut): eager four-lane single-rank steps only
        if (pl.tail_cut is not None and not self.segmented and not self.use_graph and eng.two_streams and eng.device.type == "cuda"
                and eng.options.split_adam):
            ops["adam_main"], ops["adam_tail"] = eng.adam_split_ops(pl.tail_cut[1], self.lr, self.betas, self.eps)
        return ops

    def _eager(self, pl, ops):
        eng = self.eng
        st = eng._stream()
        eng.drop_seed.add_(1)
        eng._run_lanes(pl.fwd)
        ops["zero_loss"](st)
        ops["loss"](st)
        ops["zero_grad"](st)
        if "adam_main" in ops:
            # everything but the fast stem's filter gradient; the trunk joins the other lanes as they stand, updates the
            # arena above the stems' filters while that last kernel runs on the fast pathway's lane, then the rest
            i_wg = pl.tail_cut[0]
            eng._run_lanes(pl.bwd, 0, i_wg)
            lanes = eng.lane_streams()
            for s_ in lanes[1:]:
                ev = torch.cuda.Event()
                ev.record(s_)
                lanes[0].wait_event(ev)
            ops["adam_main"](st)
            torch.sub(eng.adam_step, 1, out=eng.adam_step_tail)      # (the tail launch increments its counter: sfk_adam)
            eng._run_lanes(pl.bwd, i_wg)
            ops["adam_tail"](st)
            return
        if not self.segmented:
            eng._run_lanes(pl.bwd)
        else:
            self.reducer.begin()
            lanes = eng.lane_streams()
            issue = lanes[2] if len(lanes) > 2 else None     # the filter-gradient lane enqueues the buckets (no relay stream)
            for a, b, ranges in pl.grad_segments(self.overlap_segments):
                eng._run_lanes(pl.bwd, a, b)          # the pathway lanes and the filter-gradient lane, as the single-rank step
                self.reducer.reduce(ranges, lanes, issue_on=issue)
            self.reducer.finish()
        ops["adam"](st)

    def __call__(self, x_slow, x_fast, labels, slow_t_index=None) -> torch.Tensor:
        eng = self.eng
        pl = eng._plan_for(x_slow, x_fast, slow_t_index, True)
        key = (pl.serial, labels.data_ptr(), pl.graph_epoch)
        ent = self._cache.get(key)
        if ent is None:
            if len(self._cache) > 4:
                self._cache.clear()
            ent = {"ops": self._build(pl, labels), "graph": None, "calls": 0, "labels": labels}
            self._cache[key] = ent
        ent["calls"] += 1
        if not self.use_graph:
            if self._trunk is not None:
                cur = torch.cuda.current_stream(eng.device)
                self._trunk.wait_stream(cur)
                with torch.cuda.stream(self._trunk):
                    self._eager(pl, ent["ops"])
                cur.wait_stream(self._trunk)
            else:
                self._eager(pl, ent["ops"])
        elif ent["graph"] is not None:
            ent["graph"].replay()
        elif ent["calls"] < 2:
            self._eager(pl, ent["ops"])      # first call: eager (loads code objects, settles allocations)
        else:
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._eager(pl, ent["ops"])
            ent["graph"] = g
            g.replay()
        self.steps += 1
        return self.loss


class ModelManager:
    """Name -> (init_model, prepare_data), as reference train.py:39-60.  'slowfast*' and 'res3d' (SURVEY.md section
    8f-4: hub slow_r50 with a 5-channel stem, train.py:79-89 / (deprecated)/train_3dresnet.py:47-51) run on this
    engine; 'res2d' (torchvision's 2-D ResNet-50 over T*C stacked frames, train.py:64-76) is host plumbing only
    (res2d.py: BASELINE config 1, "CPU reference path, no GPU").  MODEL.ARCH = 'canonical8x8' swaps the SlowFast
    geometry for the hub model's ((deprecated)/(torchvideo)train.py:44-71,249)."""

    def __init__(self, cfg, device="cuda", backend=None):
        self.cfg, self.device, self.backend = cfg, device, backend
        self._pre = None
        name = cfg.MODEL.NAME
        self.arch = str(cfg.MODEL.get("ARCH", "ref")).lower()
        if self.arch not in ("ref", "canonical8x8"):
            raise ValueError(f"MODEL.ARCH={self.arch!r}: 'ref' or 'canonical8x8'")
        if name == "res2d":
            self.init_model = self._init_res2d_model
            self.prepare_data = self._prepare_res2d_data
        elif name == "res3d":
            self.init_model = self._init_res3d_model
            self.prepare_data = self._prepare_res3d_data
        elif "slowfast" in name:
            if self.arch == "canonical8x8":
                self.init_model = self._init_canonical_model
                self.prepare_data = self._prepare_canonical_data
            else:
                self.init_model = self._init_slowfast_model
                self.prepare_data = self._prepare_slowfast_data
        else:
            raise NotImplementedError()

    def _h2d(self, t: torch.Tensor) -> torch.Tensor:
        """host -> device through PINNED memory, so the copy is a DMA that overlaps the previous step's kernels (a
        pageable source makes `non_blocking=True` a synchronous staged copy: train.py:127's 1.5 GB batch would stall the
        host for its whole duration).  Loaders built by the Trainer already pin (pin_memory=True)."""
        if t.device.type == "cpu" and torch.device(self.device).type == "cuda" and not t.is_pinned():
            t = t.pin_memory()
        return t.to(self.device, non_blocking=True)

    # ---- res2d (train.py:64-76): host plumbing, see res2d.py
    def _init_res2d_model(self):
        from .res2d import resnet50_2d
        return resnet50_2d(in_channels=5 * int(self.cfg.CHALEARN.CLIP_LEN), num_classes=1000).to(
            "cpu" if torch.device(self.device).type != "cuda" else self.device)

    def _prepare_res2d_data(self, batch):
        """(N,T,21,S,S)[:, :, :5] -> (N, T*5, S, S): frames stacked on the channel axis (train.py:70-76)."""
        dev = "cpu" if torch.device(self.device).type != "cuda" else self.device
        x = batch[self.cfg.MODEL.R3D_INPUT][:, :, :5].to(dev)
        n, t, c, h, w = x.size()
        return torch.reshape(x, (n, t * c, h, w)), batch['label'].to(dev)

    # ---- canonical SlowFast-R50 8x8 (MODEL.ARCH)
    def _init_canonical_model(self):
        from .slowfast import init_canonical_slowfast
        model = init_canonical_slowfast(self.cfg, device=self.device, backend=self.backend)
        ckpt = Path('pretrained', 'SLOWFAST_8x8_R50.pyth')          # the hub checkpoint loads as it is (no surgery)
        if ckpt.is_file():
            state = torch.load(ckpt, map_location="cpu", weights_only=True)["model_state"]
            if tuple(state['blocks.6.proj.weight'].shape) != (self.cfg.CHALEARN.NUM_CLASS, 2304):
                del state['blocks.6.proj.weight'], state['blocks.6.proj.bias']
            model.load_state_dict(state, strict=False)
        return model

    def _prepare_canonical_data(self, batch):
        """(N,T,21,S,S) -> the BGR frames as one strided (N,3,T,S,S) view, handed over as BOTH pathways: the slow
        pathway's PackPathway gather (model.slow_t_index) happens inside its stem kernel."""
        x = torch.permute(self._h2d(batch[self.cfg.MODEL.R3D_INPUT]), [0, 2, 1, 3, 4])[:, 0:3]
        return [x, x], self._h2d(batch['label'])

    @staticmethod
    def delete_mismatch(state_dict):
        """The 12 Kinetics-checkpoint tensors whose shapes differ from the ChaLearn model (train.py:93-111)."""
        keys = ['blocks.0.multipathway_blocks.0.conv.weight', 'blocks.0.multipathway_blocks.1.conv.weight',
                'blocks.6.proj.weight', 'blocks.6.proj.bias']
        for b in (1, 2, 3, 4):
            keys += [f'blocks.{b}.multipathway_blocks.0.res_blocks.0.branch1_conv.weight',
                     f'blocks.{b}.multipathway_blocks.0.res_blocks.0.branch2.conv_a.weight']
        for k in keys:
            del state_dict[k]
        return state_dict

    def _init_slowfast_model(self):
        model = init_my_slowfast(self.cfg, (5, 15), (64, 8), device=self.device, backend=self.backend)
        ckpt = Path('pretrained', 'SLOWFAST_8x8_R50.pyth')          # train.py:116 (absent offline: random init)
        if ckpt.is_file():
            state = torch.load(ckpt, map_location="cpu", weights_only=True)["model_state"]
            model.load_state_dict(self.delete_mismatch(state), strict=False)
        else:
            print(f'warning: {ckpt} not found, training from the reference init scheme')
        return model

    def _init_res3d_model(self):
        """slow_r50 (400 Kinetics classes, as the hub model the reference loads) with Conv3d(5, 64, (1,7,7)) as stem;
        no pretrained file exists offline, so the reference init scheme is used."""
        from .slowfast import _DTYPES, slow_r50
        dtype = _DTYPES[str(self.cfg.MODEL.get("DTYPE", "fp32")).lower()]
        t = int(self.cfg.CHALEARN.CLIP_LEN)
        s = crop_resize_dict[self.cfg.MODEL.R3D_INPUT] // 32
        # hub head pool (8,7,7) fits 8x224^2 clips; other geometries pool the whole res5 map they produce
        pool = (8, 7, 7) if (t >= 8 and s >= 7) else (min(t, 8), s, s)
        return slow_r50(400, 5, dtype=dtype, device=self.device, backend=self.backend, head_pool_kernel=pool)

    def _prepare_res3d_data(self, batch):
        """(N,T,21,S,S) -> BGR+UV (N,5,T,S,S) strided view (train.py:85-89; 5 channels as train.py:72 / the 5-channel
        stem of :81)."""
        x = self._h2d(batch[self.cfg.MODEL.R3D_INPUT])
        x = torch.permute(x, [0, 2, 1, 3, 4])
        return x[:, 0:5], self._h2d(batch['label'])

    def _prepare_slowfast_data(self, batch):
        """(N,T,21,S,S) -> [BGR+UV (N,5,T,S,S), flow (N,15,T,S,S)] strided views of the SAME memory; the depth channel
        (20) is dropped (train.py:125-145).  The stem kernels read these views in place.
        A batch that carries ``<R3D_INPUT>_u8`` (N,T,S,S,21 uint8 frames, optional ``crop`` (N,2)) instead of the float32
        tensor takes the uint8 transport: normalise + RandomCrop run on the device (input_pipeline.py)."""
        key = self.cfg.MODEL.R3D_INPUT
        if key + "_u8" in batch:
            if self._pre is None:
                from .input_pipeline import DevicePreprocess
                self._pre = DevicePreprocess(self.device, self.backend)
            x = self._pre(batch[key + "_u8"], batch.get("crop"))
        else:
            x = self._h2d(batch[key])
        x = torch.permute(x, [0, 2, 1, 3, 4])
        y_true = self._h2d(batch['label'])
        return [x[:, 0:5], x[:, 5:20]], y_true


def _identity(x):
    return x


class SyntheticChalearn(torch.utils.data.Dataset):
    """Stand-in with the item contract of the reference's ChalearnVideoDataset (dataset/chalearn_dataset.py:162-185):
    train -> dict, test -> list of dicts (uniform windows), values normalised like ToTensor+Normalize(0.45, 0.225)."""

    def __init__(self, cfg, name_of_set: str, num_videos: int = 8, clips_per_video=(1, 3), seed: int = 0,
                 as_uint8: bool = False):
        self.cfg, self.name = cfg, name_of_set
        self.as_uint8 = as_uint8          # hand over the HWC uint8 frames (+ the train clip's crop offsets) instead
        self.key = cfg.MODEL.R3D_INPUT
        self.size = crop_resize_dict[self.key]
        self.t = cfg.CHALEARN.CLIP_LEN
        g = torch.Generator().manual_seed(seed)
        self.labels = torch.randint(0, cfg.CHALEARN.NUM_CLASS, (num_videos,), generator=g).tolist()
        self.nclips = torch.randint(clips_per_video[0], clips_per_video[1] + 1, (num_videos,), generator=g).tolist()
        self.seed = seed

    def __len__(self):
        return len(self.labels)

    def _clip(self, i, j):
        g = torch.Generator().manual_seed(self.seed * 7919 + i * 31 + j)
        u8 = torch.randint(0, 256, (self.t, 21, self.size, self.size), generator=g, dtype=torch.uint8)
        if self.as_uint8:
            item = {self.key + "_u8": u8.permute(0, 2, 3, 1).contiguous(), 'label': self.labels[i]}
            if self.name == 'train':
                from .input_pipeline import draw_crop_offsets
                item['crop'] = draw_crop_offsets(1, self.size // 10, g)[0]
            return item
        return {self.key: (u8.float() / 255.0 - 0.45) / 0.225, 'label': self.labels[i]}

    def __getitem__(self, i):
        if self.name == 'train':
            return self._clip(i, 0)
        return [self._clip(i, j) for j in range(self.nclips[i])]


class Trainer:
    def __init__(self, cfg, train_loader=None, test_loader=None, device="cuda", backend=None, use_graph: bool = False,
                 train_set=None, test_set=None, dist_backend: Optional[str] = None):
        """train_set / test_set: datasets with the reference's item contract (default: the reference's own
        ChalearnVideoDataset); the Trainer builds the loaders from them as train.py:149-170 does and, when WORLD_SIZE > 1,
        shards them over the ranks.  train_loader / test_loader: ready loaders, used as they are."""
        self.debug = cfg.DEBUG
        self.num_workers = 0 if self.debug else min(cfg.NUM_CPU, 10)
        self.cfg = cfg
        self.device = device
        self.batch_size = cfg.CHALEARN.BATCH_SIZE
        self.rank, self.world, _ = sdist.init_process_group_from_env(dist_backend)
        self.epoch = 0
        if train_loader is None or test_loader is None:
            if train_set is None or test_set is None:
                train_set, test_set = self._reference_datasets()
            train_loader, test_loader = self._make_loaders(train_set, test_set)
        self.train_loader, self.test_loader = train_loader, test_loader
        self.mm = ModelManager(cfg, device=device, backend=backend)
        self.model = self.mm.init_model()
        self.num_step = 0
        self.ckpt_dir = Path(cfg.CHALEARN.ROOT, cfg.MODEL.LOGS, cfg.MODEL.CKPT_DIR, cfg.MODEL.NAME)
        self.max_historical_acc = 0.
        self.load_ckpt()
        eng = getattr(self.model, "engine", None)
        # Adam is created AFTER the checkpoint load, its state is never saved (train.py:180-182)
        if eng is None:                                  # res2d: a plain torch module, the reference's own five lines
            from .res2d import TorchStep
            assert self.world == 1, "res2d is single-process host plumbing"
            self.step = TorchStep(self.model, lr=cfg.MODEL.LR)
        else:
            reducer = sdist.GradReducer(eng.G, bucket_mb=cfg.DIST.BUCKET_MB) if self.world > 1 else None
            self.step = TrainStep(eng, lr=cfg.MODEL.LR, use_graph=use_graph, reducer=reducer)

    def _reference_datasets(self):
        try:
            from dataset.chalearn_dataset import ChalearnVideoDataset   # the reference's module, unchanged
        except Exception as e:  # cv2 / torchvision / label files missing
            raise RuntimeError("no datasets / loaders were given and the reference's dataset.chalearn_dataset is not "
                               f"importable here ({e}); pass train_set/test_set (e.g. SyntheticChalearn)") from e
        return ChalearnVideoDataset(self.cfg, 'train'), ChalearnVideoDataset(self.cfg, 'test')

    def _make_loaders(self, tr, te):
        """train.py:164,170: train = shuffle + drop_last, test = whole videos, identity collate.  world > 1: one shuffled
        epoch cut into disjoint per-rank shards (every rank runs the same number of steps), test videos round-robin."""
        pin = torch.device(self.device).type == "cuda"
        kw = dict(num_workers=self.num_workers, pin_memory=pin)
        if self.world > 1:
            self.train_sampler = sdist.EpochShardSampler(len(tr), self.rank, self.world, seed=0)
            train = torch.utils.data.DataLoader(tr, batch_size=self.batch_size, sampler=self.train_sampler,
                                                drop_last=True, **kw)
            test = torch.utils.data.DataLoader(te, batch_size=self.batch_size, drop_last=False, collate_fn=_identity,
                                               sampler=sdist.VideoShardSampler(len(te), self.rank, self.world), **kw)
            test.sfk_shard = (self.rank, self.world, len(te))
        else:
            self.train_sampler = None
            train = torch.utils.data.DataLoader(tr, batch_size=self.batch_size, shuffle=True, drop_last=True, **kw)
            test = torch.utils.data.DataLoader(te, batch_size=self.batch_size, shuffle=False, drop_last=False,
                                               collate_fn=_identity, **kw)
        return train, test

    # ---- checkpoints: model weights only, 'acc%.3f_e%d.ckpt', newest by lexicographic sort, HTAH fallback
    def save_ckpt(self, epoch=0, acc=0.0):
        if self.rank != 0:
            return
        self.ckpt_dir.mkdir(parents=True, exist_ok=True)
        ckpt_path = Path(self.ckpt_dir, 'acc%.3f_e%d.ckpt' % (acc, epoch))
        if not self.debug:
            torch.save({k: v.cpu() for k, v in self.model.state_dict().items()}, ckpt_path)
            print(f"Checkpoint saved in {str(ckpt_path)}")
        else:
            print(f'Ignore checkpoint saving under debug mode. {str(ckpt_path)}')

    def load_ckpt(self):
        ckpt_list = sorted(glob.glob(str(self.ckpt_dir / '*.ckpt')))
        if len(ckpt_list) == 0:
            print('warning: no checkpoint found, try using HTAH ckeckpoint')
            ckpt_list = sorted(glob.glob(str(Path(self.ckpt_dir.parent, 'slowfast-HTAH', '*.ckpt'))))
            if len(ckpt_list) == 0:
                print('warning: no HTAH checkpoint found')
                return
        ckpt = ckpt_list[-1]
        print(f'loading checkpoint from {str(ckpt)}')
        self.model.load_state_dict(torch.load(ckpt, map_location="cpu", weights_only=True), strict=True)

    def train_epoch(self):
        self.step.reset_meters()
        seen = 0
        self.model.train()
        if getattr(self, "train_sampler", None) is not None:
            self.train_sampler.set_epoch(self.epoch)     # a new permutation of the epoch, the same on every rank
        for batch in self.train_loader:
            x, y_true = self.mm.prepare_data(batch)
            if torch.is_tensor(x):                       # res3d / res2d: one input tensor
                self.step(x, None, y_true)
            else:
                self.step(x[0], x[1], y_true, slow_t_index=self.model.slow_t_index)
            self.num_step += 1
            seen += int(y_true.shape[0])
            if self.debug:
                break
        # one device->host read per epoch instead of one per step
        loss_avg = float(self.step.loss_sum[0]) / max(self.step.steps, 1)
        correct = int(self.step.correct[0])
        if self.world > 1:                               # the epoch's meters over all ranks (printed by every rank)
            parts = sdist.gather_objects((loss_avg, correct, seen))
            loss_avg = sum(p[0] for p in parts) / len(parts)
            correct, seen = sum(p[1] for p in parts), sum(p[2] for p in parts)
        print(f'loss_avg: {round(loss_avg, 3)}')
        print(f'Train Accuracy: {round(correct / max(seen, 1), 3)}. ({correct} / {seen})')
        return loss_avg, correct / max(seen, 1)

    def train(self):
        max_epoch = self.cfg.MODEL.MAX_EPOCH if not self.debug else 3
        acc, epoch = 0.0, 0
        for epoch in range(max_epoch):
            print(f'========== Training epoch {epoch}')
            self.num_step = 0
            self.epoch = epoch
            self.train_epoch()
            acc = self.run_eval()['acc']
            if acc > self.max_historical_acc:
                self.max_historical_acc = acc
                self.save_ckpt(epoch, acc)
            else:
                print("Not saved. Current best acc: %.3f" % (self.max_historical_acc))
        self.save_ckpt(epoch, acc)

    def run_eval(self, dataset_loader=None):
        """Batched no-grad forward over uniform windows; softmax; per-video mean over its clips; argmax
        (train.py:287-370).  Returns {'ps','t','acc','sv'} as train_sparse.py:76-84 consumes it.
        The logits of every batch stay on the device; softmax, the per-video mean, argmax and the accuracy count are
        ONE ``sfk_eval_aggregate`` launch at the end, followed by one device->host copy of the result (the reference
        copies logits and labels to the host after every batch, train.py:308-309)."""
        loader = self.test_loader if dataset_loader is None else dataset_loader
        logit_list, true_list, batch_collect, samples_per_video = [], [], [], []
        self.model.eval()

        def test_batch(collect):
            x, y_true = self.mm.prepare_data(collect)
            with torch.no_grad():
                y_pred = self.model(x)
            logit_list.append(y_pred.float().clone())      # the engine reuses its logits buffer
            true_list.append(y_true.clone())

        for step, batch in enumerate(loader):
            for b in batch:
                samples_per_video.append(len(b))
                batch_collect.extend(b)
            if len(batch_collect) < self.batch_size:
                continue
            while len(batch_collect) > self.batch_size:          # strict '>' as the reference (train.py:322)
                test_batch(default_collate(batch_collect[:self.batch_size]))
                batch_collect = batch_collect[self.batch_size:]
            if self.debug and step > 5:
                break
        if len(batch_collect) > 0:
            test_batch(default_collate(batch_collect))
        if logit_list:
            logits = torch.cat(logit_list, dim=0).contiguous()
            labels = torch.cat(true_list, dim=0).to(torch.int64).contiguous()
        else:
            # a rank whose shard holds no video (fewer test videos than ranks): zero rows, so that it still takes part in
            # the gather below instead of raising while the other ranks wait in all_gather_object
            # (the width is a placeholder: the res2d network scores 1000 classes, reference train.py:64-76; _gather_eval never
            # concatenates a zero-row part)
            logits = torch.zeros(0, int(self.cfg.CHALEARN.NUM_CLASS), dtype=torch.float32, device=self.device)
            labels = torch.zeros(0, dtype=torch.int64, device=self.device)
        shard = getattr(loader, "sfk_shard", None)
        # (debug mode stops after a few loader steps on every rank; the shards are then partial, nothing is gathered and each
        # rank reports the accuracy of its own videos -- harmless: only rank 0 writes checkpoints, debug writes none)
        if shard is not None and shard[1] > 1 and not self.debug:
            logits, labels, samples_per_video = self._gather_eval(logits, labels, samples_per_video, shard)
        if hasattr(self.model, "engine"):
            ps, pred, ncorrect = aggregate_scores(self.model.engine.be, logits, labels, samples_per_video, softmax=True)
        else:
            from .res2d import aggregate_scores_host
            ps, pred, ncorrect = aggregate_scores_host(logits, labels, samples_per_video)
        ps, true_arr = ps.cpu().numpy(), labels.cpu().numpy()
        nvid = sum(1 for s_ in samples_per_video if s_ > 0)
        accuracy = ncorrect / max(nvid, 1)
        print(f'Test Accuracy: {round(float(accuracy), 3)}. ({ncorrect} / {nvid})')
        return {'ps': ps, 't': true_arr, 'acc': accuracy, 'sv': samples_per_video}


    def _gather_eval(self, logits, labels, sv, shard):
        """every rank evaluated videos rank, rank + world, ... (VideoShardSampler); rebuild the 1-rank order: video v is
        the (v // world)-th video of rank v % world.  Small host objects (scores, labels, counts) travel; every rank ends
        with the full result, so the best-accuracy bookkeeping of train() is the same everywhere."""
        rank, world, total = shard
        parts = sdist.gather_objects((logits.cpu(), labels.cpu(), list(sv)))
        offs = []
        for lg, lb, s_ in parts:
            o = [0]
            for c in s_:
                o.append(o[-1] + int(c))
            assert o[-1] == lg.shape[0] == lb.shape[0]
            offs.append(o)
        rows_l, rows_t, sv_all = [], [], []
        for v in range(total):
            r, j = v % world, v // world
            lg, lb, s_ = parts[r]
            if offs[r][j + 1] > offs[r][j]:         # zero-row parts (an empty shard's placeholder width) stay out of the cat
                rows_l.append(lg[offs[r][j]:offs[r][j + 1]])
                rows_t.append(lb[offs[r][j]:offs[r][j + 1]])
            sv_all.append(s_[j])
        dev = logits.device
        if not rows_l:
            return logits, labels, sv_all
        return (torch.cat(rows_l, 0).to(dev).contiguous(), torch.cat(rows_t, 0).to(dev).contiguous(), sv_all)


def aggregate_scores(be, scores: torch.Tensor, labels: torch.Tensor, samples_per_video, softmax: bool):
    """(ps, pred per video, #correct videos) through sfk_eval_aggregate; scores (rows, classes) fp32 on the device.
    Videos without clips are skipped, as the reference loops do (train.py:351-362, train_sparse.py:211-228)."""
    dev = scores.device
    seg = [0]
    for s_ in samples_per_video:
        seg.append(seg[-1] + int(s_))
    assert seg[-1] <= scores.shape[0]
    seg_off = torch.tensor(seg, dtype=torch.int32).to(dev)
    nvid = len(samples_per_video)
    ps = torch.empty_like(scores)
    pred = torch.empty(max(nvid, 1), dtype=torch.int32, device=dev)
    correct = torch.zeros(1, dtype=torch.int32, device=dev)
    if nvid > 0:
        stream = torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0
        be.eval_aggregate(scores, labels, seg_off, nvid, softmax, ps, pred, correct)(stream)
    return ps, pred[:nvid], int(correct[0])

"""Late-fusion ensemble over the per-crop SlowFast models (reference train_sparse.py).

  ResultSaver          train_sparse.py:29-85   run_eval of every part model, result dict pickled per part
  SparseFusionDataset  train_sparse.py:106-146 stacks the part dicts: 't' (sample,), 'ps' (part, sample, class), 'sv'
  SparseModel          train_sparse.py:88-104  one Linear(num_part, 1) per class -> libsfk sfk_sparse_fusion_fwd/bwd
  SparseTrainer        train_sparse.py:149-240 CE + Adam(1e-3), batch 500; test = per-video mean of the fused scores
                                               (NO softmax, :218-221) -> argmax, through sfk_eval_aggregate

The SparseModel's parameters live in one fp32 arena [w: (class, part) | b: (class)]; ``state_dict`` speaks the
reference's keys ``fcs.<c>.weight`` (1, P) / ``fcs.<c>.bias`` (1,).  One training step = 5 kernel launches
(forward, softmax-CE, zero, backward, Adam); nothing returns to the host until the epoch's accuracy is read.
The result files keep the reference's pickle format (dict of numpy arrays); they are read with an unpickler that
resolves numpy's array reconstructors only.
"""
from __future__ import annotations

import glob
import math
import pickle
from pathlib import Path
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.utils.data

class _ArraysOnlyUnpickler(pickle.Unpickler):
    """The result files keep the reference's format (a pickled dict of numpy arrays, train_sparse.py:82-84), so that a
    reference-side reader and this one exchange files; reading them must not execute what a foreign pickle names:
    only the numpy array / dtype / scalar reconstructors resolve, every other global is refused."""
    _ALLOWED = {("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
                ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
                ("numpy", "ndarray"), ("numpy", "dtype"), ("numpy.core.numeric", "_frombuffer"),
                ("numpy._core.numeric", "_frombuffer")}

    def find_class(self, module, name):
        if (module, name) in self._ALLOWED:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"sparse-fusion result file names {module}.{name}: only numpy arrays are accepted")


def load_result_file(path) -> dict:
    with Path(path).open('rb') as f:
        d = _ArraysOnlyUnpickler(f).load()
    if not isinstance(d, dict) or not {'ps', 't', 'sv'} <= set(d):
        raise ValueError(f"{path}: not a run_eval result dict")
    return d


PART_YAMLS = ['slowfast-HTAH', 'slowfast-LHandArm', 'slowfast-LHand', 'slowfast-RHandArm', 'slowfast-RHand']  # :35


class ResultSaver:
    """Dump the eval result dict of each part model under ROOT/logs/sparse_fusion/<set>/<MODEL.NAME>."""

    def __init__(self, cfgs, make_trainer, loaders):
        """cfgs: iterable of part configs; make_trainer(cfg) -> object with run_eval(loader);
        loaders(cfg, name_of_set) -> un-shuffled loader of uniformly sampled clips (train_sparse.py:56-63)."""
        self.cfgs, self.make_trainer, self.loaders = list(cfgs), make_trainer, loaders

    def save_network_output(self, sets=('train', 'test')) -> List[Path]:
        written = []
        for cfg in self.cfgs:
            trainer = self.make_trainer(cfg)
            for name_of_set in sets:
                path = Path(cfg.CHALEARN.ROOT, cfg.MODEL.LOGS, 'sparse_fusion', name_of_set, cfg.MODEL.NAME)
                y = trainer.run_eval(self.loaders(cfg, name_of_set))
                print(f"eval acc {y['acc']}")
                path.parent.mkdir(parents=True, exist_ok=True)
                with path.open('wb') as f:
                    pickle.dump({'ps': np.asarray(y['ps']), 't': np.asarray(y['t']), 'acc': float(y['acc']),
                                 'sv': list(y['sv'])}, f)
                written.append(path)
        return written


class SparseFusionDataset(torch.utils.data.Dataset):
    def __init__(self, res_folder) -> None:
        parts = []
        for p in glob.glob(str(Path(res_folder, '*'))):
            parts.append((Path(p).stem, load_result_file(p)))     # the reference's file format, arrays only
        parts.sort(key=lambda x: x[0])                                # by part name (train_sparse.py:121)
        self.part_names = [p[0] for p in parts]
        self.T_cp = np.stack([p[1]['t'] for p in parts])[0, :]        # labels are the same for every part
        self.PS_cp = np.stack([p[1]['ps'] for p in parts])            # (part, sample, class)
        self.sv = np.stack([p[1]['sv'] for p in parts])[0, :]
        self.num_part, self.num_N, self.num_class = self.PS_cp.shape

    def __len__(self):
        return self.T_cp.shape[0]

    def __getitem__(self, index):
        return {'t': self.T_cp[index], 'ps': self.PS_cp[:, index]}


class SparseModel:
    """y[n, c] = fcs[c](x[n, :, c]) for x (N, P, C); nn.Linear's default init per class (seeded)."""

    def __init__(self, num_class: int, num_part: int, device="cuda", backend=None, seed: int = 0):
        if backend is None:
            from ._lib import HipBackend
            backend = HipBackend()
        self.be, self.device = backend, torch.device(device)
        self.num_class, self.num_part = num_class, num_part
        C, P = num_class, num_part
        self.numel = C * P + C
        self.P_ = torch.zeros(self.numel, device=self.device)
        self.G = torch.zeros(self.numel, device=self.device)
        gen = torch.Generator().manual_seed(seed)
        bound = 1.0 / math.sqrt(P)            # kaiming_uniform(a=sqrt(5)) on (1, P) == U(-1/sqrt(P), 1/sqrt(P)); same for bias
        self.P_[: C * P] = ((torch.rand(C * P, generator=gen) * 2 - 1) * bound).to(self.device)
        self.P_[C * P:] = ((torch.rand(C, generator=gen) * 2 - 1) * bound).to(self.device)
        self.m = torch.zeros_like(self.P_)
        self.v = torch.zeros_like(self.P_)
        self.step_count = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.training = True

    @property
    def w(self):
        return self.P_[: self.num_class * self.num_part]

    @property
    def b(self):
        return self.P_[self.num_class * self.num_part:]

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream if self.device.type == "cuda" else 0

    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def parameters(self):
        return [self.P_]

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        assert tuple(x.shape[1:3]) == (self.num_part, self.num_class)       # train_sparse.py:98
        x = x.to(self.device, dtype=torch.float32).contiguous()
        n = x.shape[0]
        y = torch.empty(n, self.num_class, device=self.device)
        self.be.sparse_fusion_fwd(x, self.w, self.b, y, n, self.num_part, self.num_class)(self._stream())
        return y

    def train_step(self, x: torch.Tensor, labels: torch.Tensor, lr: float = 1e-3, betas=(0.9, 0.999), eps=1e-8):
        """forward -> mean CE -> zero_grad -> backward -> Adam (train_sparse.py:171-176); returns (loss (1,), logits)."""
        st = self._stream()
        x = x.to(self.device, dtype=torch.float32).contiguous()
        labels = labels.to(self.device, dtype=torch.int64).contiguous()
        n, C, P = x.shape[0], self.num_class, self.num_part
        y = self(x)
        dy = torch.empty_like(y)
        loss = torch.zeros(1, device=self.device)
        lsum = torch.zeros(1, device=self.device)
        corr = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.be.softmax_ce(y, labels, n, C, 1.0, dy, loss, lsum, corr)(st)
        self.be.fill_zero(self.G)(st)
        self.be.sparse_fusion_bwd(x, dy, self.G[: C * P], self.G[C * P:], n, P, C)(st)
        self.be.adam(self.P_, self.G, self.m, self.v, self.numel, lr, betas[0], betas[1], eps, 1.0, self.step_count)(st)
        return loss, y

    def state_dict(self) -> Dict[str, torch.Tensor]:
        C, P = self.num_class, self.num_part
        sd = {}
        w, b = self.w.view(C, P), self.b
        for c in range(C):
            sd[f'fcs.{c}.weight'] = w[c:c + 1].clone()
            sd[f'fcs.{c}.bias'] = b[c:c + 1].clone()
        return sd

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        C, P = self.num_class, self.num_part
        own = {f'fcs.{c}.{s}' for c in range(C) for s in ('weight', 'bias')}
        if strict and set(sd) != own:
            raise RuntimeError(f"Error(s) in loading state_dict: missing {sorted(own - set(sd))[:4]}, "
                               f"unexpected {sorted(set(sd) - own)[:4]}")
        w, b = self.w.view(C, P), self.b
        for c in range(C):
            if f'fcs.{c}.weight' in sd:
                w[c].copy_(sd[f'fcs.{c}.weight'].reshape(P).to(self.device))
            if f'fcs.{c}.bias' in sd:
                b[c] = float(sd[f'fcs.{c}.bias'].reshape(()))


class SparseTrainer:
    def __init__(self, cfg, train_folder=None, test_folder=None, device="cuda", backend=None, batch_size: int = 500,
                 seed: int = 0):
        root = Path(cfg.CHALEARN.ROOT, cfg.MODEL.LOGS)
        self.train_dataset = SparseFusionDataset(train_folder or root / 'sparse_fusion' / 'train')
        self.test_dataset = SparseFusionDataset(test_folder or root / 'sparse_fusion' / 'test')
        self.batch_size, self.device = batch_size, torch.device(device)
        self.sparse_model = SparseModel(self.train_dataset.num_class, self.train_dataset.num_part, device, backend, seed)
        self.max_accuracy = 0.
        self.ckpt_folder = root / 'sparse_fusion_ckpt'
        self.gen = torch.Generator().manual_seed(seed)
        # the whole score table lives on the device: (sample, part, class) -- 5 parts x 249 classes x 4 B per sample
        self.train_ps = torch.from_numpy(np.ascontiguousarray(self.train_dataset.PS_cp.transpose(1, 0, 2))).float().to(self.device)
        self.train_t = torch.from_numpy(np.asarray(self.train_dataset.T_cp)).long().to(self.device)
        self.test_ps = torch.from_numpy(np.ascontiguousarray(self.test_dataset.PS_cp.transpose(1, 0, 2))).float().to(self.device)
        self.test_t = torch.from_numpy(np.asarray(self.test_dataset.T_cp)).long().to(self.device)

    def train(self, epochs: int = 2000, test_every: int = 10):
        n = self.train_ps.shape[0]
        for epoch in range(epochs):
            perm = torch.randperm(n, generator=self.gen).to(self.device)       # DataLoader(shuffle=True), :154
            self.sparse_model.train()
            for i in range(0, n, self.batch_size):
                idx = perm[i:i + self.batch_size]
                self.sparse_model.train_step(self.train_ps[idx], self.train_t[idx])
            if (epoch + 1) % test_every == 0:
                self.test(epoch)

    def save_ckpt(self, acc, epoch):
        self.ckpt_folder.mkdir(parents=True, exist_ok=True)
        path = Path(self.ckpt_folder, 'acc-%.3f-epoch-%d' % (acc, epoch))
        torch.save({k: v.cpu() for k, v in self.sparse_model.state_dict().items()}, path)
        return path

    def test(self, epoch: int = 0) -> float:
        from .train import aggregate_scores
        self.sparse_model.eval()
        outs = [self.sparse_model(self.test_ps[i:i + self.batch_size])
                for i in range(0, self.test_ps.shape[0], self.batch_size)]
        scores = torch.cat(outs, dim=0).contiguous()
        sv = [int(s) for s in self.test_dataset.sv]
        _, _, ncorrect = aggregate_scores(self.sparse_model.be, scores, self.test_t, sv, softmax=False)
        accuracy = ncorrect / max(len(sv), 1)
        if accuracy > self.max_accuracy:
            self.save_ckpt(accuracy, epoch)
        self.max_accuracy = max(accuracy, self.max_accuracy)
        print('Max accuracy: %.3f, new test accuracy: %.3f' % (self.max_accuracy, accuracy))
        return accuracy

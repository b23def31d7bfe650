"""CPU restatements of the reference code either side of the hot path -- TEST INFRASTRUCTURE ONLY (see oracle/__init__).

  run_eval_scores        train.py:337-362        numpy softmax, per-video mean over its clips, argmax, accuracy
  sparse_test_scores     train_sparse.py:204-232 the same aggregation WITHOUT softmax on the fused scores
  SparseModelRef         train_sparse.py:88-104  one nn.Linear(num_part, 1) per class
  to_tensor_normalize    dataset/chalearn_dataset.py:41-46   ToTensor + Normalize(0.45, 0.225)   [torchvision semantics]
  random_crop            dataset/chalearn_dataset.py:73-85   RandomCrop(size, padding) at a given (top, left)

Pinning: the first three are pinned by tests/golden/aux_*.npz, captured from the reference's own classes / methods
(tests/golden/make_aux_golden.py).  The two preprocessing functions restate torchvision transforms, a third-party
package that is neither vendored by the reference nor installed here: numerically "parity unpinned".
"""
import numpy as np
import torch


def run_eval_scores(logits: np.ndarray, labels: np.ndarray, samples_per_video):
    """train.py:337-362 on the concatenated logits: returns (ps, per-video correctness list, accuracy)."""
    ps = np.exp(logits) / np.sum(np.exp(logits), axis=1, keepdims=True)
    correct, read = [], 0
    for num in samples_per_video:
        preds = np.mean(ps[read: read + num], axis=0)
        trues = labels[read: read + num]
        read += num
        assert np.all(trues == trues[0])
        correct.append(np.argmax(preds, axis=0) == trues[0])
    c = np.array(correct)
    return ps, c, c.sum() / len(c)


def sparse_test_scores(scores: np.ndarray, labels: np.ndarray, samples_per_video):
    """train_sparse.py:211-230: mean of the raw fused scores per video, argmax."""
    correct, read = [], 0
    for num in samples_per_video:
        preds = np.mean(scores[read: read + num], axis=0)
        trues = labels[read: read + num]
        read += num
        assert np.all(trues == trues[0])
        correct.append(np.argmax(preds, axis=0) == trues[0])
    return np.array(correct), float(np.mean(correct))


class SparseModelRef(torch.nn.Module):
    def __init__(self, num_class, num_part):
        super().__init__()
        self.num_class, self.num_part = num_class, num_part
        self.fcs = torch.nn.ModuleList([torch.nn.Linear(num_part, 1) for _ in range(num_class)])

    def forward(self, x):                      # (N, P, C) -> (N, C)
        assert x.size()[1:3] == (self.num_part, self.num_class)
        return torch.concat([self.fcs[c](x[:, :, c]) for c in range(self.num_class)], dim=-1)


def to_tensor_normalize(frames_u8: torch.Tensor, mean=0.45, std=0.225) -> torch.Tensor:
    """(T, H, W, C) uint8 -> (T, C, H, W) float32: ToTensor (x / 255, HWC -> CHW) then Normalize ((x - mean) / std)."""
    x = frames_u8.permute(0, 3, 1, 2).to(torch.float32).div(255)
    m = torch.tensor(mean, dtype=torch.float32)
    s = torch.tensor(std, dtype=torch.float32)
    return x.sub(m).div(s)


def random_crop(clip: torch.Tensor, padding: int, top: int, left: int) -> torch.Tensor:
    """RandomCrop(size, padding) of a (T, C, S, S) tensor at offset (top, left) of the zero-padded image."""
    s = clip.shape[-1]
    padded = torch.nn.functional.pad(clip, (padding, padding, padding, padding), mode="constant", value=0.0)
    return padded[..., top: top + s, left: left + s]

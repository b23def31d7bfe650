"""Input pipeline step before the hot path: clips cross PCIe as uint8 and are normalised,
transposed and randomly cropped ON the device by one libsfk kernel.

What it replaces (reference dataset/chalearn_dataset.py):
  :41-46   transforms.ToTensor() + Normalize(mean 0.45, std 0.225) per frame  -> a 256-entry fp32 table built here with
           the same two fp32 operations (so every byte maps to the bit pattern the reference produces), applied by
           ``sfk_u8_normalize_crop``;
  :73-85   transforms.RandomCrop(size, padding = size // 10) on the (T, 21, S, S) clip tensor: one (top, left) per clip,
           zeros (of the NORMALISED tensor) outside the frame -> the kernel's per-clip crop offsets;
  train.py:127  the 1.5 GB pageable float32 H2D copy of a 55-clip batch -> a uint8 copy a quarter of that size, staged in
           pinned host memory so that it is an asynchronous DMA.
The reference's ChalearnVideoDataset is untouched: a loader that can hand over its ``img_cat`` frames (HWC uint8, :113)
feeds ``DevicePreprocess``; loaders that deliver float32 batches keep the reference path (ModelManager.prepare_data).
torchvision is not installed here, so ToTensor / Normalize / RandomCrop are restated from their documented semantics;
numerically this step is "parity unpinned" against torchvision itself (tests/test_aux_cpu.py pins it to plain torch).
"""
from __future__ import annotations

from typing import Optional

import torch

MEAN, STD = 0.45, 0.225          # dataset/chalearn_dataset.py:43-45, all 21 channels


def normalize_lut(mean: float = MEAN, std: float = STD) -> torch.Tensor:
    """float32[256]: ToTensor (uint8 -> float32 / 255) then Normalize ((x - mean) / std), in that order, in fp32."""
    u = torch.arange(256, dtype=torch.uint8)
    x = u.to(torch.float32).div(255)
    return x.sub(torch.tensor(mean, dtype=torch.float32)).div(torch.tensor(std, dtype=torch.float32))


def draw_crop_offsets(n: int, padding: int, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """(n, 2) int32 (top, left), each uniform in [0, 2*padding]: RandomCrop.get_params on the padded (S+2p)^2 image,
    drawn top first, then left, one pair per clip (the crop is applied to the whole (T,21,S,S) tensor, :82-83)."""
    out = torch.empty(n, 2, dtype=torch.int32)
    for i in range(n):
        out[i, 0] = int(torch.randint(0, 2 * padding + 1, (1,), generator=generator))
        out[i, 1] = int(torch.randint(0, 2 * padding + 1, (1,), generator=generator))
    return out


class DevicePreprocess:
    """uint8 frames (N, T, S, S, C) -> normalised clip batch (N, T, C, S, S) on the device."""

    def __init__(self, device="cuda", backend=None, out_dtype: torch.dtype = torch.float32):
        if backend is None:
            from ._lib import HipBackend
            backend = HipBackend()              # raises when libsfk.so is missing: no CPU path
        self.be, self.device, self.out_dtype = backend, torch.device(device), out_dtype
        self.lut = normalize_lut().to(self.device)

    def _h2d(self, t: torch.Tensor) -> torch.Tensor:
        """through pinned host memory: the uint8 batch crosses PCIe as an asynchronous DMA behind the previous step's
        kernels (from pageable memory `non_blocking=True` is a synchronous staged copy)"""
        if t.device.type == "cpu" and self.device.type == "cuda" and not t.is_pinned():
            t = t.pin_memory()
        return t.to(self.device, non_blocking=True)

    def __call__(self, frames_u8: torch.Tensor, crop: Optional[torch.Tensor] = None, padding: Optional[int] = None):
        assert frames_u8.dtype == torch.uint8 and frames_u8.dim() == 5
        n, t, h, w, c = frames_u8.shape
        x = self._h2d(frames_u8).contiguous()
        if crop is not None:
            padding = h // 10 if padding is None else padding
            crop = self._h2d(crop.to(torch.int32)).contiguous()
            assert tuple(crop.shape) == (n, 2)
        out = torch.empty(n, t, c, h, w, dtype=self.out_dtype, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream if self.device.type == "cuda" else 0
        self.be.u8_normalize_crop(x, self.lut, crop, int(padding or 0), out)(stream)
        return out

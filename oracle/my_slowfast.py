"""Restatement of the reference's own SlowFast parametrisation -- TEST INFRASTRUCTURE ONLY.

Follows (semantics, not text):
  /root/reference/model/my_slowfast.py:44-126   init_my_slowfast      -> ``init_my_slowfast`` below
  /root/reference/model/my_slowfast.py:136-257  fusion builder        -> ``RefFusionBuilder``
  /root/reference/model/my_slowfast.py:260-344  FuseFastToSlow        -> ``RefFuseFastToSlow`` (live ``forward`` only)
  /root/reference/train.py:125-145              _prepare_slowfast_data -> ``prepare_slowfast_data``
  /root/reference/(deprecated)/(torchvideo)train.py:53-71  PackPathway -> ``pack_pathway``
  /root/reference/train.py:216-245              one optimisation step -> ``train_step``

``RefFuseFastToSlow`` is pinned bit-exactly against the reference class by tests/golden/fuse_fast_to_slow_*.npz.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch
import torch.nn as nn

from . import pytorchvideo_restated as pv


class RefFuseFastToSlow(nn.Module):
    """cat([x_slow, ReLU(BN(conv_{3x1x1}(x_fast)))], C).  Sub-modules sit in ModuleLists and the reference
    also constructs ``residual`` / ``res_unit`` which its live forward never touches (dead parameters that
    must still appear in the state dict: my_slowfast.py:204-213,228-236 vs 334-344)."""

    def __init__(self, conv_fast_to_slow, residual, norm, activation, res_unit):
        super().__init__()
        self.conv_fast_to_slow = conv_fast_to_slow
        self.residual = residual
        self.norm = norm
        self.activation = activation
        self.res_unit = res_unit

    def forward(self, x: List[torch.Tensor]):
        x_s, x_f = x[0], x[1]
        fuse = self.conv_fast_to_slow[0](x_f)
        if self.norm is not None:          # a ModuleList: always true in the reference (my_slowfast.py:339)
            fuse = self.norm[0](fuse)
        if self.activation is not None:
            fuse = self.activation[0](fuse)
        return [torch.cat([x_s, fuse], 1), x_f]


class RefFusionBuilder:
    """ratio 2, kernel (3,1,1), stride 1, BN + ReLU, Identity past stage 3 (my_slowfast.py:246-257)."""

    def __init__(self, reduction_ratio: int, fusion_ratio: int = 2, kernel=(3, 1, 1), stride=(1, 1, 1),
                 eps: float = 1e-5, momentum: float = 0.1, max_stage_idx: int = 3):
        self.reduction_ratio, self.fusion_ratio = reduction_ratio, fusion_ratio
        self.kernel, self.stride = tuple(kernel), tuple(stride)
        self.eps, self.momentum, self.max_stage_idx = eps, momentum, max_stage_idx

    def create_module(self, fusion_dim_in: int, stage_idx: int) -> nn.Module:
        if stage_idx > self.max_stage_idx:
            return nn.Identity()
        c_slow = fusion_dim_in
        c_fast = fusion_dim_in // self.reduction_ratio
        c_fuse = int(c_fast * self.fusion_ratio)
        c_cat = c_slow + c_fuse
        pad = tuple(k // 2 for k in self.kernel)
        conv = nn.ModuleList([nn.Conv3d(c_fast, c_fuse, self.kernel, self.stride, pad, bias=False)])
        norm = nn.ModuleList([nn.BatchNorm3d(c_fast * self.fusion_ratio, eps=self.eps, momentum=self.momentum)])
        act = nn.ModuleList([nn.ReLU()])
        # dead branches (constructed, never run): 1x1x1 conv+ReLU "residual" and a conv/ReLU/BN "res_unit"
        residual = nn.Sequential(nn.Conv3d(c_slow, c_cat, (1, 1, 1), bias=True), nn.ReLU(inplace=True))
        q = c_cat // 4
        res_unit = nn.Sequential(
            nn.Conv3d(c_cat, q, (1, 1, 1)), nn.ReLU(inplace=True), nn.BatchNorm3d(q),
            nn.Conv3d(q, q, (1, 3, 3), padding=(0, 1, 1)), nn.ReLU(inplace=True), nn.BatchNorm3d(q),
            nn.Conv3d(q, c_cat, (1, 1, 1)),
        )
        return RefFuseFastToSlow(conv, residual, norm, act, res_unit)


def init_my_slowfast(num_class: int, input_channels: Sequence[int], stem_dim_outs: Sequence[int],
                     fuse: bool = True, depth: int = 50) -> nn.Module:
    """The model ``train.py:114`` trains: ``init_my_slowfast(cfg, (5, 15), (64, 8))``.

    Same clip length on both pathways; both stems (1,7,7)/(1,2,2) + MaxPool (1,3,3)/(1,2,2); conv_a kernels
    slow ((1,1,1),(1,1,1),(3,1,1),(3,1,1)) / fast all (3,1,1); conv_b (1,3,3); spatial strides (1,2,2,2);
    head AvgPool (4,2,2) stride 1 on both pathways; depth 50 (my_slowfast.py:50-99; `depth` exists for the small
    test networks only -- the reference hard-codes 50 at :98)."""
    n = len(input_channels)
    assert n >= 2 and len(stem_dim_outs) == n
    ratios = tuple(int(stem_dim_outs[0]) // int(c) for c in stem_dim_outs[1:])
    if fuse:
        builder = RefFusionBuilder(ratios[0]).create_module
        fusion_ratio = 2 * (n - 1)
    else:
        builder = lambda fusion_dim_in, stage_idx: nn.Identity()  # noqa: E731  (my_slowfast.py:90-92)
        fusion_ratio = 0
    t3 = (3, 1, 1)
    return pv.create_slowfast(
        slowfast_channel_reduction_ratio=ratios,
        slowfast_conv_channel_fusion_ratio=fusion_ratio,
        model_depth=depth,
        model_num_class=num_class,
        input_channels=tuple(input_channels),
        fusion_builder=builder,
        stem_dim_outs=tuple(stem_dim_outs),
        stem_conv_kernel_sizes=((1, 7, 7),) * n,
        stem_conv_strides=((1, 2, 2),) * n,
        stem_pool=(nn.MaxPool3d,) * n,
        stem_pool_kernel_sizes=((1, 3, 3),) * n,
        stem_pool_strides=((1, 2, 2),) * n,
        stage_conv_a_kernel_sizes=(((1, 1, 1), (1, 1, 1), t3, t3),) + ((t3,) * 4,) * (n - 1),
        stage_conv_b_kernel_sizes=(((1, 3, 3),) * 4,) * n,
        stage_conv_b_num_groups=((1, 1, 1, 1),) * n,
        stage_conv_b_dilations=(((1, 1, 1),) * 4,) * n,
        stage_spatial_strides=((1, 2, 2, 2),) * n,
        stage_temporal_strides=((1, 1, 1, 1),) * n,
        head_pool_kernel_sizes=((4, 2, 2),) * n,
    )


def canonical_slowfast_8x8(num_class: int = 400) -> nn.Module:
    """SlowFast-R50 8x8 as torch.hub's ``slowfast_r50`` builds it ((deprecated)/(torchvideo)train.py:249):
    all ``create_slowfast`` defaults -- the BENCH geometry of BASELINE.json's metric."""
    return pv.create_slowfast(model_num_class=num_class)


def mini_slowfast(num_class: int = 7, *, ref_style: bool = True, depth: int = 18, head_pool=None,
                  input_channels=None) -> nn.Module:
    """Same wiring at depth 18 = (1,1,1,1) bottleneck blocks per stage: small enough for golden fixtures and
    fast tests.  ``ref_style`` picks the reference's parametrisation (same T on both pathways, (3,1,1) fusion,
    ModuleList fusion keys) or the canonical 8x8 one (T_fast = 4 T_slow, (7,1,1)/(4,1,1) fusion, (5,7,7) stem)."""
    if ref_style:
        ic = (5, 15) if input_channels is None else tuple(input_channels)
        t3 = (3, 1, 1)
        hp = ((2, 2, 2),) * 2 if head_pool is None else head_pool
        return pv.create_slowfast(
            slowfast_channel_reduction_ratio=(8,), slowfast_conv_channel_fusion_ratio=2, model_depth=depth,
            model_num_class=num_class, input_channels=ic, fusion_builder=RefFusionBuilder(8).create_module,
            stem_dim_outs=(64, 8), stem_conv_kernel_sizes=((1, 7, 7),) * 2,
            stage_conv_a_kernel_sizes=(((1, 1, 1), (1, 1, 1), t3, t3), (t3,) * 4),
            head_pool_kernel_sizes=hp)
    ic = (3, 3) if input_channels is None else tuple(input_channels)
    hp = ((2, 2, 2), (8, 2, 2)) if head_pool is None else head_pool
    return pv.create_slowfast(model_depth=depth, model_num_class=num_class, input_channels=ic,
                              head_pool_kernel_sizes=hp)


def slow_r50(num_class: int = 400, input_channels: int = 5, depth: int = 50, head_pool=(8, 7, 7)) -> nn.Module:
    """The reference's `res3d` network: hub `slow_r50` with its stem conv swapped for
    Conv3d(5, 64, (1,7,7), stride (1,2,2), padding (0,3,3), bias=False) ((deprecated)/train_3dresnet.py:47-51,
    train.py:79-89).  depth=18 + a small head pool give the mini version the parity tests use."""
    return pv.create_resnet(input_channel=input_channels, model_depth=depth, model_num_class=num_class,
                            stem_conv_kernel_size=(1, 7, 7), head_pool_kernel_size=tuple(head_pool))


def prepare_res3d_data(clips: torch.Tensor) -> torch.Tensor:
    """(N,T,C,S,S) dataset tensor -> (N,C,T,S,S) view (train.py:85-89; the deprecated trainer feeds the 5 BGR+UV
    channels, train_3dresnet.py:86-93)."""
    return clips.permute(0, 2, 1, 3, 4)


def prepare_slowfast_data(clips: torch.Tensor) -> List[torch.Tensor]:
    """(N,T,21,S,S) dataset tensor -> [BGR+UV (N,5,T,S,S), flow (N,15,T,S,S)] views; the depth channel (20)
    is dropped (train.py:125-145)."""
    x = clips.permute(0, 2, 1, 3, 4)
    return [x[:, 0:5], x[:, 5:20]]


def pack_pathway(frames: torch.Tensor, alpha: int = 4) -> List[torch.Tensor]:
    """(N,C,T,H,W) -> [slow = T//alpha frames picked at linspace(0,T-1) truncated, fast = all frames]."""
    t = frames.shape[2]
    idx = torch.linspace(0, t - 1, t // alpha).long()
    return [frames.index_select(2, idx), frames]


def train_step(model: nn.Module, optim: torch.optim.Optimizer, x: List[torch.Tensor],
               y_true: torch.Tensor) -> Tuple[float, int]:
    """forward, mean cross-entropy, zero_grad, backward, optimiser step, argmax bookkeeping
    (train.py:225-242).  Returns (loss, number correct)."""
    model.train()
    y_pred = model(list(x))
    loss = nn.functional.cross_entropy(y_pred, y_true)
    optim.zero_grad()
    loss.backward()
    optim.step()
    with torch.no_grad():
        correct = int((y_pred.argmax(-1) == y_true).sum())
    return float(loss.item()), correct


def conv_macs(model: nn.Module, inputs: List[torch.Tensor]) -> int:
    """Multiply-accumulates of every Conv3d/Linear in one forward (used to pin 65.709 G / 50.309 G)."""
    total = [0]
    hooks = []

    def conv_hook(m, inp, out):
        k = m.kernel_size[0] * m.kernel_size[1] * m.kernel_size[2]
        total[0] += out.numel() * (m.in_channels // m.groups) * k

    def lin_hook(m, inp, out):
        total[0] += out.numel() * m.in_features

    for mod in model.modules():
        if isinstance(mod, nn.Conv3d):
            hooks.append(mod.register_forward_hook(conv_hook))
        elif isinstance(mod, nn.Linear):
            hooks.append(mod.register_forward_hook(lin_hook))
    was_training = model.training
    model.eval()
    with torch.no_grad():
        model(list(inputs))
    model.train(was_training)
    for h in hooks:
        h.remove()
    return total[0]

"""Restatement of the pytorchvideo building blocks the reference calls -- TEST INFRASTRUCTURE ONLY.

The reference delegates the whole backbone to third-party ``pytorchvideo`` (facebookresearch, version
un-pinned; API level >= 0.1.3), imported at /root/reference/model/my_slowfast.py:14,23-24,34-38 and
/root/reference/train.py:16,29-30.  That package is absent from the reference tree and from this image, so
its published wiring is restated here from SURVEY.md appendix A1 (A1.1 create_slowfast, A1.2 stem, A1.3
res stage / bottleneck, A1.4 MultiPathWayWithFuse / Net, A1.5 head, A1.8 init).

Module attribute names are the pytorchvideo ones, because the reference's checkpoint surgery
(train.py:94-108) and ``load_state_dict(strict=True)`` (train.py:212) pin the state-dict key scheme:
  blocks.0.multipathway_blocks.{p}.{conv,norm}
  blocks.{1..4}.multipathway_blocks.{p}.res_blocks.{i}.{branch1_conv,branch1_norm,branch2.{conv,norm}_{a,b,c}}
  blocks.{0..3}.multipathway_fusion.*      blocks.6.proj.{weight,bias}

Numerics of this file are "parity unpinned" (no reference fixture exists for the backbone); structure is
pinned by tests/test_oracle_structure.py.
"""
from __future__ import annotations

import math
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

_STAGE_DEPTHS = {18: (1, 1, 1, 1), 26: (2, 2, 2, 2), 50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}


def _half(k: Sequence[int]) -> Tuple[int, ...]:
    return tuple(int(v) // 2 for v in k)


# --------------------------------------------------------------------------- stem (A1.2)
class ResNetBasicStem(nn.Module):
    """conv(bias=False) -> norm -> activation -> pool   (SURVEY A1.2)."""

    def __init__(self, conv, norm, activation, pool):
        super().__init__()
        self.conv, self.norm, self.activation, self.pool = conv, norm, activation, pool

    def forward(self, x):
        x = self.conv(x)
        if self.norm is not None:
            x = self.norm(x)
        if self.activation is not None:
            x = self.activation(x)
        if self.pool is not None:
            x = self.pool(x)
        return x


def create_res_basic_stem(*, in_channels, out_channels, conv_kernel_size=(3, 7, 7), conv_stride=(1, 2, 2),
                          conv_padding=None, pool=nn.MaxPool3d, pool_kernel_size=(1, 3, 3),
                          pool_stride=(1, 2, 2), pool_padding=None, norm=nn.BatchNorm3d, norm_eps=1e-5,
                          norm_momentum=0.1, activation=nn.ReLU):
    conv_padding = _half(conv_kernel_size) if conv_padding is None else conv_padding
    pool_padding = _half(pool_kernel_size) if pool_padding is None else pool_padding
    return ResNetBasicStem(
        conv=nn.Conv3d(in_channels, out_channels, tuple(conv_kernel_size), tuple(conv_stride),
                       tuple(conv_padding), bias=False),
        norm=None if norm is None else norm(out_channels, eps=norm_eps, momentum=norm_momentum),
        activation=None if activation is None else activation(),
        pool=None if pool is None else pool(kernel_size=tuple(pool_kernel_size), stride=tuple(pool_stride),
                                            padding=tuple(pool_padding)),
    )


# --------------------------------------------------------------------------- bottleneck / res block (A1.3)
class BottleneckBlock(nn.Module):
    def __init__(self, conv_a, norm_a, act_a, conv_b, norm_b, act_b, conv_c, norm_c):
        super().__init__()
        self.conv_a, self.norm_a, self.act_a = conv_a, norm_a, act_a
        self.conv_b, self.norm_b, self.act_b = conv_b, norm_b, act_b
        self.conv_c, self.norm_c = conv_c, norm_c

    def forward(self, x):
        x = self.act_a(self.norm_a(self.conv_a(x)))
        x = self.act_b(self.norm_b(self.conv_b(x)))
        return self.norm_c(self.conv_c(x))


def create_bottleneck_block(*, dim_in, dim_inner, dim_out, conv_a_kernel_size=(3, 1, 1), conv_a_stride=(2, 1, 1),
                            conv_a_padding=(1, 0, 0), conv_b_kernel_size=(1, 3, 3), conv_b_stride=(1, 2, 2),
                            conv_b_padding=(0, 1, 1), conv_b_num_groups=1, conv_b_dilation=(1, 1, 1),
                            norm=nn.BatchNorm3d, norm_eps=1e-5, norm_momentum=0.1, activation=nn.ReLU):
    def bn(c):
        return norm(c, eps=norm_eps, momentum=norm_momentum)

    norm_c = bn(dim_out)
    norm_c.block_final_bn = True  # init flag only (A1.8)
    return BottleneckBlock(
        conv_a=nn.Conv3d(dim_in, dim_inner, tuple(conv_a_kernel_size), tuple(conv_a_stride),
                         tuple(conv_a_padding), bias=False),
        norm_a=bn(dim_inner), act_a=activation(),
        conv_b=nn.Conv3d(dim_inner, dim_inner, tuple(conv_b_kernel_size), tuple(conv_b_stride),
                         tuple(conv_b_padding), bias=False, groups=conv_b_num_groups,
                         dilation=tuple(conv_b_dilation)),
        norm_b=bn(dim_inner), act_b=activation(),
        conv_c=nn.Conv3d(dim_inner, dim_out, (1, 1, 1), bias=False),
        norm_c=norm_c,
    )


class ResBlock(nn.Module):
    """out = ReLU(shortcut(x) + branch2(x)); shortcut is a 1x1x1 conv+BN when the shape changes (A1.3)."""

    def __init__(self, branch1_conv, branch1_norm, branch2, activation):
        super().__init__()
        self.branch1_conv, self.branch1_norm = branch1_conv, branch1_norm
        self.branch2, self.activation = branch2, activation

    def forward(self, x):
        if self.branch1_conv is None:
            s = x
        else:
            s = self.branch1_conv(x)
            if self.branch1_norm is not None:
                s = self.branch1_norm(s)
        return self.activation(s + self.branch2(x))


class ResStage(nn.Module):
    def __init__(self, res_blocks):
        super().__init__()
        self.res_blocks = res_blocks

    def forward(self, x):
        for b in self.res_blocks:
            x = b(x)
        return x


def create_res_stage(*, depth, dim_in, dim_inner, dim_out, bottleneck, conv_a_kernel_size, conv_a_stride,
                     conv_a_padding, conv_b_kernel_size, conv_b_stride, conv_b_padding, conv_b_num_groups=1,
                     conv_b_dilation=(1, 1, 1), norm=nn.BatchNorm3d, norm_eps=1e-5, norm_momentum=0.1,
                     activation=nn.ReLU):
    blocks = []
    for i in range(depth):
        d_in = dim_in if i == 0 else dim_out
        a_stride = tuple(conv_a_stride) if i == 0 else (1, 1, 1)
        b_stride = tuple(conv_b_stride) if i == 0 else (1, 1, 1)
        stride = tuple(x * y for x, y in zip(a_stride, b_stride))
        needs_proj = d_in != dim_out or math.prod(stride) != 1
        blocks.append(ResBlock(
            branch1_conv=nn.Conv3d(d_in, dim_out, (1, 1, 1), stride, bias=False) if needs_proj else None,
            branch1_norm=norm(dim_out, eps=norm_eps, momentum=norm_momentum) if needs_proj else None,
            branch2=bottleneck(dim_in=d_in, dim_inner=dim_inner, dim_out=dim_out,
                               conv_a_kernel_size=conv_a_kernel_size, conv_a_stride=a_stride,
                               conv_a_padding=conv_a_padding, conv_b_kernel_size=conv_b_kernel_size,
                               conv_b_stride=b_stride, conv_b_padding=conv_b_padding,
                               conv_b_num_groups=conv_b_num_groups, conv_b_dilation=conv_b_dilation,
                               norm=norm, norm_eps=norm_eps, norm_momentum=norm_momentum,
                               activation=activation),
            activation=activation(),
        ))
    return ResStage(nn.ModuleList(blocks))


# --------------------------------------------------------------------------- multi-pathway containers (A1.4)
class MultiPathWayWithFuse(nn.Module):
    def __init__(self, multipathway_blocks, multipathway_fusion, inplace=True):
        super().__init__()
        self.multipathway_blocks = multipathway_blocks
        self.multipathway_fusion = multipathway_fusion
        self.inplace = inplace

    def forward(self, x: List[torch.Tensor]):
        y = x if self.inplace else [None] * len(x)
        for p, blk in enumerate(self.multipathway_blocks):
            if blk is not None:
                y[p] = blk(x[p])
        if self.multipathway_fusion is not None:
            y = self.multipathway_fusion(y)
        return y


class PoolConcatPathway(nn.Module):
    def __init__(self, retain_list=False, pool=None, dim=1):
        super().__init__()
        self.retain_list, self.pool, self.dim = retain_list, pool, dim

    def forward(self, x: List[torch.Tensor]):
        if self.pool is not None:
            x = [x[p] if self.pool[p] is None else self.pool[p](x[p]) for p in range(len(x))]
        if self.retain_list:
            return x
        return torch.cat(x, self.dim)


class FuseFastToSlow(nn.Module):
    """pytorchvideo's default lateral fusion: cat([slow, ReLU(BN(conv(fast)))], C)   (A1.1 last line)."""

    def __init__(self, conv_fast_to_slow, norm=None, activation=None):
        super().__init__()
        self.conv_fast_to_slow, self.norm, self.activation = conv_fast_to_slow, norm, activation

    def forward(self, x):
        x_s, x_f = x[0], x[1]
        fuse = self.conv_fast_to_slow(x_f)
        if self.norm is not None:
            fuse = self.norm(fuse)
        if self.activation is not None:
            fuse = self.activation(fuse)
        return [torch.cat([x_s, fuse], 1), x_f]


class FastToSlowFusionBuilder:
    def __init__(self, slowfast_channel_reduction_ratio, conv_fusion_channel_ratio, conv_kernel_size, conv_stride,
                 norm=nn.BatchNorm3d, norm_eps=1e-5, norm_momentum=0.1, activation=nn.ReLU, max_stage_idx=3):
        self.ratio, self.fusion_ratio = slowfast_channel_reduction_ratio, conv_fusion_channel_ratio
        self.k, self.s = tuple(conv_kernel_size), tuple(conv_stride)
        self.norm, self.eps, self.mom, self.act = norm, norm_eps, norm_momentum, activation
        self.max_stage_idx = max_stage_idx

    def create_module(self, fusion_dim_in, stage_idx):
        if stage_idx > self.max_stage_idx:
            return nn.Identity()
        c_in = fusion_dim_in // self.ratio
        c_out = int(c_in * self.fusion_ratio)
        return FuseFastToSlow(
            conv_fast_to_slow=nn.Conv3d(c_in, c_out, self.k, self.s, _half(self.k), bias=False),
            norm=None if self.norm is None else self.norm(c_out, eps=self.eps, momentum=self.mom),
            activation=None if self.act is None else self.act(),
        )


# --------------------------------------------------------------------------- head (A1.5)
class ResNetBasicHead(nn.Module):
    """dropout -> Linear at every (T',H',W') position -> global mean -> (N, classes); returns LOGITS."""

    def __init__(self, pool=None, dropout=None, proj=None, activation=None, output_pool=None):
        super().__init__()
        self.pool, self.dropout, self.proj = pool, dropout, proj
        self.activation, self.output_pool = activation, output_pool

    def forward(self, x):
        if self.pool is not None:
            x = self.pool(x)
        if self.dropout is not None:
            x = self.dropout(x)
        if self.proj is not None:
            x = self.proj(x.permute(0, 2, 3, 4, 1)).permute(0, 4, 1, 2, 3)
        if self.activation is not None:
            x = self.activation(x)
        if self.output_pool is not None:
            x = self.output_pool(x)
            x = x.view(x.shape[0], -1)
        return x


def create_res_basic_head(*, in_features, out_features, pool=None, output_size=(1, 1, 1), dropout_rate=0.5,
                          activation=None, output_with_global_average=True):
    return ResNetBasicHead(
        pool=None if pool is None else pool,
        dropout=nn.Dropout(dropout_rate) if dropout_rate > 0 else None,
        proj=nn.Linear(in_features, out_features),
        activation=None if activation is None else activation(),
        output_pool=nn.AdaptiveAvgPool3d(tuple(output_size)) if output_with_global_average else None,
    )


class Net(nn.Module):
    def __init__(self, *, blocks):
        super().__init__()
        self.blocks = blocks
        init_net_weights(self)

    def forward(self, x):
        for b in self.blocks:
            x = b(x)
        return x


def init_net_weights(model: nn.Module, fc_init_std: float = 0.01) -> None:
    """'resnet' init (A1.8): conv <- Kaiming-normal(fan_out, relu); BN weight 1, except block-final BN <- 0;
    Linear weight ~ N(0, fc_init_std), bias 0.  Lowest-confidence item of the restatement; only matters for
    from-scratch training, never for parity (parity loads explicit seeded weights)."""
    for m in model.modules():
        if isinstance(m, nn.Conv3d):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, nn.modules.batchnorm._NormBase):
            if m.weight is not None:
                m.weight.data.fill_(0.0 if getattr(m, "block_final_bn", False) else 1.0)
            if m.bias is not None:
                m.bias.data.zero_()
        elif isinstance(m, nn.Linear):
            m.weight.data.normal_(mean=0.0, std=fc_init_std)
            if m.bias is not None:
                m.bias.data.zero_()


# --------------------------------------------------------------------------- create_slowfast (A1.1)
def create_slowfast(*, slowfast_channel_reduction_ratio=(8,), slowfast_conv_channel_fusion_ratio=2,
                    slowfast_fusion_conv_kernel_size=(7, 1, 1), slowfast_fusion_conv_stride=(4, 1, 1),
                    fusion_builder: Optional[Callable] = None, input_channels=(3, 3), model_depth=50,
                    model_num_class=400, dropout_rate=0.5, norm=nn.BatchNorm3d, activation=nn.ReLU,
                    stem_function=None, stem_dim_outs=(64, 8), stem_conv_kernel_sizes=((1, 7, 7), (5, 7, 7)),
                    stem_conv_strides=((1, 2, 2), (1, 2, 2)), stem_pool=(nn.MaxPool3d, nn.MaxPool3d),
                    stem_pool_kernel_sizes=((1, 3, 3), (1, 3, 3)), stem_pool_strides=((1, 2, 2), (1, 2, 2)),
                    stage_conv_a_kernel_sizes=(((1, 1, 1), (1, 1, 1), (3, 1, 1), (3, 1, 1)),
                                               ((3, 1, 1), (3, 1, 1), (3, 1, 1), (3, 1, 1))),
                    stage_conv_b_kernel_sizes=(((1, 3, 3),) * 4, ((1, 3, 3),) * 4),
                    stage_conv_b_num_groups=((1, 1, 1, 1), (1, 1, 1, 1)),
                    stage_conv_b_dilations=(((1, 1, 1),) * 4, ((1, 1, 1),) * 4),
                    stage_spatial_strides=((1, 2, 2, 2), (1, 2, 2, 2)),
                    stage_temporal_strides=((1, 1, 1, 1), (1, 1, 1, 1)), bottleneck=None,
                    head_pool=nn.AvgPool3d, head_pool_kernel_sizes=((8, 7, 7), (32, 7, 7)),
                    head_output_size=(1, 1, 1), head_activation=None, head_output_with_global_average=True):
    n_path = len(input_channels)
    depths = _STAGE_DEPTHS[model_depth]
    if fusion_builder is None:
        fusion_builder = FastToSlowFusionBuilder(
            slowfast_channel_reduction_ratio=slowfast_channel_reduction_ratio[0],
            conv_fusion_channel_ratio=slowfast_conv_channel_fusion_ratio,
            conv_kernel_size=slowfast_fusion_conv_kernel_size, conv_stride=slowfast_fusion_conv_stride,
            norm=norm, activation=activation, max_stage_idx=len(depths) - 1).create_module
    if bottleneck is None:
        bottleneck = tuple((create_bottleneck_block,) * 4 for _ in range(n_path))
    if stem_function is None:
        stem_function = (create_res_basic_stem,) * n_path

    blocks = []
    stems = [stem_function[p](in_channels=input_channels[p], out_channels=stem_dim_outs[p],
                              conv_kernel_size=stem_conv_kernel_sizes[p], conv_stride=stem_conv_strides[p],
                              conv_padding=_half(stem_conv_kernel_sizes[p]), pool=stem_pool[p],
                              pool_kernel_size=stem_pool_kernel_sizes[p], pool_stride=stem_pool_strides[p],
                              pool_padding=_half(stem_pool_kernel_sizes[p]), norm=norm, activation=activation)
             for p in range(n_path)]
    blocks.append(MultiPathWayWithFuse(nn.ModuleList(stems),
                                       fusion_builder(fusion_dim_in=stem_dim_outs[0], stage_idx=0)))

    stage_dim_in = stem_dim_outs[0]
    stage_dim_out = stage_dim_in * 4
    for idx, depth in enumerate(depths):
        red0 = slowfast_channel_reduction_ratio[0]
        dim_in = [stage_dim_in + stage_dim_in * slowfast_conv_channel_fusion_ratio // red0]
        dim_inner = [stage_dim_out // 4]
        dim_out = [stage_dim_out]
        for r in slowfast_channel_reduction_ratio:
            dim_in.append(stage_dim_in // r)
            dim_inner.append(stage_dim_out // 4 // r)
            dim_out.append(stage_dim_out // r)
        stages = []
        for p in range(n_path):
            ka = stage_conv_a_kernel_sizes[p][idx]
            kb = stage_conv_b_kernel_sizes[p][idx]
            dil = stage_conv_b_dilations[p][idx]
            ss, ts = stage_spatial_strides[p][idx], stage_temporal_strides[p][idx]
            stages.append(create_res_stage(
                depth=depth, dim_in=dim_in[p], dim_inner=dim_inner[p], dim_out=dim_out[p],
                bottleneck=bottleneck[p][idx], conv_a_kernel_size=ka, conv_a_stride=(ts, 1, 1),
                conv_a_padding=_half(ka), conv_b_kernel_size=kb, conv_b_stride=(1, ss, ss),
                conv_b_padding=tuple(d * (k // 2) if k > 1 else 0 for k, d in zip(kb, dil)),
                conv_b_num_groups=stage_conv_b_num_groups[p][idx], conv_b_dilation=dil,
                norm=norm, activation=activation))
        blocks.append(MultiPathWayWithFuse(nn.ModuleList(stages),
                                           fusion_builder(fusion_dim_in=stage_dim_out, stage_idx=idx + 1)))
        stage_dim_in = stage_dim_out
        stage_dim_out *= 2

    blocks.append(PoolConcatPathway(
        retain_list=False,
        pool=nn.ModuleList([head_pool(kernel_size=tuple(k), stride=(1, 1, 1), padding=(0, 0, 0))
                            for k in head_pool_kernel_sizes])))
    head_in = stage_dim_in + sum(stage_dim_in // r for r in slowfast_channel_reduction_ratio)
    blocks.append(create_res_basic_head(in_features=head_in, out_features=model_num_class, pool=None,
                                        output_size=head_output_size, dropout_rate=dropout_rate,
                                        activation=head_activation,
                                        output_with_global_average=head_output_with_global_average))
    return Net(blocks=nn.ModuleList(blocks))


# --------------------------------------------------------------------------- create_resnet (single pathway; A1.1)
def create_resnet(*, input_channel=3, model_depth=50, model_num_class=400, dropout_rate=0.5, norm=nn.BatchNorm3d,
                  activation=nn.ReLU, stem_dim_out=64, stem_conv_kernel_size=(3, 7, 7), stem_conv_stride=(1, 2, 2),
                  stem_pool=nn.MaxPool3d, stem_pool_kernel_size=(1, 3, 3), stem_pool_stride=(1, 2, 2),
                  stage_conv_a_kernel_size=((1, 1, 1), (1, 1, 1), (3, 1, 1), (3, 1, 1)),
                  stage_conv_b_kernel_size=((1, 3, 3),) * 4, stage_spatial_h_stride=(1, 2, 2, 2),
                  stage_temporal_stride=(1, 1, 1, 1), head_pool=nn.AvgPool3d, head_pool_kernel_size=(4, 7, 7),
                  head_output_size=(1, 1, 1), head_activation=None, head_output_with_global_average=True):
    """pytorchvideo.models.resnet.create_resnet (third-party, absent here; restated from its published structure):
    blocks = [stem, res2, res3, res4, res5, head]; the head (create_res_basic_head) owns its AvgPool3d, stride 1.
    The reference reaches it as torch.hub `slow_r50` ((deprecated)/train_3dresnet.py:48, train.py:80):
    create_resnet(stem_conv_kernel_size=(1,7,7), head_pool_kernel_size=(8,7,7), model_depth=50)."""
    depths = _STAGE_DEPTHS[model_depth]
    blocks = [create_res_basic_stem(in_channels=input_channel, out_channels=stem_dim_out,
                                    conv_kernel_size=stem_conv_kernel_size, conv_stride=stem_conv_stride,
                                    conv_padding=_half(stem_conv_kernel_size), pool=stem_pool,
                                    pool_kernel_size=stem_pool_kernel_size, pool_stride=stem_pool_stride,
                                    pool_padding=_half(stem_pool_kernel_size), norm=norm, activation=activation)]
    dim_in, dim_out = stem_dim_out, stem_dim_out * 4
    for idx, depth in enumerate(depths):
        ka, kb = stage_conv_a_kernel_size[idx], stage_conv_b_kernel_size[idx]
        ss, ts = stage_spatial_h_stride[idx], stage_temporal_stride[idx]
        blocks.append(create_res_stage(
            depth=depth, dim_in=dim_in, dim_inner=dim_out // 4, dim_out=dim_out, bottleneck=create_bottleneck_block,
            conv_a_kernel_size=ka, conv_a_stride=(ts, 1, 1), conv_a_padding=_half(ka), conv_b_kernel_size=kb,
            conv_b_stride=(1, ss, ss), conv_b_padding=tuple(k // 2 for k in kb), conv_b_num_groups=1,
            conv_b_dilation=(1, 1, 1), norm=norm, activation=activation))
        dim_in, dim_out = dim_out, dim_out * 2
    blocks.append(create_res_basic_head(
        in_features=dim_in, out_features=model_num_class,
        pool=head_pool(kernel_size=tuple(head_pool_kernel_size), stride=(1, 1, 1), padding=(0, 0, 0)),
        output_size=head_output_size, dropout_rate=dropout_rate, activation=head_activation,
        output_with_global_average=head_output_with_global_average))
    return Net(blocks=nn.ModuleList(blocks))

"""SlowFast wiring as data: which convs / norms exist, under which checkpoint keys, with which geometry.

Two parametrisations of the same wiring:
  ref_spec        what train.py:114 trains -- init_my_slowfast(cfg, (5,15), (64,8)) (model/my_slowfast.py:44-126):
                  same clip length on both pathways, (1,7,7) stems, (3,1,1)/(1,1,1) lateral fusion held in
                  ModuleLists (+ dead residual/res_unit parameters), head pools (4,2,2) stride 1.
  canonical_spec  SlowFast-R50 8x8 of BASELINE.json's metric: T_fast = 4 T_slow, fast stem (5,7,7), fusion
                  (7,1,1)/(4,1,1), head pools (8,7,7)/(32,7,7)  ((deprecated)/(torchvideo)train.py:44-71,249).
  slow_r50_spec   the reference's `res3d` model: pytorchvideo slow_r50 with a 5-channel (1,7,7) stem
                  ((deprecated)/train_3dresnet.py:47-51, train.py:79-89) -- ONE pathway: the slow pathway's wiring
                  without lateral fusion, keys `blocks.0.conv`, `blocks.<s>.res_blocks.<i>...`, `blocks.5.proj`.
Checkpoint key scheme: pytorchvideo's, pinned by train.py:94-108.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Tuple

from .plan import ConvGeom

Triple = Tuple[int, int, int]
STAGE_DEPTHS = {18: (1, 1, 1, 1), 26: (2, 2, 2, 2), 50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}


@dataclass(frozen=True)
class SlowFastSpec:
    num_class: int
    input_channels: Tuple[int, int] = (3, 3)
    stem_dim_outs: Tuple[int, int] = (64, 8)
    stem_kernels: Tuple[Triple, Triple] = ((1, 7, 7), (5, 7, 7))
    conv_a_kernels: Tuple[Tuple[Triple, ...], Tuple[Triple, ...]] = (
        ((1, 1, 1), (1, 1, 1), (3, 1, 1), (3, 1, 1)), ((3, 1, 1),) * 4)
    spatial_strides: Tuple[int, int, int, int] = (1, 2, 2, 2)
    depth: int = 50
    fuse: bool = True
    fusion_kernel: Triple = (7, 1, 1)
    fusion_stride: Triple = (4, 1, 1)
    fusion_ratio: int = 2
    ref_fusion_keys: bool = False      # ModuleList-indexed keys + dead residual/res_unit parameters
    head_pool_kernels: Tuple[Triple, Triple] = ((8, 7, 7), (32, 7, 7))
    dropout: float = 0.5
    bn_eps: float = 1e-5
    bn_momentum: float = 0.1

    @property
    def pathways(self) -> int:
        return len(self.input_channels)

    @property
    def reduction(self) -> int:
        return self.stem_dim_outs[0] // self.stem_dim_outs[1] if self.pathways == 2 else 1

    @property
    def depths(self) -> Tuple[int, int, int, int]:
        return STAGE_DEPTHS[self.depth]


def ref_spec(num_class: int = 249, input_channels=(5, 15), stem_dim_outs=(64, 8), fuse: bool = True,
             depth: int = 50, head_pool_kernels=((4, 2, 2), (4, 2, 2))) -> SlowFastSpec:
    return SlowFastSpec(num_class=num_class, input_channels=tuple(input_channels), stem_dim_outs=tuple(stem_dim_outs),
                        stem_kernels=((1, 7, 7), (1, 7, 7)), depth=depth, fuse=fuse, fusion_kernel=(3, 1, 1),
                        fusion_stride=(1, 1, 1), fusion_ratio=2 if fuse else 0, ref_fusion_keys=True,
                        head_pool_kernels=tuple(map(tuple, head_pool_kernels)))


def canonical_spec(num_class: int = 400, depth: int = 50, input_channels=(3, 3),
                   head_pool_kernels=((8, 7, 7), (32, 7, 7))) -> SlowFastSpec:
    return SlowFastSpec(num_class=num_class, input_channels=tuple(input_channels), depth=depth,
                        head_pool_kernels=tuple(map(tuple, head_pool_kernels)))


def slow_r50_spec(num_class: int = 400, input_channels: int = 5, depth: int = 50,
                  head_pool_kernel: Triple = (8, 7, 7)) -> SlowFastSpec:
    """pytorchvideo `slow_r50` = create_resnet(stem (1,7,7), conv_a ((1,1,1),(1,1,1),(3,1,1),(3,1,1)), head pool
    (8,7,7), dropout 0.5) with the reference's stem swap to `input_channels` (train_3dresnet.py:49)."""
    return SlowFastSpec(num_class=num_class, input_channels=(input_channels,), stem_dim_outs=(64,),
                        stem_kernels=((1, 7, 7),), conv_a_kernels=(((1, 1, 1), (1, 1, 1), (3, 1, 1), (3, 1, 1)),),
                        depth=depth, fuse=False, fusion_ratio=0, head_pool_kernels=(tuple(head_pool_kernel),))


# ----------------------------------------------------------------------------- layer records
@dataclass
class ConvBN:
    """Conv3d(bias=False) followed by BatchNorm3d; `conv_key`/`norm_key` are state-dict prefixes."""
    conv_key: str
    norm_key: str
    geom: ConvGeom                 # reference geometry (weight shape (cout, cin, kt, kh, kw))
    is_stem: bool = False
    zero_init_gamma: bool = False  # block-final BN


@dataclass
class Block:
    conv_a: ConvBN
    conv_b: ConvBN
    conv_c: ConvBN
    branch1: Optional[ConvBN]


@dataclass
class DeadParam:
    key: str
    shape: Tuple[int, ...]
    kind: str  # 'conv_w' | 'bias' | 'bn_w' | 'bn_b' | 'bn_rm' | 'bn_rv' | 'bn_nbt'


@dataclass
class Wiring:
    spec: SlowFastSpec
    stems: List[ConvBN]
    stages: List[List[List[Block]]]      # [stage 0..3][pathway][block]
    fusions: List[Optional[ConvBN]]      # after blocks.0 .. blocks.3 (None when fuse=False)
    dead: List[DeadParam] = field(default_factory=list)
    head_in: int = 0
    head_key: str = "blocks.6.proj"

    def all_convbn(self) -> List[ConvBN]:
        out: List[ConvBN] = []
        np_ = len(self.stems)
        for p in range(np_):
            out.append(self.stems[p])
        if self.fusions[0] is not None:
            out.append(self.fusions[0])
        for si, stage in enumerate(self.stages):
            for p in range(np_):
                for b in stage[p]:
                    if b.branch1 is not None:
                        out.append(b.branch1)
                    out += [b.conv_a, b.conv_b, b.conv_c]
            if si + 1 < 4 and self.fusions[si + 1] is not None:
                out.append(self.fusions[si + 1])
        return out


def _half(k: Triple) -> Triple:
    return (k[0] // 2, k[1] // 2, k[2] // 2)


def build_wiring(spec: SlowFastSpec) -> Wiring:
    red = spec.reduction
    np_ = spec.pathways
    assert np_ in (1, 2) and (np_ == 2 or not spec.fuse)

    def path(p: int) -> str:          # create_slowfast wraps every stage in MultiPathWayWithFuse, create_resnet does not
        return f".multipathway_blocks.{p}" if np_ == 2 else ""
    stems = []
    for p in range(np_):
        k = spec.stem_kernels[p]
        stems.append(ConvBN(f"blocks.0{path(p)}.conv", f"blocks.0{path(p)}.norm",
                            ConvGeom(spec.input_channels[p], spec.stem_dim_outs[p], k, (1, 2, 2), _half(k)),
                            is_stem=True))
    dead: List[DeadParam] = []

    def fusion(block_idx: int, dim_in: int) -> Optional[ConvBN]:
        if not spec.fuse:
            return None
        c_in = dim_in // red
        c_out = c_in * spec.fusion_ratio
        base = f"blocks.{block_idx}.multipathway_fusion"
        idx = ".0" if spec.ref_fusion_keys else ""
        if spec.ref_fusion_keys:
            c_cat, q = dim_in + c_out, (dim_in + c_out) // 4
            dead.extend([
                DeadParam(f"{base}.residual.0.weight", (c_cat, dim_in, 1, 1, 1), "conv_w"),
                DeadParam(f"{base}.residual.0.bias", (c_cat,), "bias"),
                DeadParam(f"{base}.res_unit.0.weight", (q, c_cat, 1, 1, 1), "conv_w"),
                DeadParam(f"{base}.res_unit.0.bias", (q,), "bias"),
                DeadParam(f"{base}.res_unit.3.weight", (q, q, 1, 3, 3), "conv_w"),
                DeadParam(f"{base}.res_unit.3.bias", (q,), "bias"),
                DeadParam(f"{base}.res_unit.6.weight", (c_cat, q, 1, 1, 1), "conv_w"),
                DeadParam(f"{base}.res_unit.6.bias", (c_cat,), "bias"),
            ])
            for j in (2, 5):
                dead.extend([DeadParam(f"{base}.res_unit.{j}.weight", (q,), "bn_w"),
                             DeadParam(f"{base}.res_unit.{j}.bias", (q,), "bn_b"),
                             DeadParam(f"{base}.res_unit.{j}.running_mean", (q,), "bn_rm"),
                             DeadParam(f"{base}.res_unit.{j}.running_var", (q,), "bn_rv"),
                             DeadParam(f"{base}.res_unit.{j}.num_batches_tracked", (), "bn_nbt")])
        return ConvBN(f"{base}.conv_fast_to_slow{idx}", f"{base}.norm{idx}",
                      ConvGeom(c_in, c_out, spec.fusion_kernel, spec.fusion_stride, _half(spec.fusion_kernel)))

    fusions: List[Optional[ConvBN]] = [fusion(0, spec.stem_dim_outs[0])]
    stages: List[List[List[Block]]] = []
    dim_in_s = spec.stem_dim_outs[0]
    dim_out_s = dim_in_s * 4
    for si, depth in enumerate(spec.depths):
        fr = spec.fusion_ratio if spec.fuse else 0
        dims_in = (dim_in_s + dim_in_s * fr // red, dim_in_s // red)
        dims_inner = (dim_out_s // 4, dim_out_s // 4 // red)
        dims_out = (dim_out_s, dim_out_s // red)
        ss = spec.spatial_strides[si]
        stage: List[List[Block]] = []
        for p in range(np_):
            ka = spec.conv_a_kernels[p][si]
            blocks: List[Block] = []
            for i in range(depth):
                d_in = dims_in[p] if i == 0 else dims_out[p]
                s_b = (1, ss, ss) if i == 0 else (1, 1, 1)
                base = f"blocks.{si + 1}{path(p)}.res_blocks.{i}"
                proj = d_in != dims_out[p] or s_b != (1, 1, 1)
                blocks.append(Block(
                    conv_a=ConvBN(f"{base}.branch2.conv_a", f"{base}.branch2.norm_a",
                                  ConvGeom(d_in, dims_inner[p], ka, (1, 1, 1), _half(ka))),
                    conv_b=ConvBN(f"{base}.branch2.conv_b", f"{base}.branch2.norm_b",
                                  ConvGeom(dims_inner[p], dims_inner[p], (1, 3, 3), s_b, (0, 1, 1))),
                    conv_c=ConvBN(f"{base}.branch2.conv_c", f"{base}.branch2.norm_c",
                                  ConvGeom(dims_inner[p], dims_out[p], (1, 1, 1)), zero_init_gamma=True),
                    branch1=ConvBN(f"{base}.branch1_conv", f"{base}.branch1_norm",
                                   ConvGeom(d_in, dims_out[p], (1, 1, 1), s_b)) if proj else None,
                ))
            stage.append(blocks)
        stages.append(stage)
        if si < 3:
            fusions.append(fusion(si + 1, dim_out_s))
        dim_in_s = dim_out_s
        dim_out_s *= 2
    if np_ == 1:
        return Wiring(spec, stems, stages, fusions, dead, head_in=dim_in_s, head_key="blocks.5.proj")
    return Wiring(spec, stems, stages, fusions, dead, head_in=dim_in_s + dim_in_s // red)

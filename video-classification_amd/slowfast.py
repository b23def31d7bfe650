"""Model facade with the surface the reference trainer uses:
``model([x_slow, x_fast]) -> (N, num_class) logits``, ``.train()/.eval()``, ``.parameters()``,
``.state_dict()/.load_state_dict()``, ``.cuda()``.

Mirrors /root/reference/model/my_slowfast.py:44-126 (``init_my_slowfast(cfg, input_channels, stem_dim_outs)``);
``slowfast_r50_8x8`` is the canonical model the reference only reaches through torch.hub
((deprecated)/(torchvideo)train.py:249).  All arithmetic runs in libsfk (HIP); there is no torch.nn fallback.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence

import torch

from . import arch
from .engine import Engine

_DTYPES = {"fp32": torch.float32, "f32": torch.float32, "float32": torch.float32, "bf16": torch.bfloat16,
           "bfloat16": torch.bfloat16}


class _SlowFastFn(torch.autograd.Function):
    """One autograd node for the whole network: forward and backward are flat kernel schedules of the engine."""

    @staticmethod
    def forward(ctx, arena, engine, x_slow, x_fast, slow_t_index):
        pl = engine.forward(x_slow, x_fast, True, slow_t_index)
        ctx.engine, ctx.pl = engine, pl
        return pl.logits.clone()

    @staticmethod
    def backward(ctx, dlogits):
        eng = ctx.engine
        eng.backward(ctx.pl, dlogits.contiguous().float())
        return eng.G, None, None, None, None


class SlowFast(torch.nn.Module):
    def __init__(self, spec: arch.SlowFastSpec, dtype=torch.float32, device="cuda", backend=None, seed: int = 0, options=None):
        super().__init__()
        self.spec = spec
        self.engine = Engine(spec, dtype=dtype, device=device, backend=backend, seed=seed, options=options)
        self.arena = self.engine.P          # the one trainable tensor: all live parameters, kernel layout
        # PackPathway ((deprecated)/(torchvideo)train.py:53-71) as a model attribute: when set, model([frames, frames])
        # reads the slow pathway's frames frames[:, :, slow_t_index] inside the stem kernel (MODEL.ARCH canonical8x8)
        self.slow_t_index: Optional[torch.Tensor] = None

    def forward(self, x, slow_t_index: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x: [x_slow, x_fast] (SlowFast) or one (N,C,T,H,W) tensor (the single-pathway res3d network)."""
        if self.spec.pathways == 1:
            x_slow, x_fast = (x if torch.is_tensor(x) else x[0]), None
        else:
            x_slow, x_fast = x[0], x[1]
        if slow_t_index is None:
            slow_t_index = self.slow_t_index
        if self.training and torch.is_grad_enabled():
            return _SlowFastFn.apply(self.arena, self.engine, x_slow, x_fast, slow_t_index)
        with torch.no_grad():
            pl = self.engine.forward(x_slow, x_fast, self.training, slow_t_index)
            return pl.logits.clone()

    # checkpoint surface: pytorchvideo key names / reference tensor shapes
    def state_dict(self, *args, **kwargs) -> Dict[str, torch.Tensor]:
        return self.engine.state_dict()

    def load_state_dict(self, state_dict, strict: bool = True):
        return self.engine.load_state_dict(state_dict, strict)

    def cuda(self, device=None):
        assert self.engine.device.type == "cuda", "the engine was created on " + str(self.engine.device)
        return self

    def num_parameters(self, live_only=False) -> int:
        return self.engine.num_parameters(live_only)


def init_my_slowfast(cfg, input_channels, stem_dim_outs, device="cuda", backend=None, seed: int = 0) -> SlowFast:
    """Same call as the reference's ``init_my_slowfast(cfg, (5, 15), (64, 8))`` (train.py:114)."""
    assert len(input_channels) == 2 and len(stem_dim_outs) == 2, "two pathways (slow, fast)"
    spec = arch.ref_spec(num_class=cfg.CHALEARN.NUM_CLASS, input_channels=input_channels,
                         stem_dim_outs=stem_dim_outs, fuse=bool(cfg.MODEL.FUSE), depth=int(cfg.MODEL.get("DEPTH", 50)))
    dtype = _DTYPES[str(cfg.MODEL.get("DTYPE", "fp32")).lower()]
    return SlowFast(spec, dtype=dtype, device=device, backend=backend, seed=seed)


def init_canonical_slowfast(cfg, device="cuda", backend=None, seed: int = 0, alpha: int = 4) -> SlowFast:
    """MODEL.ARCH = 'canonical8x8': the hub model of (deprecated)/(torchvideo)train.py:249 (SlowFast-R50 8x8, the model
    BASELINE.json's metric is quoted on) driven through the reference's config surface -- BGR frames of CLIP_LEN frames,
    the slow pathway = every alpha-th frame (PackPathway), global head pools for the crop's size."""
    t = int(cfg.CHALEARN.CLIP_LEN)
    from .config import crop_resize_dict
    s = crop_resize_dict[cfg.MODEL.R3D_INPUT] // 32
    assert t % alpha == 0 and s >= 1, (t, s)
    spec = arch.canonical_spec(num_class=cfg.CHALEARN.NUM_CLASS, depth=int(cfg.MODEL.get("DEPTH", 50)),
                               head_pool_kernels=((t // alpha, s, s), (t, s, s)))
    dtype = _DTYPES[str(cfg.MODEL.get("DTYPE", "fp32")).lower()]
    m = SlowFast(spec, dtype=dtype, device=device, backend=backend, seed=seed)
    m.slow_t_index = pack_pathway_index(t, alpha, device)
    return m


def slowfast_r50_8x8(num_class: int = 400, dtype=torch.bfloat16, device="cuda", backend=None, seed: int = 0, options=None) -> SlowFast:
    return SlowFast(arch.canonical_spec(num_class), dtype=dtype, device=device, backend=backend, seed=seed, options=options)


def slow_r50(num_class: int = 400, input_channels: int = 5, dtype=torch.float32, device="cuda", backend=None,
             seed: int = 0, depth: int = 50, head_pool_kernel=(8, 7, 7)) -> SlowFast:
    """The reference's `res3d` model (hub slow_r50 + 5-channel (1,7,7) stem, (deprecated)/train_3dresnet.py:47-51):
    the same engine with one pathway; ``model(x)`` takes the (N,C,T,H,W) tensor itself."""
    spec = arch.slow_r50_spec(num_class, input_channels, depth, tuple(head_pool_kernel))
    return SlowFast(spec, dtype=dtype, device=device, backend=backend, seed=seed)


def pack_pathway_index(num_frames: int, alpha: int = 4, device="cuda") -> torch.Tensor:
    """Frame indices of the slow pathway: linspace(0, T-1, T//alpha).long()
    ((deprecated)/(torchvideo)train.py:60-71) = [0,4,8,13,17,22,26,31] for T=32."""
    return torch.linspace(0, num_frames - 1, num_frames // alpha).long().to(torch.int32).to(device)

"""SlowFast execution engine: one flat list of C-ABI kernel calls per forward / backward, on one hipStream.

What replaces what (reference file:line):
  forward            model(x) at train.py:226,306  (pytorchvideo Net.forward over blocks.0..6)
  backward           loss.backward() at train.py:230
  lateral fusion     FuseFastToSlow.forward, model/my_slowfast.py:334-344 -- the fused branch is written straight into
                     the channel slice [C_slow, C_slow+C_fuse) of the slow pathway's buffer, so torch.cat (:343) vanishes
  parameters         one fp32 arena (master weights) + one fp32 gradient arena; conv filters are kept in the kernels'
                     [cout][tap][cin] layout and converted to/from the reference (cout,cin,kt,kh,kw) layout only in
                     state_dict()/load_state_dict() (train.py:192,212 checkpoint surface, pytorchvideo key names)
Data layout in HBM: channels-last feature maps (N,T,H,W,C) with an explicit pixel stride (include/sfk.h), bf16 or f32;
per-channel statistics, master weights, gradients and the head are fp32.
"""
from __future__ import annotations

import math
import dataclasses
import os
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Tuple

import torch

from . import arch
from ._lib import BN_FOLD_ROWS, BnBwdFuse, ConvEpilogue, ConvPass, FMap, StemSrc, WgradPass, stem_kp
from .plan import ConvGeom, dgrad_passes, fwd_pass, round_up, wgrad_taps

Run = Callable[[int], None]


def _max_parts() -> int:
    from ._lib import tuning
    return int(tuning().bn_parts)       # partial rows of the BatchNorm reductions (sfk_tuning.bn_parts)


@dataclass
class _Layer:
    cb: arch.ConvBN
    eg: ConvGeom              # geometry the kernels see (stems: temporal conv over the im2col patch matrix)
    kreal: int                # stems: kh*kw*cin before padding, else 0
    w_off: int = 0
    w_numel: int = 0
    g_off: int = 0
    b_off: int = 0
    needs_dgrad: bool = True
    rm: torch.Tensor = None
    rv: torch.Tensor = None
    nbt: torch.Tensor = None

    @property
    def c(self) -> int:
        return self.eg.cout


@dataclass
class _UnitRec:
    """What the backward of one conv+BN unit needs from its forward."""
    L: _Layer
    x: FMap
    y: FMap
    mean: torch.Tensor
    invstd: torch.Tensor
    scale: torch.Tensor
    shift: torch.Tensor


class Wait:
    """Schedule marker: stream of pathway `lane` must wait for everything issued so far on pathway `on`.
    A no-op when the schedule runs on a single stream (stream order already implies it)."""

    def __init__(self, lane: int, on: int):
        self.lane, self.on = lane, on

    def __call__(self, stream):
        return None


class OpList(list):
    """A kernel schedule; `meta[i]` describes op i for the profiler / roofline report (None for glue ops), `lane[i]`
    is the pathway stream it belongs to (0 = slow pathway / trunk, 1 = fast pathway).  The two pathways only meet at the
    lateral fusions, so their kernels run concurrently on two HIP streams (inside one hipGraph when captured)."""

    def __init__(self):
        super().__init__()
        self.meta: List[Optional[dict]] = []
        self.lane: List[int] = []
        self.cur_lane = 0

    def append(self, op, **meta):
        super().append(op)
        self.meta.append(meta or None)
        self.lane.append(op.lane if isinstance(op, Wait) else self.cur_lane)

    def sync(self, lane: int, on: int):
        self.append(Wait(lane, on))

    def insert_at(self, i: int, op, lane: int, **meta):
        """op (or a Wait) into position i of the issue order"""
        super().insert(i, op)
        self.meta.insert(i, meta or None)
        self.lane.insert(i, op.lane if isinstance(op, Wait) else lane)


class Plan:
    def __init__(self):
        self.fwd: OpList = OpList()
        self.bwd: OpList = OpList()
        self.logits: torch.Tensor = None
        self.dlogits: torch.Tensor = None
        self.key = None
        # (index one past the backward op that finishes it, (arena offset, numel)): every gradient range has ONE writer
        self.grad_marks: List[Tuple[int, Tuple[int, int]]] = []
        self.stem_state: Dict[int, dict] = {}             # per pathway: buffers + slots of the ops bound to the clip tensors
        self.tail_cut: Optional[Tuple[int, int]] = None   # (index of the fast stem's filter-gradient op in bwd, arena offset
                                                          #  below which only stem filters live): see Engine.adam_split_ops

    def grad_segments(self, nseg: int) -> List[Tuple[int, int, List[Tuple[int, int]]]]:
        """Cut the backward schedule into nseg pieces [(op_begin, op_end, finished gradient ranges)], balanced by
        gradient bytes, for overlapping the all-reduce with the rest of backward."""
        total = sum(r[1] for _, r in self.grad_marks)
        marks = sorted(self.grad_marks)
        segs, begin, acc, cur, k = [], 0, 0, [], 1
        for end, rng in marks:
            cur.append(rng)
            acc += rng[1]
            if acc >= total * k / nseg and k < nseg:
                segs.append((begin, end, cur))
                begin, cur, k = end, [], k + 1
        segs.append((begin, len(self.bwd), cur))
        return segs


@dataclasses.dataclass
class EngineOptions:
    """Every schedule / fusion switch of the engine, with the measured best as the default.  The product never reads the
    environment: Engine() without options runs the defaults, and whoever wants an A/B (bench.py, tools/) fills an
    EngineOptions -- EngineOptions.from_env() maps the SFK_* variables of tools/gpu_ab_env.sh onto the fields, explicitly, at
    the caller's request.  `ablate_kinds` (timing-only: the scheduler SKIPS those kernel classes, results are garbage) has
    no environment variable at all: bench.py --ablate sets it and says so in its output line."""
    fuse_bn_bwd: bool = False          # BatchNorm-backward reduce in the dgrad epilogues (sfk_conv_desc.bnb): neutral .. -0.75 %
    tail_dual: bool = True             # sfk_conv_pw_dual for the narrowest block tails
    shortcut_lane: str = "f"           # projection shortcuts beside branch2 on the filter-gradient lane: f forward, b backward, 1 both, 0 neither
    wgrad_lanes: int = 1               # 0: filter gradients on the pathway lanes; 1: one lane per pathway; 2: ONE lane for both
    relu_bits: bool = True             # block-output ReLU masks kept as bitmaps
    relu_out_mask: bool = True         # ... and applied by the data-gradient pass that finishes an identity block's output gradient
    deterministic_wgrad: bool = False  # split sums through the workspace + ordered reduce everywhere (bit-reproducible dW)
    fuse_tail: bool = True             # conv_c -> norm_c -> (+ shortcut) -> ReLU without the conv output in HBM
    tail_min_c: int = 8
    tail_max_c: int = 128
    fuse_stem_tail: bool = True        # the stems' BatchNorm -> ReLU -> MaxPool as one forward pass / two-pass backward
    fuse_tail_dg: bool = True          # R and the first dgrad pass of the tail backward in one kernel
    tail_r_lane: int = 0               # R = dz^T a beside the first dgrad pass: 0 pathway lane, 2 filter-gradient lane, 4 own lanes
    split_refresh: bool = True         # filter refresh: stems on the trunk, the rest on the idle filter-gradient lane
    refresh_behind_stems: bool = True  # ... issued BEHIND the two stem convs (it waits for both) instead of beside them
    split_adam: bool = True            # Adam beside the last kernel of the step (TrainStep)
    trunk_priority: bool = True        # the trunk lane on a high-priority HIP stream (TrainStep)
    mfma_wgrad_trunk: bool = False     # MFMA-bound filter gradients on the pathway's own lane, directly behind their data gradient
    dist_wgrad_one_lane: bool = True   # world > 1: all filter gradients on ONE lane, the collective's stream is the fourth queue
    fuse_finalize: bool = False        # BatchNorm finalize as the prologue of its bn_apply / bn_bwd_apply launch (sfk_bn_finalize_apply):
                                       # bit-identical, 5 us per launch alone -- and 27.7 -> 27.9 .. 28.1 ms in the step (waiting workgroups
                                       # hold CUs the other lanes would use): off
    fuse_finalize_max_c: int = 64      # ... for BatchNorms of at most this many channels (the pairs are claimed through one counter)
    lane_cus: str = ""                 # EXPERIMENT: CUs the side lanes may use, "fast,wgrad_slow,wgrad_fast" (0 / empty = all): the
                                       # side streams are created with hipExtStreamCreateWithCUMask (eager schedule only)
    ablate_kinds: frozenset = frozenset()

    _ENV = {"SFK_FUSE_BNB": ("fuse_bn_bwd", "1"), "SFK_TAIL_DUAL": ("tail_dual", "!0"), "SFK_SHORTCUT_LANE": ("shortcut_lane", "s"),
            "SFK_WGRAD_LANES": ("wgrad_lanes", "i"), "SFK_RELU_BITS": ("relu_bits", "!0"), "SFK_RELU_OUT": ("relu_out_mask", "!0"),
            "SFK_WGWS": ("deterministic_wgrad", "1"), "SFK_TAIL": ("fuse_tail", "!0"), "SFK_TAIL_MINC": ("tail_min_c", "i"),
            "SFK_TAIL_MAXC": ("tail_max_c", "i"), "SFK_STEM_TAIL": ("fuse_stem_tail", "!0"), "SFK_TAIL_DG": ("fuse_tail_dg", "!0"),
            "SFK_TAIL_RLANE": ("tail_r_lane", "i"), "SFK_SPLIT_REFRESH": ("split_refresh", "!0"), "SFK_REFRESH_BEHIND": ("refresh_behind_stems", "!0"),
            "SFK_SPLIT_ADAM": ("split_adam", "!0"), "SFK_TRUNK_PRIO": ("trunk_priority", "!0"),
            "SFK_WGRAD_TRUNK": ("mfma_wgrad_trunk", "1"), "SFK_DIST_ONE_LANE": ("dist_wgrad_one_lane", "!0"),
            "SFK_LANE_CUS": ("lane_cus", "s"), "SFK_FUSE_FIN": ("fuse_finalize", "!0"),
            "SFK_FUSE_FIN_MAXC": ("fuse_finalize_max_c", "i")}

    @classmethod
    def from_env(cls, env=None) -> "EngineOptions":
        """The SFK_* experiment variables -> options (benchmark / tools only; the library and the engine never call this)."""
        env = os.environ if env is None else env
        o = cls()
        for var, (field, kind) in cls._ENV.items():
            if var not in env:
                continue
            v = env[var]
            setattr(o, field, v == "1" if kind == "1" else v != "0" if kind == "!0" else int(v) if kind == "i" else v)
        return o

    def non_default(self) -> Dict[str, object]:
        d = EngineOptions()
        return {f.name: getattr(self, f.name) for f in dataclasses.fields(self)
                if not f.name.startswith("_") and getattr(self, f.name) != getattr(d, f.name)}


def _masked_stream(device, spec: int):
    """EXPERIMENT (EngineOptions.lane_cus): a HIP stream whose kernels may only use some CUs.  spec = ncu (the first ncu mask
    bits) or -ncu (the LAST ncu bits); the runtime deals mask bits round-robin over the 8 XCDs, so any contiguous run of bits
    is spread evenly.  Wrapped as a torch ExternalStream; lives as long as the process."""
    import ctypes as C
    total = torch.cuda.get_device_properties(device).multi_processor_count
    n = min(abs(spec), total)
    bits = range(n) if spec > 0 else range(total - n, total)
    words = (C.c_uint32 * ((total + 31) // 32))()
    for b in bits:
        words[b // 32] |= 1 << (b % 32)
    hip = C.CDLL("libamdhip64.so")            # the runtime torch already loaded (one instance per process)
    st = C.c_void_p()
    with torch.cuda.device(device):
        rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), C.c_uint32(len(words)), words)
    if rc != 0 or not st.value:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask({spec}) failed: {rc}")
    return torch.cuda.ExternalStream(st.value, device)


class Engine:
    def __init__(self, spec: arch.SlowFastSpec, dtype: torch.dtype = torch.bfloat16, device="cuda", backend=None,
                 seed: int = 0, options: Optional[EngineOptions] = None):
        assert dtype in (torch.bfloat16, torch.float32)
        self.spec, self.dtype, self.device = spec, dtype, torch.device(device)
        if backend is None:
            from ._lib import HipBackend
            backend = HipBackend()        # raises if libsfk.so is missing: no fallback path exists
        self.be = backend
        self.wiring = arch.build_wiring(spec)
        self.vec = 8                      # channel alignment both precisions are built for
        self._layers: Dict[str, _Layer] = {}
        self._bufs: Dict[str, torch.Tensor] = {}
        self._plans: Dict[tuple, Plan] = {}
        self._pending_fin: Dict[int, tuple] = {}     # deferred BatchNorm finalizes by id(scale buffer): consumed by the next _apply
        self._build_params(seed)
        self.max_parts = _max_parts() if getattr(self.be, "name", "") == "hip" else 1024
        self.two_streams = True           # slow / fast pathway on two HIP streams (see OpList)
        # every switch comes from the options object (defaults = the measured best); nothing here reads the environment
        o = self.options = options if options is not None else EngineOptions()
        assert o.shortcut_lane in ("f", "b", "1", "0") and o.tail_r_lane in (0, 2, 4) and o.wgrad_lanes in (0, 1, 2)
        # BatchNorm-backward reduce folded into the dgrad epilogues (sfk_conv_desc.bnb): removes 3.3 ms of reduce kernels,
        # adds 3.0 ms to the conv class -- measured neutral on the step (877 vs 879 clips/s), so it is opt-in
        self.fuse_bn_bwd = o.fuse_bn_bwd
        self.tail_dual = o.tail_dual
        # projection shortcuts beside branch2 on the pathway's filter-gradient lane: f = forward (default: that lane is idle in the
        # forward, +0.5 %), b = backward too (the lane carries the filter gradients there: -0.7 %), 1 = both, 0 = neither
        self.shortcut_lane_f, self.shortcut_lane_b = o.shortcut_lane in ("1", "f"), o.shortcut_lane in ("1", "b")
        # diagnostic: kernel classes (OpList meta kinds) that the lane scheduler SKIPS -- what does the step time owe to one
        # class?  (bench.py --ablate; results are garbage with anything skipped)
        self._ablate_kinds = frozenset(o.ablate_kinds)
        self.wgrad_lanes = o.wgrad_lanes != 0                               # filter gradients on their own streams
        # both pathways' filter gradients on ONE lane (lane 2): three compute streams, so that the collective's stream of a
        # world > 1 step is the fourth hardware queue (dist.GradReducer; single-rank: 1078 vs 1077 clips/s, neutral)
        self.wgrad_one_lane = o.wgrad_lanes == 2
        self.relu_bits = o.relu_bits                                        # block-output ReLU masks kept as bitmaps
        # ... and applied by the data-gradient pass that finishes the gradient of an identity-shortcut block's output
        # (sfk_conv_desc.out_relu_bits): that block's BatchNorm backward then reads dz as it is, no mask, no rewrite
        self.relu_out_mask = self.relu_bits and o.relu_out_mask
        self.kvec = 8 if self.dtype == torch.bfloat16 else 4               # channels per 16-byte lane of the BN kernels
        # split sums of the filter gradients: fp32 atomics (default: on their own lanes the atomic latency hides behind
        # the pathway's chain, 897 vs 886 clips/s) or the partial-tile workspace + ordered reduce (bit-reproducible dW)
        self.deterministic_wgrad = o.deterministic_wgrad
        # The bottleneck tail conv_c -> norm_c -> (+ shortcut) -> ReLU without the conv output in HBM (include/sfk.h,
        # sfk_bn_tail_*): statistics from the Gram matrix of conv_c's input, BatchNorm + shortcut + ReLU in the conv
        # epilogue, and a backward that needs neither y_c nor dy_c.  Per block with conv_c inputs of >= tail_min_c channels.
        self.fuse_tail = o.fuse_tail and hasattr(self.be, "bn_tail_fwd")
        self.tail_min_c = o.tail_min_c
        # upper bound: the tail's fixed cost is O(cout * c^2) (T = W G, W^T B W) whatever the map size, what it saves is
        # O(pixels * cout) -- it pays on the large maps of the early stages (c <= 128), not on res4 / res5 (c = 256 / 512)
        self.tail_max_c = o.tail_max_c
        # the stems' BatchNorm -> ReLU -> MaxPool as one forward pass and a two-pass backward (sfk_bn_maxpool_*)
        self.fuse_stem_tail = (o.fuse_stem_tail and hasattr(self.be, "bn_maxpool_fwd") and self.be.bn_maxpool_supported(3, 2, 1))
        # R = dz^T a beside the first dgrad pass (_tail_bwd): 0 = on the pathway's lane (default), 2 = on its filter-gradient lane
        # (neutral), 4 = on lanes of its own (an experiment: 948 clips/s with the default 4 hardware queues, 719 with 8)
        self.fuse_tail_dg = o.fuse_tail_dg                                  # R and the first dgrad pass in one kernel (_tail_bwd)
        self.tail_r_lane = o.tail_r_lane
        # 0 slow pathway / trunk, 1 fast pathway, 2 / 3 filter gradients of the slow / fast pathway.  Not more: a process gets
        # 4 hardware queues (GPU_MAX_HW_QUEUES), streams beyond that share one and serialise (6 lanes: 948 vs 1033 clips/s)
        self.NLANES = 6 if o.tail_r_lane == 4 else 4
        self._side = None
        self.drop_seed = torch.full((1,), 0x5EED0000 + seed, dtype=torch.int64, device=self.device)

    # ------------------------------------------------------------------ parameters
    def _new(self, *shape, dtype=torch.float32) -> torch.Tensor:
        return torch.zeros(*shape, dtype=dtype, device=self.device)

    def _build_params(self, seed: int):
        W = self.wiring
        off = 0
        layers: List[_Layer] = []
        for cb in W.all_convbn():
            g = cb.geom
            if cb.is_stem:
                # stem filters live in the stem layout [co][((f*cin+ci)*7+kh)*8+kw] (sfk_stem_conv_fwd)
                assert g.k[1:] == (7, 7) and g.s == (1, 2, 2) and g.p == (g.k[0] // 2, 3, 3), g
                kp = stem_kp(g.cin, g.k[0])
                eg = ConvGeom(kp, g.cout, (1, 1, 1))          # one row of kp "channels" per output channel
                L = _Layer(cb, eg, g.k[0] * g.cin * 7 * 8, needs_dgrad=False)
            else:
                assert g.cin % self.vec == 0 and g.cout % 4 == 0, (cb.conv_key, g)
                L = _Layer(cb, g, 0)
            L.w_off, L.w_numel = off, L.eg.cout * L.eg.wtaps * L.eg.cin
            off += round_up(L.w_numel, self.vec)
            layers.append(L)
            self._layers[cb.conv_key] = L
        self.conv_total = off
        self.fc_w_off = off
        self.fc_in, self.fc_out = W.head_in, self.spec.num_class
        off += round_up(self.fc_in * self.fc_out, self.vec)
        self.fc_b_off = off
        off += round_up(self.fc_out, self.vec)
        for L in layers:
            L.g_off = off
            off += round_up(L.c, self.vec)
            L.b_off = off
            off += round_up(L.c, self.vec)
        self.layers = layers
        self.arena_numel = off
        self.P = torch.nn.Parameter(self._new(off), requires_grad=True)
        self.G = self._new(off)
        self.adam_m = None
        self.adam_v = None
        self.adam_step = None
        # compute-precision copies of the filters: S = [cout][tap][cin] (forward, filter-gradient layout),
        # St = [cin][tap][cout] (data-gradient pass)
        self.S = self.P.data if self.dtype == torch.float32 else self._new(self.conv_total, dtype=self.dtype)
        self.St = self._new(self.conv_total, dtype=self.dtype)
        for L in layers:
            L.rm = self._new(L.c)
            L.rv = torch.ones(L.c, dtype=torch.float32, device=self.device)
            L.nbt = self._new(1, dtype=torch.int64)
        self.dead: Dict[str, torch.Tensor] = {}
        self.reset_parameters(seed)

    def reset_parameters(self, seed: int = 0):
        """'resnet' init of the reference stack: conv Kaiming-normal(fan_out), BN weight 1
        (0 for the block-final BN), BN bias 0, Linear N(0, 0.01) / bias 0."""
        gen = torch.Generator(device="cpu").manual_seed(seed)
        P = self.P.data
        P.zero_()
        for L in self.layers:
            g = L.cb.geom
            std = math.sqrt(2.0 / (g.cout * g.wtaps))
            w = torch.randn(g.cout, g.cin, *g.k, generator=gen) * std
            P[L.w_off:L.w_off + L.w_numel] = self._to_engine_layout(L, w).to(self.device)
            P[L.g_off:L.g_off + L.c] = 0.0 if L.cb.zero_init_gamma else 1.0
            L.rm.zero_()
            L.rv.fill_(1.0)
            L.nbt.zero_()
        P[self.fc_w_off:self.fc_w_off + self.fc_in * self.fc_out] = (
            torch.randn(self.fc_out * self.fc_in, generator=gen) * 0.01).to(self.device)
        for d in self.wiring.dead:
            if d.kind == "conv_w":
                fan_out = d.shape[0] * d.shape[2] * d.shape[3] * d.shape[4]
                t = torch.randn(*d.shape, generator=gen) * math.sqrt(2.0 / fan_out)
            elif d.kind in ("bn_w", "bn_rv"):
                t = torch.ones(*d.shape)
            elif d.kind == "bn_nbt":
                t = torch.zeros((), dtype=torch.int64)
            else:
                t = torch.zeros(*d.shape)
            self.dead[d.key] = t.to(self.device)

    def _to_engine_layout(self, L: _Layer, w: torch.Tensor) -> torch.Tensor:
        """(cout, cin, kt, kh, kw) -> flat [cout][tap][cin]  (stems: [cout][kt][(kh,kw,cin) zero-padded])."""
        g = L.cb.geom
        w = w.reshape(g.cout, g.cin, *g.k)
        if L.cb.is_stem:
            w = torch.nn.functional.pad(w.permute(0, 2, 1, 3, 4), (0, 1))           # co, kt, ci, kh, kw(7 -> 8)
            w = torch.nn.functional.pad(w.reshape(g.cout, L.kreal), (0, L.eg.cin - L.kreal))
            return w.reshape(-1).float()
        return w.permute(0, 2, 3, 4, 1).reshape(-1).float()              # co, kt, kh, kw, ci

    def _from_engine_layout(self, L: _Layer, flat: torch.Tensor) -> torch.Tensor:
        g = L.cb.geom
        if L.cb.is_stem:
            w = flat.reshape(g.cout, L.eg.cin)[:, :L.kreal].reshape(g.cout, g.k[0], g.cin, 7, 8)[..., :7]
            return w.permute(0, 2, 1, 3, 4).contiguous()
        w = flat.reshape(g.cout, g.k[0], g.k[1], g.k[2], g.cin)
        return w.permute(0, 4, 1, 2, 3).contiguous()

    # checkpoint surface: flat dict with pytorchvideo key names, fp32, reference tensor shapes
    def state_dict(self) -> Dict[str, torch.Tensor]:
        sd: Dict[str, torch.Tensor] = {}
        P = self.P.data
        for L in self.layers:
            sd[L.cb.conv_key + ".weight"] = self._from_engine_layout(L, P[L.w_off:L.w_off + L.w_numel]).clone()
            nk = L.cb.norm_key
            sd[nk + ".weight"] = P[L.g_off:L.g_off + L.c].clone()
            sd[nk + ".bias"] = P[L.b_off:L.b_off + L.c].clone()
            sd[nk + ".running_mean"] = L.rm.clone()
            sd[nk + ".running_var"] = L.rv.clone()
            sd[nk + ".num_batches_tracked"] = L.nbt[0].clone()
        sd[self.wiring.head_key + ".weight"] = P[self.fc_w_off:self.fc_w_off + self.fc_in * self.fc_out].reshape(
            self.fc_out, self.fc_in).clone()
        sd[self.wiring.head_key + ".bias"] = P[self.fc_b_off:self.fc_b_off + self.fc_out].clone()
        for k, v in self.dead.items():
            sd[k] = v.clone()
        return sd

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        own = self.state_dict()
        missing = [k for k in own if k not in sd]
        unexpected = [k for k in sd if k not in own]
        bad = [k for k in own if k in sd and tuple(sd[k].shape) != tuple(own[k].shape)]
        if bad:
            raise RuntimeError("size mismatch for " + ", ".join(bad))
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict: missing {missing}, unexpected {unexpected}")
        P = self.P.data
        dev = self.device

        def get(k):
            return sd[k].to(dev) if k in sd else None

        with torch.no_grad():
            for L in self.layers:
                w = get(L.cb.conv_key + ".weight")
                if w is not None:
                    P[L.w_off:L.w_off + L.w_numel] = self._to_engine_layout(L, w.float())
                nk = L.cb.norm_key
                for suffix, dst in ((".weight", P[L.g_off:L.g_off + L.c]), (".bias", P[L.b_off:L.b_off + L.c]),
                                    (".running_mean", L.rm), (".running_var", L.rv)):
                    v = get(nk + suffix)
                    if v is not None:
                        dst.copy_(v.float())
                v = get(nk + ".num_batches_tracked")
                if v is not None:
                    L.nbt.fill_(int(v))
            v = get(self.wiring.head_key + ".weight")
            if v is not None:
                P[self.fc_w_off:self.fc_w_off + self.fc_in * self.fc_out] = v.float().reshape(-1)
            v = get(self.wiring.head_key + ".bias")
            if v is not None:
                P[self.fc_b_off:self.fc_b_off + self.fc_out] = v.float()
            for k in self.dead:
                if k in sd:
                    self.dead[k] = sd[k].to(dev).clone()

        class _Keys:
            missing_keys = missing
            unexpected_keys = unexpected
        return _Keys()

    def num_parameters(self, live_only: bool = False) -> int:
        n = sum(L.cb.geom.cout * L.cb.geom.cin * L.cb.geom.wtaps + 2 * L.c for L in self.layers)
        n += self.fc_in * self.fc_out + self.fc_out
        if not live_only:
            n += sum(int(v.numel()) for k, v in self.dead.items()
                     if not k.endswith(("running_mean", "running_var", "num_batches_tracked")))
        return n

    # ------------------------------------------------------------------ buffers
    def _buf(self, tag: str, numel: int, dtype=None) -> torch.Tensor:
        dtype = self.dtype if dtype is None else dtype
        key = f"{tag}|{dtype}"
        t = self._bufs.get(key)
        if t is None or t.numel() < numel:
            t = torch.zeros(max(numel, 1), dtype=dtype, device=self.device)
            self._bufs[key] = t
        return t

    def _fmap(self, tag: str, n, t, h, w, c, ld=None) -> FMap:
        ld = c if ld is None else ld
        return FMap(self._buf(tag, n * t * h * w * ld), n, t, h, w, c, ld, 0)

    def _fold_ws(self, tag: str, c: int) -> torch.Tensor:
        """scratch of the two-level BatchNorm partial fold (sfk_bn_finalize); one per unit: the pathways overlap"""
        return self._buf(f"foldws.{tag}", BN_FOLD_ROWS * c * 2, torch.float32)

    def _pslice(self, off: int, n: int) -> torch.Tensor:
        return self.P.data[off:off + n]

    def _gslice(self, off: int, n: int) -> torch.Tensor:
        return self.G[off:off + n]

    # ------------------------------------------------------------------ plan construction
    def _conv(self, pl: Plan, L: _Layer, x: FMap, y: FMap, stats_tag: Optional[str]):
        """forward conv pass x -> y (+ BatchNorm partial statistics when stats_tag). returns (stats, mtiles)"""
        sp = fwd_pass(L.eg, (x.t, x.h, x.w))
        w = self.S[L.w_off:L.w_off + L.w_numel]
        p = ConvPass(x, y, sp.rows, sp.gs, sp.os, sp.oo, list(sp.taps), w, L.eg.wtaps, L.eg.cin, L.eg.cout)
        stats, mt = None, 0
        if stats_tag is not None:
            mt = self.be.conv_igemm_mtiles(p)       # rows the pass will leave in `stats` (kernel-family specific)
            stats = self._buf(stats_tag, mt * L.c * 2, torch.float32)
            p.stats = stats
        rows = x.n * sp.rows[0] * sp.rows[1] * sp.rows[2]
        esz = 2 if self.dtype == torch.bfloat16 else 4
        pl.fwd.append(self.be.conv_igemm(p), kind="conv_fwd", layer=L.cb.conv_key, cout=L.eg.cout,
                      flops=2.0 * rows * L.eg.cout * L.eg.cin * L.eg.wtaps,
                      bytes=float(esz * (x.pixels * L.eg.cin + rows * L.eg.cout + L.w_numel)))
        return stats, mt

    def _unit_fwd(self, pl: Plan, L: _Layer, x: FMap, tag: str, train: bool, n: int, defer: bool = False):
        """conv + BatchNorm coefficients.  returns (y, scale, shift, rec).  defer: the caller's next op is _apply(y, scale, shift,
        ...) on the same lane -- the finalize then rides that launch as its prologue (EngineOptions.fuse_finalize)"""
        od = L.eg.out_dims((x.t, x.h, x.w))
        y = self._fmap(f"y.{tag}", n, od[0], od[1], od[2], L.c)
        scale = self._buf(f"scale.{tag}", L.c, torch.float32)
        shift = self._buf(f"shift.{tag}", L.c, torch.float32)
        gamma, beta = self._pslice(L.g_off, L.c), self._pslice(L.b_off, L.c)
        rec = None
        if train:
            stats, mt = self._conv(pl, L, x, y, f"stats.{tag}")
            mean = self._buf(f"mean.{tag}", L.c, torch.float32)
            invstd = self._buf(f"invstd.{tag}", L.c, torch.float32)
            if defer and self.options.fuse_finalize and L.c <= min(512, self.options.fuse_finalize_max_c) and hasattr(self.be, "bn_finalize_apply"):
                self._pending_fin[id(scale)] = (stats, mt, y.pixels, gamma, beta, self.spec.bn_eps, self.spec.bn_momentum, L.rm, L.rv,
                                                L.nbt, mean, invstd, self._fold_ws(tag, L.c), self._buf(f"finsync.{tag}", 2144, torch.int32))
            else:
                pl.fwd.append(self.be.bn_finalize(stats, mt, L.c, y.pixels, gamma, beta, self.spec.bn_eps,
                                                  self.spec.bn_momentum, L.rm, L.rv, L.nbt, mean, invstd, scale, shift,
                                                  self._fold_ws(tag, L.c)), kind="bn_finalize")
            rec = _UnitRec(L, x, y, mean, invstd, scale, shift)
        else:
            self._conv(pl, L, x, y, None)
            pl.fwd.append(self.be.bn_eval_coeffs(gamma, beta, L.rm, L.rv, self.spec.bn_eps, L.c, scale, shift))
        return y, scale, shift, rec

    def _apply(self, pl: Plan, y: FMap, scale, shift, res, res_scale, res_shift, relu: bool, out: FMap, bits=None):
        esz = 2 if self.dtype == torch.bfloat16 else 4
        fin = self._pending_fin.pop(id(scale), None)
        if fin is not None:     # this BatchNorm's finalize was deferred to here (_unit_fwd(defer=True)): one launch for both
            run = self.be.bn_finalize_apply(*fin, y, scale, shift, res, res_scale, res_shift, relu, out, relu_bits=bits)
        else:
            run = self.be.bn_apply(y, scale, shift, res, res_scale, res_shift, relu, out, relu_bits=bits)
        pl.fwd.append(run,
                      kind="bn_apply",
                      bytes=float(y.pixels * y.c * esz * (2 + (1 if res is not None else 0))
                                  + (y.pixels * y.c // self.kvec if bits is not None else 0)))

    def _bn_bwd(self, pl: Plan, rec: _UnitRec, da: FMap, tag: str, relu: bool, mask_src: Optional[FMap],
                dz_inplace: bool, dy: FMap, reduced=None, bits=None):
        """BatchNorm(+ReLU) backward of one unit: da -> dy (may alias da), accumulates dgamma/dbeta.
        bits: the ReLU mask the forward bn_apply left (1 bit per element), read instead of mask_src (needs dz_inplace).
        reduced = (partials, rows): the pass that produced da already masked it (da holds dz) and left the partial
        sums (fused epilogue, _dgrad(fuse=...)), so the reduce kernel is skipped."""
        L = rec.L
        coef = self._buf(f"coef.{tag}", L.c * 3, torch.float32)
        esz = 2 if self.dtype == torch.bfloat16 else 4
        el = float(rec.y.pixels * L.c * esz)
        if reduced is not None:
            parts, np_ = reduced
            dz_inplace = True
        else:
            parts = self._buf(f"bparts.{tag}", self.max_parts * L.c * 2, torch.float32)
            if bits is not None:
                assert dz_inplace
                mask_src = None
            run, np_ = self.be.bn_bwd_reduce(da, rec.y, mask_src, rec.mean, rec.invstd, rec.scale, rec.shift, relu,
                                             da if dz_inplace else None, parts, self.max_parts, relu_bits=bits)
            pl.bwd.append(run, kind="bn_bwd_reduce", layer=L.cb.norm_key,
                          bytes=el * (2 + (1 if mask_src is not None else 0) + (1 if dz_inplace else 0))
                          + (rec.y.pixels * L.c // self.kvec if bits is not None else 0))
        if self.options.fuse_finalize and L.c <= min(512, self.options.fuse_finalize_max_c) and hasattr(self.be, "bn_bwd_finalize_apply"):
            # the finalize rides the apply launch as its prologue (sfk_bn_bwd_finalize_apply): dgamma / dbeta are complete behind it
            fin = (parts, np_, rec.y.pixels, self._pslice(L.g_off, L.c), self._gslice(L.g_off, L.c), self._gslice(L.b_off, L.c), coef,
                   self._fold_ws(tag, L.c), self._buf(f"bfinsync.{tag}", 2144, torch.int32))
            if dz_inplace:   # the mask is already applied to da
                run, nb = self.be.bn_bwd_finalize_apply(*fin, da, rec.y, None, rec.mean, rec.invstd, rec.scale, rec.shift, False, dy), 3
            else:
                run = self.be.bn_bwd_finalize_apply(*fin, da, rec.y, mask_src, rec.mean, rec.invstd, rec.scale, rec.shift, relu, dy)
                nb = 3 + (1 if mask_src is not None else 0)
            pl.bwd.append(run, kind="bn_bwd_apply", layer=L.cb.norm_key, bytes=el * nb)
            pl.grad_marks.append((len(pl.bwd), (L.g_off, L.b_off + round_up(L.c, self.vec) - L.g_off)))
            return
        pl.bwd.append(self.be.bn_bwd_finalize(parts, np_, L.c, rec.y.pixels, self._pslice(L.g_off, L.c), rec.invstd,
                                              self._gslice(L.g_off, L.c), self._gslice(L.b_off, L.c), coef,
                                              self._fold_ws(tag, L.c)), kind="bn_finalize")
        pl.grad_marks.append((len(pl.bwd), (L.g_off, L.b_off + round_up(L.c, self.vec) - L.g_off)))
        if dz_inplace:   # the mask is already applied to da
            pl.bwd.append(self.be.bn_bwd_apply(da, rec.y, None, rec.mean, rec.invstd, rec.scale, rec.shift, False,
                                               coef, dy), kind="bn_bwd_apply", layer=L.cb.norm_key, bytes=el * 3)
        else:
            pl.bwd.append(self.be.bn_bwd_apply(da, rec.y, mask_src, rec.mean, rec.invstd, rec.scale, rec.shift, relu,
                                               coef, dy), kind="bn_bwd_apply", layer=L.cb.norm_key,
                          bytes=el * (3 + (1 if mask_src is not None else 0)))

    def _wgrad(self, pl: Plan, rec: _UnitRec, dy: FMap):
        L = rec.L
        esz = 2 if self.dtype == torch.bfloat16 else 4
        wp = WgradPass(rec.x, dy, L.eg.s, list(wgrad_taps(L.eg)), self._gslice(L.w_off, L.w_numel), L.eg.wtaps,
                       L.eg.cin, L.eg.cout)
        # optional scratch for the split sums (deterministic_wgrad): one buffer per lane (launches on a stream are ordered), grown to the
        # largest request; the descriptor keeps the pointer, so size it before binding
        # Filter gradients feed nothing downstream in the step (only Adam reads dW), so they leave the pathway's
        # dependency chain (BN backward -> dgrad -> BN backward ...) for a lane of their own and fill the gaps that chain
        # leaves on the chip; the lane waits for the producer of dy, everything joins before the optimiser.
        home = pl.bwd.cur_lane
        wants = hasattr(self.be, "conv_wgrad_wants_workspace") and self.be.conv_wgrad_wants_workspace(wp)
        wl = (2 if self.wgrad_one_lane else home + 2) if self.wgrad_lanes else home
        # the MFMA-bound layers' 256 x 256 tiles own every CU they run on (one workgroup per CU): options.mfma_wgrad_trunk keeps
        # them on the pathway's own lane, directly behind the data gradient that read the same dY (still in L2 / MALL), instead of
        # beside the trunk's kernels
        if wants and self.options.mfma_wgrad_trunk:
            wl = home
        if wl != home:
            pl.bwd.sync(wl, home)
            pl.bwd.cur_lane = wl
        need = self.be.conv_wgrad_workspace_bytes(wp) if (self.deterministic_wgrad or wants) else 0
        if need > 0:
            cap = self._wg_ws_need.get(wl, 0)
            self._wg_ws_need[wl] = max(cap, need)
            self._wg_pending.append((pl.bwd, len(pl.bwd), wp, wl))
        pl.bwd.append(self.be.conv_wgrad(wp),
                      kind="conv_wgrad", layer=L.cb.conv_key, cout=L.eg.cout,
                      flops=2.0 * dy.pixels * L.eg.cout * L.eg.cin * L.eg.wtaps,
                      bytes=float(esz * (rec.x.pixels * L.eg.cin + dy.pixels * L.eg.cout) + 4 * L.w_numel))
        pl.grad_marks.append((len(pl.bwd), (L.w_off, round_up(L.w_numel, self.vec))))
        pl.bwd.cur_lane = home

    def _dgrad(self, pl: Plan, rec: _UnitRec, dy: FMap, dx: FMap, accumulate: bool, fuse=None, out_bits=None,
               out_sum_tag=None):
        """data gradient of rec's conv: dy -> dx (+= when accumulate).
        fuse = (bn unit whose activation gradient dx is, mask_src | None, relu, tag): fold that unit's BatchNorm-backward
        reduce into this pass's epilogue when the backend can (single stride-1 pass, bf16, > 16 channels); returns
        (partials, rows) then -- dx holds dz and the caller skips the reduce kernel -- else None.
        out_bits: dx is the gradient w.r.t. a ReLU output whose mask is this bitmap: the pass stores result * mask when
        the backend can (single pass over all of dx); returns "masked" then."""
        L = rec.L
        assert L.needs_dgrad
        passes, needs_zero = dgrad_passes(L.eg, (rec.x.t, rec.x.h, rec.x.w))
        if needs_zero and not accumulate:
            assert dx.ld == dx.c and dx.c_off == 0, "zero-fill of a channel slice is not supported"
            pl.bwd.append(self.be.fill_zero(dx.buf[: dx.pixels * dx.ld]))
        wt = self.St[L.w_off:L.w_off + L.w_numel]
        esz = 2 if self.dtype == torch.bfloat16 else 4
        reduced = None
        for sp in passes:
            rows = dy.n * sp.rows[0] * sp.rows[1] * sp.rows[2]
            cp = ConvPass(dy, dx, sp.rows, sp.gs, sp.os, sp.oo, list(sp.taps), wt, L.eg.wtaps, L.eg.cout, L.eg.cin,
                          accumulate=accumulate)
            extra = 0.0
            if fuse is not None and self.fuse_bn_bwd and len(passes) == 1 and self.be.conv_bnb_supported(cp):
                brec, mask_src, relu, tag = fuse
                # the tile -- hence the number of partial rows -- depends on the epilogue: ask with the descriptor AS LAUNCHED
                cp.bnb = BnBwdFuse(brec.y, mask_src, brec.mean, brec.invstd, brec.scale, brec.shift, relu,
                                   self._buf(f"bparts.{tag}", brec.L.c * 2, torch.float32))
                mt = self.be.conv_igemm_mtiles(cp)
                parts = self._buf(f"bparts.{tag}", max(mt, 1) * brec.L.c * 2, torch.float32)
                cp.bnb.partials = parts
                reduced = (parts, mt)
                extra = float(esz * rows * L.eg.cin * (2 if mask_src is not None else 1))
            elif out_sum_tag is not None:
                # dx is the output gradient of a block with a fused tail: its backward wants dz AND sum dz.  One pass that can
                # (bitmap mask + per-tile sums in the epilogue: sfk_bn_bwd_fuse with y_bn = NULL) does both; otherwise dx
                # stays unmasked and that block's own mask pass (sfk_bn_bwd_reduce, y = NULL) does it
                if out_bits is not None and len(passes) == 1 and self.be.conv_relu_out_supported(cp) \
                        and self.be.conv_bnb_supported(cp):
                    cp.relu_out_bits = out_bits
                    cp.bnb = BnBwdFuse(None, None, None, None, None, None, True,
                                       self._buf(f"bparts.{out_sum_tag}.c", self.max_parts * L.eg.cin * 2, torch.float32))
                    # the row count is a property of the descriptor AS LAUNCHED (the tile depends on the epilogue; the streaming
                    # pointwise kernel leaves one row per wave, at most max_parts): ask now that the epilogue is set
                    mt = self.be.conv_igemm_mtiles(cp)
                    parts = self._buf(f"bparts.{out_sum_tag}.c", max(mt, self.max_parts) * L.eg.cin * 2, torch.float32)
                    cp.bnb.partials = parts
                    assert parts.numel() >= mt * L.eg.cin * 2
                    reduced = ("masked+sum", parts, mt)
                    extra = float(rows * L.eg.cin // self.kvec)
            elif out_bits is not None and len(passes) == 1 and self.be.conv_relu_out_supported(cp):
                cp.relu_out_bits = out_bits
                reduced = "masked"
                extra = float(rows * L.eg.cin // self.kvec)
            pl.bwd.append(self.be.conv_igemm(cp),
                          kind="conv_dgrad", layer=L.cb.conv_key, cout=L.eg.cin,
                          flops=2.0 * rows * L.eg.cout * L.eg.cin * len(sp.taps),
                          bytes=float(esz * (dy.pixels * L.eg.cout / len(passes) + rows * L.eg.cin * (2 if accumulate else 1)
                                             + L.w_numel)) + extra)
        return reduced

    # ---- stem: direct (kt,7,7)/(1,2,2) conv from the clip -> BN -> ReLU -> MaxPool
    def _stem_ops(self, pl: Plan, p: int, x5: torch.Tensor, t_index, rec=None):
        """(re)create the two ops that hold the clip's address: forward conv and filter gradient"""
        L = self._layers[self.wiring.stems[p].conv_key]
        src = StemSrc(x5, t_index, L.cb.geom.k[0])
        st = pl.stem_state[p]
        esz = 2 if self.dtype == torch.bfloat16 else 4
        flops = 2.0 * st["y"].pixels * L.c * L.cb.geom.cin * L.cb.geom.wtaps
        fwd = self.be.stem_conv_fwd(src, self.S[L.w_off:L.w_off + L.w_numel], st["y"], st["stats"])
        meta = dict(kind="stem_fwd", layer=L.cb.conv_key, cout=L.c, flops=flops,
                    bytes=float(x5.numel() * x5.element_size() + st["y"].pixels * L.c * esz))
        if st["fwd_slot"] is None:
            st["fwd_slot"] = len(pl.fwd)
            pl.fwd.append(fwd, **meta)
        else:
            pl.fwd[st["fwd_slot"]] = fwd
        if st.get("da") is not None:
            bwd = self.be.stem_conv_wgrad(src, st["da"], self._gslice(L.w_off, L.w_numel))
            if st["bwd_slot"] is None:
                st["bwd_slot"] = len(pl.bwd)
                if p == 1:
                    # the fast stem's filter gradient is the LAST kernel of the step (the trunk has finished ~0.4 ms before it):
                    # TrainStep updates everything above `cut` beside it and only the stems' filters after it (adam_split_ops)
                    cut = L.w_off + round_up(L.w_numel, self.vec)
                    if all(M.w_off >= cut or M.cb.is_stem for M in self.layers):
                        pl.tail_cut = (len(pl.bwd), cut)
                pl.bwd.append(bwd, kind="stem_wgrad", layer=L.cb.conv_key, cout=L.c, flops=flops, bytes=meta["bytes"])
                pl.grad_marks.append((len(pl.bwd), (L.w_off, round_up(L.w_numel, self.vec))))
            else:
                pl.bwd[st["bwd_slot"]] = bwd

    def _stem_fwd(self, pl, p: int, x5: torch.Tensor, t_index, out: FMap, train: bool):
        L = self._layers[self.wiring.stems[p].conv_key]
        g = L.cb.geom
        n, _, t_in, h_in, w_in = x5.shape
        t_out = t_in if t_index is None else int(t_index.numel())
        ho = (h_in + 2 * g.p[1] - g.k[1]) // g.s[1] + 1
        wo = (w_in + 2 * g.p[2] - g.k[2]) // g.s[2] + 1
        tag = f"stem{p}"
        y = self._fmap(f"y.{tag}", n, t_out, ho, wo, L.c)
        scale = self._buf(f"scale.{tag}", L.c, torch.float32)
        shift = self._buf(f"shift.{tag}", L.c, torch.float32)
        gamma, beta = self._pslice(L.g_off, L.c), self._pslice(L.b_off, L.c)
        st = {"y": y, "stats": None, "fwd_slot": None, "bwd_slot": None, "da": None}
        pl.stem_state[p] = st
        rec = None
        if train:
            mt = self.be.stem_conv_tiles(StemSrc(x5, t_index, g.k[0]), y)
            st["stats"] = self._buf(f"stats.{tag}", mt * L.c * 2, torch.float32)
            self._stem_ops(pl, p, x5, t_index)
            mean = self._buf(f"mean.{tag}", L.c, torch.float32)
            invstd = self._buf(f"invstd.{tag}", L.c, torch.float32)
            pl.fwd.append(self.be.bn_finalize(st["stats"], mt, L.c, y.pixels, gamma, beta, self.spec.bn_eps,
                                              self.spec.bn_momentum, L.rm, L.rv, L.nbt, mean, invstd, scale, shift,
                                              self._fold_ws(tag, L.c)), kind="bn_finalize")
            rec = _UnitRec(L, None, y, mean, invstd, scale, shift)
        else:
            self._stem_ops(pl, p, x5, t_index)
            pl.fwd.append(self.be.bn_eval_coeffs(gamma, beta, L.rm, L.rv, self.spec.bn_eps, L.c, scale, shift))
        assert (out.h, out.w) == ((y.h + 2 - 3) // 2 + 1, (y.w + 2 - 3) // 2 + 1) and out.t == y.t
        argmax = self._buf(f"argmax.{p}", out.pixels * L.c, torch.uint8)
        esz = 2 if self.dtype == torch.bfloat16 else 4
        if self.fuse_stem_tail:
            # BatchNorm + ReLU + MaxPool in one pass: the activation map (411 MB for the slow stem of the metric) is never
            # written, and the backward rebuilds its gradient from d_out and the argmax bytes (sfk_bn_maxpool_*)
            pl.fwd.append(self.be.bn_maxpool_fwd(y, scale, shift, out, argmax, 3, 2, 1), kind="bn_apply",
                          bytes=float(esz * (y.pixels + out.pixels) * L.c + out.pixels * L.c))
            return (rec, None, argmax, out, x5, t_index)
        a = self._fmap(f"a.stem{p}", n, y.t, y.h, y.w, L.c)
        self._apply(pl, y, scale, shift, None, None, None, True, a)
        pl.fwd.append(self.be.maxpool_fwd(a, out, argmax, 3, 2, 1))
        return (rec, a, argmax, out, x5, t_index)

    def _stem_bwd(self, pl, p: int, srec, d_out: FMap):
        rec, a, argmax, out, x5, t_index = srec
        y, L = rec.y, rec.L
        da = self._fmap(f"da.stem{p}", y.n, y.t, y.h, y.w, y.c)
        if a is None:     # fused stem tail: pool routing + ReLU mask + BatchNorm backward in two passes over (d_out, argmax, y)
            tag = f"stem{p}"
            esz = 2 if self.dtype == torch.bfloat16 else 4
            rd = float(esz * (y.pixels + d_out.pixels) * L.c + d_out.pixels * L.c)
            coef = self._buf(f"coef.{tag}", L.c * 3, torch.float32)
            parts = self._buf(f"bparts.{tag}", self.max_parts * L.c * 2, torch.float32)
            run, np_ = self.be.bn_maxpool_bwd_reduce(d_out, argmax, y, rec.mean, rec.invstd, rec.scale, rec.shift, parts,
                                                     self.max_parts)
            pl.bwd.append(run, kind="bn_bwd_reduce", layer=L.cb.norm_key, bytes=rd)
            pl.bwd.append(self.be.bn_bwd_finalize(parts, np_, L.c, y.pixels, self._pslice(L.g_off, L.c), rec.invstd,
                                                  self._gslice(L.g_off, L.c), self._gslice(L.b_off, L.c), coef,
                                                  self._fold_ws(tag, L.c)), kind="bn_finalize")
            pl.grad_marks.append((len(pl.bwd), (L.g_off, L.b_off + round_up(L.c, self.vec) - L.g_off)))
            pl.bwd.append(self.be.bn_maxpool_bwd_apply(d_out, argmax, y, rec.mean, rec.invstd, rec.scale, rec.shift, coef, da),
                          kind="bn_bwd_apply", layer=L.cb.norm_key, bytes=rd + float(esz * y.pixels * L.c))
        else:
            pl.bwd.append(self.be.maxpool_bwd(d_out, argmax, da, 3, 2, 1))
            self._bn_bwd(pl, rec, da, f"stem{p}", True, None, False, da)
        pl.stem_state[p]["da"] = da
        self._stem_ops(pl, p, x5, t_index)

    # ---- lateral fusion: conv over the fast pathway -> BN -> ReLU -> channel slice of the slow buffer
    def _fusion_fwd(self, pl, bi: int, xf: FMap, out_slice: FMap, train: bool):
        L = self._layers[self.wiring.fusions[bi].conv_key]
        y, scale, shift, rec = self._unit_fwd(pl, L, xf, f"fuse{bi}", train, xf.n, defer=True)
        assert (y.t, y.h, y.w, y.c) == (out_slice.t, out_slice.h, out_slice.w, out_slice.c), \
            "lateral fusion: fast pathway does not line up with the slow pathway (T_fast / stride != T_slow?)"
        self._apply(pl, y, scale, shift, None, None, None, True, out_slice)
        return rec

    def _fusion_bwd(self, pl, bi: int, rec: _UnitRec, d_slice: FMap, d_xf: FMap):
        dy = self._fmap(f"dy.fuse{bi}", rec.y.n, rec.y.t, rec.y.h, rec.y.w, rec.y.c)
        self._bn_bwd(pl, rec, d_slice, f"fuse{bi}", True, None, False, dy)
        self._wgrad(pl, rec, dy)
        self._dgrad(pl, rec, dy, d_xf, accumulate=True)

    # ---- bottleneck tail: conv_c -> norm_c -> + shortcut -> ReLU with no conv output in HBM (sfk.h: sfk_bn_tail_*)
    TAP0 = [(0, 0, 0, 0)]

    def _tail_ok(self, Lc: _Layer) -> bool:
        g = Lc.cb.geom
        return (self.fuse_tail and self.relu_bits and g.k == (1, 1, 1) and g.s == (1, 1, 1)
                and self.tail_min_c <= g.cin <= min(512, self.tail_max_c) and g.cout > 16 and g.cout % self.kvec == 0)

    def _ws_wgrad(self, oplist: "OpList", wp: WgradPass, key, **meta):
        """a filter-gradient call that sums its pixel splits through the lane's scratch (deterministic); the scratch is
        sized and bound at the end of plan construction (_build_plan)"""
        need = self.be.conv_wgrad_workspace_bytes(wp)
        if need > 0:
            self._wg_ws_need[key] = max(self._wg_ws_need.get(key, 0), need)
            self._wg_pending.append((oplist, len(oplist), wp, key))
        oplist.append(self.be.conv_wgrad(wp), **meta)

    def _tail_fwd(self, pl, blk, Lc: _Layer, yb: FMap, sb, hb, res: FMap, rs, rh, out: FMap, tag: str, train: bool):
        """norm_b apply -> [Gram matrix -> statistics] -> conv_c with BatchNorm + shortcut + ReLU in its epilogue"""
        n, c4, C = yb.n, Lc.eg.cin, Lc.eg.cout
        V = self.kvec
        esz = 2 if self.dtype == torch.bfloat16 else 4
        ab = self._fmap(f"a.{tag}.b", n, yb.t, yb.h, yb.w, c4)
        scale = self._buf(f"scale.{tag}.c", C, torch.float32)
        shift = self._buf(f"shift.{tag}.c", C, torch.float32)
        gamma, beta = self._pslice(Lc.g_off, C), self._pslice(Lc.b_off, C)
        w = self.S[Lc.w_off:Lc.w_off + Lc.w_numel]
        tail = None
        if train:
            # norm_b apply writes a_b AND leaves its column sums g = 1^T a_b as partial rows (no constant-1 channel beside a_b:
            # on the fast pathway's 8 .. 32-channel maps the widened records doubled .. 1.25x-ed every pass over a_b)
            asums = self._buf(f"asums.{tag}", self.max_parts * c4 * 2, torch.float32)
            run, a_np = self.be.bn_apply(yb, sb, hb, None, None, None, True, ab, out_sums=asums, max_parts=self.max_parts)
            pl.fwd.append(run, kind="bn_apply", bytes=float(yb.pixels * c4 * esz * 2))
            gram = self._tail_zero("f", c4 * c4)
            lane = pl.fwd.cur_lane
            self._ws_wgrad(pl.fwd, WgradPass(ab, ab, (1, 1, 1), self.TAP0, gram, 1, c4, c4), f"f{lane}",
                           kind="conv_wgrad", layer=Lc.cb.conv_key + ":gram", cout=c4,
                           flops=2.0 * ab.pixels * c4 * c4, bytes=float(esz * ab.pixels * c4 + 4 * c4 * c4))
            mean = self._buf(f"mean.{tag}.c", C, torch.float32)
            invstd = self._buf(f"invstd.{tag}.c", C, torch.float32)
            t = self._buf(f"tailT.{tag}", C * c4, torch.float32)
            gvec = self._buf(f"tailg.{tag}", c4, torch.float32)
            wd = self._buf(f"tailWd.{tag}", C * c4)       # (A W)^T, A = gamma * invstd: the backward's first dgrad filter
            pl.fwd.append(self.be.bn_tail_fwd(gram, asums, a_np, ab.pixels, gvec, c4, w, C, gamma, beta, self.spec.bn_eps,
                                              self.spec.bn_momentum, Lc.rm, Lc.rv, Lc.nbt, mean, invstd, scale, shift, t, wd))
            tail = dict(ab=ab, g=gvec, count=ab.pixels, t=t, mean=mean, invstd=invstd, wd=wd)
        else:
            self._apply(pl, yb, sb, hb, None, None, None, True, ab)
            pl.fwd.append(self.be.bn_eval_coeffs(gamma, beta, Lc.rm, Lc.rv, self.spec.bn_eps, C, scale, shift))
        bits = self._buf(f"relubits.{tag}", out.pixels * (C // V), torch.uint8) if train else None
        cp = ConvPass(ab, out, (ab.t, ab.h, ab.w), (1, 1, 1), (1, 1, 1), (0, 0, 0), self.TAP0, w, 1, c4, C,
                      ep=ConvEpilogue(scale=scale, shift=shift, res=res, res_scale=rs, res_shift=rh, relu=True, relu_bits=bits))
        pl.fwd.append(self.be.conv_igemm(cp), kind="conv_fwd", layer=Lc.cb.conv_key, cout=C,
                      flops=2.0 * ab.pixels * C * c4,
                      bytes=float(esz * (ab.pixels * c4 + 2 * ab.pixels * C + Lc.w_numel) + (ab.pixels * C // V if train else 0)))
        return tail, bits

    def _tail_zero_layout(self, train: bool):
        """fp32 scratch that must be ZERO when its producer (a += filter-gradient call) runs -- the Gram matrices of the
        forward, R and W^T B W of the backward -- is carved out of one flat buffer per schedule, cleared by ONE fill at the
        schedule's start.  Sizes depend on the wiring only, so the buffers exist before the first op is built."""
        V = self.kvec
        nf = nb = 0
        for stage in self.wiring.stages:
            for blocks in stage:
                for blk in blocks:
                    Lc = self._layers[blk.conv_c.conv_key]
                    if self._tail_ok(Lc):
                        c4, C = Lc.eg.cin, Lc.eg.cout
                        nf += round_up(c4 ** 2, 64)
                        nb += round_up(C * c4, 64)
        self._tailz = {"f": self._buf("tailz.f", nf, torch.float32)[:nf] if (train and nf) else None,
                       "b": self._buf("tailz.b", nb, torch.float32)[:nb] if (train and nb) else None}
        self._tailz_off = {"f": 0, "b": 0}

    def _tail_zero(self, key: str, numel: int) -> torch.Tensor:
        off = self._tailz_off[key]
        self._tailz_off[key] = off + round_up(numel, 64)
        return self._tailz[key][off:off + numel]

    def _tail_bwd(self, pl, Lc: _Layer, tail: dict, d_out: FMap, dab: FMap, tag: str, dz_parts):
        """dz (in d_out) -> dgamma, dbeta, dW of conv_c / norm_c and d(a_b) in dab, without y_c or dy_c:
        R = dz^T a_b, s = sum dz as the partial rows dz_parts = (rows, count) of the kernel that wrote dz, the small algebra of sfk_bn_tail_bwd, then  d(a_b) = dz (A W) + a_b (W^T B W) + C W."""
        c4, C = Lc.eg.cin, Lc.eg.cout
        ab = tail["ab"]
        esz = 2 if self.dtype == torch.bfloat16 else 4
        w = self.S[Lc.w_off:Lc.w_off + Lc.w_numel]
        r = self._tail_zero("b", C * c4)
        wp = WgradPass(ab, d_out, (1, 1, 1), self.TAP0, r, 1, c4, C)
        meta = dict(kind="conv_wgrad", layer=Lc.cb.conv_key, cout=C, flops=2.0 * d_out.pixels * C * c4,
                    bytes=float(esz * (ab.pixels * c4 + d_out.pixels * C) + 4 * C * c4))
        # R = dz^T a_b may run BESIDE the first data-gradient pass dz (A W), whose filter the forward left (A = gamma * invstd
        # needs no gradient statistics): tail_r_lane puts it on the pathway's filter-gradient lane and the pathway waits for
        # it only before the small algebra that needs it.  Measured: see DESIGN.md section 4b (the lane carries a backlog of
        # earlier filter gradients; lanes of its own exceed the 4 hardware queues a process gets and serialise).
        # ... and where the filter-gradient tile has idle waves (slow res2: 256 x 64) those compute that first pass from the dz
        # rows the tile already holds: dz is read ONCE for R and dz (A W) (sfk_wgrad_desc.dg_w / dg_y)
        rows = (d_out.t, d_out.h, d_out.w)
        wp.dg_w, wp.dg_y = tail["wd"], dab
        fused_dg = self.fuse_tail_dg and hasattr(self.be, "conv_wgrad_dg_supported") and self.be.conv_wgrad_dg_supported(wp)
        if fused_dg:
            meta = dict(meta, flops=2.0 * meta["flops"], bytes=meta["bytes"] + float(esz * d_out.pixels * c4))
        else:
            wp.dg_w = wp.dg_y = None
        home = pl.bwd.cur_lane
        rl = home + self.tail_r_lane if (self.wgrad_lanes and self.tail_r_lane and not fused_dg) else home
        if rl != home:
            pl.bwd.sync(rl, home)
            pl.bwd.cur_lane = rl
        if self.deterministic_wgrad:
            self._ws_wgrad(pl.bwd, wp, f"b{rl}", **meta)
        else:
            pl.bwd.append(self.be.conv_wgrad(wp), **meta)
        pl.bwd.cur_lane = home
        # the narrowest maps (fast res2 / res3: 8 / 16 channels): both passes in ONE streaming kernel after the small algebra
        # (sfk_conv_pw_dual: da written once; two MFMA tiles that are all epilogue become one pass at HBM speed)
        dual = (not fused_dg and rl == home and self.tail_dual and hasattr(self.be, "conv_pw_dual_supported")
                and self.be.conv_pw_dual_supported(d_out, ab, dab))
        if not fused_dg and not dual:
            pl.bwd.append(self.be.conv_igemm(ConvPass(d_out, dab, rows, (1, 1, 1), (1, 1, 1), (0, 0, 0), self.TAP0, tail["wd"], 1, C, c4)),
                          kind="conv_dgrad", layer=Lc.cb.conv_key, cout=c4, flops=2.0 * d_out.pixels * C * c4,
                          bytes=float(esz * (d_out.pixels * C + d_out.pixels * c4 + Lc.w_numel)))
        if rl != home:
            pl.bwd.sync(home, rl)
        bias = self._buf(f"tailbias.{tag}", c4, torch.float32)
        coef = self._buf(f"tailcoef.{tag}", C * 4, torch.float32)
        m = self._buf(f"tailM.{tag}", c4 * c4)             # W^T diag(B) W, the filter of the second data-gradient pass
        pl.bwd.append(self.be.bn_tail_bwd(r, dz_parts[0], dz_parts[1], tail["g"], tail["count"], tail["t"], c4, w, C, self._pslice(Lc.g_off, C), tail["mean"],
                                          tail["invstd"], self._gslice(Lc.g_off, C), self._gslice(Lc.b_off, C),
                                          self._gslice(Lc.w_off, Lc.w_numel), m, bias, coef))
        pl.grad_marks.append((len(pl.bwd), (Lc.g_off, Lc.b_off + round_up(C, self.vec) - Lc.g_off)))
        pl.grad_marks.append((len(pl.bwd), (Lc.w_off, round_up(Lc.w_numel, self.vec))))
        if dual:
            pl.bwd.append(self.be.conv_pw_dual(d_out, tail["wd"], ab, m, bias, dab),
                          kind="conv_dgrad", layer=Lc.cb.conv_key + ":dual", cout=c4, flops=2.0 * d_out.pixels * (C + c4) * c4,
                          bytes=float(esz * d_out.pixels * (C + 2 * c4)))
            return
        pl.bwd.append(self.be.conv_igemm(ConvPass(ab, dab, rows, (1, 1, 1), (1, 1, 1), (0, 0, 0), self.TAP0, m, 1, c4, c4,
                                                  accumulate=True, ep=ConvEpilogue(shift=bias))),
                      kind="conv_dgrad", layer=Lc.cb.conv_key + ":m", cout=c4, flops=2.0 * d_out.pixels * c4 * c4,
                      bytes=float(esz * 3 * d_out.pixels * c4))

    # ---- bottleneck residual block
    def _block_fwd(self, pl, blk: arch.Block, x: FMap, out: FMap, tag: str, train: bool):
        n = x.n
        La, Lb, Lc = (self._layers[b.conv_key] for b in (blk.conv_a, blk.conv_b, blk.conv_c))
        rec1 = None
        y1 = s1 = h1 = None
        sl = None          # lane of the projection shortcut when it runs beside branch2 (below)
        if blk.branch1 is not None:
            L1 = self._layers[blk.branch1.conv_key]
            # The projection shortcut (conv + BatchNorm statistics) only meets branch2 at the block's last op: on the
            # four-lane training schedule it runs on the pathway's filter-gradient lane (idle in the forward) instead of in
            # front of conv_a on the pathway's own chain -- the trunk is the step's critical path
            home = pl.fwd.cur_lane
            if self.shortcut_lane_f and train and self.two_streams and self.wgrad_lanes and self.device.type == "cuda":
                sl = home + 2
                pl.fwd.sync(sl, home)
                pl.fwd.cur_lane = sl
            y1, s1, h1, rec1 = self._unit_fwd(pl, L1, x, f"{tag}.b1", train, n)
            pl.fwd.cur_lane = home
        ya, sa, ha, reca = self._unit_fwd(pl, La, x, f"{tag}.a", train, n, defer=True)
        aa = self._fmap(f"a.{tag}.a", n, ya.t, ya.h, ya.w, La.c)
        self._apply(pl, ya, sa, ha, None, None, None, True, aa)
        yb, sb, hb, recb = self._unit_fwd(pl, Lb, aa, f"{tag}.b", train, n, defer=not self._tail_ok(Lc))
        if sl is not None:
            pl.fwd.sync(pl.fwd.cur_lane, sl)          # the shortcut map and its coefficients are ready
        if self._tail_ok(Lc):
            assert (yb.t, yb.h, yb.w, Lc.c) == (out.t, out.h, out.w, out.c)
            res, rs, rh = (y1, s1, h1) if blk.branch1 is not None else (x, None, None)
            tail, bits = self._tail_fwd(pl, blk, Lc, yb, sb, hb, res, rs, rh, out, tag, train)
            return (blk, tag, x, out, rec1, reca, recb, tail, bits)
        ab = self._fmap(f"a.{tag}.b", n, yb.t, yb.h, yb.w, Lb.c)
        self._apply(pl, yb, sb, hb, None, None, None, True, ab)
        yc, sc, hc, recc = self._unit_fwd(pl, Lc, ab, f"{tag}.c", train, n, defer=True)
        assert (yc.t, yc.h, yc.w, yc.c) == (out.t, out.h, out.w, out.c)
        # the block output's ReLU mask, 1 bit per element, for the backward pass (which otherwise re-reads `out`)
        bits = self._buf(f"relubits.{tag}", out.pixels * (out.c // self.kvec), torch.uint8) if (train and self.relu_bits) else None
        if blk.branch1 is not None:
            self._apply(pl, yc, sc, hc, y1, s1, h1, True, out, bits)
        else:
            self._apply(pl, yc, sc, hc, x, None, None, True, out, bits)
        return (blk, tag, x, out, rec1, reca, recb, recc, bits)

    def _block_bwd(self, pl, brec, d_out: FMap, reduced_c=None, prev=None):
        """d_out: gradient w.r.t. the block output (clobbered).  returns (gradient w.r.t. the block input, reduced)
        reduced_c: (partials, rows) -- this block's final-BatchNorm reduce was already done by the pass that finished d_out
        (d_out holds dz) -- or "masked": that pass applied this block's ReLU bitmap (d_out holds dz, the reduce is to do);
        prev: the record of the block that feeds this one through an identity path -- its final-BatchNorm reduce is
        folded into this block's last data-gradient pass, whose result IS the gradient of prev's output; `reduced`
        is then what to hand to prev's _block_bwd."""
        blk, tag, x, out, rec1, reca, recb, recc, bits = brec
        n = x.n
        dab = self._fmap(f"da.{tag}.b", n, recb.y.t, recb.y.h, recb.y.w, recb.y.c)
        if isinstance(recc, dict):    # fused tail: no y_c, no dy_c (sfk_bn_tail_bwd)
            if isinstance(reduced_c, tuple) and reduced_c[0] == "masked+sum":
                dz_parts = reduced_c[1:]          # the pass that finished d_out masked it and left the partial sums of dz
            else:
                assert reduced_c is None
                esz = 2 if self.dtype == torch.bfloat16 else 4
                parts = self._buf(f"bparts.{tag}.c", self.max_parts * d_out.c * 2, torch.float32)
                run, np_ = self.be.bn_bwd_reduce(d_out, None, None, None, None, None, None, True, d_out, parts,
                                                 self.max_parts, relu_bits=bits)      # mask in place + partial sums
                pl.bwd.append(run, kind="bn_bwd_reduce", layer=f"{tag}.mask",
                              bytes=float(d_out.pixels * d_out.c * (2 * esz) + d_out.pixels * d_out.c // self.kvec))
                dz_parts = (parts, np_)
            self._tail_bwd(pl, self._layers[blk.conv_c.conv_key], recc, d_out, dab, tag, dz_parts)
            red_b = None
        else:
            # ReLU mask of the block output applied in place (d_out becomes dz, shared by branch2 and the shortcut)
            dyc = self._fmap(f"dy.{tag}.c", n, recc.y.t, recc.y.h, recc.y.w, recc.y.c)
            if reduced_c == "masked":     # the pass that finished d_out applied this block's ReLU bitmap: d_out holds dz
                self._bn_bwd(pl, recc, d_out, f"{tag}.c", False, None, False, dyc)
            else:
                self._bn_bwd(pl, recc, d_out, f"{tag}.c", True, out, True, dyc, reduced=reduced_c, bits=bits)
            self._wgrad(pl, recc, dyc)
            red_b = self._dgrad(pl, recc, dyc, dab, accumulate=False, fuse=(recb, None, True, f"{tag}.b"))
        # projection shortcut: its BatchNorm backward only needs dz (final by now: the conv_c part above masked / consumed it)
        # and only meets branch2 at the last data-gradient pass -- on the four-lane schedule it runs on the pathway's
        # filter-gradient lane into a buffer of its own instead of in place at the end of the pathway's chain
        sc_lane, dy1 = None, None
        if rec1 is not None and self.shortcut_lane_b and self.two_streams and self.wgrad_lanes and not self.wgrad_one_lane \
                and self.device.type == "cuda":
            home = pl.bwd.cur_lane
            sc_lane = home + 2
            dy1 = self._fmap(f"dy.{tag}.b1", n, d_out.t, d_out.h, d_out.w, d_out.c)
            pl.bwd.sync(sc_lane, home)
            pl.bwd.cur_lane = sc_lane
            self._bn_bwd(pl, rec1, d_out, f"{tag}.b1", False, None, False, dy1)
            pl.bwd.cur_lane = home
            self._wgrad(pl, rec1, dy1)
        self._bn_bwd(pl, recb, dab, f"{tag}.b", True, None, False, dab, reduced=red_b)
        self._wgrad(pl, recb, dab)
        daa = self._fmap(f"da.{tag}.a", n, reca.y.t, reca.y.h, reca.y.w, reca.y.c)
        red_a = self._dgrad(pl, recb, dab, daa, accumulate=False, fuse=(reca, None, True, f"{tag}.a"))
        self._bn_bwd(pl, reca, daa, f"{tag}.a", True, None, False, daa, reduced=red_a)
        self._wgrad(pl, reca, daa)
        if rec1 is None:
            # identity shortcut: dX = dz + dgrad_a; dX is the gradient of the previous block's output
            fuse = None
            if prev is not None and not isinstance(prev[7], dict):
                p_tag, p_out, p_recc = prev[1], prev[3], prev[7]
                fuse = (p_recc, p_out, True, f"{p_tag}.c")
            p_bits = prev[8] if (prev is not None and self.relu_out_mask) else None
            p_tail = prev is not None and isinstance(prev[7], dict)
            red_prev = self._dgrad(pl, reca, daa, d_out, accumulate=True, fuse=fuse, out_bits=p_bits,
                                   out_sum_tag=(prev[1] if p_tail else None))
            return d_out, red_prev
        dx = self._fmap(f"dx.{tag}", n, x.t, x.h, x.w, x.c)
        self._dgrad(pl, reca, daa, dx, accumulate=False)
        if sc_lane is not None:
            pl.bwd.sync(pl.bwd.cur_lane, sc_lane)      # the shortcut's BatchNorm backward ran beside branch2 (above)
        else:
            self._bn_bwd(pl, rec1, d_out, f"{tag}.b1", False, None, False, d_out)   # d_out already holds dz
            self._wgrad(pl, rec1, d_out)
            dy1 = d_out
        self._dgrad(pl, rec1, dy1, dx, accumulate=True)
        return dx, None

    def _stage_bwd(self, pl, brecs, d: FMap) -> FMap:
        """backward of one pathway's stage, last block first; each block's final data-gradient pass also reduces the
        previous block's last BatchNorm (identity shortcuts only: the first block of a stage ends in its projection)"""
        reduced = None
        for i in range(len(brecs) - 1, -1, -1):
            prev = brecs[i - 1] if i > 0 else None
            d, reduced = self._block_bwd(pl, brecs[i], d, reduced_c=reduced, prev=prev)
        return d

    def _build_plan(self, x_slow: torch.Tensor, x_fast: torch.Tensor, slow_t_index, train: bool) -> Plan:
        be, spec, W = self.be, self.spec, self.wiring
        pl = Plan()
        self._wg_ws_need, self._wg_pending = {}, []
        self._tail_zero_layout(train)
        NP = spec.pathways                              # 2: SlowFast; 1: the single-pathway `res3d` network (slow_r50)
        n = x_slow.shape[0]
        # ---- refresh the compute-precision filter copies from the fp32 master arena
        #      (one launch: cast into the forward layout and, for training, the data-gradient transposes)
        # On the four-lane training schedule the launch is split: the stems' filters on the trunk (a few KB), everything else
        # on the idle filter-gradient lane BESIDE the stems, which are the first 0.6 ms of the step and read no other filter;
        # the pathways wait for it before their first non-stem conv (the 110 us of the single launch sat in front of the step
        # with nothing to overlap them)
        refresh_lane = None
        deferred_refresh = None
        if self.dtype != torch.float32 or train:
            s_ = self.S if self.dtype != torch.float32 else None
            ent = lambda Ls: [(L.w_off, L.eg.cout, L.eg.wtaps, L.eg.cin, L.needs_dgrad) for L in Ls]
            stem_keys = {st.conv_key for st in W.stems}
            split = (train and NP == 2 and self.two_streams and self.wgrad_lanes and self.device.type == "cuda"
                     and self.options.split_refresh)
            if split:
                pl.fwd.append(be.filter_refresh(self.P.data, s_, self.St, ent([L for L in self.layers if L.cb.conv_key in stem_keys])))
                refresh_lane = 2
                rest_refresh = be.filter_refresh(self.P.data, s_, self.St, ent([L for L in self.layers if L.cb.conv_key not in stem_keys]))
                if self.options.refresh_behind_stems:
                    deferred_refresh = rest_refresh          # issued behind the stem convs, below
                else:
                    pl.fwd.sync(refresh_lane, 0)              # after the previous step's optimiser (trunk order)
                    pl.fwd.cur_lane = refresh_lane
                    pl.fwd.append(rest_refresh)
                    pl.fwd.cur_lane = 0
            else:
                pl.fwd.append(be.filter_refresh(self.P.data, s_, self.St if train else None, ent(self.layers)))
        if self._tailz["f"] is not None:
            pl.fwd.append(be.fill_zero(self._tailz["f"]))
        # ---- geometry after the stems
        def stem_out(x5, p, t_idx):
            g = W.stems[p].geom
            t = x5.shape[2] if t_idx is None else int(t_idx.numel())
            t = (t + 2 * g.p[0] - g.k[0]) // g.s[0] + 1
            h = (x5.shape[3] + 2 * g.p[1] - g.k[1]) // g.s[1] + 1
            w = (x5.shape[4] + 2 * g.p[2] - g.k[2]) // g.s[2] + 1
            return t, (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
        ts, hs, ws = stem_out(x_slow, 0, slow_t_index)
        c_s = spec.stem_dim_outs[0]
        fuse = spec.fuse

        def slow_buffer(tag, t, h, w, c, bi):
            """slow pathway activation of block index bi; wide enough to also hold the fused fast channels"""
            extra = W.fusions[bi].geom.cout if (fuse and bi < 4 and W.fusions[bi] is not None) else 0
            full = self._fmap(tag, n, t, h, w, c + extra)
            return full, full.channels(0, c)

        cat0, s0 = slow_buffer("cat.0", ts, hs, ws, c_s, 0)
        F_, B_ = pl.fwd, pl.bwd
        xf = None
        if NP == 2:
            tf, hf, wf = stem_out(x_fast, 1, None)
            xf = self._fmap("xf.0", n, tf, hf, wf, spec.stem_dim_outs[1])
            F_.sync(1, 0)                               # fork: the fast pathway starts after the filter refresh
        F_.cur_lane = 0
        stem_recs = [self._stem_fwd(pl, 0, x_slow, slow_t_index, s0, train)]
        if NP == 2:
            F_.cur_lane = 1
            stem_recs.append(self._stem_fwd(pl, 1, x_fast, None, xf, train))
        if deferred_refresh is not None:
            # The non-stem refresh (123 us, 0.27 GB) beside the stems slowed BOTH of them (stem class 0.56 -> 0.87 ms in the step):
            # it now waits for the two stem convs -- a Wait right behind each conv in the issue order, not behind the stems'
            # BatchNorm / pool kernels -- and runs on the filter-gradient lane under those; the pathways wait for it as before.
            s0, s1 = pl.stem_state[0]["fwd_slot"], pl.stem_state[1]["fwd_slot"]
            assert s0 < s1
            F_.insert_at(s1 + 1, Wait(refresh_lane, 1), refresh_lane)      # (the later position first: s0 stays valid)
            F_.insert_at(s1 + 2, deferred_refresh, refresh_lane)
            F_.insert_at(s0 + 1, Wait(refresh_lane, 0), refresh_lane)
            pl.stem_state[1]["fwd_slot"] = s1 + 1                          # the fast stem's conv moved down by that Wait
        fusion_recs = [None] * 4
        if refresh_lane is not None:                    # the non-stem filter copies are ready (split refresh, above)
            F_.sync(1, refresh_lane)
            F_.sync(0, refresh_lane)
        if fuse:
            fusion_recs[0] = self._fusion_fwd(pl, 0, xf, cat0.channels(c_s, cat0.c - c_s), train)
            F_.sync(0, 1)                               # the slow pathway reads the fused channels
        xs_full = cat0
        stage_recs = []
        xs_fulls, xfs = [cat0], [xf]
        for si in range(4):
            recs_sp = []
            for p in range(NP):
                F_.cur_lane = p
                blocks = W.stages[si][p]
                x = xs_full if p == 0 else xf
                brecs = []
                for i, blk in enumerate(blocks):
                    od = blk.conv_b.geom.out_dims((x.t, x.h, x.w))
                    cout = blk.conv_c.geom.cout
                    last = i == len(blocks) - 1
                    tag = f"s{si}p{p}b{i}"
                    if p == 0 and last:
                        full, out = slow_buffer(f"cat.{si + 1}", od[0], od[1], od[2], cout, si + 1)
                    else:
                        full = out = self._fmap(f"out.{tag}", n, od[0], od[1], od[2], cout)
                    brecs.append(self._block_fwd(pl, blk, x, out, tag, train))
                    x = out
                    if p == 0 and last:
                        xs_full_next = full
                recs_sp.append(brecs)
                if p == 1:
                    xf = x
            xs_full = xs_full_next
            c_slow = W.stages[si][0][-1].conv_c.geom.cout
            if fuse and si < 3:
                F_.cur_lane = 1
                fusion_recs[si + 1] = self._fusion_fwd(pl, si + 1, xf, xs_full.channels(c_slow, xs_full.c - c_slow),
                                                       train)
                F_.sync(0, 1)
            stage_recs.append(recs_sp)
            xs_fulls.append(xs_full)
            xfs.append(xf)
        # ---- head: each pathway pools its own output on its own lane (disjoint column ranges of `feat`), then the trunk
        #      joins the fast pathway for the Linear
        xs_out, xf_out = xs_full, xf
        F = self.fc_in
        feat = self._buf("feat", n * F, torch.float32)
        rate = float(spec.dropout) if train else 0.0
        ks = spec.head_pool_kernels[0]
        if NP == 2:
            kf = spec.head_pool_kernels[1]
            F_.cur_lane = 1
            pl.fwd.append(be.head_pool_fwd(xf_out, kf, rate, self.drop_seed, feat, F, xs_out.c))
        F_.cur_lane = 0
        pl.fwd.append(be.head_pool_fwd(xs_out, ks, rate, self.drop_seed, feat, F, 0))
        if NP == 2:
            F_.sync(0, 1)
        assert xs_out.c + (xf_out.c if NP == 2 else 0) == F
        K = self.fc_out
        pl.logits = self._buf("logits", n * K, torch.float32)[: n * K].view(n, K)
        fcw, fcb = self._pslice(self.fc_w_off, F * K), self._pslice(self.fc_b_off, K)
        pl.fwd.append(be.fc_fwd(feat, fcw, fcb, pl.logits, n, F, K))
        if not train:
            self._bind_wgrad_scratch()
            return pl

        # ================================================================= backward schedule
        pl.dlogits = self._buf("dlogits", n * K, torch.float32)[: n * K].view(n, K)
        dfeat = self._buf("dfeat", n * F, torch.float32)
        if self._tailz["b"] is not None:
            pl.bwd.append(be.fill_zero(self._tailz["b"]))
        # the Linear's data gradient starts the backward chain; its filter / bias gradient feeds nothing downstream and goes
        # to the filter-gradient lane like every other one (16 us off the turn-around between forward and backward)
        if self.wgrad_lanes:
            pl.bwd.append(be.fc_bwd(pl.dlogits, feat, fcw, dfeat, None, None, n, F, K))
            pl.bwd.sync(2, 0)
            pl.bwd.cur_lane = 2
            pl.bwd.append(be.fc_bwd(pl.dlogits, feat, fcw, None, self._gslice(self.fc_w_off, F * K),
                                    self._gslice(self.fc_b_off, K), n, F, K))
            pl.bwd.cur_lane = 0
        else:
            pl.bwd.append(be.fc_bwd(pl.dlogits, feat, fcw, dfeat, self._gslice(self.fc_w_off, F * K),
                                    self._gslice(self.fc_b_off, K), n, F, K))
        pl.grad_marks.append((len(pl.bwd), (self.fc_w_off, self.layers[0].g_off - self.fc_w_off)))
        d_xs = self._fmap("d.cat.4", n, xs_out.t, xs_out.h, xs_out.w, xs_out.c)
        d_xf = None
        if NP == 2:                                     # fork: the fast pathway un-pools its own gradient on its own lane
            d_xf = self._fmap("d.xf.4", n, xf_out.t, xf_out.h, xf_out.w, xf_out.c)
            B_.sync(1, 0)
            B_.cur_lane = 1
            pl.bwd.append(be.head_pool_bwd(dfeat, F, xs_out.c, kf, rate, self.drop_seed, d_xf))
            B_.cur_lane = 0
        pl.bwd.append(be.head_pool_bwd(dfeat, F, 0, ks, rate, self.drop_seed, d_xs))
        for si in range(3, -1, -1):
            # slow pathway of this stage: d_xs is the gradient of its last block's output
            B_.cur_lane = 0
            d = self._stage_bwd(pl, stage_recs[si][0], d_xs)
            d_cat = d                                   # gradient of the (concatenated) slow input of this stage
            if NP == 2:
                B_.cur_lane = 1
                d_xf = self._stage_bwd(pl, stage_recs[si][1], d_xf)
            c_prev = xs_fulls[si].c - (W.fusions[si].geom.cout if fuse else 0)
            if fuse:
                B_.sync(1, 0)                           # the fusion's gradient comes out of the slow pathway's d_cat
                self._fusion_bwd(pl, si, fusion_recs[si], d_cat.channels(c_prev, d_cat.c - c_prev), d_xf)
            d_xs = d_cat.channels(0, c_prev)
        B_.cur_lane = 0
        self._stem_bwd(pl, 0, stem_recs[0], d_xs)
        if NP == 2:
            B_.cur_lane = 1
            self._stem_bwd(pl, 1, stem_recs[1], d_xf)
            B_.sync(0, 1)                               # join before the optimiser
        if self.wgrad_lanes:
            B_.sync(0, 2)
            if NP == 2:
                B_.sync(0, 3)
        B_.cur_lane = 0
        self._bind_wgrad_scratch()
        return pl

    def _bind_wgrad_scratch(self):
        """bind the filter-gradient ops to their lane's scratch now that its size is known"""
        for oplist, slot, wp, lane in self._wg_pending:
            wp.workspace = self._buf(f"wgws.{lane}", (self._wg_ws_need[lane] + 3) // 4, torch.float32)
            oplist[slot] = self.be.conv_wgrad(wp)
        self._wg_pending = []

    # ------------------------------------------------------------------ execution
    def _plan_for(self, x_slow, x_fast, slow_t_index, train: bool) -> Plan:
        """Plans are keyed on geometry (shapes, strides, dtypes); only the two im2col ops hold the input
        addresses, so a new batch tensor re-binds those two ops instead of rebuilding ~1000 descriptors."""
        def geo(t):
            return None if t is None else (tuple(t.shape), tuple(t.stride()), t.dtype)

        def ptrs():
            return tuple(None if t is None else t.data_ptr() for t in (x_slow, x_fast, slow_t_index))
        key = (geo(x_slow), geo(x_fast), geo(slow_t_index), train)
        pl = self._plans.get(key)
        if pl is None:
            if len(self._plans) >= 6:
                self._plans.clear()
            pl = self._build_plan(x_slow, x_fast, slow_t_index, train)
            assert not self._pending_fin, "a deferred BatchNorm finalize was never consumed by an _apply"
            pl.key = key
            Engine._plan_serial += 1
            pl.serial = Engine._plan_serial          # never reused (id() of a dropped plan can be)
            pl.bound = ptrs()
            pl.inputs = (x_slow, x_fast, slow_t_index)   # keep the bound tensors alive
            pl.graph_epoch = 0
            self._plans[key] = pl
        elif pl.bound != ptrs():
            self._stem_ops(pl, 0, x_slow, slow_t_index)
            if x_fast is not None:
                self._stem_ops(pl, 1, x_fast, None)
            pl.bound = ptrs()
            pl.inputs = (x_slow, x_fast, slow_t_index)
            pl.graph_epoch += 1                          # a captured hipGraph of this plan is stale now
        return pl

    def _stream(self) -> int:
        if self.device.type == "cuda":
            return torch.cuda.current_stream(self.device).cuda_stream
        return 0

    @staticmethod
    def _run(ops: List[Run], stream: int):
        for op in ops:
            op(stream)

    _plan_serial = 0

    def lane_streams(self):
        """torch streams the schedule's lanes run on (one entry when the schedule is single-stream / on the CPU)"""
        if self.device.type != "cuda":
            return []
        main = torch.cuda.current_stream(self.device)
        if not self.two_streams:
            return [main]
        if self._side is None:
            cus = [int(v) for v in self.options.lane_cus.split(",")] if self.options.lane_cus else []
            self._side = [_masked_stream(self.device, cus[i]) if i < len(cus) and cus[i] > 0 else torch.cuda.Stream(self.device)
                          for i in range(self.NLANES - 1)]
        return [main] + self._side

    def _run_lanes(self, ops: "OpList", begin: int = 0, end: Optional[int] = None):
        """Run ops[begin:end] of a schedule on its lanes: lane 0 on the current stream, the others on side streams,
        ordered by the Wait markers (events).  Capturable: the side streams fork from / join into the capturing
        stream."""
        end = len(ops) if end is None else end
        if self.device.type != "cuda" or not self.two_streams:
            return self._run(ops[begin:end], self._stream())
        streams = self.lane_streams()
        handles = [s.cuda_stream for s in streams]
        skip = self._ablate_kinds
        for i in range(begin, end):
            op, lane = ops[i], ops.lane[i]
            if skip and not isinstance(op, Wait) and ((ops.meta[i] is not None and ops.meta[i].get("kind") in skip)
                                                      or f"lane{lane}" in skip):
                continue                 # TIMING ABLATION ONLY (SFK_ABLATE): the step's results are wrong
            if isinstance(op, Wait):
                ev = torch.cuda.Event()
                ev.record(streams[op.on])
                streams[op.lane].wait_event(ev)
            else:
                op(handles[lane])

    def forward(self, x_slow: torch.Tensor, x_fast: torch.Tensor, train: bool, slow_t_index=None) -> Plan:
        """x_*: (N, C, T, H, W) views with ANY strides (the dataset's N,T,C,H,W memory is read in place).
        If slow_t_index is given, the slow pathway reads frames x_slow[:, :, slow_t_index] (PackPathway)."""
        assert x_slow.dim() == 5 and x_slow.shape[1] == self.spec.input_channels[0]
        if self.spec.pathways == 2:
            assert x_fast.dim() == 5 and x_slow.shape[0] == x_fast.shape[0]
            assert x_fast.shape[1] == self.spec.input_channels[1]
        else:
            assert x_fast is None, "single-pathway network: forward(x, None, ...)"
        pl = self._plan_for(x_slow, x_fast, slow_t_index, train)
        if train:
            self.drop_seed.add_(1)
        self._run_lanes(pl.fwd)
        return pl

    def backward(self, pl: Plan, dlogits: Optional[torch.Tensor] = None, zero_grad: bool = True):
        """Fills the gradient arena self.G from pl.dlogits (or the given dlogits)."""
        if dlogits is not None:
            pl.dlogits.copy_(dlogits)
        if zero_grad:
            self.G.zero_()
        self._run_lanes(pl.bwd)

    # ---- fused loss + optimiser (train.py:228-231 without the per-step .item() sync of :236)
    def loss_ops(self, pl: Plan, labels: torch.Tensor, loss_out, loss_sum, correct, gscale: float = 1.0) -> Run:
        n, k = pl.logits.shape
        return self.be.softmax_ce(pl.logits, labels, n, k, gscale, pl.dlogits, loss_out, loss_sum, correct)

    def _adam_state(self):
        if self.adam_m is None:
            self.adam_m = self._new(self.arena_numel)
            self.adam_v = self._new(self.arena_numel)
            self.adam_step = self._new(1, dtype=torch.int64)
            # scratch counter of the split update's second launch (adam_split_ops): ONE tensor per engine, so that every cached
            # TrainStep entry bakes in the same pointer and the value TrainStep writes before the launch is the one it reads
            self.adam_step_tail = self._new(1, dtype=torch.int64)

    def adam_ops(self, lr: float, betas=(0.9, 0.999), eps: float = 1e-8, grad_scale: float = 1.0) -> Run:
        self._adam_state()
        return self.be.adam(self.P.data, self.G, self.adam_m, self.adam_v, self.arena_numel, lr, betas[0], betas[1],
                            eps, grad_scale, self.adam_step, None)

    def adam_split_ops(self, cut: int, lr: float, betas=(0.9, 0.999), eps: float = 1e-8, grad_scale: float = 1.0):
        """(main, tail): the same update as adam_ops in two launches -- arena [cut:] (increments the step counter) and [:cut]
        (the stems' filters; sfk_adam increments the counter it is given, so the tail gets a scratch counter that
        TrainStep sets to step - 1 first).  Element-wise identical to the single launch."""
        self._adam_state()
        n = self.arena_numel
        main = self.be.adam(self.P.data[cut:], self.G[cut:], self.adam_m[cut:], self.adam_v[cut:], n - cut, lr, betas[0],
                            betas[1], eps, grad_scale, self.adam_step, None)
        tail = self.be.adam(self.P.data[:cut], self.G[:cut], self.adam_m[:cut], self.adam_v[:cut], cut, lr, betas[0],
                            betas[1], eps, grad_scale, self.adam_step_tail, None)
        return main, tail

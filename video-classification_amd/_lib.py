"""ctypes binding of include/sfk.h and the HIP backend the engine drives.

There is NO fallback: if libsfk.so is missing or fails to load, ``load()`` raises.  The engine talks to a
*backend* object whose methods mirror the C entry points and return zero-allocation closures ``run(stream)``
with every descriptor pre-built, so a training step is a flat list of C calls on one hipStream (and can be
captured into a hipGraph).
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SFK_LIB") or os.path.join(HERE, "libsfk.so")   # SFK_LIB: experiment builds (tools/gpu_ab_lib.sh)
SFK_F32, SFK_BF16 = 0, 1
SFK_MAX_TAPS = 16
BN_SYNC_INTS = 2144       # include/sfk.h SFK_BN_SYNC_INTS
ABI_VERSION = 20       # include/sfk.h SFK_ABI_VERSION
BN_FOLD_ROWS = 64      # include/sfk.h SFK_BN_FOLD_ROWS
_DT = {torch.float32: SFK_F32, torch.bfloat16: SFK_BF16}


class SfkError(RuntimeError):
    pass


# ----------------------------------------------------------------------------- python-side descriptors
@dataclass
class FMap:
    """Channels-last feature map view: element (n,t,h,w,c) at buf[(((n*T+t)*H+h)*W+w)*ld + c_off + c]."""
    buf: torch.Tensor
    n: int
    t: int
    h: int
    w: int
    c: int
    ld: int = 0
    c_off: int = 0

    def __post_init__(self):
        if self.ld == 0:
            self.ld = self.c
        assert self.ld >= self.c_off + self.c
        assert self.buf.numel() >= self.pixels * self.ld, (self.buf.numel(), self.pixels, self.ld)

    @property
    def pixels(self) -> int:
        return self.n * self.t * self.h * self.w

    @property
    def dtype(self) -> torch.dtype:
        return self.buf.dtype

    def channels(self, c0: int, c: int) -> "FMap":
        assert 0 <= c0 and c0 + c <= self.c
        return FMap(self.buf, self.n, self.t, self.h, self.w, c, self.ld, self.c_off + c0)

    def view5(self) -> torch.Tensor:
        """(N,T,H,W,C) strided torch view of this map (for tests and host-side glue)."""
        return self.buf[: self.pixels * self.ld].view(self.n, self.t, self.h, self.w, self.ld)[
            ..., self.c_off:self.c_off + self.c]

    def like(self, buf: torch.Tensor, c: Optional[int] = None) -> "FMap":
        c = self.c if c is None else c
        return FMap(buf, self.n, self.t, self.h, self.w, c)


Tap = Tuple[int, int, int, int]  # (dt, dh, dw, widx)


@dataclass
class BnBwdFuse:
    """sfk_bn_bwd_fuse: the BatchNorm-backward reduce folded into the data-gradient pass that produces dA."""
    y_bn: Optional[FMap]             # conv output the BatchNorm normalised; None (+ relu_out_bits): bitmap mask, sum dz only
    mask_src: Optional[FMap]         # activation whose sign is the ReLU mask, or None
    mean: Optional[torch.Tensor]
    invstd: Optional[torch.Tensor]
    scale: Optional[torch.Tensor]
    shift: Optional[torch.Tensor]
    relu: bool
    partials: torch.Tensor           # fp32 [mtiles][cout][2]


@dataclass
class ConvEpilogue:
    """sfk_conv_epilogue: v = acc*scale + shift (+ old y) (+ res*res_scale + res_shift); ReLU (+ bitmap); y = v."""
    scale: Optional[torch.Tensor] = None
    shift: Optional[torch.Tensor] = None
    res: Optional[FMap] = None
    res_scale: Optional[torch.Tensor] = None
    res_shift: Optional[torch.Tensor] = None
    relu: bool = False
    relu_bits: Optional[torch.Tensor] = None


@dataclass
class ConvPass:
    x: FMap
    y: FMap
    rows: Tuple[int, int, int]
    gs: Tuple[int, int, int]
    os: Tuple[int, int, int]
    oo: Tuple[int, int, int]
    taps: List[Tap]
    w: torch.Tensor          # [cout][wtaps][cin] flat, dtype of x
    wtaps: int
    cin: int
    cout: int
    accumulate: bool = False
    stats: Optional[torch.Tensor] = None  # fp32 [mtiles][cout][2]
    bnb: Optional[BnBwdFuse] = None
    relu_out_bits: Optional[torch.Tensor] = None   # uint8 bitmap of bn_apply: the pass stores result * mask
    ep: Optional[ConvEpilogue] = None


@dataclass
class WgradPass:
    x: FMap
    dy: FMap
    gs: Tuple[int, int, int]
    taps: List[Tap]
    dw: torch.Tensor         # fp32 [cout][wtaps][cin] flat view into the gradient arena
    wtaps: int
    cin: int
    cout: int
    workspace: Optional[torch.Tensor] = None   # fp32 scratch for the partial-tile path (sfk_conv_wgrad_workspace_bytes)
    dg_w: Optional[torch.Tensor] = None        # fused data gradient of the same dY: [cin][cout] matrix (include/sfk.h)
    dg_y: Optional[FMap] = None                # ... its output map (cin channels)


@dataclass
class StemSrc:
    src: torch.Tensor        # any strided view indexed (n, c, t, h, w), f32 or bf16, read in place
    t_index: Optional[torch.Tensor]   # int32 frame indices (PackPathway) or None
    kt: int

    @property
    def t_len(self) -> int:
        return int(self.t_index.numel()) if self.t_index is not None else int(self.src.shape[2])


def stem_kp(cin: int, kt: int) -> int:
    """row length of the stem filter layout [co][((f*cin+ci)*7+kh)*8+kw] (mirrors sfk_stem_kp)"""
    return (kt * cin * 7 + 3) // 4 * 4 * 8


# ----------------------------------------------------------------------------- ctypes mirror of sfk.h
class _FMap(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("dtype", C.c_int32), ("n", C.c_int32), ("t", C.c_int32), ("h", C.c_int32),
                ("w", C.c_int32), ("c", C.c_int32), ("ld", C.c_int32), ("c_off", C.c_int32)]


class _Tap(C.Structure):
    _fields_ = [("dt", C.c_int8), ("dh", C.c_int8), ("dw", C.c_int8), ("widx", C.c_uint8)]


class _BnBwdFuse(C.Structure):
    _fields_ = [("y_bn", _FMap), ("mask_src", _FMap), ("mean", C.c_void_p), ("invstd", C.c_void_p),
                ("scale", C.c_void_p), ("shift", C.c_void_p), ("relu", C.c_int32), ("partials", C.c_void_p)]


class _ConvEpilogue(C.Structure):
    _fields_ = [("scale", C.c_void_p), ("shift", C.c_void_p), ("res", _FMap), ("res_scale", C.c_void_p),
                ("res_shift", C.c_void_p), ("relu", C.c_int32), ("reserved", C.c_int32), ("relu_bits", C.c_void_p)]


class _ConvDesc(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("reserved0", C.c_uint32), ("x", _FMap), ("y", _FMap), ("rt", C.c_int32), ("rh", C.c_int32), ("rw", C.c_int32),
                ("gs", C.c_int32 * 3), ("os", C.c_int32 * 3), ("oo", C.c_int32 * 3), ("ntaps", C.c_int32),
                ("taps", _Tap * SFK_MAX_TAPS), ("w", C.c_void_p), ("wtaps", C.c_int32), ("cin", C.c_int32),
                ("cout", C.c_int32), ("accumulate", C.c_int32), ("stats", C.c_void_p), ("bnb", _BnBwdFuse),
                ("out_relu_bits", C.c_void_p), ("ep", _ConvEpilogue)]


class _WgradDesc(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("reserved0", C.c_uint32), ("x", _FMap), ("dy", _FMap), ("gs", C.c_int32 * 3), ("ntaps", C.c_int32),
                ("taps", _Tap * SFK_MAX_TAPS), ("dw", C.c_void_p), ("wtaps", C.c_int32), ("cin", C.c_int32),
                ("cout", C.c_int32), ("workspace", C.c_void_p), ("workspace_bytes", C.c_int64),
                ("dg_w", C.c_void_p), ("dg_y", _FMap)]


class _StemSrc(C.Structure):
    _fields_ = [("src", C.c_void_p), ("src_dtype", C.c_int32), ("sn", C.c_int64), ("sc", C.c_int64),
                ("st", C.c_int64), ("sh", C.c_int64), ("sw", C.c_int64), ("cin", C.c_int32), ("t_in", C.c_int32),
                ("h_in", C.c_int32), ("w_in", C.c_int32), ("t_index", C.c_void_p), ("t_len", C.c_int32),
                ("kt", C.c_int32)]


class _Tuning(C.Structure):
    """sfk_tuning: the write-once kernel-selection table of sfk_init (defaults = the measured best)."""
    _fields_ = [("struct_size", C.c_uint32), ("igemm_short_k", C.c_int32), ("igemm_small_k", C.c_int32), ("igemm_wide_store", C.c_int32),
                ("wgrad_target_8w", C.c_int32), ("wgrad_target_4w", C.c_int32), ("wgrad_use_workspace", C.c_int32),
                ("wgrad_wide_co", C.c_int32), ("bn_parts", C.c_int32), ("nt_apply_mb", C.c_int32),
                ("nt_reduce_mb", C.c_int32), ("nt_bwd_apply_mb", C.c_int32), ("igemm_pw_stream", C.c_int32),
                ("pool_blocks", C.c_int64), ("igemm_tile256", C.c_int32), ("wgrad_target_gen", C.c_int32),
                ("wgrad_target_256", C.c_int32), ("wgrad_min_stages_256", C.c_int32), ("igemm_p8", C.c_int32),
                ("wgrad_p8", C.c_int32), ("igemm_halo", C.c_int32), ("wgrad_band", C.c_int32), ("stem_v3", C.c_int32)]


# experiment knobs (tools/gpu_ab_env.sh): read HERE, once, on the host side of the boundary -- the library itself never
# reads the environment (include/sfk.h); name -> sfk_tuning field
TUNING_ENV = {"SFK_KSHORT": "igemm_short_k", "SFK_SMALLK": "igemm_small_k", "SFK_WIDE": "igemm_wide_store",
              "SFK_WGT8": "wgrad_target_8w", "SFK_WGT4": "wgrad_target_4w", "SFK_WGWS_LIB": "wgrad_use_workspace",
              "SFK_WG_WIDECO": "wgrad_wide_co", "SFK_BN_PARTS": "bn_parts", "SFK_NT_APPLY_MB": "nt_apply_mb",
              "SFK_NT_RED_MB": "nt_reduce_mb", "SFK_NT_BAPP_MB": "nt_bwd_apply_mb", "SFK_POOL_BLOCKS": "pool_blocks",
              "SFK_PW_STREAM": "igemm_pw_stream", "SFK_TILE256": "igemm_tile256", "SFK_WGTG": "wgrad_target_gen",
              "SFK_P8": "igemm_p8",
              "SFK_WGP8": "wgrad_p8", "SFK_HALO": "igemm_halo", "SFK_WGBAND": "wgrad_band", "SFK_STEM3": "stem_v3"}

_PF, _PV, _I32, _I64, _F = C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_float
_P_FMAP = C.POINTER(_FMap)

# name -> argument types (return type is int unless listed in _RESTYPE)
SIGNATURES = {
    "sfk_conv_igemm": [C.POINTER(_ConvDesc), _PV],
    "sfk_conv_igemm_mtiles": [C.POINTER(_ConvDesc)],
    "sfk_conv_bnb_supported": [C.POINTER(_ConvDesc)],
    "sfk_conv_relu_out_supported": [C.POINTER(_ConvDesc)],
    "sfk_conv_epilogue_supported": [C.POINTER(_ConvDesc)],
    "sfk_conv_igemm_family": [C.POINTER(_ConvDesc)],
    "sfk_bn_tail_fwd": [_PF, _PF, _I32, _I64, _PF, _I32, _PV, _I32, _I32, _PF, _PF, _F, _F, _PF, _PF, _PV, _PF, _PF, _PF, _PF, _PF, _PV, _PV],
    "sfk_bn_tail_bwd": [_PF, _PF, _I32, _PF, _I64, _PF, _I32, _PV, _I32, _I32, _PF, _PF, _PF, _PF, _PF, _PF, _PV, _PF, _PF, _PV],
    "sfk_conv_pw_dual_supported": [_P_FMAP, _P_FMAP, _P_FMAP],
    "sfk_conv_pw_dual": [_P_FMAP, _PV, _P_FMAP, _PV, _PF, _P_FMAP, _PV],
    "sfk_conv_wgrad": [C.POINTER(_WgradDesc), _PV],
    "sfk_conv_wgrad_workspace_bytes": [C.POINTER(_WgradDesc)],
    "sfk_conv_wgrad_dg_supported": [C.POINTER(_WgradDesc)],
    "sfk_conv_wgrad_wants_workspace": [C.POINTER(_WgradDesc)],
    "sfk_stem_kp": [_I32, _I32],
    "sfk_stem_conv_tiles": [C.POINTER(_StemSrc), _P_FMAP],
    "sfk_stem_conv_fwd": [C.POINTER(_StemSrc), _PV, _P_FMAP, _PF, _PV],
    "sfk_stem_conv_wgrad": [C.POINTER(_StemSrc), _P_FMAP, _PF, _PV],
    "sfk_bn_finalize": [_PF, _I32, _I32, _I64, _PF, _PF, _F, _F, _PF, _PF, _PV, _PF, _PF, _PF, _PF, _PF, _PV],
    "sfk_bn_eval_coeffs": [_PF, _PF, _PF, _PF, _F, _I32, _PF, _PF, _PV],
    "sfk_bn_stats": [_P_FMAP, _PF, _I32, C.POINTER(C.c_int32), _PV],
    "sfk_bn_apply": [_P_FMAP, _PF, _PF, _P_FMAP, _PF, _PF, _I32, _P_FMAP, _PV, _PF, _I32, C.POINTER(C.c_int32), _PV],
    "sfk_bn_finalize_apply": [_PF, _I32, _I64, _PF, _PF, _F, _F, _PF, _PF, _PV, _PF, _PF, _PF, _PV, _P_FMAP, _PF, _PF, _P_FMAP, _PF, _PF,
                              _I32, _P_FMAP, _PV, _PV],
    "sfk_bn_bwd_finalize_apply": [_PF, _I32, _I64, _PF, _PF, _PF, _PF, _PF, _PV, _P_FMAP, _P_FMAP, _P_FMAP, _PF, _PF, _PF, _PF, _I32,
                                  _P_FMAP, _PV],
    "sfk_bn_bwd_reduce": [_P_FMAP, _P_FMAP, _P_FMAP, _PF, _PF, _PF, _PF, _I32, _P_FMAP, _PF, _I32,
                          C.POINTER(C.c_int32), _PV, _PV],
    "sfk_bn_bwd_finalize": [_PF, _I32, _I32, _I64, _PF, _PF, _PF, _PF, _PF, _PF, _PV],
    "sfk_bn_bwd_apply": [_P_FMAP, _P_FMAP, _P_FMAP, _PF, _PF, _PF, _PF, _I32, _PF, _P_FMAP, _PV],
    "sfk_maxpool_fwd": [_P_FMAP, _P_FMAP, _PV, _I32, _I32, _I32, _PV],
    "sfk_maxpool_bwd": [_P_FMAP, _PV, _P_FMAP, _I32, _I32, _I32, _PV],
    "sfk_bn_maxpool_fwd": [_P_FMAP, _PF, _PF, _P_FMAP, _PV, _I32, _I32, _I32, _PV],
    "sfk_bn_maxpool_bwd_reduce": [_P_FMAP, _PV, _P_FMAP, _PF, _PF, _PF, _PF, _PF, _I32, C.POINTER(C.c_int32), _PV],
    "sfk_bn_maxpool_bwd_apply": [_P_FMAP, _PV, _P_FMAP, _PF, _PF, _PF, _PF, _PF, _P_FMAP, _PV],
    "sfk_head_pool_fwd": [_P_FMAP, _I32, _I32, _I32, _F, _PV, _PF, _I32, _I32, _PV],
    "sfk_head_pool_bwd": [_PF, _I32, _I32, _I32, _I32, _I32, _F, _PV, _P_FMAP, _PV],
    "sfk_head_dropout_mask": [_I32, _I32, _I32, _I32, _F, _PV, _PV, _PV],
    "sfk_fc_fwd": [_PF, _PF, _PF, _PF, _I32, _I32, _I32, _PV],
    "sfk_fc_bwd": [_PF, _PF, _PF, _PF, _PF, _PF, _I32, _I32, _I32, _PV],
    "sfk_softmax_ce": [_PF, _PV, _I32, _I32, _F, _PF, _PF, _PF, _PV, _PV],
    "sfk_adam": [_PF, _PF, _PF, _PF, _I64, _F, _F, _F, _F, _F, _PV, _PV, _I32, _PV],
    "sfk_filter_transpose": [_PV, _I32, _PV, _I32, _I32, _I32, _I32, _PV],
    "sfk_cast": [_PV, _I32, _PV, _I32, _I64, _PV],
    "sfk_fill_zero": [_PV, C.c_size_t, _PV],
    "sfk_u8_normalize_crop": [_PV, _PF, _PV, _I32, _PV, _I32, _I32, _I32, _I32, _I32, _I32, _PV],
    "sfk_eval_aggregate": [_PF, _PV, _PV, _I32, _I32, _I32, _PF, _PV, _PV, _PV],
    "sfk_sparse_fusion_fwd": [_PF, _PF, _PF, _PF, _I32, _I32, _I32, _PV],
    "sfk_sparse_fusion_bwd": [_PF, _PF, _PF, _PF, _I32, _I32, _I32, _PV],
    "sfk_filter_refresh": [_PF, _PV, _PV, _I32, _PV, _I32, _I32, _PV],
    "sfk_default_tuning": [C.POINTER(_Tuning)],
    "sfk_get_tuning": [C.POINTER(_Tuning)],
    "sfk_init": [C.POINTER(_Tuning)],
    "sfk_abi_version": [],
    "sfk_status_string": [C.c_int],
}
_RESTYPE = {"sfk_status_string": C.c_char_p, "sfk_conv_wgrad_workspace_bytes": C.c_int64}


def new_conv_desc() -> "_ConvDesc":
    """a zeroed sfk_conv_desc carrying the ABI handshake (struct_size = the layout THIS binding was written for)"""
    d = _ConvDesc()
    d.struct_size = C.sizeof(_ConvDesc)
    return d


def new_wgrad_desc() -> "_WgradDesc":
    d = _WgradDesc()
    d.struct_size = C.sizeof(_WgradDesc)
    return d


def new_tuning() -> "_Tuning":
    t = _Tuning()
    t.struct_size = C.sizeof(_Tuning)
    return t

_lib = None


def load(path: str = LIB_PATH) -> C.CDLL:
    """dlopen libsfk.so and type every entry point of include/sfk.h.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise SfkError(f"{path} not found: build it with `python video-classification_amd/build.py` "
                       "(there is no CPU or PyTorch fallback for the SlowFast path)")
    lib = C.CDLL(path)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = _RESTYPE[name] if name in _RESTYPE else C.c_int
    if lib.sfk_abi_version() != ABI_VERSION:
        raise SfkError(f"libsfk ABI version mismatch: library {lib.sfk_abi_version()}, binding {ABI_VERSION}")
    t = new_tuning()
    if lib.sfk_default_tuning(C.byref(t)) != 0:
        raise SfkError("sfk_default_tuning refused this binding's sfk_tuning layout")
    for env, fld in TUNING_ENV.items():
        if os.environ.get(env) is not None:
            setattr(t, fld, int(os.environ[env]))
    st = lib.sfk_init(C.byref(t))                # once per process, before the first launch
    if st != 0:
        raise SfkError(f"sfk_init: {lib.sfk_status_string(st).decode()}")
    _lib = lib
    return lib


def tuning() -> _Tuning:
    """the table the loaded library runs with (sfk_get_tuning)"""
    t = new_tuning()
    _check(load().sfk_get_tuning(C.byref(t)), "sfk_get_tuning")
    return t


def _check(st: int, what: str):
    if st != 0:
        msg = load().sfk_status_string(st).decode()
        raise SfkError(f"{what}: {msg} ({st})")


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _c_fmap(f: FMap) -> _FMap:
    return _FMap(f.buf.data_ptr(), _DT[f.buf.dtype], f.n, f.t, f.h, f.w, f.c, f.ld, f.c_off)


def _c_taps(taps: Sequence[Tap]):
    arr = (_Tap * SFK_MAX_TAPS)()
    assert 0 < len(taps) <= SFK_MAX_TAPS
    for i, (dt, dh, dw, wi) in enumerate(taps):
        arr[i] = _Tap(dt, dh, dw, wi)
    return arr


def _c_conv(p: ConvPass) -> _ConvDesc:
    d = new_conv_desc()
    d.x, d.y = _c_fmap(p.x), _c_fmap(p.y)
    d.rt, d.rh, d.rw = p.rows
    d.gs = (C.c_int32 * 3)(*p.gs)
    d.os = (C.c_int32 * 3)(*p.os)
    d.oo = (C.c_int32 * 3)(*p.oo)
    d.ntaps = len(p.taps)
    d.taps = _c_taps(p.taps)
    d.w = p.w.data_ptr()
    d.wtaps, d.cin, d.cout = p.wtaps, p.cin, p.cout
    d.accumulate = 1 if p.accumulate else 0
    d.stats = _ptr(p.stats)
    if p.bnb is not None:
        b = p.bnb
        if b.y_bn is not None:
            d.bnb.y_bn = _c_fmap(b.y_bn)
        if b.mask_src is not None:
            d.bnb.mask_src = _c_fmap(b.mask_src)
        d.bnb.mean, d.bnb.invstd = _ptr(b.mean), _ptr(b.invstd)
        d.bnb.scale, d.bnb.shift = _ptr(b.scale), _ptr(b.shift)
        d.bnb.relu = 1 if b.relu else 0
        d.bnb.partials = _ptr(b.partials)
    if p.relu_out_bits is not None:
        vec = 8 if p.y.dtype == torch.bfloat16 else 4
        assert p.relu_out_bits.dtype == torch.uint8 and p.relu_out_bits.numel() >= p.y.pixels * (p.cout // vec)
        d.out_relu_bits = p.relu_out_bits.data_ptr()
    if p.ep is not None:
        e = p.ep
        assert e.scale is not None or e.shift is not None
        d.ep.scale, d.ep.shift = _ptr(e.scale), _ptr(e.shift)
        if e.res is not None:
            d.ep.res = _c_fmap(e.res)
        d.ep.res_scale, d.ep.res_shift = _ptr(e.res_scale), _ptr(e.res_shift)
        d.ep.relu = 1 if e.relu else 0
        if e.relu_bits is not None:
            vec = 8 if p.y.dtype == torch.bfloat16 else 4
            assert e.relu_bits.dtype == torch.uint8 and e.relu_bits.numel() >= p.y.pixels * (p.cout // vec)
            d.ep.relu_bits = e.relu_bits.data_ptr()
    return d


class HipBackend:
    """Every method returns ``run(stream)``; descriptors are built once, here.  Tensors referenced by a closure
    are kept alive by it."""

    name = "hip"

    def __init__(self):
        self.lib = load()

    # -- convolution
    def conv_igemm_mtiles(self, p: ConvPass) -> int:
        d = _c_conv(p)
        r = self.lib.sfk_conv_igemm_mtiles(C.byref(d))
        if r < 0:
            _check(r, "sfk_conv_igemm_mtiles")
        return r

    def conv_bnb_supported(self, p: ConvPass) -> bool:
        """can this pass take a BnBwdFuse (sfk_conv_bnb_supported)?"""
        return bool(self.lib.sfk_conv_bnb_supported(C.byref(_c_conv(p))))

    def conv_relu_out_supported(self, p: ConvPass) -> bool:
        """can this pass apply a ReLU bitmap to what it stores (sfk_conv_relu_out_supported)?"""
        q = ConvPass(**{**p.__dict__, "relu_out_bits": None})
        return bool(self.lib.sfk_conv_relu_out_supported(C.byref(_c_conv(q))))

    def conv_epilogue_supported(self, p: ConvPass) -> bool:
        """can this pass run with its fused output transform p.ep (sfk_conv_epilogue_supported)?"""
        return bool(self.lib.sfk_conv_epilogue_supported(C.byref(_c_conv(p))))

    def conv_family(self, p: ConvPass) -> int:
        """0 register-staged igemm, 1 LDS-DMA igemm, 3 streaming pointwise kernel with the fused epilogue, 4 the deep-pipelined
        256 x 256 tile (conv_igemm_p8.hip), 5 the LDS-band 3 x 3 kernel (conv_halo.hip)"""
        return int(self.lib.sfk_conv_igemm_family(C.byref(_c_conv(p))))

    def conv_igemm(self, p: ConvPass):
        d, fn, keep = _c_conv(p), self.lib.sfk_conv_igemm, p

        def run(stream, _d=C.byref(d), _keep=(d, keep)):
            st = fn(_d, stream)
            if st:
                _check(st, "sfk_conv_igemm")
        return run

    def _wgrad_desc(self, p: WgradPass):
        d = new_wgrad_desc()
        d.x, d.dy = _c_fmap(p.x), _c_fmap(p.dy)
        d.gs = (C.c_int32 * 3)(*p.gs)
        d.ntaps, d.taps = len(p.taps), _c_taps(p.taps)
        d.dw, d.wtaps, d.cin, d.cout = p.dw.data_ptr(), p.wtaps, p.cin, p.cout
        if p.dg_w is not None:
            d.dg_w, d.dg_y = p.dg_w.data_ptr(), _c_fmap(p.dg_y)
        return d

    def conv_wgrad_dg_supported(self, p: WgradPass) -> bool:
        """the fused data gradient (p.dg_w, p.dg_y) can run with this filter-gradient pass"""
        return p.dg_w is not None and bool(self.lib.sfk_conv_wgrad_dg_supported(C.byref(self._wgrad_desc(p))))

    def conv_wgrad_wants_workspace(self, p: WgradPass) -> bool:
        """this pass would run the 256-column tile, which sums its pixel splits only through a workspace"""
        return bool(self.lib.sfk_conv_wgrad_wants_workspace(C.byref(self._wgrad_desc(p))))

    def conv_wgrad_workspace_bytes(self, p: WgradPass) -> int:
        """bytes of scratch with which sfk_conv_wgrad sums its pixel splits without atomics (deterministically)"""
        r = self.lib.sfk_conv_wgrad_workspace_bytes(C.byref(self._wgrad_desc(p)))
        if r < 0:
            _check(int(r), "sfk_conv_wgrad_workspace_bytes")
        return int(r)

    def conv_wgrad(self, p: WgradPass):
        d = self._wgrad_desc(p)
        if p.workspace is not None:
            d.workspace, d.workspace_bytes = p.workspace.data_ptr(), p.workspace.numel() * p.workspace.element_size()
        fn = self.lib.sfk_conv_wgrad

        def run(stream, _d=C.byref(d), _keep=(d, p)):
            st = fn(_d, stream)
            if st:
                _check(st, "sfk_conv_wgrad")
        return run

    @staticmethod
    def _c_stem(p: StemSrc) -> _StemSrc:
        s = p.src
        assert s.dim() == 5
        d = _StemSrc()
        d.src, d.src_dtype = s.data_ptr(), _DT[s.dtype]
        d.sn, d.sc, d.st, d.sh, d.sw = s.stride()
        d.cin, d.t_in, d.h_in, d.w_in = s.shape[1], s.shape[2], s.shape[3], s.shape[4]
        d.t_index = _ptr(p.t_index)
        d.t_len = p.t_len
        d.kt = p.kt
        return d

    def stem_conv_tiles(self, p: StemSrc, y: FMap) -> int:
        d, fy = self._c_stem(p), _c_fmap(y)
        r = self.lib.sfk_stem_conv_tiles(C.byref(d), C.byref(fy))
        if r < 0:
            _check(r, "sfk_stem_conv_tiles")
        return r

    def stem_conv_fwd(self, p: StemSrc, w: torch.Tensor, y: FMap, stats: Optional[torch.Tensor]):
        d, fy = self._c_stem(p), _c_fmap(y)
        return self._plain("sfk_stem_conv_fwd", C.byref(d), _ptr(w), C.byref(fy), _ptr(stats), keep=(d, fy, p, w, y, stats))

    def stem_conv_wgrad(self, p: StemSrc, dy: FMap, dw: torch.Tensor):
        d, fy = self._c_stem(p), _c_fmap(dy)
        return self._plain("sfk_stem_conv_wgrad", C.byref(d), C.byref(fy), _ptr(dw), keep=(d, fy, p, dy, dw))

    # -- generic plain-argument entry points
    def _plain(self, name, *args, keep=()):
        fn = getattr(self.lib, name)
        cargs = tuple(args)

        def run(stream, _a=cargs, _keep=keep):
            st = fn(*_a, stream)
            if st:
                _check(st, name)
        run.sfk_name = name          # for the profiler's per-op dump
        return run

    def bn_finalize(self, partials, nparts, c, count, gamma, beta, eps, momentum, running_mean, running_var, nbt,
                    mean, invstd, scale, shift, workspace=None):
        """workspace: optional BN_FOLD_ROWS * c * 2 floats (parallel first-level fold of many partial rows)"""
        ts = (partials, gamma, beta, running_mean, running_var, nbt, mean, invstd, scale, shift, workspace)
        return self._plain("sfk_bn_finalize", _ptr(partials), nparts, c, count, _ptr(gamma), _ptr(beta), eps, momentum,
                           _ptr(running_mean), _ptr(running_var), _ptr(nbt), _ptr(mean), _ptr(invstd), _ptr(scale),
                           _ptr(shift), _ptr(workspace), keep=ts)

    def bn_finalize_apply(self, partials, nparts, count, gamma, beta, eps, momentum, running_mean, running_var, nbt, mean, invstd,
                          workspace, sync, y: FMap, scale, shift, res: Optional[FMap], res_scale, res_shift, relu: bool, out: FMap,
                          relu_bits=None):
        """bn_finalize + bn_apply in one launch (sfk_bn_finalize_apply); sync: BN_SYNC_INTS zeroed int32 of this BatchNorm's own"""
        fy, fo = _c_fmap(y), _c_fmap(out)
        fr = _c_fmap(res) if res is not None else None
        assert sync.dtype == torch.int32 and sync.numel() >= BN_SYNC_INTS
        ts = (partials, gamma, beta, running_mean, running_var, nbt, mean, invstd, workspace, sync, fy, fo, fr, y, out, res, scale, shift,
              res_scale, res_shift, relu_bits)
        return self._plain("sfk_bn_finalize_apply", _ptr(partials), nparts, count, _ptr(gamma), _ptr(beta), eps, momentum,
                           _ptr(running_mean), _ptr(running_var), _ptr(nbt), _ptr(mean), _ptr(invstd), _ptr(workspace), _ptr(sync),
                           C.byref(fy), _ptr(scale), _ptr(shift), C.byref(fr) if fr else None, _ptr(res_scale), _ptr(res_shift),
                           1 if relu else 0, C.byref(fo), _ptr(relu_bits), keep=ts)

    def bn_bwd_finalize_apply(self, partials, nparts, count, gamma, dgamma, dbeta, coef, workspace, sync, da: FMap, y: FMap,
                              mask_src: Optional[FMap], mean, invstd, scale, shift, relu: bool, dy: FMap):
        """bn_bwd_finalize + bn_bwd_apply in one launch (sfk_bn_bwd_finalize_apply)"""
        fa, fy, fo = _c_fmap(da), _c_fmap(y), _c_fmap(dy)
        fm = _c_fmap(mask_src) if mask_src is not None else None
        assert sync.dtype == torch.int32 and sync.numel() >= BN_SYNC_INTS
        ts = (partials, gamma, dgamma, dbeta, coef, workspace, sync, fa, fy, fo, fm, da, y, dy, mask_src, mean, invstd, scale, shift)
        return self._plain("sfk_bn_bwd_finalize_apply", _ptr(partials), nparts, count, _ptr(gamma), _ptr(dgamma), _ptr(dbeta), _ptr(coef),
                           _ptr(workspace), _ptr(sync), C.byref(fa), C.byref(fy), C.byref(fm) if fm else None, _ptr(mean), _ptr(invstd),
                           _ptr(scale), _ptr(shift), 1 if relu else 0, C.byref(fo), keep=ts)

    def bn_eval_coeffs(self, gamma, beta, rm, rv, eps, c, scale, shift):
        return self._plain("sfk_bn_eval_coeffs", _ptr(gamma), _ptr(beta), _ptr(rm), _ptr(rv), eps, c, _ptr(scale),
                           _ptr(shift), keep=(gamma, beta, rm, rv, scale, shift))

    def bn_stats(self, y: FMap, partials, max_parts):
        """returns (run, nparts)"""
        fy = _c_fmap(y)
        np_ = C.c_int32(0)
        run = self._plain("sfk_bn_stats", C.byref(fy), _ptr(partials), max_parts, C.byref(np_),
                          keep=(fy, np_, y, partials))
        return run, self._dry_parts(y, max_parts)

    @staticmethod
    def _dry_parts(y: FMap, max_parts: int) -> int:
        # mirrors chan_grid() in csrc/bn.hip
        vec = 8 if y.dtype == torch.bfloat16 else 4
        cgs = y.c // vec
        cgs_b = min(cgs, 256)
        rows_b = 256 // cgs_b
        parts = (y.pixels + rows_b * 16 - 1) // (rows_b * 16)
        cchunks = (cgs + 255) // 256
        parts = min(parts, max(1, int(tuning().bn_parts) // cchunks))
        if max_parts > 0:
            parts = min(parts, max_parts)
        return max(1, parts)

    def bn_apply(self, y: FMap, scale, shift, res: Optional[FMap], res_scale, res_shift, relu: bool, out: FMap,
                 relu_bits=None, out_sums=None, max_parts: int = 0):
        """relu_bits: optional uint8 tensor of pixels * c / V bytes (V = 8 bf16, 4 f32) that receives the ReLU mask.
        out_sums: optional fp32 [max_parts][c][2] that receives the partial column sums of the output; then the call returns
        (run, nparts) -- nparts is fixed by the geometry, the library reports it when the launch is configured"""
        fy, fo = _c_fmap(y), _c_fmap(out)
        fr = _c_fmap(res) if res is not None else None
        if relu_bits is not None:
            vec = 8 if y.dtype == torch.bfloat16 else 4
            assert relu_bits.dtype == torch.uint8 and relu_bits.numel() >= y.pixels * (y.c // vec)
        np_ = C.c_int32(0)
        if out_sums is not None:
            assert max_parts > 0 and out_sums.numel() >= max_parts * y.c * 2
        run = self._plain("sfk_bn_apply", C.byref(fy), _ptr(scale), _ptr(shift), C.byref(fr) if fr else None,
                          _ptr(res_scale), _ptr(res_shift), 1 if relu else 0, C.byref(fo), _ptr(relu_bits), _ptr(out_sums),
                          max_parts, C.byref(np_) if out_sums is not None else None,
                          keep=(fy, fo, fr, np_, y, out, res, scale, shift, res_scale, res_shift, relu_bits, out_sums))
        if out_sums is None:
            return run
        return run, self._dry_parts(y, max_parts)

    def bn_bwd_reduce(self, da: FMap, y: FMap, mask_src: Optional[FMap], mean, invstd, scale, shift, relu: bool,
                      dz_out: Optional[FMap], partials, max_parts, relu_bits=None):
        """returns (run, nparts); relu_bits: the mask bn_apply wrote (then mask_src must be None)."""
        if relu_bits is not None:
            vec = 8 if da.dtype == torch.bfloat16 else 4
            assert relu_bits.dtype == torch.uint8 and relu_bits.numel() >= da.pixels * (da.c // vec)
        fa, fy = _c_fmap(da), (_c_fmap(y) if y is not None else None)
        fm = _c_fmap(mask_src) if mask_src is not None else None
        fz = _c_fmap(dz_out) if dz_out is not None else None
        np_ = C.c_int32(0)
        run = self._plain("sfk_bn_bwd_reduce", C.byref(fa), C.byref(fy) if fy else None, C.byref(fm) if fm else None, _ptr(mean),
                          _ptr(invstd), _ptr(scale), _ptr(shift), 1 if relu else 0, C.byref(fz) if fz else None,
                          _ptr(partials), max_parts, C.byref(np_), _ptr(relu_bits),
                          keep=(fa, fy, fm, fz, np_, da, y, mask_src, dz_out, mean, invstd, scale, shift, partials,
                                relu_bits))
        return run, self._dry_parts(da, max_parts)

    def bn_bwd_finalize(self, partials, nparts, c, count, gamma, invstd, dgamma, dbeta, coef, workspace=None):
        return self._plain("sfk_bn_bwd_finalize", _ptr(partials), nparts, c, count, _ptr(gamma), _ptr(invstd),
                           _ptr(dgamma), _ptr(dbeta), _ptr(coef), _ptr(workspace),
                           keep=(partials, gamma, invstd, dgamma, dbeta, coef, workspace))

    def bn_bwd_apply(self, da: FMap, y: FMap, mask_src: Optional[FMap], mean, invstd, scale, shift, relu: bool, coef,
                     dy: FMap):
        fa, fy, fo = _c_fmap(da), _c_fmap(y), _c_fmap(dy)
        fm = _c_fmap(mask_src) if mask_src is not None else None
        return self._plain("sfk_bn_bwd_apply", C.byref(fa), C.byref(fy), C.byref(fm) if fm else None, _ptr(mean),
                           _ptr(invstd), _ptr(scale), _ptr(shift), 1 if relu else 0, _ptr(coef), C.byref(fo),
                           keep=(fa, fy, fo, fm, da, y, dy, mask_src, mean, invstd, scale, shift, coef))

    # -- the bottleneck tail (conv_c -> norm_c without the conv output in HBM)
    def bn_tail_fwd(self, gram, a_sums, a_nparts, count, g, c, w, cout, gamma, beta, eps, momentum, running_mean, running_var, nbt,
                    mean, invstd, scale, shift, t, wd=None):
        """gram [c][c]; a_sums: the partial column sums bn_apply left for `a` (a_nparts rows), folded into g [c] (kept for the
        backward); count = pixels.  wd: optional [c][cout] filter (A W)^T of the backward's first data-gradient pass"""
        ts = (gram, a_sums, g, w, gamma, beta, running_mean, running_var, nbt, mean, invstd, scale, shift, t, wd)
        return self._plain("sfk_bn_tail_fwd", _ptr(gram), _ptr(a_sums), a_nparts, count, _ptr(g), c, _ptr(w), _DT[w.dtype], cout,
                           _ptr(gamma), _ptr(beta), eps, momentum, _ptr(running_mean), _ptr(running_var), _ptr(nbt), _ptr(mean),
                           _ptr(invstd), _ptr(scale), _ptr(shift), _ptr(t), _ptr(wd), keep=ts)

    def bn_tail_bwd(self, r, dz_partials, nparts, g, count, t, c, w, cout, gamma, mean, invstd, dgamma, dbeta, dw, m,
                    bias, coef):
        """g [c], count: what bn_tail_fwd folded / was given.  m: [c][c] filter (compute precision) of the second
        data-gradient pass, W^T diag(B) W (include/sfk.h)"""
        ts = (r, dz_partials, g, t, w, gamma, mean, invstd, dgamma, dbeta, dw, m, bias, coef)
        return self._plain("sfk_bn_tail_bwd", _ptr(r), _ptr(dz_partials), nparts, _ptr(g), count, _ptr(t), c, _ptr(w),
                           _DT[w.dtype], cout, _ptr(gamma), _ptr(mean), _ptr(invstd), _ptr(dgamma), _ptr(dbeta), _ptr(dw),
                           _ptr(m), _ptr(bias), _ptr(coef), keep=ts)

    def conv_pw_dual_supported(self, x1: FMap, x2: FMap, y: FMap) -> bool:
        f1, f2, fy = _c_fmap(x1), _c_fmap(x2), _c_fmap(y)
        return bool(load().sfk_conv_pw_dual_supported(C.byref(f1), C.byref(f2), C.byref(fy)))

    def conv_pw_dual(self, x1: FMap, w1, x2: FMap, w2, bias, y: FMap):
        """y = x1 w1^T + x2 w2^T + bias in one pass (both data-gradient passes of a narrow block tail: include/sfk.h)"""
        f1, f2, fy = _c_fmap(x1), _c_fmap(x2), _c_fmap(y)
        return self._plain("sfk_conv_pw_dual", C.byref(f1), _ptr(w1), C.byref(f2), _ptr(w2), _ptr(bias), C.byref(fy),
                           keep=(f1, f2, fy, x1, x2, y, w1, w2, bias))

    # -- pooling / head / loss
    def maxpool_fwd(self, x: FMap, y: FMap, argmax, k, s, p):
        fx, fy = _c_fmap(x), _c_fmap(y)
        return self._plain("sfk_maxpool_fwd", C.byref(fx), C.byref(fy), _ptr(argmax), k, s, p, keep=(fx, fy, x, y, argmax))

    def maxpool_bwd(self, dy: FMap, argmax, dx: FMap, k, s, p):
        fy, fx = _c_fmap(dy), _c_fmap(dx)
        return self._plain("sfk_maxpool_bwd", C.byref(fy), _ptr(argmax), C.byref(fx), k, s, p, keep=(fx, fy, dx, dy, argmax))

    # -- the stem's BatchNorm -> ReLU -> MaxPool without the activation map (include/sfk.h sfk_bn_maxpool_*)
    @staticmethod
    def bn_maxpool_supported(k, s, p) -> bool:
        return (k, s, p) == (3, 2, 1)

    def bn_maxpool_fwd(self, y: FMap, scale, shift, out: FMap, argmax, k, s, p):
        fy, fo = _c_fmap(y), _c_fmap(out)
        return self._plain("sfk_bn_maxpool_fwd", C.byref(fy), _ptr(scale), _ptr(shift), C.byref(fo), _ptr(argmax), k, s, p,
                           keep=(fy, fo, y, out, scale, shift, argmax))

    def bn_maxpool_bwd_reduce(self, d_out: FMap, argmax, y: FMap, mean, invstd, scale, shift, partials, max_parts):
        """returns (run, nparts)"""
        fd, fy = _c_fmap(d_out), _c_fmap(y)
        np_ = C.c_int32(0)
        run = self._plain("sfk_bn_maxpool_bwd_reduce", C.byref(fd), _ptr(argmax), C.byref(fy), _ptr(mean), _ptr(invstd),
                          _ptr(scale), _ptr(shift), _ptr(partials), max_parts, C.byref(np_),
                          keep=(fd, fy, np_, d_out, y, argmax, mean, invstd, scale, shift, partials))
        return run, self._dry_parts(y, max_parts)

    def bn_maxpool_bwd_apply(self, d_out: FMap, argmax, y: FMap, mean, invstd, scale, shift, coef, dy: FMap):
        fd, fy, fo = _c_fmap(d_out), _c_fmap(y), _c_fmap(dy)
        return self._plain("sfk_bn_maxpool_bwd_apply", C.byref(fd), _ptr(argmax), C.byref(fy), _ptr(mean), _ptr(invstd),
                           _ptr(scale), _ptr(shift), _ptr(coef), C.byref(fo),
                           keep=(fd, fy, fo, d_out, y, dy, argmax, mean, invstd, scale, shift, coef))

    def head_pool_fwd(self, x: FMap, k, rate, seed, feat, feat_ld, f_off):
        fx = _c_fmap(x)
        return self._plain("sfk_head_pool_fwd", C.byref(fx), k[0], k[1], k[2], rate, _ptr(seed), _ptr(feat), feat_ld,
                           f_off, keep=(fx, x, seed, feat))

    def head_pool_bwd(self, dfeat, feat_ld, f_off, k, rate, seed, dx: FMap):
        fx = _c_fmap(dx)
        return self._plain("sfk_head_pool_bwd", _ptr(dfeat), feat_ld, f_off, k[0], k[1], k[2], rate, _ptr(seed),
                           C.byref(fx), keep=(fx, dx, seed, dfeat))

    def head_dropout_mask(self, n, c, f_off, positions, rate, seed, mask):
        return self._plain("sfk_head_dropout_mask", n, c, f_off, positions, rate, _ptr(seed), _ptr(mask), keep=(seed, mask))

    def fc_fwd(self, feat, w, b, logits, n, f, k):
        return self._plain("sfk_fc_fwd", _ptr(feat), _ptr(w), _ptr(b), _ptr(logits), n, f, k, keep=(feat, w, b, logits))

    def fc_bwd(self, dlogits, feat, w, dfeat, dw, db, n, f, k):
        return self._plain("sfk_fc_bwd", _ptr(dlogits), _ptr(feat), _ptr(w), _ptr(dfeat), _ptr(dw), _ptr(db), n, f, k,
                           keep=(dlogits, feat, w, dfeat, dw, db))

    def softmax_ce(self, logits, labels, n, k, gscale, dlogits, loss_out, loss_sum, correct):
        return self._plain("sfk_softmax_ce", _ptr(logits), _ptr(labels), n, k, gscale, _ptr(dlogits), _ptr(loss_out),
                           _ptr(loss_sum), _ptr(correct), keep=(logits, labels, dlogits, loss_out, loss_sum, correct))

    # -- optimiser / misc
    def adam(self, p, g, m, v, count, lr, b1, b2, eps, gscale, step, shadow=None):
        sd = _DT[shadow.dtype] if shadow is not None else SFK_F32
        return self._plain("sfk_adam", _ptr(p), _ptr(g), _ptr(m), _ptr(v), count, lr, b1, b2, eps, gscale, _ptr(step),
                           _ptr(shadow), sd, keep=(p, g, m, v, step, shadow))

    def filter_transpose(self, src, dst, cout, wtaps, cin):
        return self._plain("sfk_filter_transpose", _ptr(src), _DT[src.dtype], _ptr(dst), _DT[dst.dtype], cout, wtaps,
                           cin, keep=(src, dst))

    def cast(self, src, dst, count):
        return self._plain("sfk_cast", _ptr(src), _DT[src.dtype], _ptr(dst), _DT[dst.dtype], count, keep=(src, dst))

    def filter_refresh(self, master, s, st, layers):
        """layers: [(arena offset, cout, wtaps, cin, transpose)], one launch for all of them (sfk_filter_refresh)."""
        import numpy as np
        ent = np.zeros(len(layers), dtype=np.dtype([("off", "<i8"), ("cout", "<i4"), ("wtaps", "<i4"), ("cin", "<i4"),
                                                    ("first_block", "<i4"), ("transpose", "<i4"), ("reserved", "<i4")]))
        fb = 0
        for i, (off, cout, wtaps, cin, tr) in enumerate(layers):
            ent[i] = (off, cout, wtaps, cin, fb, 1 if tr else 0, 0)
            fb += wtaps * ((cout + 31) // 32) * ((cin + 31) // 32)
        table = torch.from_numpy(ent.view(np.uint8).copy()).to(master.device)
        ref = st if st is not None else s
        return self._plain("sfk_filter_refresh", _ptr(master), _ptr(s), _ptr(st), _DT[ref.dtype], _ptr(table),
                           len(layers), fb, keep=(master, s, st, table))

    def u8_normalize_crop(self, src_u8, lut, crop, pad: int, out):
        """src_u8 (N,T,H,W,C) uint8 -> out (N,T,C,H,W) f32|bf16 = lut[byte], shifted by the per-clip crop (or None)"""
        n, t, h, w, c = src_u8.shape
        assert src_u8.dtype == torch.uint8 and src_u8.is_contiguous() and out.is_contiguous()
        assert tuple(out.shape) == (n, t, c, h, w)
        return self._plain("sfk_u8_normalize_crop", _ptr(src_u8), _ptr(lut), _ptr(crop), pad, _ptr(out), _DT[out.dtype],
                           n, t, c, h, w, keep=(src_u8, lut, crop, out))

    def eval_aggregate(self, logits, labels, seg_off, nvideos: int, softmax: bool, ps_out, pred, correct):
        return self._plain("sfk_eval_aggregate", _ptr(logits), _ptr(labels), _ptr(seg_off), nvideos, logits.shape[1],
                           1 if softmax else 0, _ptr(ps_out), _ptr(pred), _ptr(correct),
                           keep=(logits, labels, seg_off, ps_out, pred, correct))

    def sparse_fusion_fwd(self, x, w, b, y, n, p, c):
        return self._plain("sfk_sparse_fusion_fwd", _ptr(x), _ptr(w), _ptr(b), _ptr(y), n, p, c, keep=(x, w, b, y))

    def sparse_fusion_bwd(self, x, dy, dw, db, n, p, c):
        return self._plain("sfk_sparse_fusion_bwd", _ptr(x), _ptr(dy), _ptr(dw), _ptr(db), n, p, c, keep=(x, dy, dw, db))

    def fill_zero(self, t: torch.Tensor):
        return self._plain("sfk_fill_zero", t.data_ptr(), t.numel() * t.element_size(), keep=(t,))

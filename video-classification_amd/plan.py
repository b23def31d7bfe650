"""Conv3d geometry -> implicit-GEMM pass descriptors for sfk_conv_igemm / sfk_conv_wgrad (include/sfk.h).

Pure host logic (no GPU): unit-tested on CPU against torch's conv3d and its autograd through a numpy
restatement of the C-ABI contract (tests/emu_backend.py).

A pass is   Y[row*os + oo] (+)= sum_taps X[row*gs + tap.d] . W[:, tap.widx, :]   (see sfk_conv_desc).
"""
from __future__ import annotations

import itertools
from dataclasses import dataclass
from typing import List, Tuple

Triple = Tuple[int, int, int]
Tap = Tuple[int, int, int, int]


@dataclass(frozen=True)
class ConvGeom:
    """nn.Conv3d(cin, cout, k, stride=s, padding=p, bias=False), dilation 1, groups 1."""
    cin: int
    cout: int
    k: Triple
    s: Triple = (1, 1, 1)
    p: Triple = (0, 0, 0)

    @property
    def wtaps(self) -> int:
        return self.k[0] * self.k[1] * self.k[2]

    def out_dims(self, dims: Triple) -> Triple:
        return tuple((d + 2 * p - k) // s + 1 for d, p, k, s in zip(dims, self.p, self.k, self.s))

    def widx(self, kt: int, kh: int, kw: int) -> int:
        return (kt * self.k[1] + kh) * self.k[2] + kw

    def macs(self, dims: Triple) -> int:
        o = self.out_dims(dims)
        return o[0] * o[1] * o[2] * self.cout * self.cin * self.wtaps


@dataclass(frozen=True)
class PassSpec:
    rows: Triple
    gs: Triple
    os: Triple
    oo: Triple
    taps: Tuple[Tap, ...]


def fwd_pass(g: ConvGeom, in_dims: Triple) -> PassSpec:
    """rows = output pixels; gathered pixel = o*s - p + k."""
    taps = tuple((kt - g.p[0], kh - g.p[1], kw - g.p[2], g.widx(kt, kh, kw))
                 for kt in range(g.k[0]) for kh in range(g.k[1]) for kw in range(g.k[2]))
    return PassSpec(g.out_dims(in_dims), g.s, (1, 1, 1), (0, 0, 0), taps)


def dgrad_passes(g: ConvGeom, in_dims: Triple) -> Tuple[List[PassSpec], bool]:
    """Data gradient dX (extent in_dims) from dY (extent out_dims), one pass per stride-parity class.

    Input pixel i = a + s*r (class a in [0,s)) receives dY[o] through tap k iff o*s - p + k == i, i.e.
    (a + p - k) % s == 0 and o = r + (a + p - k)//s: inside a class every tap is a constant offset, so the
    class is a dense stride-1 gather over r, scattered to i = r*s + a.  The filter is used with (co, ci)
    swapped ([cin][wtaps][cout], sfk_filter_transpose).  Returns (passes, needs_zero_fill): a class without
    taps (1x1x1 stride-2 shortcuts) leaves its pixels untouched, so dX must be zeroed first unless the
    first pass to touch dX accumulates onto something."""
    passes: List[PassSpec] = []
    needs_zero = False
    for a in itertools.product(range(g.s[0]), range(g.s[1]), range(g.s[2])):
        rows = tuple((d - ai + si - 1) // si if d > ai else 0 for d, ai, si in zip(in_dims, a, g.s))
        if min(rows) <= 0:
            continue
        taps = []
        for kt in range(g.k[0]):
            for kh in range(g.k[1]):
                for kw in range(g.k[2]):
                    num = (a[0] + g.p[0] - kt, a[1] + g.p[1] - kh, a[2] + g.p[2] - kw)
                    if any(nm % si for nm, si in zip(num, g.s)):
                        continue
                    taps.append((num[0] // g.s[0], num[1] // g.s[1], num[2] // g.s[2], g.widx(kt, kh, kw)))
        if not taps:
            needs_zero = True
            continue
        passes.append(PassSpec(rows, (1, 1, 1), g.s, a, tuple(taps)))
    return passes, needs_zero


def wgrad_taps(g: ConvGeom) -> Tuple[Tap, ...]:
    return fwd_pass(g, (1, 1, 1)).taps  # tap table does not depend on the extent


def round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m

#!/usr/bin/env python
"""Micro-benchmark of the fast res5 conv_a data gradient (64 -> 256 channels, taps (3,1,1), accumulate, bitmap mask +
per-tile sums in the epilogue) under different operand values -- looks for data-dependent run time."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_classification_amd._lib import BnBwdFuse, ConvPass, FMap, HipBackend
from video_classification_amd.plan import ConvGeom, dgrad_passes

be = HipBackend()
dev = "cuda"
SHAPES = {"fast5": (32, 32, 7, 7, 256, 64, (3, 1, 1)), "slow2": (32, 8, 56, 56, 256, 64, (1, 1, 1)), "slow3": (32, 8, 28, 28, 512, 128, (1, 1, 1))}
n, t, h, w, cin, cout, kk = SHAPES[sys.argv[1] if len(sys.argv) > 1 else "fast5"]
g = ConvGeom(cin, cout, kk, (1, 1, 1), (kk[0] // 2, 0, 0))
sp = dgrad_passes(g, (t, h, w))[0][0]
px = n * t * h * w
st = torch.cuda.current_stream().cuda_stream


def run(name, dy_t, dx_t, bits_t):
    dy = FMap(dy_t, n, t, h, w, cout)
    dx = FMap(dx_t, n, t, h, w, cin)
    wt = (torch.randn(cout * g.wtaps * cin, device=dev) * 0.05).bfloat16()
    for mode in ("plain", "acc", "acc+bits+sum"):
        cp = ConvPass(dy, dx, sp.rows, sp.gs, sp.os, sp.oo, list(sp.taps), wt, g.wtaps, cout, cin, accumulate=mode != "plain")
        if mode == "acc+bits+sum":
            parts = torch.zeros(4096 * cin * 2, device=dev)
            cp.relu_out_bits = bits_t
            cp.bnb = BnBwdFuse(None, None, None, None, None, None, True, parts)
        r = be.conv_igemm(cp)
        for _ in range(3): r(st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): r(st)
        e1.record(); torch.cuda.synchronize()
        print(f"{name:28s} {mode:14s} {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us")


rnd = lambda c, s=1.0: (torch.randn(px * c, device=dev) * s).bfloat16()
bits_r = torch.randint(0, 256, (px * cin // 8,), dtype=torch.uint8, device=dev)
run("randn", rnd(cout), rnd(cin), bits_r)
if len(sys.argv) <= 1:
    run("randn * 1e-30", rnd(cout, 1e-30), rnd(cin, 1e-30), bits_r)
    run("randn * 1e-39 (denormal)", rnd(cout, 1e-39), rnd(cin, 1e-39), bits_r)
    run("zeros", torch.zeros(px * cout, device=dev).bfloat16(), torch.zeros(px * cin, device=dev).bfloat16(), bits_r)
    run("dy with inf/nan", rnd(cout).fill_(float("nan")), rnd(cin), bits_r)

#!/usr/bin/env python
"""Micro-benchmarks of the block-tail pieces at one layer shape (default: slow res2, 64 -> 256 over 32 x 8 x 56 x 56 pixels).
usage: python tools/bench_tail.py [c4 C t h w]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_classification_amd._lib import ConvEpilogue, ConvPass, FMap, HipBackend, WgradPass

args = [int(a) for a in sys.argv[1:]]
c4, C, t, h, w = args if len(args) == 5 else (64, 256, 8, 56, 56)
n, dev = 32, "cuda"
be = HipBackend()
st = torch.cuda.current_stream().cuda_stream
ONE, TAP0 = (1, 1, 1), [(0, 0, 0, 0)]
pix = n * t * h * w
bf = lambda *s: torch.randn(*s, device=dev).bfloat16()
f32 = lambda *s: torch.zeros(*s, device=dev)


def timeit(name, runs, bytes_, reps=20):
    for r in runs: r(st)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        for r in runs: r(st)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    print(f"{name:44s} {ms*1e3:8.1f} us   {bytes_/ms/1e6:7.0f} GB/s")


for ld in (c4, c4 + 8):
    xb = bf(pix * ld)
    full = FMap(xb, n, t, h, w, ld, ld, 0)
    ab = full.channels(0, c4)
    dz = FMap(bf(pix * C), n, t, h, w, C)
    out = FMap(bf(pix * C), n, t, h, w, C)
    res = FMap(bf(pix * C), n, t, h, w, C)
    wq = (torch.randn(C * c4, device=dev) * c4 ** -0.5).bfloat16()
    sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    bits = torch.zeros(pix * C // 8, dtype=torch.uint8, device=dev)
    B = 2.0
    print(f"--- a_b pixel stride {ld} channels ({pix} pixels, {c4} -> {C})")
    timeit("R-wgrad (dz^T a_b)", [be.conv_wgrad(WgradPass(ab, dz, ONE, TAP0, f32(C * c4), 1, c4, C))], B * pix * (c4 + C))
    parts = f32(1024 * C * 2)
    timeit("mask + sum dz (bn_bwd_reduce, y = NULL)", [be.bn_bwd_reduce(dz, None, None, None, None, None, None, True, dz, parts, 1024, relu_bits=bits)[0]], B * pix * 2 * C + pix * C / 8)
    if ld > c4:
        wp = WgradPass(full, full, ONE, TAP0, f32(ld * ld), 1, ld, ld)
        need = be.conv_wgrad_workspace_bytes(wp)
        wp.workspace = torch.zeros(need // 4 + 4, device=dev)
        timeit("gram (widened, workspace)", [be.conv_wgrad(wp)], B * pix * ld)
    wp = WgradPass(ab, ab, ONE, TAP0, f32(c4 * c4), 1, c4, c4)
    need = be.conv_wgrad_workspace_bytes(wp)
    wp.workspace = torch.zeros(need // 4 + 4, device=dev)
    timeit("gram (c4 only, workspace)", [be.conv_wgrad(wp)], B * pix * c4)
    mk = lambda **kw: ConvPass(ab, out, (t, h, w), ONE, ONE, (0, 0, 0), TAP0, wq, 1, c4, C, **kw)
    timeit("conv_c plain", [be.conv_igemm(mk())], B * pix * (c4 + C))
    timeit("conv_c ep scale/shift", [be.conv_igemm(mk(ep=ConvEpilogue(scale=sc, shift=sh)))], B * pix * (c4 + C))
    timeit("conv_c ep scale/shift+relu+bits", [be.conv_igemm(mk(ep=ConvEpilogue(scale=sc, shift=sh, relu=True, relu_bits=bits)))], B * pix * (c4 + C) + pix * C / 8)
    timeit("conv_c ep full (res, relu, bits)", [be.conv_igemm(mk(ep=ConvEpilogue(scale=sc, shift=sh, res=res, relu=True, relu_bits=bits)))], B * pix * (c4 + 2 * C) + pix * C / 8)
    y = FMap(bf(pix * C), n, t, h, w, C)
    timeit("bn_apply c (y, res -> out, bits)", [be.bn_apply(y, sc, sh, res, None, None, True, out, relu_bits=bits)], B * pix * 3 * C + pix * C / 8)
    da = FMap(bf(pix * c4), n, t, h, w, c4)
    wd = (torch.randn(C * c4, device=dev) * C ** -0.5).bfloat16()
    m = (torch.randn(c4 * c4, device=dev) * c4 ** -0.5).bfloat16()
    timeit("dgrad pass 1 (dz -> da)", [be.conv_igemm(ConvPass(dz, da, (t, h, w), ONE, ONE, (0, 0, 0), TAP0, wd, 1, C, c4))], B * pix * (c4 + C))
    timeit("dgrad pass 2 (a_b -> da +=, bias)", [be.conv_igemm(ConvPass(ab, da, (t, h, w), ONE, ONE, (0, 0, 0), TAP0, m, 1, c4, c4, accumulate=True, ep=ConvEpilogue(shift=torch.zeros(c4, device=dev))))], B * pix * 3 * c4)

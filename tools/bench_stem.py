#!/usr/bin/env python
"""Micro-benchmark of the stem convolutions of the metric geometry through the C ABI (batch 32, 224 x 224, bf16 clip N,C,T,H,W).
usage: python tools/bench_stem.py <fwd|wgrad>_<fast|slow> [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_classification_amd._lib import FMap, HipBackend, StemSrc, stem_kp

kind = sys.argv[1] if len(sys.argv) > 1 else "fwd_fast"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
op, which = kind.split("_")
n, h, w = 32, 224, 224
t, kt, cout = (32, 5, 8) if which == "fast" else (8, 1, 64)
be = HipBackend()
dev = "cuda"
clip = torch.randn(n, 3, t, h, w, device=dev).bfloat16()
src = StemSrc(clip, None, kt)
ho, wo = h // 2, w // 2
y = FMap(torch.randn(n * t * ho * wo * cout, device=dev).bfloat16(), n, t, ho, wo, cout)
kp = stem_kp(3, kt)
wl = (torch.randn(cout * kp, device=dev) * 0.05).bfloat16()
if op == "fwd":
    stats = torch.zeros(be.stem_conv_tiles(src, y) * cout * 2, device=dev)
    run = be.stem_conv_fwd(src, wl, y, stats)
else:
    run = be.stem_conv_wgrad(src, y, torch.zeros(cout * kp, device=dev))
st = torch.cuda.current_stream().cuda_stream
run(st); torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(reps):
    run(st)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / reps
fl = 2.0 * n * t * ho * wo * cout * 3 * kt * 49
by = 2.0 * (clip.numel() + y.buf.numel())
print(f"{kind}: {ms*1e3:.1f} us  {fl/ms/1e9:.1f} TFLOP/s  {by/ms/1e6:.0f} GB/s algorithmic")

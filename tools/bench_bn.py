#!/usr/bin/env python
"""Micro-benchmark of the BatchNorm streaming kernels through the C ABI on one map: apply, backward reduce, backward apply.
usage: python tools/bench_bn.py <pixels> <channels> [out_ld]     (bf16; out_ld > channels: the output is a channel slice)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_classification_amd._lib import FMap, HipBackend

px, c = int(sys.argv[1]), int(sys.argv[2])
old = int(sys.argv[3]) if len(sys.argv) > 3 else c
be, dev = HipBackend(), "cuda"
bf = torch.bfloat16
y = FMap(torch.randn(px * c, device=dev).to(bf), 1, 1, 1, px, c)
da = FMap(torch.randn(px * c, device=dev).to(bf), 1, 1, 1, px, c)
out = FMap(torch.zeros(px * old, device=dev, dtype=bf), 1, 1, 1, px, c, old, 0)
dy = FMap(torch.zeros(px * c, device=dev, dtype=bf), 1, 1, 1, px, c)
v = lambda: torch.rand(c, device=dev) + 0.5
mean, invstd, scale, shift, coef = v(), v(), v(), v() - 1.0, torch.rand(c * 3, device=dev)
parts = torch.zeros(1024 * c * 2, device=dev)
red, _ = be.bn_bwd_reduce(da, y, None, mean, invstd, scale, shift, True, None, parts, 1024)
ops = {"apply": (be.bn_apply(y, scale, shift, None, None, None, True, out), 2),
       "bwd_reduce": (red, 2),
       "bwd_apply": (be.bn_bwd_apply(da, y, None, mean, invstd, scale, shift, True, coef, dy), 3)}
st = torch.cuda.current_stream().cuda_stream
for name, (run, units) in ops.items():
    for _ in range(3):
        run(st)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(20):
        run(st)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / 20 * 1e3
    print(f"px {px} c {c} ld_out {old}: {name:11s} {us:7.1f} us  {units * px * c * 2 / us / 1e6:6.2f} TB/s", flush=True)
# finalize + apply as two launches against the fused launch (sfk_bn_finalize_apply); partial rows as a conv epilogue leaves them
nparts = max(1, min(4096, px // 256))
stats = torch.rand(nparts * c * 2, device=dev)
gamma, beta = v(), v() - 1.0
rm, rv, nbt = torch.zeros(c, device=dev), torch.ones(c, device=dev), torch.zeros(1, dtype=torch.int64, device=dev)
sync = torch.zeros(2144, dtype=torch.int32, device=dev)
fin = be.bn_finalize(stats, nparts, c, px, gamma, beta, 1e-5, 0.1, rm, rv, nbt, mean, invstd, scale, shift)
app = be.bn_apply(y, scale, shift, None, None, None, True, out)
fused = be.bn_finalize_apply(stats, nparts, px, gamma, beta, 1e-5, 0.1, rm, rv, nbt, mean, invstd, None, sync, y, scale, shift, None, None,
                             None, True, out)
for name, runs in (("finalize+apply", (fin, app)), ("fused", (fused,))):
    for _ in range(3):
        for r in runs: r(st)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(20):
        for r in runs: r(st)
    b.record()
    torch.cuda.synchronize()
    print(f"px {px} c {c} rows {nparts}: {name:15s} {a.elapsed_time(b) / 20 * 1e3:7.1f} us", flush=True)

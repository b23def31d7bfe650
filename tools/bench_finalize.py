#!/usr/bin/env python
"""Micro-benchmark of sfk_bn_finalize / sfk_bn_bwd_finalize on the partial-row shapes of the metric step (rows x channels).
usage: python tools/bench_finalize.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_classification_amd._lib import BN_FOLD_ROWS, HipBackend

be, dev = HipBackend(), "cuda"
st = torch.cuda.current_stream().cuda_stream
for nparts, c in [(3136, 256), (3136, 64), (1024, 64), (1024, 8), (1024, 32), (784, 512), (392, 1024), (196, 2048), (98, 512), (50176, 8)]:
    f = lambda n, **k: torch.zeros(n, device=dev, **k)
    parts = torch.rand(nparts * c * 2, device=dev) + 0.5
    g, b = f(c) + 1.0, f(c)
    rm, rv, nbt = f(c), f(c) + 1.0, f(1, dtype=torch.int64)
    mean, invstd, scale, shift, ws = f(c), f(c), f(c), f(c), f(BN_FOLD_ROWS * c * 2)
    dg, db, coef = f(c), f(c), f(3 * c)
    ops = {"finalize": be.bn_finalize(parts, nparts, c, nparts * 64, g, b, 1e-5, 0.1, rm, rv, nbt, mean, invstd, scale, shift, ws),
           "bwd_finalize": be.bn_bwd_finalize(parts, nparts, c, nparts * 64, g, invstd, dg, db, coef, ws)}
    for name, run in ops.items():
        for _ in range(3):
            run(st)
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        for _ in range(50):
            run(st)
        e.record()
        torch.cuda.synchronize()
        print(f"rows {nparts:6d} c {c:5d}: {name:13s} {a.elapsed_time(e) / 50 * 1e3:7.1f} us", flush=True)

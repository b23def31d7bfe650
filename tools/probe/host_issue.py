#!/usr/bin/env python3
"""Is the host in the step's critical path?  Times (a) how long the Python thread needs to ISSUE one training step (no sync
inside the loop) against (b) the step's wall time, at the metric batch and at a tiny batch (same ~1,060 launches, 1/16 of the
GPU work: whatever a step costs there is launch-side).   python tools/probe/host_issue.py [batch ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from video_classification_amd.slowfast import pack_pathway_index, slowfast_r50_8x8   # noqa: E402
from video_classification_amd.train import TrainStep                                  # noqa: E402

dev = torch.device("cuda", 0)
for B in [int(a) for a in sys.argv[1:]] or [32, 2]:
    model = slowfast_r50_8x8(400, dtype=torch.bfloat16, device=dev, seed=0)
    model.train()
    frames = torch.randn(B, 3, 32, 224, 224).to(torch.bfloat16).to(dev)
    labels = torch.randint(0, 400, (B,)).to(dev)
    idx = pack_pathway_index(32, 4, dev)
    step = TrainStep(model.engine, lr=2e-4)
    for _ in range(4):
        step(frames, frames, labels, slow_t_index=idx)
    torch.cuda.synchronize()
    K = 10
    t0 = time.perf_counter()
    marks = []
    for _ in range(K):
        step(frames, frames, labels, slow_t_index=idx)
        marks.append(time.perf_counter())
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    per = [round((b - a) * 1e3, 2) for a, b in zip([t0] + marks, marks)]
    print(f"batch {B}: issue {t_issue / K * 1e3:.2f} ms/step, wall {t_all / K * 1e3:.2f} ms/step; per-step issue ms {per}", flush=True)
    del step, model, frames
    torch.cuda.empty_cache()

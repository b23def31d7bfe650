"""cold vs warm first step in ONE process: which forward buffer differs first?"""
import os, sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import test_gpu_model as t
from video_classification_amd import arch
from video_classification_amd.slowfast import SlowFast, pack_pathway_index
from video_classification_amd.train import TrainStep
DEV='cuda'
gen = torch.Generator().manual_seed(77)
frames = torch.randn(2, 3, 32, 224, 224, generator=gen).to(torch.bfloat16).to(DEV)
labels = torch.tensor([3, 250], device=DEV)
idx = pack_pathway_index(32, 4, DEV)
def run():
    m = SlowFast(arch.canonical_spec(400), dtype=torch.bfloat16, device=DEV, backend=t.hip_backend(), seed=5)
    step = TrainStep(m.engine, lr=0.0, use_graph=False)
    loss = float(step(frames, frames, labels, slow_t_index=idx))
    torch.cuda.synchronize()
    snap = {k: v.detach().float().cpu().clone() for k, v in m.engine._bufs.items()
            if k.split('|')[0].split('.')[0] in ('mean', 'invstd', 'stats', 'y', 'a', 'out', 'cat', 'xf', 'feat', 'logits', 'scale', 'shift')}
    return loss, snap, list(m.engine._bufs.keys())
l0, s0, order = run()
l1, s1, _ = run()
l2, s2, _ = run()
print('LOSS cold', repr(l0), 'warm', repr(l1), repr(l2))
bad = [k for k in order if k in s0 and not torch.equal(s0[k], s1[k])]
bad12 = [k for k in order if k in s1 and not torch.equal(s1[k], s2[k])]
print('DIFF cold-vs-warm:', len(bad), bad[:12])
print('DIFF warm-vs-warm:', len(bad12), bad12[:6])
for k in bad[:4]:
    d = (s0[k] - s1[k]).abs()
    print('  ', k, 'numel', d.numel(), 'ndiff', int((d > 0).sum()), 'max', float(d.max()), 'first idx', int((d > 0).nonzero()[0]))

"""cold first training step in a fresh process; prints the loss (5.892194747924805 is the reproducible value)"""
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
m = SlowFast(arch.canonical_spec(400), dtype=torch.bfloat16, device=DEV, backend=t.hip_backend(), seed=5)
m.engine.two_streams = os.environ.get("ONE_STREAM", "0") != "1"
step = TrainStep(m.engine, lr=0.0, use_graph=False)
loss = float(step(frames, frames, labels, slow_t_index=idx))
torch.cuda.synchronize()
print('LOSS', repr(loss))

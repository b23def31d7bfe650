import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from video_classification_amd import arch
from video_classification_amd.slowfast import SlowFast, pack_pathway_index
from video_classification_amd.train import TrainStep
DEV='cuda'
two = sys.argv[1] == '2'
gen = torch.Generator().manual_seed(77)
frames = torch.randn(2, 3, 32, 224, 224, generator=gen).to(torch.bfloat16).to(DEV)
labels = torch.tensor([3, 250], device=DEV)
idx = pack_pathway_index(32, 4, DEV)
m = SlowFast(arch.canonical_spec(400), dtype=torch.bfloat16, device=DEV, seed=5)
m.engine.two_streams = two
step = TrainStep(m.engine, lr=0.0, use_graph=False)
loss = float(step(frames, frames, labels, slow_t_index=idx))
torch.cuda.synchronize()
print('streams', sys.argv[1], 'cold loss', repr(loss), 'G norm', repr(float(m.engine.G.double().norm())))

import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from video_classification_amd import arch
from video_classification_amd.slowfast import SlowFast, pack_pathway_index
DEV='cuda'
gen = torch.Generator().manual_seed(77)
frames = torch.randn(2, 3, 32, 224, 224, generator=gen).to(torch.bfloat16).to(DEV)
idx = pack_pathway_index(32, 4, DEV)
def run(two, train):
    m = SlowFast(arch.canonical_spec(400), dtype=torch.bfloat16, device=DEV, seed=5)
    m.engine.two_streams = two
    m.train(train)
    y = m([frames, frames], slow_t_index=idx).clone()
    torch.cuda.synchronize()
    return y
for train in (False, True):
    for two in (False, True):
        a = run(two, train); b = run(two, train); c = run(two, train)
        print('train', train, 'two_streams', two, 'max|a-b|', float((a-b).abs().max()), float((a-c).abs().max()))
a = run(False, True); b = run(True, True)
print('one vs two streams', float((a-b).abs().max()))

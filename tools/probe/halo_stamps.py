"""Diagnostic build only (-DHALO_STAMP): s_memtime per band of conv_halo_kernel: barrier wait / K loop / epilogue."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from video_classification_amd._lib import ConvPass, FMap, HipBackend
from video_classification_amd.plan import ConvGeom, fwd_pass
n, t, h, w, c = 32, 8, 56, 56, 64
be = HipBackend()
g = ConvGeom(c, c, (1, 3, 3), (1, 1, 1), (0, 1, 1))
x = FMap(torch.randn(n * t * h * w * c, device="cuda").bfloat16(), n, t, h, w, c)
y = FMap(torch.zeros(n * t * h * w * c, device="cuda").bfloat16(), n, t, h, w, c)
wt = (torch.randn(c * 9 * c, device="cuda") * (9 * c) ** -0.5).bfloat16()
sp = fwd_pass(g, (t, h, w))
ps = ConvPass(x, y, sp.rows, sp.gs, sp.os, sp.oo, list(sp.taps), wt, 9, c, c)
ps.stats = torch.zeros(4 << 20, device="cuda")
run = be.conv_igemm(ps)
st = torch.cuda.current_stream().cuda_stream
for _ in range(3): run(st)
torch.cuda.synchronize()
nb = n * t * h // 4
d = ps.stats.view(torch.int32)[2 << 20:(2 << 20) + nb * 16].view(nb, 4, 4).cpu().double()
print("per band (cycles, mean over bands and waves): wait+barrier %.0f  K loop %.0f  epilogue %.0f" % tuple(d[:, :, :3].mean((0, 1))))
first = d[:256, :, :3].mean((0, 1)); rest = d[256:, :, :3].mean((0, 1))
print("first band of a workgroup: %.0f %.0f %.0f   later bands: %.0f %.0f %.0f" % (*first, *rest))

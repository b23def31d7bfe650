"""Diagnostic build only (-DP8_STAMP): per-segment s_memtime sums of conv_igemm_p8_kernel's main loop, averaged over waves.
usage: SFK_LIB=.../libsfk_stamp.so SFK_P8=1 python tools/probe/p8_stamps.py [cin cout kt n t h w]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from video_classification_amd._lib import ConvPass, FMap, HipBackend
from video_classification_amd.plan import ConvGeom, fwd_pass

a = [int(v) for v in sys.argv[1:]] or [1024, 256, 3, 32, 8, 16, 16]
cin, cout, kt, n, t, h, w = a
be = HipBackend()
g = ConvGeom(cin, cout, (kt, 1, 1), (1, 1, 1), (kt // 2, 0, 0))
x = FMap(torch.randn(n * t * h * w * cin, device="cuda").bfloat16(), n, t, h, w, cin)
y = FMap(torch.zeros(n * t * h * w * cout, device="cuda").bfloat16(), n, t, h, w, cout)
wt = (torch.randn(cout * kt * cin, device="cuda") * (kt * cin) ** -0.5).bfloat16()
sp = fwd_pass(g, (t, h, w))
ps = ConvPass(x, y, sp.rows, sp.gs, sp.os, sp.oo, list(sp.taps), wt, g.wtaps, cin, cout)
ps.stats = torch.zeros(4 << 20, device="cuda")
run = be.conv_igemm(ps)
st = torch.cuda.current_stream().cuda_stream
for _ in range(3): run(st)
torch.cuda.synchronize()
nblk = be.conv_igemm_mtiles(ps) * ((cout + 255) // 256)
d = ps.stats.view(torch.int32)[2 << 20:(2 << 20) + nblk * 8 * 16].view(nblk, 8, 16).cpu().double()
kt_total = kt * cin // 64
names = ["L issue (+ds_read return)", "vmcnt+barrier1+lgkm", "M (mfma issue)", "barrier2"]
for grp in (0, 1):
    m = d[:, 4 * grp:4 * grp + 4].mean((0, 1)) / kt_total
    print(f"group {grp}: cycles per K-tile {m.sum():.0f}")
    for q in range(4):
        print("   q%d: " % q + "  ".join(f"{names[i]} {m[4 * q + i]:.0f}" for i in range(4)))

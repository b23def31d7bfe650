"""diagnostic: error map of the band filter-gradient kernel by (co block, tap, ci block)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from emu_backend import EmuBackend
from test_gpu_kernels import fmap_pair, DEV, stream
from video_classification_amd._lib import HipBackend, WgradPass
from video_classification_amd.plan import ConvGeom, wgrad_taps
n, t, h, w = [int(a) for a in (sys.argv[1] if len(sys.argv) > 1 else "2,4,56,56").split(",")]
hip = HipBackend()
cin = cout = 64
gen = torch.Generator().manual_seed(5)
g = ConvGeom(cin, cout, (1, 3, 3), (1, 1, 1), (0, 1, 1))
xc, xg = fmap_pair(n, cin, t, h, w, torch.bfloat16, gen, ld=cin, c_off=0)
dyc, dyg = fmap_pair(n, cout, t, h, w, torch.bfloat16, gen, ld=cout, c_off=0)
dwc = torch.zeros(cout * 9 * cin)
EmuBackend().conv_wgrad(WgradPass(xc, dyc, g.s, list(wgrad_taps(g)), dwc, g.wtaps, cin, cout))(0)
wp = WgradPass(xg, dyg, g.s, list(wgrad_taps(g)), None, g.wtaps, cin, cout)
wp.dw = torch.zeros(cout * 9 * cin, device=DEV)
need = hip.conv_wgrad_workspace_bytes(wp)
wp.workspace = torch.zeros(need // 4 + 4, device=DEV)
hip.conv_wgrad(wp)(stream()); torch.cuda.synchronize()
got = wp.dw.cpu().view(4, 16, 9, 4, 16)
ref = dwc.view(4, 16, 9, 4, 16)
err = (got - ref).abs()
print("taps", list(wgrad_taps(g)))
print("scale", float(ref.abs().max()))
for cbo in range(4):
    for tp in range(9):
        print(cbo, tp, " ".join(f"{float(err[cbo, :, tp, cib].max()):9.3f}" for cib in range(4)))
bad = err[3, :, 8, 3]
print("err by co%16 (rows) x ci%16 (cols) of block (3, 8, 3):")
for r in range(16): print(" ".join(f"{float(v):7.2f}" for v in bad[r]))

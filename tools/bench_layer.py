#!/usr/bin/env python
"""Micro-benchmark of ONE conv layer of the metric model through the C ABI (for rocprofv3 --pmc runs); custom shapes: fwd_x1024,256,3,1,1,32,8,16,16
usage: python tools/bench_layer.py <kind> [reps]   kind in fwd_a4, fwd_b4, fwd_c4, wgrad_a4, wgrad_b4, dgrad_a4, fwd_c2"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from video_classification_amd._lib import ConvPass, FMap, HipBackend, WgradPass
from video_classification_amd.plan import ConvGeom, dgrad_passes, fwd_pass, wgrad_taps

LAYERS = {  # cin, cout, k, s, p, (t, h, w)     batch 32
    "a4": (1024, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0), (8, 14, 14)),
    "b4": (256, 256, (1, 3, 3), (1, 1, 1), (0, 1, 1), (8, 14, 14)),
    "c4": (256, 1024, (1, 1, 1), (1, 1, 1), (0, 0, 0), (8, 14, 14)),
    "c2": (64, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0), (8, 56, 56)),
    "b2": (64, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), (8, 56, 56)),
    "b3": (128, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1), (8, 28, 28)),
    "a2": (256, 64, (1, 1, 1), (1, 1, 1), (0, 0, 0), (8, 56, 56)),
    "a5": (2048, 512, (3, 1, 1), (1, 1, 1), (1, 0, 0), (8, 7, 7)),
    "b5": (512, 512, (1, 3, 3), (1, 1, 1), (0, 1, 1), (8, 7, 7)),
    # fast pathway (T = 32)
    "fb2": (8, 8, (1, 3, 3), (1, 1, 1), (0, 1, 1), (32, 56, 56)),
    "fa2": (32, 8, (3, 1, 1), (1, 1, 1), (1, 0, 0), (32, 56, 56)),
    "fc2": (8, 32, (1, 1, 1), (1, 1, 1), (0, 0, 0), (32, 56, 56)),
    "fb3": (16, 16, (1, 3, 3), (1, 1, 1), (0, 1, 1), (32, 28, 28)),
    "fa4": (128, 32, (3, 1, 1), (1, 1, 1), (1, 0, 0), (32, 14, 14)),
}
kind = sys.argv[1] if len(sys.argv) > 1 else "fwd_a4"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
op, name = kind.split("_", 1)
n = 32
if name.startswith("x"):            # custom: x<cin>,<cout>,<kt>,<kh>,<kw>,<n>,<t>,<h>,<w>  (stride 1, same padding)
    v = [int(a) for a in name[1:].split(",")]
    cin, cout, k, dims, n = v[0], v[1], tuple(v[2:5]), tuple(v[6:9]), v[5]
    s, p = (1, 1, 1), tuple(a // 2 for a in k)
else:
    cin, cout, k, s, p, dims = LAYERS[name]
be = HipBackend()
dev = "cuda"
g = ConvGeom(cin, cout, k, s, p)
od = g.out_dims(dims)
x = FMap(torch.randn(n * dims[0] * dims[1] * dims[2] * cin, device=dev).bfloat16(), n, *dims, cin)
y = FMap(torch.randn(n * od[0] * od[1] * od[2] * cout, device=dev).bfloat16(), n, *od, cout)
w = (torch.randn(cout * g.wtaps * cin, device=dev) * (g.wtaps * cin) ** -0.5).bfloat16()
runs = []
if op in ("fwd", "fwdns"):          # fwdns: without the BatchNorm partial sums
    sp = fwd_pass(g, dims)
    ps = ConvPass(x, y, sp.rows, sp.gs, sp.os, sp.oo, list(sp.taps), w, g.wtaps, cin, cout)
    if op == "fwd":
        ps.stats = torch.zeros(8192 * cout * 2, device=dev)
        assert be.conv_igemm_mtiles(ps) <= 8192
    runs = [be.conv_igemm(ps)]
elif op == "dgradacc":              # data gradient accumulated into dx (identity shortcut)
    for sp in dgrad_passes(g, dims)[0]:
        runs.append(be.conv_igemm(ConvPass(y, x, sp.rows, sp.gs, sp.os, sp.oo, list(sp.taps), w, g.wtaps, cout, cin, accumulate=True)))
elif op == "dgrad":
    for sp in dgrad_passes(g, dims)[0]:
        runs.append(be.conv_igemm(ConvPass(y, x, sp.rows, sp.gs, sp.os, sp.oo, list(sp.taps), w, g.wtaps, cout, cin)))
else:
    dw = torch.zeros(cout * g.wtaps * cin, device=dev)
    wp = WgradPass(x, y, g.s, list(wgrad_taps(g)), dw, g.wtaps, cin, cout)
    # as the engine binds it: a workspace where the kernel wants one (the 256-column tile) or with SFK_WGWS=1 (deterministic sums)
    if be.conv_wgrad_wants_workspace(wp) or os.environ.get("SFK_WGWS", "0") == "1":
        wp.workspace = torch.zeros(be.conv_wgrad_workspace_bytes(wp) // 4 + 4, device=dev)
    runs = [be.conv_wgrad(wp)]
st = torch.cuda.current_stream().cuda_stream
for r in runs: r(st)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(reps):
    for r in runs: r(st)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / reps
fl = 2.0 * n * od[0] * od[1] * od[2] * cout * cin * g.wtaps
by = 2.0 * (x.buf.numel() + y.buf.numel())
print(f"{kind}: {ms*1e3:.1f} us  {fl/ms/1e9:.1f} TFLOP/s  {by/ms/1e6:.0f} GB/s algorithmic")

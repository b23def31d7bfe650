"""GPU parity, kernel by kernel, THROUGH THE C ABI: every libsfk entry point against the torch-CPU restatement of
its contract (tests/emu_backend.py, itself pinned to torch's conv3d / autograd in test_plan_cpu.py) on identical
seeded inputs.  fp32 results must agree to fp32 rounding; bf16 feature maps to one bf16 ulp (2^-8 relative),
with their fp32 side outputs (BN partial sums, filter gradients) held to fp32 accuracy."""
import pytest
import torch

from emu_backend import EmuBackend
from helpers import rel_err, rel_l2
from video_classification_amd._lib import ConvPass, FMap, StemSrc, WgradPass, stem_kp
from video_classification_amd.plan import ConvGeom, dgrad_passes, fwd_pass, wgrad_taps

pytestmark = pytest.mark.gpu
DEV = "cuda"
DTYPES = [torch.float32, torch.bfloat16]
TOL = {torch.float32: 2e-5, torch.bfloat16: 8e-3}


@pytest.fixture(scope="module")
def hip():
    from video_classification_amd._lib import HipBackend
    return HipBackend()


def stream():
    return torch.cuda.current_stream().cuda_stream


def mk(shape, dtype, gen, scale=1.0):
    return (torch.randn(shape, generator=gen) * scale).to(dtype)


def fmap_pair(n, c, t, h, w, dtype, gen, ld=None, c_off=0, fill=None):
    """Identical feature maps on CPU and GPU (optionally a channel slice of a wider record)."""
    ld = c if ld is None else ld
    if fill is None:
        buf = mk((n * t * h * w * ld,), dtype, gen)
    else:
        buf = torch.full((n * t * h * w * ld,), fill, dtype=dtype)
    return FMap(buf, n, t, h, w, c, ld, c_off), FMap(buf.to(DEV), n, t, h, w, c, ld, c_off)


CONV_CASES = [
    # cin, cout, k, s, p, (n, t, h, w)                      what it exercises
    (8, 8, (3, 1, 1), (1, 1, 1), (1, 0, 0), (2, 5, 9, 7)),       # BN=16 tile, half-empty co fragment, ragged M
    (16, 16, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 3, 10, 11)),   # BN=16, 9 taps, cin < BK
    (32, 32, (1, 3, 3), (1, 2, 2), (0, 1, 1), (2, 2, 12, 14)),   # BN=32, stride-2
    (80, 64, (1, 1, 1), (1, 1, 1), (0, 0, 0), (2, 3, 9, 9)),     # BN=64, cin = 2.5 K-steps (slow res2 conv_a)
    (64, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0), (2, 2, 7, 7)),    # BN=128, 2 co tiles, M = 196
    (320, 128, (1, 1, 1), (1, 2, 2), (0, 0, 0), (1, 2, 8, 8)),   # strided shortcut, 10 K-steps
    (128, 136, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 4, 5, 6)),   # cout not a multiple of 16 / of the tile
    (8, 16, (7, 1, 1), (4, 1, 1), (3, 0, 0), (2, 16, 6, 5)),     # canonical lateral fusion
    (160, 8, (5, 1, 1), (1, 1, 1), (2, 0, 0), (1, 8, 6, 6)),     # fast stem temporal part over the patch matrix
    (8, 8, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 1, 1, 1)),       # a single pixel
    (64, 128, (3, 1, 1), (1, 1, 1), (1, 0, 0), (2, 4, 64, 66)),  # M = 33792 rows: the 256x128 LDS-DMA tile, uniform-K walk
    (80, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 8, 72, 60)),  # M = 34560, cin = 2.5 K-steps: 256x128 tile, packed-K walk
    (128, 96, (1, 3, 3), (1, 2, 2), (0, 1, 1), (2, 2, 20, 22)),  # 128x128 LDS-DMA tile, 9 taps, stride 2, ragged cout
    (64, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), (2, 3, 28, 28)),   # conv_b of slow res3 at reduced size
    (64, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1), (1, 2, 56, 56)),  # 56-wide rows, 2 co tiles
    (128, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1), (3, 2, 14, 14)),  # 14x14 frames, 2 ci tiles
    (64, 128, (1, 1, 1), (1, 1, 1), (0, 0, 0), (2, 4, 120, 160)),  # 600 row tiles: more tiles than resident workgroups (persistent kernel: several tiles per block)
    (32, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0), (3, 5, 96, 112)),  # 630 x 2 tiles, 3 taps, ragged last tile
    (512, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0), (2, 4, 70, 72)),   # 256-channel LDS-DMA tile with 224 computed rows (180 tiles: one generation), ragged
    (256, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0), (4, 4, 64, 64)),   # 256 x 256 tile (exactly 256 tiles), 3 taps
]


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("case", CONV_CASES, ids=[f"c{c[0]}-{c[1]}-k{''.join(map(str, c[2]))}-s{''.join(map(str, c[3]))}" for c in CONV_CASES])
def test_conv_forward_and_stats(hip, dtype, case):
    cin, cout, k, s, p, (n, t, h, w) = case
    gen = torch.Generator().manual_seed(hash(case) % 1000)
    emu = EmuBackend()
    g = ConvGeom(cin, cout, k, s, p)
    sp = fwd_pass(g, (t, h, w))
    xc, xg = fmap_pair(n, cin, t, h, w, dtype, gen, ld=cin + 8, c_off=8)
    yc, yg = fmap_pair(n, cout, *sp.rows, dtype, gen, ld=cout + 4, c_off=4, fill=3.0)
    wt = mk((cout * g.wtaps * cin,), dtype, gen, scale=(g.wtaps * cin) ** -0.5)
    pc = ConvPass(xc, yc, sp.rows, sp.gs, sp.os, sp.oo, list(sp.taps), wt, g.wtaps, cin, cout)
    pg = ConvPass(xg, yg, sp.rows, sp.gs, sp.os, sp.oo, list(sp.taps), wt.to(DEV), g.wtaps, cin, cout)
    mt, mtc = hip.conv_igemm_mtiles(pg), emu.conv_igemm_mtiles(pc)      # row-tile counts are backend-specific
    pc.stats = torch.zeros(mtc * cout * 2)
    pg.stats = torch.full((mt * cout * 2,), float("nan"), device=DEV)
    emu.conv_igemm(pc)(0)
    hip.conv_igemm(pg)(stream())
    torch.cuda.synchronize()
    assert rel_err(yg.view5().float().cpu(), yc.view5().float()) < TOL[dtype]
    wide = yg.buf.cpu().float().view(-1, cout + 4)
    assert torch.all(wide[:, :4] == 3.0)                            # the neighbouring channel slice is untouched
    sg, sc = pg.stats.cpu().view(mt, cout, 2), pc.stats.view(mtc, cout, 2)
    assert torch.isfinite(sg).all()
    assert rel_err(sg.sum(0), sc.sum(0)) < 1e-4                     # fp32 partial sums even in bf16 mode


DGRAD_CASES = CONV_CASES[:9] + [CONV_CASES[-1], CONV_CASES[-2]]      # + the 256-output-channel tile (plain and +=), 512 outputs


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("case", DGRAD_CASES, ids=[f"c{c[0]}-{c[1]}-k{''.join(map(str, c[2]))}-s{''.join(map(str, c[3]))}" for c in DGRAD_CASES])
def test_conv_data_gradient(hip, dtype, case):
    cin, cout, k, s, p, (n, t, h, w) = case
    gen = torch.Generator().manual_seed(7 + hash(case) % 1000)
    emu = EmuBackend()
    g = ConvGeom(cin, cout, k, s, p)
    od = g.out_dims((t, h, w))
    passes, needs_zero = dgrad_passes(g, (t, h, w))
    dyc, dyg = fmap_pair(n, cout, *od, dtype, gen)
    wt = mk((cin * g.wtaps * cout,), dtype, gen, scale=(g.wtaps * cout) ** -0.5)     # [ci][tap][co]
    for accumulate in (False, True):
        dxc, dxg = fmap_pair(n, cin, t, h, w, dtype, gen, ld=cin + 8, c_off=0)
        if not accumulate:
            dxc.view5().zero_(); dxg.view5().zero_()
        for sp_ in passes:
            emu.conv_igemm(ConvPass(dyc, dxc, sp_.rows, sp_.gs, sp_.os, sp_.oo, list(sp_.taps), wt, g.wtaps, cout,
                                    cin, accumulate=accumulate))(0)
            hip.conv_igemm(ConvPass(dyg, dxg, sp_.rows, sp_.gs, sp_.os, sp_.oo, list(sp_.taps), wt.to(DEV), g.wtaps,
                                    cout, cin, accumulate=accumulate))(stream())
        torch.cuda.synchronize()
        assert rel_err(dxg.view5().float().cpu(), dxc.view5().float()) < TOL[dtype]
        assert torch.equal(dxg.buf.cpu().float().view(-1, cin + 8)[:, cin:], dxc.buf.float().view(-1, cin + 8)[:, cin:])


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("case", CONV_CASES, ids=[f"c{c[0]}-{c[1]}-k{''.join(map(str, c[2]))}-s{''.join(map(str, c[3]))}" for c in CONV_CASES])
def test_conv_filter_gradient(hip, dtype, case):
    cin, cout, k, s, p, (n, t, h, w) = case
    gen = torch.Generator().manual_seed(11 + hash(case) % 1000)
    emu = EmuBackend()
    g = ConvGeom(cin, cout, k, s, p)
    od = g.out_dims((t, h, w))
    xc, xg = fmap_pair(n, cin, t, h, w, dtype, gen, ld=cin + 8, c_off=8)
    dyc, dyg = fmap_pair(n, cout, *od, dtype, gen, ld=cout + 8, c_off=0)
    base = torch.randn(cout * g.wtaps * cin, generator=gen)        # dW is accumulated INTO
    dwc, dwg = base.clone(), base.clone().to(DEV)
    emu.conv_wgrad(WgradPass(xc, dyc, g.s, list(wgrad_taps(g)), dwc, g.wtaps, cin, cout))(0)
    hip.conv_wgrad(WgradPass(xg, dyg, g.s, list(wgrad_taps(g)), dwg, g.wtaps, cin, cout))(stream())
    torch.cuda.synchronize()
    assert rel_err(dwg.cpu(), dwc) < 5e-5                            # fp32 atomics: order differs, accuracy does not
    # the workspace path: pixel splits stored as partial tiles and summed in split order -> same values, and two runs
    # agree BIT FOR BIT (the atomic path does not promise that)
    wp = WgradPass(xg, dyg, g.s, list(wgrad_taps(g)), None, g.wtaps, cin, cout)
    wp.dw = base.clone().to(DEV)
    need = hip.conv_wgrad_workspace_bytes(wp)
    assert need > 0
    outs = []
    for _ in range(2):
        wp.dw = base.clone().to(DEV)
        wp.workspace = torch.full((need // 4 + 4,), float("nan"), device=DEV)
        hip.conv_wgrad(wp)(stream())
        torch.cuda.synchronize()
        outs.append(wp.dw.cpu())
    assert rel_err(outs[0], dwc) < 5e-5
    assert torch.equal(outs[0], outs[1])
    # a workspace that is too small falls back to atomics instead of overrunning it
    wp.dw = base.clone().to(DEV)
    wp.workspace = torch.zeros(4, device=DEV)
    hip.conv_wgrad(wp)(stream())
    torch.cuda.synchronize()
    assert rel_err(wp.dw.cpu(), dwc) < 5e-5


P8_CASES = [
    # cin, cout, k, s, p, (n, t, h, w)             conv_igemm_p8.hip: 256 x 256 tile, 64-channel K-tiles, staggered wave groups
    (512, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0), (4, 4, 64, 64)),     # exactly 256 tiles, 3 taps x 8 K-tiles (even count)
    (192, 256, (1, 3, 3), (1, 1, 1), (0, 1, 1), (2, 3, 57, 61)),     # 9 taps x 3 K-tiles = 27 (odd), ragged M, padding on every side
    (128, 512, (1, 3, 3), (1, 2, 2), (0, 1, 1), (4, 3, 90, 94)),     # two K-tiles per tap, stride 2 (strided data gradient passes), 2 co tiles
    (1024, 768, (1, 1, 1), (1, 1, 1), (0, 0, 0), (2, 4, 50, 52)),    # pointwise, 16 K-tiles, 3 co tiles
    (384, 1024, (3, 1, 1), (1, 1, 1), (1, 0, 0), (1, 9, 48, 40)),    # 4 co tiles share a pixel tile, 224-row tiles (68 x 4 vs 78 x 4)
    (256, 256, (1, 3, 3), (1, 1, 1), (0, 1, 1), (2, 2, 64, 66)),     # 9 taps x 4 K-tiles forward AND data gradient on the tile
]


@pytest.mark.parametrize("case", P8_CASES, ids=[f"c{c[0]}-{c[1]}-k{''.join(map(str, c[2]))}-s{''.join(map(str, c[3]))}" for c in P8_CASES])
def test_conv_p8_tile_bf16(hip, case):
    """conv_igemm_p8_kernel (sfk_tuning.igemm_p8): forward with BatchNorm partial sums into a channel slice, the data gradient
    (plain and +=, strided parity passes) and the ReLU-bitmap epilogue, against the CPU restatement; two runs agree bit for bit
    (the staggered barrier schedule has no data-dependent order)."""
    from video_classification_amd._lib import tuning
    if not tuning().igemm_p8:
        pytest.skip("the deep-pipelined tile is off in this process (SFK_P8=0)")
    cin, cout, k, s, p, (n, t, h, w) = case
    dtype = torch.bfloat16
    gen = torch.Generator().manual_seed(31 + cin + cout)
    emu = EmuBackend()
    g = ConvGeom(cin, cout, k, s, p)
    sp = fwd_pass(g, (t, h, w))
    xc, xg = fmap_pair(n, cin, t, h, w, dtype, gen, ld=cin + 8, c_off=8)
    wt = mk((cout * g.wtaps * cin,), dtype, gen, scale=(g.wtaps * cin) ** -0.5)
    outs = []
    for rep in range(2):
        yc, yg = fmap_pair(n, cout, *sp.rows, dtype, torch.Generator().manual_seed(5), ld=cout + 8, c_off=8, fill=3.0)
        pc = ConvPass(xc, yc, sp.rows, sp.gs, sp.os, sp.oo, list(sp.taps), wt, g.wtaps, cin, cout)
        pg = ConvPass(xg, yg, sp.rows, sp.gs, sp.os, sp.oo, list(sp.taps), wt.to(DEV), g.wtaps, cin, cout)
        assert hip.conv_family(pg) == 4
        mt, mtc = hip.conv_igemm_mtiles(pg), emu.conv_igemm_mtiles(pc)
        pc.stats = torch.zeros(mtc * cout * 2)
        pg.stats = torch.full((mt * cout * 2,), float("nan"), device=DEV)
        if rep == 0:
            emu.conv_igemm(pc)(0)
        hip.conv_igemm(pg)(stream())
        torch.cuda.synchronize()
        outs.append((yg.buf.cpu(), pg.stats.cpu()))
        if rep == 0:
            assert rel_err(yg.view5().float().cpu(), yc.view5().float()) < TOL[dtype]
            assert torch.all(yg.buf.cpu().float().view(-1, cout + 8)[:, :8] == 3.0)
            sg, sc = pg.stats.cpu().view(mt, cout, 2), pc.stats.view(mtc, cout, 2)
            assert torch.isfinite(sg).all() and rel_err(sg.sum(0), sc.sum(0)) < 1e-4
    assert torch.equal(outs[0][0].view(torch.int16), outs[1][0].view(torch.int16)) and torch.equal(outs[0][1], outs[1][1])
    # data gradient: x <- dy (cout channels) through the transposed filter; plain, then +=
    od = g.out_dims((t, h, w))
    passes, _ = dgrad_passes(g, (t, h, w))
    dyc, dyg = fmap_pair(n, cout, *od, dtype, gen)
    wtt = mk((cin * g.wtaps * cout,), dtype, gen, scale=(g.wtaps * cout) ** -0.5)
    for accumulate in (False, True):
        dxc, dxg = fmap_pair(n, cin, t, h, w, dtype, gen, ld=cin + 8, c_off=0)
        if not accumulate:
            dxc.view5().zero_(); dxg.view5().zero_()
        fams = set()
        for sp_ in passes:
            emu.conv_igemm(ConvPass(dyc, dxc, sp_.rows, sp_.gs, sp_.os, sp_.oo, list(sp_.taps), wtt, g.wtaps, cout, cin,
                                    accumulate=accumulate))(0)
            pgd = ConvPass(dyg, dxg, sp_.rows, sp_.gs, sp_.os, sp_.oo, list(sp_.taps), wtt.to(DEV), g.wtaps, cout, cin,
                           accumulate=accumulate)
            fams.add(hip.conv_family(pgd))
            hip.conv_igemm(pgd)(stream())
        torch.cuda.synchronize()
        assert rel_err(dxg.view5().float().cpu(), dxc.view5().float()) < TOL[dtype], (accumulate, fams)
        assert torch.equal(dxg.buf.cpu().float().view(-1, cin + 8)[:, cin:], dxc.buf.float().view(-1, cin + 8)[:, cin:])
    # output ReLU bitmap (+=): stride-1 layers only
    if s == (1, 1, 1):
        px = n * t * h * w
        base = mk((px * (cout + 8),), dtype, gen)
        keep = torch.rand(px, cout, generator=gen) > 0.45
        bits = (keep.reshape(px, cout // 8, 8).to(torch.int32) << torch.arange(8, dtype=torch.int32)).sum(-1).to(torch.uint8).reshape(-1)
        ya, yb = FMap(base.clone().to(DEV), n, t, h, w, cout, cout + 8, 0), FMap(base.clone().to(DEV), n, t, h, w, cout, cout + 8, 0)
        plain = ConvPass(xg, ya, sp.rows, sp.gs, sp.os, sp.oo, list(sp.taps), wt.to(DEV), g.wtaps, cin, cout, accumulate=True)
        masked = ConvPass(xg, yb, sp.rows, sp.gs, sp.os, sp.oo, list(sp.taps), wt.to(DEV), g.wtaps, cin, cout, accumulate=True,
                          relu_out_bits=bits.to(DEV))
        assert hip.conv_family(masked) == 4
        hip.conv_igemm(plain)(stream()); hip.conv_igemm(masked)(stream())
        torch.cuda.synchronize()
        want = ya.buf.cpu().view(px, cout + 8).clone()
        want[:, :cout] = torch.where(keep, want[:, :cout], torch.zeros((), dtype=dtype))
        assert torch.equal(yb.buf.cpu().view(px, cout + 8).view(torch.int16), want.view(torch.int16))


P8_WGRAD_CASES = [
    # cin, cout, k, s, p, (n, t, h, w)             conv_wgrad_p8.hip: 256 x 256 tile, 64-pixel K-tiles, pixel splits through the workspace
    (256, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0), (4, 8, 64, 64)),     # one tap per column tile, 3 column tiles x 85 splits
    (128, 256, (1, 3, 3), (1, 1, 1), (0, 1, 1), (3, 8, 50, 49)),     # two taps per column tile, 4.5 column tiles, ragged last K-tile
    (256, 512, (1, 3, 3), (1, 2, 2), (0, 1, 1), (4, 4, 60, 62)),     # stride 2, two co tiles, 17 K-tiles per workgroup
    (640, 256, (3, 1, 1), (1, 1, 1), (1, 0, 0), (6, 8, 28, 28)),     # column tiles straddle taps at 128-column granularity (cin = 2.5 tiles)
    (384, 768, (1, 1, 1), (1, 1, 1), (0, 0, 0), (8, 4, 40, 36)),     # pointwise, 3 co tiles x 1.5 column tiles
]


@pytest.mark.parametrize("case", P8_WGRAD_CASES, ids=[f"c{c[0]}-{c[1]}-k{''.join(map(str, c[2]))}-s{''.join(map(str, c[3]))}" for c in P8_WGRAD_CASES])
def test_conv_filter_gradient_p8_bf16(hip, case):
    """conv_wgrad_p8_kernel (sfk_tuning.wgrad_p8): against the CPU restatement, into a dW that already holds values; two runs
    agree bit for bit (ordered sum of the splits); without a workspace the call still gives the same gradient (ring kernels)."""
    from video_classification_amd._lib import tuning
    if not tuning().wgrad_p8:
        pytest.skip("the deep-pipelined filter-gradient tile is off in this process (SFK_WGP8=0)")
    cin, cout, k, s, p, (n, t, h, w) = case
    dtype = torch.bfloat16
    gen = torch.Generator().manual_seed(41 + cin + cout)
    emu = EmuBackend()
    g = ConvGeom(cin, cout, k, s, p)
    od = g.out_dims((t, h, w))
    xc, xg = fmap_pair(n, cin, t, h, w, dtype, gen, ld=cin + 8, c_off=8)
    dyc, dyg = fmap_pair(n, cout, *od, dtype, gen, ld=cout + 8, c_off=0)
    base = torch.randn(cout * g.wtaps * cin, generator=gen)
    dwc = base.clone()
    emu.conv_wgrad(WgradPass(xc, dyc, g.s, list(wgrad_taps(g)), dwc, g.wtaps, cin, cout))(0)
    wp = WgradPass(xg, dyg, g.s, list(wgrad_taps(g)), None, g.wtaps, cin, cout)
    wp.dw = base.clone().to(DEV)
    nkt = (n * od[0] * od[1] * od[2] + 63) // 64
    tiles = ((cout + 255) // 256) * ((g.wtaps * cin + 255) // 256)
    assert nkt // (256 // tiles) >= tuning().wgrad_p8, "the cases are sized for the default threshold (16 K-tiles per workgroup)"
    assert hip.conv_wgrad_wants_workspace(wp)
    need = hip.conv_wgrad_workspace_bytes(wp)
    assert need == -(-nkt // -(-nkt // (256 // tiles))) * tiles * 8 * 32 * 64 * 16      # splits x tiles x 256 KB
    outs = []
    for _ in range(2):
        wp.dw = base.clone().to(DEV)
        wp.workspace = torch.full((need // 4 + 4,), float("nan"), device=DEV)
        hip.conv_wgrad(wp)(stream())
        torch.cuda.synchronize()
        outs.append(wp.dw.cpu())
    scale = float((dwc - base).abs().max())
    assert float((outs[0] - dwc).abs().max()) < 2e-5 * scale + 1e-4, (float((outs[0] - dwc).abs().max()), scale)
    assert torch.equal(outs[0], outs[1])
    wp.dw = base.clone().to(DEV)                                   # no workspace: the ring kernels with atomics
    wp.workspace = None
    hip.conv_wgrad(wp)(stream())
    torch.cuda.synchronize()
    assert float((wp.dw.cpu() - dwc).abs().max()) < 2e-5 * scale + 1e-4


BAND_WGRAD_CASES = [
    # c, (n, t, h, w)       conv_wgrad_band.hip: the (1,3,3) filter gradient out of LDS bands, whole dW in the workgroups' accumulators
    (64, (2, 4, 56, 56)),   # 112 bands, one per workgroup: the K-parity-1 waves run the odd K-steps only
    (64, (3, 8, 56, 56)),   # 336 bands on 256 workgroups: one or two bands each (both buffers)
    (64, (5, 8, 8, 56)),    # frames of 8 rows: every band is the first or the last of its frame
    (64, (9, 8, 56, 56)),   # 1008 bands: three or four per workgroup, the buffers alternate
    (128, (2, 5, 28, 28)),  # 70 bands x 2 input-channel halves on 140 workgroups (half K-step at the end of every band)
    (128, (9, 8, 28, 28)),  # 1008 units: three or four per workgroup
    (128, (10, 4, 8, 28)),  # frames of 8 rows
]


@pytest.mark.parametrize("case", BAND_WGRAD_CASES, ids=[f"c{c[0]}-n{c[1][0]}t{c[1][1]}h{c[1][2]}" for c in BAND_WGRAD_CASES])
def test_conv_filter_gradient_band_bf16(hip, case):
    """conv_wgrad_band_kernel (sfk_tuning.wgrad_band): against the CPU restatement, out of / into channel slices, into a dW that
    already holds values; two runs agree bit for bit (ordered sum of the workgroups' partials); without a workspace the call
    still gives the same gradient (implicit-GEMM kernels)."""
    from video_classification_amd._lib import tuning
    if not tuning().wgrad_band:
        pytest.skip("the LDS-band filter-gradient kernel is off in this process (SFK_WGBAND=0)")
    cin, (n, t, h, w) = case
    cout = cin
    if not (tuning().wgrad_band & (1 if cin == 64 else 2)):
        pytest.skip("this width of the LDS-band filter-gradient kernel is off in this process (SFK_WGBAND)")
    dtype = torch.bfloat16
    gen = torch.Generator().manual_seed(67 + n + h)
    emu = EmuBackend()
    g = ConvGeom(cin, cout, (1, 3, 3), (1, 1, 1), (0, 1, 1))
    xc, xg = fmap_pair(n, cin, t, h, w, dtype, gen, ld=cin + 8, c_off=8)
    dyc, dyg = fmap_pair(n, cout, t, h, w, dtype, gen, ld=cout + 16, c_off=8)
    base = torch.randn(cout * g.wtaps * cin, generator=gen)
    dwc = base.clone()
    emu.conv_wgrad(WgradPass(xc, dyc, g.s, list(wgrad_taps(g)), dwc, g.wtaps, cin, cout))(0)
    wp = WgradPass(xg, dyg, g.s, list(wgrad_taps(g)), None, g.wtaps, cin, cout)
    wp.dw = base.clone().to(DEV)
    assert hip.conv_wgrad_wants_workspace(wp)
    need = hip.conv_wgrad_workspace_bytes(wp)
    assert need == min(256, n * t * (h // 4) * (cin // 64)) * (cout // 64) * 144 * 1024    # one (cout x 576) fp32 partial per workgroup
    outs = []
    for _ in range(2):
        wp.dw = base.clone().to(DEV)
        wp.workspace = torch.full((need // 4 + 4,), float("nan"), device=DEV)
        hip.conv_wgrad(wp)(stream())
        torch.cuda.synchronize()
        outs.append(wp.dw.cpu())
    scale = float((dwc - base).abs().max())
    assert float((outs[0] - dwc).abs().max()) < 2e-5 * scale + 1e-4, (float((outs[0] - dwc).abs().max()), scale)
    assert torch.equal(outs[0], outs[1])
    wp.dw = base.clone().to(DEV)                                   # no workspace: the implicit-GEMM kernels with atomics
    wp.workspace = None
    hip.conv_wgrad(wp)(stream())
    torch.cuda.synchronize()
    assert float((wp.dw.cpu() - dwc).abs().max()) < 2e-5 * scale + 1e-4


@pytest.mark.parametrize("cin,dims", [(64, (32, 8, 56, 56)), (128, (32, 8, 28, 28))], ids=["res2", "res3"])
def test_conv_filter_gradient_band_at_the_metric_size(hip, cin, dims):
    """the band kernel at the benchmark's own layer sizes (slow res2 / res3 conv_b, batch 32): no CPU restatement finishes there, so
    the check is a property -- the SAME gradient out of two independent kernels (LDS bands + ordered partial sums against the
    implicit GEMM with atomics: different staging, different K order, different reduction), linearity in dY (dW(2 dY) = 2 dW(dY)
    bit for bit: powers of two commute with every rounding) and a repeat that is bit-identical."""
    from video_classification_amd._lib import tuning
    if not (tuning().wgrad_band & (1 if cin == 64 else 2)):
        pytest.skip("this width of the LDS-band filter-gradient kernel is off in this process (SFK_WGBAND)")
    n, t, h, w = dims
    g = ConvGeom(cin, cin, (1, 3, 3), (1, 1, 1), (0, 1, 1))
    gen = torch.Generator(device=DEV).manual_seed(3)
    x = FMap((torch.randn(n * t * h * w * cin, device=DEV, generator=gen)).to(torch.bfloat16), n, t, h, w, cin)
    dy = FMap((torch.randn(n * t * h * w * cin, device=DEV, generator=gen) * 0.05).to(torch.bfloat16), n, t, h, w, cin)
    dy2 = FMap((dy.buf.float() * 2).to(torch.bfloat16), n, t, h, w, cin)
    numel = cin * g.wtaps * cin

    def grad(dy_, workspace):
        wp = WgradPass(x, dy_, g.s, list(wgrad_taps(g)), torch.zeros(numel, device=DEV), g.wtaps, cin, cin)
        if workspace:
            wp.workspace = torch.empty(hip.conv_wgrad_workspace_bytes(wp) // 4 + 4, device=DEV)
        hip.conv_wgrad(wp)(stream())
        torch.cuda.synchronize()
        return wp.dw
    band, band_again, band2, ring = grad(dy, True), grad(dy, True), grad(dy2, True), grad(dy, False)
    assert torch.equal(band, band_again)
    assert torch.equal(band2, band * 2)
    scale = float(ring.abs().max())
    assert float((band - ring).abs().max()) < 2e-4 * scale, (float((band - ring).abs().max()), scale)
    assert rel_l2(band.cpu(), ring.cpu()) < 2e-5


HALO_CASES = [
    # cin = cout, (n, t, h, w)            conv_halo.hip: the (1,3,3) stride-1 conv of slow res2 out of an LDS band, filter in registers
    (64, (2, 3, 56, 56)),       # 84 bands: one per workgroup
    (64, (3, 8, 56, 56)),       # 336 bands: persistent workgroups take a second band (double-buffered band, slab ring across bands)
    (64, (1, 2, 8, 56)),        # 4 bands: frames of 8 rows (first and last band of a frame adjacent)
]


@pytest.mark.parametrize("case", HALO_CASES, ids=[f"c{c[0]}-n{c[1][0]}t{c[1][1]}" for c in HALO_CASES])
def test_conv_halo_bf16(hip, case):
    """conv_halo_kernel (sfk_tuning.igemm_halo): forward with BatchNorm partial sums into / out of channel slices, and the plain
    data gradient (flipped taps, transposed filter), against the CPU restatement; two runs agree bit for bit."""
    from video_classification_amd._lib import tuning
    if not tuning().igemm_halo:
        pytest.skip("the LDS-band kernel is off in this process (SFK_HALO=0)")
    c, (n, t, h, w) = case
    dtype = torch.bfloat16
    gen = torch.Generator().manual_seed(53 + c + n)
    emu = EmuBackend()
    g = ConvGeom(c, c, (1, 3, 3), (1, 1, 1), (0, 1, 1))
    sp = fwd_pass(g, (t, h, w))
    xc, xg = fmap_pair(n, c, t, h, w, dtype, gen, ld=c + 8, c_off=8)
    wt = mk((c * 9 * c,), dtype, gen, scale=(9 * c) ** -0.5)
    outs = []
    for rep in range(2):
        yc, yg = fmap_pair(n, c, *sp.rows, dtype, torch.Generator().manual_seed(5), ld=c + 8, c_off=8, fill=3.0)
        pc = ConvPass(xc, yc, sp.rows, sp.gs, sp.os, sp.oo, list(sp.taps), wt, 9, c, c)
        pg = ConvPass(xg, yg, sp.rows, sp.gs, sp.os, sp.oo, list(sp.taps), wt.to(DEV), 9, c, c)
        assert hip.conv_family(pg) == 5
        mt, mtc = hip.conv_igemm_mtiles(pg), emu.conv_igemm_mtiles(pc)
        assert mt == n * t * (h // 4)
        pc.stats = torch.zeros(mtc * c * 2)
        pg.stats = torch.full((mt * c * 2,), float("nan"), device=DEV)
        if rep == 0:
            emu.conv_igemm(pc)(0)
        hip.conv_igemm(pg)(stream())
        torch.cuda.synchronize()
        outs.append((yg.buf.cpu(), pg.stats.cpu()))
        if rep == 0:
            assert rel_err(yg.view5().float().cpu(), yc.view5().float()) < TOL[dtype]
            assert torch.all(yg.buf.cpu().float().view(-1, c + 8)[:, :8] == 3.0)
            sg, sc = pg.stats.cpu().view(mt, c, 2), pc.stats.view(mtc, c, 2)
            assert torch.isfinite(sg).all() and rel_err(sg.sum(0), sc.sum(0)) < 1e-4
    assert torch.equal(outs[0][0].view(torch.int16), outs[1][0].view(torch.int16)) and torch.equal(outs[0][1], outs[1][1])
    # data gradient (the engine's conv_b -> conv_a hand-over: plain stores)
    passes, _ = dgrad_passes(g, (t, h, w))
    assert len(passes) == 1
    dyc, dyg = fmap_pair(n, c, t, h, w, dtype, gen)
    wtt = mk((c * 9 * c,), dtype, gen, scale=(9 * c) ** -0.5)
    dxc, dxg = fmap_pair(n, c, t, h, w, dtype, gen, ld=c + 8, c_off=0)
    sp_ = passes[0]
    emu.conv_igemm(ConvPass(dyc, dxc, sp_.rows, sp_.gs, sp_.os, sp_.oo, list(sp_.taps), wtt, 9, c, c))(0)
    pgd = ConvPass(dyg, dxg, sp_.rows, sp_.gs, sp_.os, sp_.oo, list(sp_.taps), wtt.to(DEV), 9, c, c)
    assert hip.conv_family(pgd) == 5
    hip.conv_igemm(pgd)(stream())
    torch.cuda.synchronize()
    assert rel_err(dxg.view5().float().cpu(), dxc.view5().float()) < TOL[dtype]
    assert torch.equal(dxg.buf.cpu().float().view(-1, c + 8)[:, c:], dxc.buf.float().view(-1, c + 8)[:, c:])
    # += keeps the implicit GEMM (no accumulate epilogue in the band kernel)
    assert hip.conv_family(ConvPass(dyg, dxg, sp_.rows, sp_.gs, sp_.os, sp_.oo, list(sp_.taps), wtt.to(DEV), 9, c, c, accumulate=True)) != 5


def test_conv_rejects_bad_descriptors(hip):
    from video_classification_amd._lib import SfkError
    gen = torch.Generator().manual_seed(0)
    _, x = fmap_pair(1, 8, 1, 4, 4, torch.bfloat16, gen)
    _, y = fmap_pair(1, 8, 1, 4, 4, torch.bfloat16, gen)
    w = torch.zeros(64, dtype=torch.bfloat16, device=DEV)
    bad_cin = ConvPass(x, y, (1, 4, 4), (1, 1, 1), (1, 1, 1), (0, 0, 0), [(0, 0, 0, 0)], w, 1, 16, 8)
    with pytest.raises(SfkError):
        hip.conv_igemm(bad_cin)(stream())
    oob = ConvPass(x, y, (1, 4, 4), (1, 1, 1), (2, 2, 2), (0, 0, 0), [(0, 0, 0, 0)], w, 1, 8, 8)   # scatter leaves y
    with pytest.raises(SfkError):
        hip.conv_igemm(oob)(stream())
    _, x6 = fmap_pair(1, 6, 1, 4, 4, torch.bfloat16, gen)                                           # cin % 8 != 0
    with pytest.raises(SfkError):
        hip.conv_igemm(ConvPass(x6, y, (1, 4, 4), (1, 1, 1), (1, 1, 1), (0, 0, 0), [(0, 0, 0, 0)], w, 1, 6, 8))(stream())


STEM_CASES = [
    # cin, cout, kt, (n, t, h, w), channel slice of the 21-channel dataset record, frame index
    (3, 8, 5, (2, 6, 36, 44), None, None),                  # canonical fast stem, ragged 18x22 output (partial tiles)
    (3, 64, 1, (1, 8, 64, 32), None, [0, 3, 7]),            # canonical slow stem reading frames through PackPathway's index
    (5, 64, 1, (2, 3, 40, 40), (0, 5), None),               # reference slow stem: BGR+UV slice of N,T,21,H,W memory
    (15, 8, 1, (2, 3, 40, 40), (5, 20), None),              # reference fast stem: flow slice
    (3, 8, 5, (2, 6, 40, 48), None, None),                  # fast stem, 16-byte rows: the frame-stationary kernels (bf16)
    (3, 8, 5, (1, 9, 72, 64), None, [0, 2, 3, 5, 8, 8, 1]), # the same through a frame index, 36x32 output, 3x2 tiles
    (3, 8, 3, (2, 4, 32, 32), None, None),                  # kt = 3 on the frame-stationary path
    (3, 8, 5, (1, 21, 40, 48), None, None),                 # 11 output pairs = two temporal units (8 + 3), odd clip length
    (3, 8, 3, (1, 19, 32, 40), None, None),                 # the same for kt = 3
]


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("src_dtype", DTYPES, ids=["src_f32", "src_bf16"])
@pytest.mark.parametrize("case", STEM_CASES, ids=["fast5x7x7", "slow_tindex", "ref_slow", "ref_fast", "fast5_rows16", "fast5_tindex", "fast3_rows16", "fast5_units", "fast3_units"])
def test_stem_conv_direct(hip, dtype, src_dtype, case):
    cin, cout, kt, (n, t, h, w), chan, tidx = case
    gen = torch.Generator().manual_seed(17)
    emu = EmuBackend()
    if chan is None:
        clip = mk((n, cin, t, h, w), src_dtype, gen)                       # N,C,T,H,W
        vc, vg = clip, clip.to(DEV)
    else:
        mem = mk((n, t, 21, h, w), src_dtype, gen)                         # dataset memory N,T,C,H,W
        vc = mem.permute(0, 2, 1, 3, 4)[:, chan[0]:chan[1]]
        vg = mem.to(DEV).permute(0, 2, 1, 3, 4)[:, chan[0]:chan[1]]
    ti_c = None if tidx is None else torch.tensor(tidx, dtype=torch.int32)
    ti_g = None if tidx is None else ti_c.to(DEV)
    t_out = t if tidx is None else len(tidx)
    ho, wo = (h + 6 - 7) // 2 + 1, (w + 6 - 7) // 2 + 1
    kp = stem_kp(cin, kt)
    # filters in the stem layout, zero padding kept zero
    wref = torch.randn(cout, cin, kt, 7, 7, generator=gen) * (cin * kt * 49) ** -0.5
    wl = torch.nn.functional.pad(wref.permute(0, 2, 1, 3, 4), (0, 1)).reshape(cout, kt * cin * 56)
    wl = torch.nn.functional.pad(wl, (0, kp - wl.shape[1])).reshape(-1).to(dtype)
    yc, yg = fmap_pair(n, cout, t_out, ho, wo, dtype, gen, ld=cout + 4, c_off=4, fill=2.0)
    sc_, sg_ = StemSrc(vc, ti_c, kt), StemSrc(vg, ti_g, kt)
    mt = hip.stem_conv_tiles(sg_, yg)
    assert mt == emu.stem_conv_tiles(sc_, yc)
    stc, stg = torch.zeros(mt * cout * 2), torch.full((mt * cout * 2,), float("nan"), device=DEV)
    emu.stem_conv_fwd(sc_, wl, yc, stc)(0)
    hip.stem_conv_fwd(sg_, wl.to(DEV), yg, stg)(stream())
    torch.cuda.synchronize()
    assert rel_err(yg.view5().float().cpu(), yc.view5().float()) < TOL[dtype]
    assert torch.all(yg.buf.cpu().float().view(-1, cout + 4)[:, :4] == 2.0)
    assert torch.isfinite(stg).all()
    assert rel_err(stg.cpu().view(mt, cout, 2).sum(0), stc.view(mt, cout, 2).sum(0)) < 1e-4
    # filter gradient, accumulated into a non-zero arena; padding entries must stay untouched
    dyc, dyg = fmap_pair(n, cout, t_out, ho, wo, dtype, gen, ld=cout + 8, c_off=8)
    base = torch.randn(cout * kp, generator=gen)
    dwc, dwg = base.clone(), base.clone().to(DEV)
    emu.stem_conv_wgrad(sc_, dyc, dwc)(0)
    hip.stem_conv_wgrad(sg_, dyg, dwg)(stream())
    torch.cuda.synchronize()
    assert rel_err(dwg.cpu(), dwc) < 5e-5
    pad = torch.ones(cout, kp, dtype=torch.bool)
    pad[:, : kt * cin * 56].view(cout, kt * cin * 7, 8)[..., :7] = False
    assert torch.equal(dwg.cpu().view(cout, kp)[pad], base.view(cout, kp)[pad])


SLOW_STEM_CASES = [
    # (n, t, h, w), frame index          stem_fwd_s3_kernel: canonical slow stem (3 -> 64, kt = 1), bf16, 16-byte output records
    ((2, 5, 72, 64), None),               # odd clip length (a half pair), 36 x 32 output = 3 x 2 tiles with a ragged last row of tiles
    ((1, 8, 40, 48), [0, 3, 7, 7, 2]),    # frames through PackPathway's index, 20 x 24 output (partial tiles both ways)
    ((3, 4, 32, 32), None),               # whole tiles, two pairs = one unit
]


@pytest.mark.parametrize("case", SLOW_STEM_CASES, ids=["odd_t", "tindex", "whole"])
def test_stem_conv_slow_register_filter_bf16(hip, case):
    """the slow stem's forward on the register-filter kernel (sfk_tuning.stem_v3 bit 1): output into a channel slice with 16-byte
    pixel records, BatchNorm partial sums, against the CPU restatement; the bytes beside the slice stay untouched."""
    from video_classification_amd._lib import tuning
    if not (tuning().stem_v3 & 2):
        pytest.skip("the slow stem's register-filter kernel is off in this process (SFK_STEM3)")
    (n, t, h, w), tidx = case
    cin, cout, kt, dtype = 3, 64, 1, torch.bfloat16
    gen = torch.Generator().manual_seed(29 + h)
    emu = EmuBackend()
    clip = mk((n, cin, t, h, w), dtype, gen)
    ti_c = None if tidx is None else torch.tensor(tidx, dtype=torch.int32)
    ti_g = None if tidx is None else ti_c.to(DEV)
    t_out = t if tidx is None else len(tidx)
    ho, wo = (h + 6 - 7) // 2 + 1, (w + 6 - 7) // 2 + 1
    kp = stem_kp(cin, kt)
    wref = torch.randn(cout, cin, kt, 7, 7, generator=gen) * (cin * kt * 49) ** -0.5
    wl = torch.nn.functional.pad(wref.permute(0, 2, 1, 3, 4), (0, 1)).reshape(cout, kt * cin * 56)
    wl = torch.nn.functional.pad(wl, (0, kp - wl.shape[1])).reshape(-1).to(dtype)
    yc, yg = fmap_pair(n, cout, t_out, ho, wo, dtype, gen, ld=cout + 8, c_off=8, fill=2.0)
    sc_, sg_ = StemSrc(clip, ti_c, kt), StemSrc(clip.to(DEV), ti_g, kt)
    mt = hip.stem_conv_tiles(sg_, yg)
    stc, stg = torch.zeros(mt * cout * 2), torch.full((mt * cout * 2,), float("nan"), device=DEV)
    emu.stem_conv_fwd(sc_, wl, yc, stc)(0)
    hip.stem_conv_fwd(sg_, wl.to(DEV), yg, stg)(stream())
    torch.cuda.synchronize()
    assert rel_err(yg.view5().float().cpu(), yc.view5().float()) < TOL[dtype]
    assert torch.all(yg.buf.cpu().float().view(-1, cout + 8)[:, :8] == 2.0)
    assert torch.isfinite(stg).all()
    assert rel_err(stg.cpu().view(mt, cout, 2).sum(0), stc.view(mt, cout, 2).sum(0)) < 1e-4   # (the restatement's rows are not tiles)


def test_conv_halo_at_the_metric_size(hip):
    """the LDS-band 3 x 3 kernel on slow res2 conv_b at batch 32: against the implicit GEMM on the same operands (the `+=` form of
    the call into a zeroed map is not eligible for the band kernel and takes the implicit-GEMM family), exact linearity in the
    filter and a bit-identical repeat; the BatchNorm partial rows add up to the sums of the stored map."""
    from video_classification_amd._lib import tuning
    if not tuning().igemm_halo:
        pytest.skip("the LDS-band kernel is off in this process (SFK_HALO=0)")
    n, t, h, w, c = 32, 8, 56, 56, 64
    g = ConvGeom(c, c, (1, 3, 3), (1, 1, 1), (0, 1, 1))
    gen = torch.Generator(device=DEV).manual_seed(4)
    x = FMap(torch.randn(n * t * h * w * c, device=DEV, generator=gen).to(torch.bfloat16), n, t, h, w, c)
    wt = (torch.randn(c * g.wtaps * c, device=DEV, generator=gen) * (g.wtaps * c) ** -0.5).to(torch.bfloat16)
    sp = fwd_pass(g, (t, h, w))

    def run(filt, accumulate, stats):
        y = FMap(torch.zeros(n * t * h * w * c, dtype=torch.bfloat16, device=DEV), n, t, h, w, c)
        ps = ConvPass(x, y, sp.rows, sp.gs, sp.os, sp.oo, list(sp.taps), filt, g.wtaps, c, c, accumulate=accumulate)
        fam = hip.conv_family(ps)
        if stats:
            ps.stats = torch.full((hip.conv_igemm_mtiles(ps) * c * 2,), float("nan"), device=DEV)
        hip.conv_igemm(ps)(stream())
        torch.cuda.synchronize()
        return y.buf, fam, ps.stats
    ya, fam_a, st = run(wt, False, True)
    yb, fam_b, _ = run(wt, False, False)
    y2, _, _ = run((wt.float() * 2).to(torch.bfloat16), False, False)
    yg, fam_g, _ = run(wt, True, False)
    assert fam_a == 5 and fam_b == 5 and fam_g != 5                    # band kernel / implicit GEMM
    assert torch.equal(ya, yb)
    assert torch.equal(y2.view(torch.int16), (ya.float() * 2).to(torch.bfloat16).view(torch.int16))
    assert rel_l2(ya.float().cpu(), yg.float().cpu()) < 2e-3
    v = ya.float().view(-1, c)
    stv = st.view(-1, c, 2)
    assert torch.isfinite(stv).all()
    assert rel_err(stv.sum(0)[:, 0].cpu(), v.sum(0).cpu()) < 2e-3 and rel_err(stv.sum(0)[:, 1].cpu(), (v * v).sum(0).cpu()) < 2e-3


@pytest.mark.parametrize("which", ["fast", "slow"])
def test_stem_forward_at_the_metric_size(hip, which):
    """the register-filter stem kernels at the benchmark's clip (32 x 3 x T x 224 x 224): the same convolution out of two
    independent kernels -- the frame-stationary half-tile kernel on the bf16 clip against the generic patch kernel on an fp32
    copy of the same values (another staging, another K order) -- plus exact linearity in the filter (2 w -> 2 y bit for bit) and
    BatchNorm partial sums that add up to the sums of the stored map."""
    from video_classification_amd._lib import tuning
    if not (tuning().stem_v3 & (1 if which == "fast" else 2)):
        pytest.skip("this stem's register-filter kernel is off in this process (SFK_STEM3)")
    n, h, w = 32, 224, 224
    t, kt, cout = (32, 5, 8) if which == "fast" else (8, 1, 64)
    gen = torch.Generator(device=DEV).manual_seed(9)
    clip = torch.randn(n, 3, t, h, w, device=DEV, generator=gen).to(torch.bfloat16)
    kp = stem_kp(3, kt)
    wref = torch.randn(cout, 3, kt, 7, 7, generator=torch.Generator().manual_seed(2)) * (3 * kt * 49) ** -0.5
    wl = torch.nn.functional.pad(wref.permute(0, 2, 1, 3, 4), (0, 1)).reshape(cout, kt * 3 * 56)
    wl = torch.nn.functional.pad(wl, (0, kp - wl.shape[1])).reshape(-1).to(torch.bfloat16).to(DEV)
    ho, wo = h // 2, w // 2

    def fwd(src, filt):
        y = FMap(torch.zeros(n * t * ho * wo * cout, dtype=torch.bfloat16, device=DEV), n, t, ho, wo, cout)
        ssrc = StemSrc(src, None, kt)
        mt = hip.stem_conv_tiles(ssrc, y)
        stats = torch.full((mt * cout * 2,), float("nan"), device=DEV)
        hip.stem_conv_fwd(ssrc, filt, y, stats)(stream())
        torch.cuda.synchronize()
        return y.buf, stats.view(mt, cout, 2)
    y1, st1 = fwd(clip, wl)
    y2, _ = fwd(clip, (wl.float() * 2).to(torch.bfloat16))
    assert torch.equal(y2.view(torch.int16), (y1.float() * 2).to(torch.bfloat16).view(torch.int16))
    yg, stg = fwd(clip.float(), wl)                     # fp32 source: the generic kernel
    assert rel_l2(y1.float().cpu(), yg.float().cpu()) < 2e-3
    assert torch.isfinite(st1).all()
    v = y1.float().view(-1, cout)
    # (the kernel sums the fp32 accumulators, the check sums the bf16-rounded map: 2^-9 relative per element, far less in the sum)
    assert rel_err(st1.sum(0)[:, 0].cpu(), v.sum(0).cpu()) < 2e-3
    assert rel_err(st1.sum(0)[:, 1].cpu(), (v * v).sum(0).cpu()) < 2e-3
    assert rel_err(st1.sum(0).cpu(), stg.sum(0).cpu()) < 1e-3


@pytest.mark.parametrize("c,nparts", [(8, 50176), (256, 3136), (80, 129), (2048, 300), (64, 128)],
                         ids=lambda v: str(v))
def test_batchnorm_partial_fold_two_level(hip, c, nparts):
    """sfk_bn_finalize / sfk_bn_bwd_finalize with the fold workspace (thousands of partial rows, as a conv over the
    benchmark's stem / res2 maps leaves them) == without it == a double-precision fold on the host."""
    from video_classification_amd._lib import BN_FOLD_ROWS
    gen = torch.Generator().manual_seed(c + nparts)
    parts = (torch.rand(nparts, c, 2, generator=gen) + 0.5)
    parts[:, :, 1] += 3.0                      # sumsq rows > sum^2 / count
    count = nparts * 4
    gamma, beta = torch.rand(c, generator=gen) + 0.5, torch.randn(c, generator=gen)
    f = lambda *sh, **k: torch.zeros(*sh, device=DEV, **k)
    pg = parts.reshape(-1).to(DEV)
    outs = []
    for ws in (None, f(BN_FOLD_ROWS * c * 2)):
        rm, rv, nbt = f(c), torch.ones(c, device=DEV), f(1, dtype=torch.int64)
        mean, invstd, scale, shift = f(c), f(c), f(c), f(c)
        hip.bn_finalize(pg, nparts, c, count, gamma.to(DEV), beta.to(DEV), 1e-5, 0.1, rm, rv, nbt, mean, invstd, scale,
                        shift, ws)(stream())
        dgamma, dbeta, coef = f(c), f(c), f(c * 3)
        hip.bn_bwd_finalize(pg, nparts, c, count, gamma.to(DEV), invstd, dgamma, dbeta, coef, ws)(stream())
        torch.cuda.synchronize()
        outs.append([t.cpu() for t in (mean, invstd, scale, shift, rm, rv, dgamma, dbeta, coef)])
        assert int(nbt) == 1
    tot = parts.double().sum(0)
    mu = tot[:, 0] / count
    var = tot[:, 1] / count - mu * mu
    for a, b in zip(*outs):
        assert rel_err(a, b) < 1e-6
    for o in outs:
        assert rel_err(o[0], mu.float()) < 1e-6
        assert rel_err(o[1], (1.0 / torch.sqrt(var + 1e-5)).float()) < 1e-5
        assert rel_err(o[6], tot[:, 1].float()) < 1e-6 and rel_err(o[7], tot[:, 0].float()) < 1e-6


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("c,dims", [(8, (2, 3, 40, 56)), (64, (2, 2, 28, 28)), (256, (1, 2, 14, 14)), (512, (1, 1, 3, 3))],
                         ids=lambda v: str(v) if isinstance(v, int) else "x".join(map(str, v)))
def test_batchnorm_finalize_rides_the_apply_launch(hip, dtype, c, dims):
    """sfk_bn_finalize_apply / sfk_bn_bwd_finalize_apply: the consumer's first workgroups fold the partial rows, the others wait
    for their counter.  Everything the two stand-alone calls write must come out bit for bit -- coefficients, running statistics,
    outputs, ReLU bitmap, dgamma / dbeta (accumulated into non-zero values) -- three launches in a row on the same counters (the last
    workgroup resets them), with and without a shortcut, on grids smaller and larger than the number of channel pairs."""
    n, t, h, w = dims
    px = n * t * h * w
    gen = torch.Generator().manual_seed(11 + c)
    _, y = fmap_pair(n, c, t, h, w, dtype, gen, ld=c + 8, c_off=8)
    _, res = fmap_pair(n, c, t, h, w, dtype, gen)
    gamma, beta = (torch.rand(c, generator=gen) + 0.5).to(DEV), (torch.randn(c, generator=gen) * 0.3).to(DEV)
    f = lambda *s_, **k: torch.zeros(*s_, device=DEV, **k)
    st = stream()
    parts = f(2048 * c * 2)
    run, npart = hip.bn_stats(y, parts, 2048)
    run(st)
    vec = 8 if dtype == torch.bfloat16 else 4

    def forward(fused, sync, with_res):
        rm, rv, nbt = f(c) + 0.25, torch.ones(c, device=DEV), f(1, dtype=torch.int64)
        mean, invstd, scale, shift = f(c), f(c), f(c), f(c)
        out = FMap(torch.zeros(px * c, dtype=dtype, device=DEV), n, t, h, w, c)
        bits = torch.zeros(px * c // vec, dtype=torch.uint8, device=DEV)
        r = res if with_res else None
        for _ in range(3):          # running statistics move three times; the counters come back to zero every time
            if fused:
                hip.bn_finalize_apply(parts, npart, px, gamma, beta, 1e-5, 0.1, rm, rv, nbt, mean, invstd, None, sync, y, scale, shift,
                                      r, None, None, True, out, relu_bits=bits)(st)
            else:
                hip.bn_finalize(parts, npart, c, px, gamma, beta, 1e-5, 0.1, rm, rv, nbt, mean, invstd, scale, shift)(st)
                hip.bn_apply(y, scale, shift, r, None, None, True, out, relu_bits=bits)(st)
        return dict(rm=rm, rv=rv, nbt=nbt, mean=mean, invstd=invstd, scale=scale, shift=shift, out=out.buf, bits=bits)

    def backward(fused, sync, fw, masked):
        da = FMap(mk((px * c,), dtype, torch.Generator().manual_seed(5)).to(DEV), n, t, h, w, c)
        bparts = f(2048 * c * 2)
        run, nb = hip.bn_bwd_reduce(da, y, None, fw["mean"], fw["invstd"], fw["scale"], fw["shift"], True, da if masked else None, bparts, 2048)
        run(st)
        dgamma, dbeta, coef = f(c) + 1.5, f(c) - 0.5, f(c * 3)
        dy = FMap(torch.zeros(px * c, dtype=dtype, device=DEV), n, t, h, w, c)
        for _ in range(2):
            if fused:
                hip.bn_bwd_finalize_apply(bparts, nb, px, gamma, dgamma, dbeta, coef, None, sync, da, y, None, fw["mean"], fw["invstd"],
                                          fw["scale"], fw["shift"], not masked, dy)(st)
            else:
                hip.bn_bwd_finalize(bparts, nb, c, px, gamma, fw["invstd"], dgamma, dbeta, coef)(st)
                hip.bn_bwd_apply(da, y, None, fw["mean"], fw["invstd"], fw["scale"], fw["shift"], not masked, coef, dy)(st)
        return dict(dgamma=dgamma, dbeta=dbeta, coef=coef, dy=dy.buf)

    sync = torch.zeros(2144, dtype=torch.int32, device=DEV)
    for with_res in (False, True):
        a, b = forward(False, None, with_res), forward(True, sync, with_res)
        torch.cuda.synchronize()
        for k_ in a:
            assert torch.equal(a[k_], b[k_]), ("forward", with_res, k_)
        assert int(sync.abs().sum()) == 0
    for masked in (False, True):
        ba, bb = backward(False, None, a, masked), backward(True, sync, a, masked)
        torch.cuda.synchronize()
        for k_ in ba:
            assert torch.equal(ba[k_], bb[k_]), ("backward", masked, k_)
        assert int(sync.abs().sum()) == 0


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("c", [8, 64, 80, 2048], ids=lambda c: f"c{c}")
def test_batchnorm_forward_backward(hip, dtype, c):
    gen = torch.Generator().manual_seed(c)
    emu = EmuBackend()
    n, t, h, w = (2, 3, 5, 7) if c < 2048 else (2, 2, 3, 3)
    px = n * t * h * w
    yc, yg = fmap_pair(n, c, t, h, w, dtype, gen, ld=c + 8, c_off=8)
    rc, rg = fmap_pair(n, c, t, h, w, dtype, gen)
    gamma, beta = torch.rand(c, generator=gen) + 0.5, torch.randn(c, generator=gen) * 0.3

    def side(be, y, res, dev, st):
        f = lambda *s, **k: torch.zeros(*s, device=dev, **k)
        parts = f(2048 * c * 2)
        run, npart = be.bn_stats(y, parts, 2048)
        run(st)
        rm, rv, nbt = f(c), torch.ones(c, device=dev), f(1, dtype=torch.int64)
        mean, invstd, scale, shift = f(c), f(c), f(c), f(c)
        be.bn_finalize(parts, npart, c, px, gamma.to(dev), beta.to(dev), 1e-5, 0.1, rm, rv, nbt, mean, invstd, scale, shift)(st)
        out = FMap(torch.zeros(px * c, dtype=dtype, device=dev), n, t, h, w, c)
        be.bn_apply(y, scale, shift, res, None, None, True, out)(st)
        out2 = FMap(torch.zeros(px * c, dtype=dtype, device=dev), n, t, h, w, c)
        be.bn_apply(y, scale, shift, res, scale, shift, False, out2)(st)
        # backward of out = relu(bn(y) + res): mask from the activation, dz written in place
        da = FMap(mk((px * c,), dtype, torch.Generator().manual_seed(5)).to(dev), n, t, h, w, c)
        bparts = f(2048 * c * 2)
        run, nb = be.bn_bwd_reduce(da, y, out, mean, invstd, scale, shift, True, da, bparts, 2048)
        run(st)
        bp1 = bparts[: nb * c * 2].clone()
        dgamma, dbeta, coef = f(c), f(c), f(c * 3)
        be.bn_bwd_finalize(bparts, nb, c, px, gamma.to(dev), invstd, dgamma, dbeta, coef)(st)
        dy = FMap(torch.zeros(px * c, dtype=dtype, device=dev), n, t, h, w, c)
        be.bn_bwd_apply(da, y, None, mean, invstd, scale, shift, False, coef, dy)(st)
        # backward of relu(bn(y)) with the mask recomputed from y
        da2 = FMap(mk((px * c,), dtype, torch.Generator().manual_seed(6)).to(dev), n, t, h, w, c)
        run, nb2 = be.bn_bwd_reduce(da2, y, None, mean, invstd, scale, shift, True, None, bparts, 2048)
        run(st)
        coef2 = f(c * 3)
        be.bn_bwd_finalize(bparts, nb2, c, px, gamma.to(dev), invstd, None, None, coef2)(st)
        be.bn_bwd_apply(da2, y, None, mean, invstd, scale, shift, True, coef2, da2)(st)   # in place
        ev_s, ev_h = f(c), f(c)
        be.bn_eval_coeffs(gamma.to(dev), beta.to(dev), rm, rv, 1e-5, c, ev_s, ev_h)(st)
        # the same block-output backward through the 1-bit ReLU mask bn_apply can leave behind
        vec = 8 if dtype == torch.bfloat16 else 4
        bits = torch.zeros(px * c // vec + 16, dtype=torch.uint8, device=dev)
        out3 = FMap(torch.zeros(px * c, dtype=dtype, device=dev), n, t, h, w, c)
        be.bn_apply(y, scale, shift, res, None, None, True, out3, relu_bits=bits)(st)
        da3 = FMap(mk((px * c,), dtype, torch.Generator().manual_seed(5)).to(dev), n, t, h, w, c)
        bparts3 = f(2048 * c * 2)
        run, nb3 = be.bn_bwd_reduce(da3, y, None, mean, invstd, scale, shift, True, da3, bparts3, 2048, relu_bits=bits)
        run(st)
        assert nb3 == nb
        return dict(mean=mean, invstd=invstd, scale=scale, shift=shift, rm=rm, rv=rv, out=out.buf, out2=out2.buf,
                    dz=da.buf, dgamma=dgamma, dbeta=dbeta, dy=dy.buf, dy2=da2.buf, ev_s=ev_s, ev_h=ev_h,
                    nbt=nbt.float(), out3=out3.buf, dz3=da3.buf, bits=bits, bparts=bp1,
                    bparts3=bparts3[: nb * c * 2])

    a = side(emu, yc, rc, "cpu", 0)
    b = side(hip, yg, rg, DEV, stream())
    torch.cuda.synchronize()
    for kname in a:
        if kname in ("bits", "bparts", "bparts3"):
            continue
        tol = TOL[dtype] if kname in ("out", "out2", "dz", "dy", "dy2", "out3", "dz3") else 2e-4
        assert rel_err(b[kname].float().cpu(), a[kname].float()) < tol, kname
    # the bitmap route is the mask_src route bit for bit (outputs, dz and the partial sums), and the bytes are the
    # packed signs of the activation
    for side_ in (a, b):
        assert torch.equal(side_["out3"], side_["out"]) and torch.equal(side_["dz3"], side_["dz"])
        vec = 8 if dtype == torch.bfloat16 else 4
        pos = (side_["out"].float() > 0).reshape(-1, vec).to(torch.int32)
        packed = (pos << torch.arange(vec, dtype=torch.int32, device=pos.device)).sum(-1).to(torch.uint8)
        assert torch.equal(side_["bits"][: packed.numel()], packed)
        assert int(side_["bits"][packed.numel():].sum()) == 0
    assert torch.equal(b["bparts3"], b["bparts"])
    # and against torch's own BatchNorm for the statistics
    v = yc.view5().float().reshape(-1, c)
    assert rel_err(b["mean"].cpu(), v.mean(0)) < 1e-5
    assert rel_err(b["rv"].cpu(), 0.9 + 0.1 * v.var(0, unbiased=True)) < 1e-5


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
def test_maxpool_with_ties(hip, dtype):
    gen = torch.Generator().manual_seed(2)
    emu = EmuBackend()
    for (h, w) in ((12, 14), (7, 9)):
        xc, xg = fmap_pair(2, 16, 3, h, w, dtype, gen)
        xc.view5().clamp_(min=0); xg.view5().clamp_(min=0)          # post-ReLU: many exact ties at 0
        ho, wo = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
        yc, yg = fmap_pair(2, 16, 3, ho, wo, dtype, gen, ld=24, c_off=8)
        ac = torch.zeros(yc.pixels * 16, dtype=torch.uint8)
        ag = torch.zeros(yc.pixels * 16, dtype=torch.uint8, device=DEV)
        emu.maxpool_fwd(xc, yc, ac, 3, 2, 1)(0)
        hip.maxpool_fwd(xg, yg, ag, 3, 2, 1)(stream())
        torch.cuda.synchronize()
        assert torch.equal(yg.view5().float().cpu(), yc.view5().float())
        assert torch.equal(ag.cpu(), ac)
        # torch's own pool agrees on values and on which element wins a tie
        ref, idx = torch.nn.functional.max_pool3d(xc.view5().float().permute(0, 4, 1, 2, 3), (1, 3, 3), (1, 2, 2),
                                                  (0, 1, 1), return_indices=True)
        assert torch.equal(ref.permute(0, 2, 3, 4, 1), yc.view5().float())
        dyc, dyg = fmap_pair(2, 16, 3, ho, wo, dtype, gen)
        dxc, dxg = fmap_pair(2, 16, 3, h, w, dtype, gen)
        emu.maxpool_bwd(dyc, ac, dxc, 3, 2, 1)(0)
        hip.maxpool_bwd(dyg, ag, dxg, 3, 2, 1)(stream())
        torch.cuda.synchronize()
        assert rel_err(dxg.view5().float().cpu(), dxc.view5().float()) < TOL[dtype]
        xt = xc.view5().float().permute(0, 4, 1, 2, 3).requires_grad_(True)
        torch.nn.functional.max_pool3d(xt, (1, 3, 3), (1, 2, 2), (0, 1, 1)).backward(dyc.view5().float().permute(0, 4, 1, 2, 3))
        assert rel_err(dxc.view5().float(), xt.grad.permute(0, 2, 3, 4, 1)) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("dims", [(2, 16, 3, 12, 14), (3, 8, 2, 7, 9), (1, 64, 2, 16, 16)], ids=str)
def test_stem_tail_bn_relu_maxpool_fused(hip, dtype, dims):
    """sfk_bn_maxpool_fwd / _bwd_reduce / _bwd_apply == sfk_bn_apply(relu) -> sfk_maxpool_fwd and sfk_maxpool_bwd ->
    sfk_bn_bwd_reduce -> sfk_bn_bwd_apply, without the activation map or its gradient: bit-identical pooled values and argmax
    bytes (ties at the ReLU's zeros included), the same dy."""
    n, c, t, h, w = dims
    gen = torch.Generator().manual_seed(sum(dims))
    emu = EmuBackend()
    yc, yg = fmap_pair(n, c, t, h, w, dtype, gen)
    scale, shift = torch.rand(c, generator=gen) + 0.5, torch.randn(c, generator=gen) * 0.3 - 0.8
    ho, wo = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
    oc, og = fmap_pair(n, c, t, ho, wo, dtype, gen)
    ac = torch.zeros(oc.pixels * c, dtype=torch.uint8)
    ag = torch.zeros(oc.pixels * c, dtype=torch.uint8, device=DEV)
    emu.bn_maxpool_fwd(yc, scale, shift, oc, ac, 3, 2, 1)(0)
    hip.bn_maxpool_fwd(yg, scale.to(DEV), shift.to(DEV), og, ag, 3, 2, 1)(stream())
    # the two stand-alone kernels on the same device: the fused pass must reproduce them bit for bit
    a_g = FMap(torch.zeros(yg.pixels * c, dtype=dtype, device=DEV), n, t, h, w, c)
    o2 = FMap(torch.zeros(og.pixels * c, dtype=dtype, device=DEV), n, t, ho, wo, c)
    a2 = torch.zeros_like(ag)
    hip.bn_apply(yg, scale.to(DEV), shift.to(DEV), None, None, None, True, a_g)(stream())
    hip.maxpool_fwd(a_g, o2, a2, 3, 2, 1)(stream())
    torch.cuda.synchronize()
    assert torch.equal(og.view5(), o2.view5()) and torch.equal(ag, a2)
    assert rel_err(og.view5().float().cpu(), oc.view5().float()) < TOL[dtype]
    assert float((ag.cpu() != ac).float().mean()) < 2e-3                   # only where an fma rounding flips a tie
    ac = ag.cpu()                                                          # the backward below: same routing on both sides
    assert float((oc.view5().float() == 0).float().mean()) > 0.01          # the case has all-zero windows (ties)
    # backward
    v = yc.view5().float().reshape(-1, c)
    mean, invstd = v.mean(0), 1.0 / torch.sqrt(v.var(0, unbiased=False) + 1e-5)
    gamma = scale / invstd
    dc, dg = fmap_pair(n, c, t, ho, wo, dtype, gen)
    px = yc.pixels
    res = []
    for be, dev, y_, d_, a_ in ((emu, "cpu", yc, dc, ac), (hip, DEV, yg, dg, ag)):
        f = lambda *sh: torch.zeros(*sh, device=dev)
        to = lambda x: x.to(dev)
        parts, coef, dgamma, dbeta = f(2048 * c * 2), f(c * 3), f(c), f(c)
        run, nb = be.bn_maxpool_bwd_reduce(d_, a_, y_, to(mean), to(invstd), to(scale), to(shift), parts, 2048)
        st = stream() if dev != "cpu" else 0
        run(st)
        be.bn_bwd_finalize(parts, nb, c, px, to(gamma), to(invstd), dgamma, dbeta, coef)(st)
        dy = FMap(torch.zeros(px * c, dtype=dtype, device=dev), n, t, h, w, c)
        be.bn_maxpool_bwd_apply(d_, a_, y_, to(mean), to(invstd), to(scale), to(shift), coef, dy)(st)
        if dev != "cpu":
            torch.cuda.synchronize()
        res.append([x.float().cpu() for x in (dgamma, dbeta, coef, dy.view5())])
    for a, b in zip(*res):
        assert rel_err(b, a) < TOL[dtype]
    # unsupported windows are refused, not mis-computed
    with pytest.raises(Exception):
        hip.bn_maxpool_bwd_apply(dg, ag, dg, mean.to(DEV), invstd.to(DEV), scale.to(DEV), shift.to(DEV), coef, dg)(stream())


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("geom", [((4, 2, 2), (6, 4, 4)), ((8, 7, 7), (8, 7, 7)), ((1, 1, 1), (2, 2, 2))], ids=["ref", "global", "unit"])
def test_head_pool_dropout_fc_loss(hip, dtype, geom):
    k, (t, h, w) = geom
    gen = torch.Generator().manual_seed(4)
    emu = EmuBackend()
    n, c, F, K = 3, 24, 40, 11
    xc, xg = fmap_pair(n, c, t, h, w, dtype, gen)
    P = (t - k[0] + 1) * (h - k[1] + 1) * (w - k[2] + 1)
    seed = torch.tensor([1234567], dtype=torch.int64)
    wfc, bfc = torch.randn(K * F, generator=gen) * 0.2, torch.randn(K, generator=gen)
    labels = torch.tensor([3, 0, 10])
    for rate in (0.0, 0.5):
        def side(be, x, dev, st):
            f = lambda *s, **kw: torch.zeros(*s, device=dev, **kw)
            feat = torch.full((n * F,), 0.25, device=dev)
            be.head_pool_fwd(x, k, rate, seed.to(dev), feat, F, 8)(st)
            mask = f(n * c * P, dtype=torch.uint8)
            be.head_dropout_mask(n, c, 8, P, rate, seed.to(dev), mask)(st)
            logits = f(n, K)
            be.fc_fwd(feat, wfc.to(dev), bfc.to(dev), logits, n, F, K)(st)
            dl, loss, lsum, corr = f(n, K), f(1), f(1), f(1, dtype=torch.int32)
            be.softmax_ce(logits, labels.to(dev), n, K, 1.0, dl, loss, lsum, corr)(st)
            dfeat, dw, db = f(n * F), f(K * F), f(K)
            be.fc_bwd(dl, feat, wfc.to(dev), dfeat, dw, db, n, F, K)(st)
            dx = FMap(torch.zeros(x.pixels * c, dtype=dtype, device=dev), n, t, h, w, c)
            be.head_pool_bwd(dfeat, F, 8, k, rate, seed.to(dev), dx)(st)
            return dict(feat=feat, mask=mask.float(), logits=logits, dl=dl, loss=loss, corr=corr.float(), dfeat=dfeat,
                        dw=dw, db=db, dx=dx.buf)
        a = side(emu, xc, "cpu", 0)
        b = side(hip, xg, DEV, stream())
        torch.cuda.synchronize()
        assert torch.equal(b["mask"].cpu(), a["mask"])              # the counter-based mask is reproducible bit for bit
        if rate > 0:
            assert 0.35 < float(a["mask"].mean()) < 0.65
        for kname in a:
            tol = TOL[dtype] if kname == "dx" else 1e-4
            assert rel_err(b[kname].float().cpu(), a[kname].float()) < tol, (kname, rate)
        # the loss against torch
        want = torch.nn.functional.cross_entropy(a["logits"], labels)
        assert abs(float(b["loss"][0]) - float(want)) < 1e-5


def test_adam_matches_torch(hip):
    gen = torch.Generator().manual_seed(8)
    count = 10_007
    p0 = torch.randn(count + 1, generator=gen)[:count + 1]
    p_t = torch.nn.Parameter(p0[:count].clone())
    opt = torch.optim.Adam([p_t], lr=2e-4)
    pg = torch.zeros(10_016, device=DEV); pg[:count] = p0[:count].to(DEV)
    m, v = torch.zeros_like(pg), torch.zeros_like(pg)
    step = torch.zeros(1, dtype=torch.int64, device=DEV)
    for it in range(5):
        g = torch.randn(count, generator=gen)
        p_t.grad = g.clone()
        opt.step()
        gg = torch.zeros_like(pg); gg[:count] = g.to(DEV)
        hip.adam(pg, gg, m, v, count, 2e-4, 0.9, 0.999, 1e-8, 1.0, step, None)(stream())
    torch.cuda.synchronize()
    assert int(step[0]) == 5
    assert rel_err(pg[:count].cpu(), p_t.detach()) < 1e-6
    assert float(pg[count:].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
def test_filter_transpose_and_cast(hip, dtype):
    gen = torch.Generator().manual_seed(9)
    cout, taps, cin = 24, 3, 40
    src = torch.randn(cout * taps * cin, generator=gen)
    dst = torch.zeros(cout * taps * cin, dtype=dtype, device=DEV)
    hip.filter_transpose(src.to(DEV), dst, cout, taps, cin)(stream())
    sh = torch.zeros(src.numel(), dtype=dtype, device=DEV)
    hip.cast(src.to(DEV), sh, src.numel())(stream())
    torch.cuda.synchronize()
    assert torch.equal(dst.cpu().view(cin, taps, cout), src.view(cout, taps, cin).permute(2, 1, 0).to(dtype))
    assert torch.equal(sh.cpu(), src.to(dtype))


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
def test_filter_refresh_one_launch(hip, dtype):
    """sfk_filter_refresh == per-layer cast + transpose: several layers (one without a data-gradient copy, one larger
    than a 2048-element block, ragged tails) from one arena in one launch; bytes outside the layers stay untouched."""
    gen = torch.Generator().manual_seed(10)
    shapes = [(24, 3, 40, True), (8, 1, 864, False), (64, 9, 64, True), (4, 1, 8, True)]
    layers, off = [], 16
    for cout, taps, cin, tr in shapes:
        layers.append((off, cout, taps, cin, tr))
        off += (cout * taps * cin + 7) // 8 * 8 + 8
    master = torch.randn(off, generator=gen)
    s = torch.full((off,), 7.0, dtype=dtype, device=DEV)
    st = torch.full((off,), 7.0, dtype=dtype, device=DEV)
    hip.filter_refresh(master.to(DEV), s, st, layers)(stream())
    torch.cuda.synchronize()
    ws, wst = torch.full((off,), 7.0, dtype=dtype), torch.full((off,), 7.0, dtype=dtype)
    for o, cout, taps, cin, tr in layers:
        n = cout * taps * cin
        ws[o:o + n] = master[o:o + n].to(dtype)
        if tr:
            wst[o:o + n] = master[o:o + n].view(cout, taps, cin).permute(2, 1, 0).reshape(-1).to(dtype)
    assert torch.equal(s.cpu(), ws) and torch.equal(st.cpu(), wst)
    # transposes only (the parity precision keeps the master arena as its forward copy)
    st2 = torch.full((off,), 7.0, dtype=dtype, device=DEV)
    hip.filter_refresh(master.to(DEV), None, st2, layers)(stream())
    torch.cuda.synchronize()
    assert torch.equal(st2.cpu(), wst)


BNB_CASES = [
    # cin (of the pass), cout (channels of dA), taps/geometry as a forward conv descriptor, dims, accumulate, mask mode
    (64, 256, (1, 1, 1), (1, 1, 1), (0, 0, 0), (2, 4, 24, 28), True, "mask_src"),     # identity-shortcut data gradient
    (256, 64, (1, 1, 1), (1, 1, 1), (0, 0, 0), (1, 8, 72, 60), False, "relu"),        # 256x128 DMA tile? no: cout 64 -> 256x64
    (128, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1), (2, 2, 20, 22), False, "relu"),       # 3x3, 128x128 DMA tile
    (64, 32, (3, 1, 1), (1, 1, 1), (1, 0, 0), (2, 6, 9, 11), False, "relu"),          # narrow: 256x32 tile, ragged M
    (512, 128, (1, 1, 1), (1, 1, 1), (0, 0, 0), (2, 8, 48, 48), True, "none"),        # M = 36864: 256x128 DMA tile
]


@pytest.mark.parametrize("case", BNB_CASES, ids=[f"c{c[0]}-{c[1]}-{c[7]}-acc{int(c[6])}" for c in BNB_CASES])
def test_conv_fused_bn_backward_reduce(hip, case):
    """sfk_conv_desc.bnb: the pass that produces dA stores dz = dA * mask and leaves (sum dz, sum dz*x_hat) partial rows
    == the plain pass followed by sfk_bn_bwd_reduce (dz bit for bit, the folded sums to fp32 accuracy)."""
    from video_classification_amd._lib import BnBwdFuse
    cin, cout, k, s, p, (n, t, h, w), accumulate, mode = case
    dtype = torch.bfloat16
    gen = torch.Generator().manual_seed(31 + cin + cout)
    g = ConvGeom(cin, cout, k, s, p)
    od = g.out_dims((t, h, w))
    sp = fwd_pass(g, (t, h, w))
    _, xg = fmap_pair(n, cin, t, h, w, dtype, gen)
    wgt = mk((cout * g.wtaps * cin,), dtype, gen, (g.wtaps * cin) ** -0.5).to(DEV)
    base = mk((n * od[0] * od[1] * od[2] * (cout + 8),), dtype, gen)                  # dA buffer (accumulated into), slice of a wider record
    _, ybn = fmap_pair(n, cout, *od, dtype, gen)
    _, msk = fmap_pair(n, cout, *od, dtype, gen, ld=cout + 8, c_off=8)
    mean, invstd = (torch.randn(cout, generator=gen) * 0.2).to(DEV), (torch.rand(cout, generator=gen) + 0.5).to(DEV)
    scale, shift = (torch.randn(cout, generator=gen)).to(DEV), (torch.randn(cout, generator=gen) * 0.3).to(DEV)
    mask_src = msk if mode == "mask_src" else None
    relu = mode != "none"

    def dA():
        return FMap(base.clone().to(DEV), n, *od, cout, cout + 8, 0)
    # reference: plain pass, then the stand-alone reduce writing dz in place
    ya = dA()
    hip.conv_igemm(ConvPass(xg, ya, sp.rows, sp.gs, sp.os, sp.oo, list(sp.taps), wgt, g.wtaps, cin, cout, accumulate=accumulate))(stream())
    parts_a = torch.zeros(2048 * cout * 2, device=DEV)
    run, npa = hip.bn_bwd_reduce(ya, ybn, mask_src, mean, invstd, scale, shift, relu, ya, parts_a, 2048)
    run(stream())
    # fused
    yb = dA()
    cp = ConvPass(xg, yb, sp.rows, sp.gs, sp.os, sp.oo, list(sp.taps), wgt, g.wtaps, cin, cout, accumulate=accumulate)
    assert hip.conv_bnb_supported(cp)
    cp.bnb = BnBwdFuse(ybn, mask_src, mean, invstd, scale, shift, relu, torch.zeros(cout * 2, device=DEV))
    mt = hip.conv_igemm_mtiles(cp)          # rows of the descriptor AS LAUNCHED: the tile depends on the epilogue (include/sfk.h)
    parts_b = torch.full((mt * cout * 2,), float("nan"), device=DEV)
    cp.bnb.partials = parts_b
    hip.conv_igemm(cp)(stream())
    torch.cuda.synchronize()
    assert torch.equal(ya.buf.cpu().view(torch.int16), yb.buf.cpu().view(torch.int16))      # dz, and the untouched 8 pad channels
    sa = parts_a[: npa * cout * 2].view(npa, cout, 2).double().sum(0).cpu()
    sb = parts_b.view(mt, cout, 2).double().sum(0).cpu()
    assert torch.isfinite(sb).all()
    assert rel_err(sb[:, 0].float(), sa[:, 0].float()) < 1e-5 and rel_err(sb[:, 1].float(), sa[:, 1].float()) < 1e-5
    # what is not supported says so
    small = ConvPass(xg, FMap(torch.zeros(n * od[0] * od[1] * od[2] * 16, dtype=dtype, device=DEV), n, *od, 16), sp.rows, sp.gs,
                     sp.os, sp.oo, list(sp.taps), wgt[: 16 * g.wtaps * cin], g.wtaps, cin, 16)
    assert not hip.conv_bnb_supported(small)


RELU_OUT_CASES = [
    # cin, cout, kernel, pad, (n, t, h, w), accumulate, dtype
    (64, 256, (1, 1, 1), (0, 0, 0), (2, 4, 24, 28), True, torch.bfloat16),      # identity-shortcut dgrad, 256x128 DMA tile
    (256, 1024, (3, 1, 1), (1, 0, 0), (2, 4, 14, 14), True, torch.bfloat16),    # res4 conv_a dgrad shape
    (8, 32, (3, 1, 1), (1, 0, 0), (2, 6, 9, 11), True, torch.bfloat16),         # fast pathway: 256x32 tile, ragged M
    (16, 64, (3, 1, 1), (1, 0, 0), (1, 5, 13, 10), False, torch.bfloat16),      # 256x64 tile, plain store
    (16, 40, (1, 1, 1), (0, 0, 0), (2, 3, 7, 9), True, torch.float32),          # parity precision: 4-channel nibbles
    (64, 256, (1, 1, 1), (0, 0, 0), (2, 4, 120, 150), True, torch.bfloat16),    # 563 x 2 tiles (several per persistent block), ragged
]


@pytest.mark.parametrize("case", RELU_OUT_CASES, ids=[f"c{c[0]}-{c[1]}-acc{int(c[5])}-{'bf16' if c[6] == torch.bfloat16 else 'f32'}"
                                                      for c in RELU_OUT_CASES])
def test_conv_output_relu_bitmap(hip, case):
    """sfk_conv_desc.out_relu_bits: the pass stores (old +) result where the bitmap bit is set and 0 elsewhere ==
    the plain pass followed by the mask, bit for bit; pad channels of a wider pixel record stay untouched."""
    cin, cout, k, p, (n, t, h, w), accumulate, dtype = case
    gen = torch.Generator().manual_seed(7 + cin + cout)
    g = ConvGeom(cin, cout, k, (1, 1, 1), p)
    sp = fwd_pass(g, (t, h, w))
    xc, xg = fmap_pair(n, cin, t, h, w, dtype, gen)
    wgt = mk((cout * g.wtaps * cin,), dtype, gen, (g.wtaps * cin) ** -0.5)
    px = n * t * h * w
    base = mk((px * (cout + 8),), dtype, gen)
    vec = 8 if dtype == torch.bfloat16 else 4
    keep = torch.rand(px, cout, generator=gen) > 0.45
    packed = (keep.reshape(px, cout // vec, vec).to(torch.int32) << torch.arange(vec, dtype=torch.int32)).sum(-1).to(torch.uint8)
    if vec == 4:
        packed |= 0xA0                                                  # the high nibble of an f32 byte is ignored
    bits = packed.reshape(-1)

    def out(dev):
        return FMap(base.clone().to(dev), n, t, h, w, cout, cout + 8, 0)
    ya, yb, yc = out(DEV), out(DEV), out("cpu")
    plain = ConvPass(xg, ya, sp.rows, sp.gs, sp.os, sp.oo, list(sp.taps), wgt.to(DEV), g.wtaps, cin, cout, accumulate=accumulate)
    assert hip.conv_relu_out_supported(plain)
    hip.conv_igemm(plain)(stream())
    masked = ConvPass(xg, yb, sp.rows, sp.gs, sp.os, sp.oo, list(sp.taps), wgt.to(DEV), g.wtaps, cin, cout, accumulate=accumulate,
                      relu_out_bits=bits.to(DEV))
    hip.conv_igemm(masked)(stream())
    torch.cuda.synchronize()
    want = ya.buf.cpu().view(px, cout + 8).clone()
    want[:, :cout] = torch.where(keep, want[:, :cout], torch.zeros((), dtype=dtype))
    got = yb.buf.cpu().view(px, cout + 8)
    assert torch.equal(got.view(torch.int16 if dtype == torch.bfloat16 else torch.int32),
                       want.view(torch.int16 if dtype == torch.bfloat16 else torch.int32))
    # and the restated contract (emu) agrees with the kernel
    EmuBackend().conv_igemm(ConvPass(xc, yc, sp.rows, sp.gs, sp.os, sp.oo, list(sp.taps), wgt, g.wtaps, cin, cout,
                                     accumulate=accumulate, relu_out_bits=bits))(0)
    assert rel_err(got[:, :cout].float(), yc.buf.view(px, cout + 8)[:, :cout].float()) < TOL[dtype]
    # a strided (scattering) pass cannot take a bitmap and says so
    sc = ConvPass(xg, ya, (t, (h + 1) // 2, (w + 1) // 2), (1, 1, 1), (1, 2, 2), (0, 0, 0), list(sp.taps), wgt.to(DEV), g.wtaps, cin, cout)
    assert not hip.conv_relu_out_supported(sc)


def test_conv_operand_beyond_2gib_runs_register_staged(hip):
    """An activation map of >= 2 GiB (batches beyond the benchmark's) cannot use the LDS-DMA kernel (31-bit offsets); the
    128-wide bf16 tile then runs register-staged instead of failing mid-step.  Checked against the SAME layer run per sample
    (each half is < 2 GiB and takes the DMA kernel), statistics rows included."""
    n, t, h, w, cin, cout = 2, 32, 128, 128, 1024, 128
    g = torch.Generator(device=DEV).manual_seed(1)
    xb = torch.randn(n * t * h * w * cin, generator=g, device=DEV, dtype=torch.bfloat16)
    assert xb.numel() * 2 >= 0x7FF00000
    wt = (torch.randn(cout * cin, generator=g, device=DEV) * 0.03).to(torch.bfloat16)
    taps, rows, one = [(0, 0, 0, 0)], (t, h, w), (1, 1, 1)

    def run(x: FMap):
        y = FMap(torch.zeros(x.pixels * cout, dtype=torch.bfloat16, device=DEV), x.n, t, h, w, cout)
        p = ConvPass(x, y, rows, one, one, (0, 0, 0), taps, wt, 1, cin, cout)
        mt = hip.conv_igemm_mtiles(p)
        p.stats = torch.zeros(mt * cout * 2, device=DEV)
        hip.conv_igemm(p)(stream())
        torch.cuda.synchronize()
        return y.buf, p.stats.view(mt, cout, 2).sum(0)

    y_all, st_all = run(FMap(xb, n, t, h, w, cin))
    per = t * h * w * cin
    parts = [run(FMap(xb[i * per:(i + 1) * per], 1, t, h, w, cin)) for i in range(n)]
    y_ref = torch.cat([p_[0] for p_ in parts])
    assert rel_err(y_all.float().cpu(), y_ref.float().cpu()) < 1e-2 and float(y_ref.float().abs().max()) > 1.0
    assert rel_err(st_all.cpu(), (parts[0][1] + parts[1][1]).cpu()) < 1e-4


# ------------------------------------------------------------------ the bottleneck tail without the conv output in HBM
@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 2, 9, 10, 8, 32), (2, 3, 7, 7, 16, 64), (1, 2, 14, 14, 64, 256), (2, 2, 7, 7, 512, 2048),
                                   (1, 4, 28, 28, 128, 512)],
                         ids=["fast-res2", "fast-res3", "slow-res2", "slow-res5", "slow-res3-dma"])
def test_bottleneck_tail_forward_backward(hip, dtype, shape):
    """sfk_conv_epilogue + sfk_bn_tail_fwd/bwd + sfk_relu_bits_mask + the two sfk_conv_wgrad / sfk_conv_igemm calls around
    them: the call sequence of tests/test_tail_cpu.py on the GPU -- against the same sequence on the CPU restatement
    (same precision) and, in fp32, against torch autograd of conv -> BatchNorm3d -> + shortcut -> ReLU."""
    from test_tail_cpu import make_case, run_tail, tail_reference
    n, t, h, w, c, cout = shape
    case = make_case(n, t, h, w, c, cout, seed=c + cout)
    if dtype == torch.bfloat16:            # both sides start from the same bf16-representable operands
        case = tuple(x.to(torch.bfloat16).double() for x in case)
    got = run_tail(hip, DEV, dtype, *case, stream=stream())
    emu = run_tail(EmuBackend(), "cpu", dtype, *case)
    tol = 2e-5 if dtype == torch.float32 else 1.6e-2
    for k in ("out", "mean", "var", "rm", "rv", "dz", "dW", "dgamma", "dbeta", "da", "m", "bias"):
        assert rel_err(got[k], emu[k]) < (tol if k not in ("var", "rv") else 10 * tol), (k, rel_err(got[k], emu[k]))
    assert got["nbt"] == 1
    if dtype == torch.float32:
        ref = tail_reference(*case)
        assert rel_err(got["out"], ref["out"].float()) < 1e-4
        for k in ("dgamma", "dbeta", "dW", "da"):
            assert rel_err(got[k], ref[k].float()) < 5e-4, k


def test_bottleneck_tail_variance_of_mean_dominated_channels(hip):
    """sfk_bn_tail_fwd on channels whose mean dwarfs their spread (mean^2 / var = 1e3 .. 1e4, e.g. a ReLU output far from zero
    under a same-sign filter row): the variance comes from the centred Gram matrix, so it keeps its digits where
    E[y^2] - E[y]^2 on an fp32 T = W G loses mean^2 / var of them.  Checked against the float64 variance of y = a W^T with the
    Gram matrix and the column sums as exact as fp32 holds them (rounded once from float64)."""
    gen = torch.Generator().manual_seed(5)
    px, c, cout = 4096, 64, 32
    for ratio in (1e3, 1e4):
        a = torch.randn(px, c, generator=gen, dtype=torch.float64) + ratio ** 0.5 * 3.0      # every column: mean^2 / var ~ 9 ratio
        W = torch.rand(cout, c, generator=gen, dtype=torch.float64) * 0.2 + 0.05                 # same-sign rows: the means add up
        gram = (a.t() @ a).float().to(DEV).reshape(-1)
        asums = torch.zeros(c * 2, device=DEV)
        asums.view(c, 2)[:, 0] = a.sum(0).float().to(DEV)                                       # ONE partial row
        y = a @ W.t()
        want_mu, want_var = y.mean(0), y.var(0, unbiased=False)
        assert float((want_mu ** 2 / want_var).min()) > ratio
        f = lambda k: torch.zeros(k, device=DEV)
        mean, invstd, scale, shift, t, gvec = f(cout), f(cout), f(cout), f(cout), f(cout * c), f(c)
        rm, rv, nbt = f(cout), torch.ones(cout, device=DEV), torch.zeros(1, dtype=torch.int64, device=DEV)
        hip.bn_tail_fwd(gram, asums, 1, px, gvec, c, W.float().to(DEV).reshape(-1), cout, torch.ones(cout, device=DEV), f(cout),
                        1e-5, 0.1, rm, rv, nbt, mean, invstd, scale, shift, t)(stream())
        torch.cuda.synchronize()
        assert rel_err(gvec.cpu(), a.sum(0).float()) < 1e-6
        got_var = 1.0 / invstd.cpu().double() ** 2 - 1e-5
        # what limits it now is G itself: one fp32 rounding of entries ~ n mean^2 is 6e-8 mean^2 / var relative to the variance
        bound = 4 * 6e-8 * float((want_mu ** 2 / want_var).max()) + 1e-5
        assert rel_err(got_var.float(), want_var.float()) < bound, (ratio, rel_err(got_var.float(), want_var.float()), bound)
        assert rel_err(mean.cpu(), want_mu.float()) < 1e-6
        assert rel_err(t.cpu().view(cout, c), (W @ (a.t() @ a)).float()) < 1e-5


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "bf16"])
def test_conv_epilogue_variants(hip, dtype):
    """sfk_conv_epilogue on a 3x3 conv and on narrow outputs: scale/shift only, += with a bias, ReLU without a shortcut,
    a shortcut living in a channel slice of a wider record"""
    from video_classification_amd._lib import ConvEpilogue
    gen = torch.Generator().manual_seed(21)
    emu = EmuBackend()
    n, t, h, w = 2, 2, 9, 7
    for cin, cout, k3 in ((16, 8, False), (8, 16, True), (32, 40, True), (24, 136, False)):
        g = ConvGeom(cin, cout, (1, 3, 3) if k3 else (1, 1, 1), (1, 1, 1), (0, 1, 1) if k3 else (0, 0, 0))
        sp = fwd_pass(g, (t, h, w))
        xc, xg = fmap_pair(n, cin, t, h, w, dtype, gen)
        wt = mk((cout * g.wtaps * cin,), dtype, gen, 0.2)
        sc, sh = torch.rand(cout, generator=gen) + 0.5, torch.randn(cout, generator=gen)
        rs, rh = torch.rand(cout, generator=gen) + 0.5, torch.randn(cout, generator=gen)
        wide_ok = cout > 16
        variants = [dict(scale=True, shift=True), dict(shift=True, accumulate=True)]
        if wide_ok:
            variants += [dict(scale=True, shift=True, relu=True, bits=True), dict(scale=True, shift=True, res=True, relu=True, bits=True),
                         dict(shift=True, res=True, res_affine=True, relu=True)]
        for v in variants:
            yc, yg = fmap_pair(n, cout, t, h, w, dtype, gen, ld=cout + 8, c_off=8)
            rc, rg = fmap_pair(n, cout, t, h, w, dtype, gen, ld=cout + 16, c_off=16)
            vec = 8 if dtype == torch.bfloat16 else 4
            outs = []
            for be, x, y, r, dev, st in ((emu, xc, yc, rc, "cpu", 0), (hip, xg, yg, rg, DEV, stream())):
                bits = torch.zeros(y.pixels * (cout // vec), dtype=torch.uint8, device=dev) if v.get("bits") else None
                ep = ConvEpilogue(scale=sc.to(dev) if v.get("scale") else None, shift=sh.to(dev) if v.get("shift") else None,
                                  res=r if v.get("res") else None, relu=bool(v.get("relu")), relu_bits=bits)
                if v.get("res_affine"):
                    ep.res_scale, ep.res_shift = rs.to(dev), rh.to(dev)
                p = ConvPass(x, y, sp.rows, sp.gs, sp.os, sp.oo, list(sp.taps), wt.to(dev), g.wtaps, cin, cout,
                             accumulate=bool(v.get("accumulate")), ep=ep)
                assert be.conv_epilogue_supported(p)
                be.conv_igemm(p)(st)
                outs.append((y, bits))
            torch.cuda.synchronize()
            (yc_, bc), (yg_, bg) = outs
            assert rel_err(yg_.buf.float().cpu(), yc_.buf.float()) < TOL[dtype], (cin, cout, k3, v)
            if bc is not None:      # bits may differ only where the pre-activation is within rounding of zero
                assert float((bg.cpu() != bc).float().mean()) < (2e-3 if dtype == torch.float32 else 3e-2)
    # unsupported combinations are rejected on the host
    xc, xg = fmap_pair(1, 8, 1, 4, 4, torch.bfloat16, gen)
    yc, yg = fmap_pair(1, 8, 1, 4, 4, torch.bfloat16, gen)
    p = ConvPass(xg, yg, (1, 4, 4), (1, 1, 1), (1, 1, 1), (0, 0, 0), [(0, 0, 0, 0)], torch.zeros(64, dtype=torch.bfloat16, device=DEV),
                 1, 8, 8, ep=ConvEpilogue(shift=torch.zeros(8, device=DEV), relu=True))
    assert not hip.conv_epilogue_supported(p)            # ReLU needs two co fragments per wave in bf16 (cout > 16)


@pytest.mark.parametrize("shape", [(64, 64), (128, 128), (256, 64), (128, 320)], ids=str)
def test_conv_pointwise_streaming_plain_bf16(hip, shape):
    """the streaming pointwise kernel (conv_pw.hip) behind sfk_conv_igemm for the K = cout = 64 / 128 passes of the block tail's
    backward: store, +=, += with a bias; with BatchNorm partial rows (and for other shapes) the implicit GEMM runs"""
    from video_classification_amd._lib import ConvEpilogue
    cin, cout = shape
    gen = torch.Generator().manual_seed(cin + cout)
    emu = EmuBackend()
    dtype = torch.bfloat16
    n, t, h, w = 2, 3, 13, 11                                     # 858 pixels: ragged last 16-pixel tile
    one, tap0 = (1, 1, 1), [(0, 0, 0, 0)]
    wt = mk((cout * cin,), dtype, gen, scale=cin ** -0.5)
    bias = torch.randn(cout, generator=gen)
    xc, xg = fmap_pair(n, cin, t, h, w, dtype, gen, ld=cin + 8, c_off=8)
    for mode in ("stats", "plain", "acc", "acc+bias"):
        yc, yg = fmap_pair(n, cout, t, h, w, dtype, gen, ld=cout + 8, c_off=8)
        res = []
        for be, x, y, dev in ((emu, xc, yc, "cpu"), (hip, xg, yg, DEV)):
            p = ConvPass(x, y, (t, h, w), one, one, (0, 0, 0), tap0, wt.to(dev), 1, cin, cout, accumulate=mode.startswith("acc"))
            if mode == "acc+bias":
                p.ep = ConvEpilogue(shift=bias.to(dev))
            mt = 0
            if mode == "stats":
                mt = be.conv_igemm_mtiles(p)
                p.stats = torch.full((mt * cout * 2 + 64,), 5.0, device=dev)
                assert be.conv_igemm_mtiles(p) == mt                 # same answer once the pointer is set
            if be is hip:
                # (128 -> 320: the plain store of slow res3's first conv_a data gradient; its += stays on the implicit GEMM)
                assert (be.conv_family(p) == 3) == ((mode != "stats" and cin == cout) or ((cin, cout) == (128, 320) and mode == "plain"))
            be.conv_igemm(p)(stream() if dev != "cpu" else 0)
            if dev != "cpu":
                torch.cuda.synchronize()
            res.append((y.buf.float().cpu(), None if not mt else p.stats.cpu()[: mt * cout * 2].view(mt, cout, 2), mt,
                        None if not mt else p.stats.cpu()[mt * cout * 2:]))
        (yc_, sc_, _, _), (yg_, sg_, mtg, tailg) = res
        assert rel_err(yg_, yc_) < TOL[dtype], mode
        assert torch.equal(yg_.view(-1, cout + 8)[:, :8], yc_.view(-1, cout + 8)[:, :8])      # the neighbouring slice is untouched
        if mode == "stats":
            assert rel_err(sg_.sum(0), sc_.sum(0)) < 1e-4 and float(tailg.min()) == 5.0 == float(tailg.max())


@pytest.mark.parametrize("shape", [(32, 8, 8), (64, 16, 16)], ids=str)
def test_conv_pw_dual_bf16(hip, shape):
    """sfk_conv_pw_dual: both data-gradient passes of a narrow block tail in one streaming kernel, against the restated
    contract (fp32 accumulation, one rounding: 1 bf16 ulp), channel slices of wider pixel records, ragged pixel count"""
    c1, c2, co = shape
    gen = torch.Generator().manual_seed(sum(shape))
    emu = EmuBackend()
    dtype = torch.bfloat16
    n, t, h, w = 3, 3, 11, 13                                     # 1287 pixels: not a multiple of the 256-thread blocks
    x1c, x1g = fmap_pair(n, c1, t, h, w, dtype, gen, ld=c1 + 8, c_off=8)
    x2c, x2g = fmap_pair(n, c2, t, h, w, dtype, gen)
    w1 = mk((co * c1,), dtype, gen, scale=c1 ** -0.5)
    w2 = mk((co * c2,), dtype, gen, scale=c2 ** -0.5)
    bias = torch.randn(co, generator=gen)
    for use_bias in (True, False):
        yc, yg = fmap_pair(n, co, t, h, w, dtype, gen, ld=co + 8, c_off=8)
        assert hip.conv_pw_dual_supported(x1g, x2g, yg) and emu.conv_pw_dual_supported(x1c, x2c, yc)
        emu.conv_pw_dual(x1c, w1, x2c, w2, bias if use_bias else None, yc)(0)
        hip.conv_pw_dual(x1g, w1.to(DEV), x2g, w2.to(DEV), bias.to(DEV) if use_bias else None, yg)(stream())
        torch.cuda.synchronize()
        got, want = yg.buf.float().cpu(), yc.buf.float()
        assert rel_err(got, want) < TOL[dtype]
        assert torch.equal(got.view(-1, co + 8)[:, :8], want.view(-1, co + 8)[:, :8])          # the neighbouring slice is untouched
    # other channel counts keep the two sfk_conv_igemm passes
    ac, ag = fmap_pair(1, 128, 1, 4, 4, dtype, gen)
    bc, bg = fmap_pair(1, 32, 1, 4, 4, dtype, gen)
    assert not hip.conv_pw_dual_supported(ag, bg, bg)


def test_conv_masked_store_with_dz_sums_bf16(hip):
    """sfk_bn_bwd_fuse with y_bn = NULL + out_relu_bits: the data-gradient pass that finishes a block's output gradient
    stores dz = (old + result) * bitmap and leaves the per-tile partial sums of the STORED dz (what sfk_bn_tail_bwd folds)"""
    from video_classification_amd._lib import BnBwdFuse
    gen = torch.Generator().manual_seed(31)
    emu = EmuBackend()
    dtype = torch.bfloat16
    # (64 -> 256) and (128 -> 512) pointwise: the streaming kernel of conv_pw.hip (one partial row per wave), ragged pixel counts
    for (n, t, h, w, cin, cout, k) in ((2, 2, 9, 7, 16, 40, (3, 1, 1)), (1, 3, 12, 12, 64, 256, (1, 1, 1)), (2, 2, 7, 7, 8, 32, (3, 1, 1)),
                                       (3, 5, 21, 19, 64, 256, (1, 1, 1)), (2, 3, 13, 11, 128, 512, (1, 1, 1)), (1, 1, 3, 3, 128, 512, (1, 1, 1))):
        g = ConvGeom(cout, cin, k, (1, 1, 1), (k[0] // 2, 0, 0))            # forward conv cout -> cin; this is its dgrad
        passes, _ = dgrad_passes(g, (t, h, w))
        assert len(passes) == 1
        sp = passes[0]
        dyc, dyg = fmap_pair(n, cin, t, h, w, dtype, gen)
        wt = mk((cout * g.wtaps * cin,), dtype, gen, 0.2)                   # [cin_g = cout][tap][cout_g = cin] as St
        bits = torch.randint(0, 256, (n * t * h * w * cout // 8,), generator=gen, dtype=torch.uint8)
        outs = []
        for be, dy, dev, st in ((emu, dyc, "cpu", 0), (hip, dyg, DEV, stream())):
            gen2 = torch.Generator().manual_seed(5)
            dx = FMap(mk((n * t * h * w * cout,), dtype, gen2).to(dev), n, t, h, w, cout)        # the += target
            p = ConvPass(dy, dx, sp.rows, sp.gs, sp.os, sp.oo, list(sp.taps), wt.to(dev), g.wtaps, cin, cout, accumulate=True)
            assert be is emu or (be.conv_relu_out_supported(p) and be.conv_bnb_supported(p))
            parts = torch.full((max(be.conv_igemm_mtiles(p), 1024) * cout * 2,), 7.0, device=dev)
            p.relu_out_bits = bits.to(dev)
            p.bnb = BnBwdFuse(None, None, None, None, None, None, True, parts)
            mt = be.conv_igemm_mtiles(p)                 # rows of the descriptor as launched
            assert 0 < mt <= 1024
            be.conv_igemm(p)(st)
            assert float(parts[mt * cout * 2:].min()) == 7.0     # nothing written past the rows the library announced
            outs.append((dx.buf, parts[: mt * cout * 2].view(mt, cout, 2)))
        torch.cuda.synchronize()
        (xc, pc), (xg, pg) = outs
        assert rel_err(xg.float().cpu(), xc.float()) < TOL[dtype]
        assert rel_err(pg.sum(0)[:, 0].cpu(), pc.sum(0)[:, 0]) < 2e-3 and float(pg[:, :, 1].abs().max()) == 0.0
        assert rel_err(pg.sum(0)[:, 0].cpu(), xg.float().cpu().view(-1, cout).sum(0)) < 1e-3     # sums of what was stored



@pytest.mark.parametrize("dims", [(2, 3, 13, 11), (1, 2, 56, 56), (1, 1, 3, 3)], ids=str)
def test_filter_gradient_with_fused_data_gradient_bf16(hip, dims):
    """sfk_wgrad_desc.dg_w / dg_y: R = dY^T X (256 x 64) and the data gradient dY dg_w^T from ONE pass over dY -- the idle column
    waves of the LDS-DMA filter-gradient tile; both results against the CPU restatement, with atomics and with the workspace"""
    n, t, h, w = dims
    cin, cout = 64, 256
    gen = torch.Generator().manual_seed(sum(dims))
    emu = EmuBackend()
    dtype = torch.bfloat16
    xc, xg = fmap_pair(n, cin, t, h, w, dtype, gen, ld=cin + 8, c_off=0)
    dyc, dyg = fmap_pair(n, cout, t, h, w, dtype, gen)
    wd = mk((cin * cout,), dtype, gen, scale=cout ** -0.5)
    res = []
    for be, x, dy, dev, use_ws in ((emu, xc, dyc, "cpu", False), (hip, xg, dyg, DEV, False), (hip, xg, dyg, DEV, True)):
        dw = torch.ones(cout * cin, device=dev)                      # accumulated into
        out = FMap(torch.full((n * t * h * w * (cin + 8),), 5.0, dtype=dtype, device=dev), n, t, h, w, cin, cin + 8, 8)
        p = WgradPass(x, dy, (1, 1, 1), [(0, 0, 0, 0)], dw, 1, cin, cout, dg_w=wd.to(dev), dg_y=out)
        assert be.conv_wgrad_dg_supported(p)
        if use_ws:
            p.workspace = torch.zeros(be.conv_wgrad_workspace_bytes(p) // 4 + 4, device=dev)
        be.conv_wgrad(p)(stream() if dev != "cpu" else 0)
        if dev != "cpu":
            torch.cuda.synchronize()
        res.append((dw.cpu(), out.buf.float().cpu()))
    for dwg, og in res[1:]:
        assert rel_err(dwg - 1.0, res[0][0] - 1.0) < 2e-3
        assert rel_err(og, res[0][1]) < TOL[dtype]
        assert torch.equal(og.view(-1, cin + 8)[:, :8], res[0][1].view(-1, cin + 8)[:, :8])      # the neighbouring slice is untouched
    # other shapes are announced as unsupported, and a descriptor that asks for the fusion there is refused
    x2c, x2g = fmap_pair(n, 32, t, h, w, dtype, gen)
    p = WgradPass(x2g, dyg, (1, 1, 1), [(0, 0, 0, 0)], torch.zeros(cout * 32, device=DEV), 1, 32, cout,
                  dg_w=wd.to(DEV), dg_y=FMap(torch.zeros(n * t * h * w * 32, dtype=dtype, device=DEV), n, t, h, w, 32))
    assert not hip.conv_wgrad_dg_supported(p)
    with pytest.raises(Exception):
        hip.conv_wgrad(p)(stream())

"""plan.py (conv geometry -> implicit-GEMM passes) against torch's conv3d and its autograd, through the
C-ABI contract restated in tests/emu_backend.py.  CPU only."""
import pytest
import torch

from emu_backend import EmuBackend
from helpers import empty_fmap, from_fmap, to_fmap
from video_classification_amd._lib import ConvPass, WgradPass
from video_classification_amd.plan import ConvGeom, dgrad_passes, fwd_pass, wgrad_taps

GEOMS = [
    # (cin, cout, k, s, p, dims)
    (8, 16, (1, 1, 1), (1, 1, 1), (0, 0, 0), (3, 5, 6)),        # K1 pointwise
    (16, 8, (1, 1, 1), (1, 2, 2), (0, 0, 0), (2, 6, 6)),        # K2 strided shortcut (even extent)
    (16, 8, (1, 1, 1), (1, 2, 2), (0, 0, 0), (2, 7, 5)),        # K2 strided shortcut (odd extent)
    (8, 8, (3, 1, 1), (1, 1, 1), (1, 0, 0), (5, 3, 4)),         # K3 temporal
    (8, 12, (1, 3, 3), (1, 1, 1), (0, 1, 1), (2, 5, 6)),        # K4 spatial
    (8, 12, (1, 3, 3), (1, 2, 2), (0, 1, 1), (2, 8, 6)),        # K4 spatial stride 2 (even)
    (8, 12, (1, 3, 3), (1, 2, 2), (0, 1, 1), (2, 7, 9)),        # K4 spatial stride 2 (odd)
    (8, 16, (7, 1, 1), (4, 1, 1), (3, 0, 0), (16, 3, 3)),       # K6 canonical lateral fusion
    (8, 16, (5, 1, 1), (1, 1, 1), (2, 0, 0), (6, 3, 3)),        # fast stem temporal part
]


def _ref(cin, cout, k, s, p, dims, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(2, cin, *dims, generator=g, requires_grad=True)
    w = torch.randn(cout, cin, *k, generator=g, requires_grad=True)
    y = torch.nn.functional.conv3d(x, w, None, s, p)
    gy = torch.randn(y.shape, generator=g)
    y.backward(gy)
    return x, w, y.detach(), gy, x.grad, w.grad


def _engine_w(w):  # (co,ci,kt,kh,kw) -> flat [co][tap][ci]
    return w.detach().permute(0, 2, 3, 4, 1).reshape(-1).contiguous()


@pytest.mark.parametrize("cin,cout,k,s,p,dims", GEOMS)
def test_forward_pass(cin, cout, k, s, p, dims):
    be = EmuBackend()
    x, w, y, *_ = _ref(cin, cout, k, s, p, dims)
    g = ConvGeom(cin, cout, k, s, p)
    sp = fwd_pass(g, dims)
    assert sp.rows == tuple(y.shape[2:])
    fx = to_fmap(x.detach())
    # write into a channel slice of a wider buffer (the concat-elimination path)
    fy = empty_fmap(2, cout, *sp.rows, ld=cout + 8, c_off=4, fill=7.0)
    ps = ConvPass(fx, fy, sp.rows, sp.gs, sp.os, sp.oo, list(sp.taps), _engine_w(w), g.wtaps, cin, cout)
    mt = be.conv_igemm_mtiles(ps)
    ps.stats = torch.zeros(mt * cout * 2)
    be.conv_igemm(ps)(0)
    assert torch.allclose(from_fmap(fy), y, atol=1e-4, rtol=1e-4)
    wide = fy.buf.view(-1, cout + 8)
    assert torch.all(wide[:, :4] == 7.0) and torch.all(wide[:, 4 + cout:] == 7.0)   # neighbours untouched
    st = ps.stats.view(mt, cout, 2).sum(0)
    assert torch.allclose(st[:, 0], y.sum((0, 2, 3, 4)), atol=1e-3, rtol=1e-4)
    assert torch.allclose(st[:, 1], (y * y).sum((0, 2, 3, 4)), atol=1e-3, rtol=1e-4)


@pytest.mark.parametrize("cin,cout,k,s,p,dims", GEOMS)
def test_dgrad_passes(cin, cout, k, s, p, dims):
    be = EmuBackend()
    x, w, y, gy, gx, gw = _ref(cin, cout, k, s, p, dims)
    g = ConvGeom(cin, cout, k, s, p)
    passes, needs_zero = dgrad_passes(g, dims)
    wt = w.detach().permute(1, 2, 3, 4, 0).reshape(-1).contiguous()      # [ci][tap][co]
    fdy = to_fmap(gy)
    base = torch.randn(2, cin, *dims)
    for accumulate in (False, True):
        fdx = to_fmap(base) if accumulate else empty_fmap(2, cin, *dims, fill=float("nan") if not needs_zero else 0.0)
        for sp in passes:
            be.conv_igemm(ConvPass(fdy, fdx, sp.rows, sp.gs, sp.os, sp.oo, list(sp.taps), wt, g.wtaps, cout, cin,
                                   accumulate=accumulate))(0)
        want = gx + base if accumulate else gx
        assert torch.allclose(from_fmap(fdx), want, atol=1e-4, rtol=1e-4)
    # every input pixel is covered by exactly one class unless the class has no taps
    if not needs_zero:
        cover = torch.zeros(dims)
        for sp in passes:
            sl = tuple(slice(o, o + (r - 1) * st + 1, st) for o, st, r in zip(sp.oo, sp.os, sp.rows))
            cover[sl] += 1
        assert torch.all(cover == 1)


@pytest.mark.parametrize("cin,cout,k,s,p,dims", GEOMS)
def test_wgrad_pass(cin, cout, k, s, p, dims):
    be = EmuBackend()
    x, w, y, gy, gx, gw = _ref(cin, cout, k, s, p, dims)
    g = ConvGeom(cin, cout, k, s, p)
    dw = torch.zeros(cout * g.wtaps * cin)
    be.conv_wgrad(WgradPass(to_fmap(x.detach()), to_fmap(gy), g.s, list(wgrad_taps(g)), dw, g.wtaps, cin, cout))(0)
    got = dw.view(cout, *k, cin).permute(0, 4, 1, 2, 3)
    assert torch.allclose(got, gw, atol=1e-3, rtol=1e-4)


def test_tap_tables_fit_the_abi():
    from video_classification_amd._lib import SFK_MAX_TAPS
    for cin, cout, k, s, p, dims in GEOMS:
        g = ConvGeom(cin, cout, k, s, p)
        assert len(fwd_pass(g, dims).taps) <= SFK_MAX_TAPS
        for sp in dgrad_passes(g, dims)[0]:
            assert 0 < len(sp.taps) <= SFK_MAX_TAPS
            assert all(-128 <= d <= 127 for t in sp.taps for d in t[:3])

"""The bottleneck tail without the conv output in HBM (include/sfk.h: sfk_conv_epilogue, sfk_bn_tail_fwd/bwd): the algebra,
on the CPU restatement of the C-ABI contract (tests/emu_backend.py), against torch autograd of the literal chain
    a -> Conv3d 1x1x1 (bias=False) -> BatchNorm3d (train) -> + shortcut -> ReLU
(pytorchvideo ResBlock / BottleneckBlock tail; reference construction model/my_slowfast.py:94-125, autograd train.py:230).
The same call sequence runs on the GPU in tests/test_gpu_kernels.py::test_bottleneck_tail_*."""
import pytest
import torch

from emu_backend import EmuBackend
from helpers import rel_err
from video_classification_amd._lib import ConvEpilogue, ConvPass, FMap, WgradPass

ONE = (1, 1, 1)
TAP0 = [(0, 0, 0, 0)]


def tail_reference(a, W, gamma, beta, res, gout, eps=1e-5):
    """torch autograd of the literal chain; a (n,t,h,w,c), W (cout,c), res/gout (n,t,h,w,cout), all float64"""
    a = a.clone().requires_grad_(True)
    W = W.clone().requires_grad_(True)
    gamma = gamma.clone().requires_grad_(True)
    beta = beta.clone().requires_grad_(True)
    res = res.clone().requires_grad_(True)
    y = a @ W.t()
    mu = y.mean((0, 1, 2, 3))
    var = y.var((0, 1, 2, 3), unbiased=False)
    z = (y - mu) / torch.sqrt(var + eps) * gamma + beta
    out = torch.relu(z + res)
    (out * gout).sum().backward()
    n = y[..., 0].numel()
    return dict(out=out.detach(), mean=mu.detach(), var=var.detach(), unb=var.detach() * n / (n - 1), da=a.grad, dW=W.grad,
                dgamma=gamma.grad, dbeta=beta.grad, dres=res.grad)


def run_tail(be, dev, dtype, a, W, gamma, beta, res, gout, stream=0, eps=1e-5, momentum=0.1, res_affine=None):
    """The call sequence of the engine for one block tail; returns what tail_reference returns (fp32, on the CPU)."""
    n, t, h, w, c = a.shape
    cout = W.shape[0]
    vec = 8 if dtype == torch.bfloat16 else 4
    f32 = lambda *s: torch.zeros(*s, device=dev)
    av = FMap(a.to(dtype).to(dev).reshape(-1).contiguous(), n, t, h, w, c)
    wq = W.to(dtype).to(dev).reshape(-1).contiguous()                  # compute-precision filter [cout][c]
    g_, b_ = gamma.float().to(dev), beta.float().to(dev)
    rmap = FMap(res.to(dtype).to(dev).reshape(-1).contiguous(), n, t, h, w, cout)
    out = FMap(torch.zeros(n * t * h * w * cout, dtype=dtype, device=dev), n, t, h, w, cout)
    bits = torch.zeros(out.pixels * (cout // vec), dtype=torch.uint8, device=dev)
    # ---- forward: G = a^T a (one filter-gradient call), the column sums of a as partial rows (the engine gets them from the
    # sfk_bn_apply pass that writes a; here a is given, so a statistics pass leaves the same rows: component 0 = sum)
    gram = f32(c * c)
    be.conv_wgrad(WgradPass(av, av, ONE, TAP0, gram, 1, c, c))(stream)
    asums = f32(1024 * c * 2)
    run, a_np = be.bn_stats(av, asums, 1024)
    run(stream)
    gvec = f32(c)
    mean, invstd, scale, shift, T = f32(cout), f32(cout), f32(cout), f32(cout), f32(cout * c)
    rm, rv, nbt = f32(cout), torch.ones(cout, device=dev), torch.zeros(1, dtype=torch.int64, device=dev)
    wd = torch.zeros(cout * c, dtype=dtype, device=dev)                # (A W)^T: known once the statistics are
    be.bn_tail_fwd(gram, asums, a_np, av.pixels, gvec, c, wq, cout, g_, b_, eps, momentum, rm, rv, nbt, mean, invstd, scale, shift,
                   T, wd)(stream)
    ep = ConvEpilogue(scale=scale, shift=shift, res=rmap, relu=True, relu_bits=bits)
    if res_affine is not None:
        ep.res_scale, ep.res_shift = res_affine
    fwd = ConvPass(av, out, (t, h, w), ONE, ONE, (0, 0, 0), TAP0, wq, 1, c, cout, ep=ep)
    assert be.conv_epilogue_supported(fwd)
    be.conv_igemm(fwd)(stream)
    # ---- backward of sum(out * gout)
    dz = FMap(gout.to(dtype).to(dev).reshape(-1).contiguous(), n, t, h, w, cout)
    parts = f32(1024 * cout * 2)
    run, nparts = be.bn_bwd_reduce(dz, None, None, None, None, None, None, True, dz, parts, 1024, relu_bits=bits)
    run(stream)                                                          # dz *= mask (in place) and partial sums of dz
    r = f32(cout * c)
    be.conv_wgrad(WgradPass(av, dz, ONE, TAP0, r, 1, c, cout))(stream)
    dgamma, dbeta, dw, bias, coef = f32(cout), f32(cout), f32(cout * c), f32(c), f32(cout * 4)
    m = torch.zeros(c * c, dtype=dtype, device=dev)
    be.bn_tail_bwd(r, parts, nparts, gvec, av.pixels, T, c, wq, cout, g_, mean, invstd, dgamma, dbeta, dw, m, bias, coef)(stream)
    da = FMap(torch.zeros(n * t * h * w * c, dtype=dtype, device=dev), n, t, h, w, c)
    be.conv_igemm(ConvPass(dz, da, (t, h, w), ONE, ONE, (0, 0, 0), TAP0, wd, 1, cout, c))(stream)
    be.conv_igemm(ConvPass(av, da, (t, h, w), ONE, ONE, (0, 0, 0), TAP0, m, 1, c, c, accumulate=True,
                           ep=ConvEpilogue(shift=bias)))(stream)
    if dev != "cpu":
        torch.cuda.synchronize()
    cpu = lambda x: x.float().cpu()
    return dict(out=cpu(out.view5()), mean=cpu(mean), var=cpu(1.0 / invstd ** 2 - eps), rm=cpu(rm), rv=cpu(rv), nbt=int(nbt[0]),
                da=cpu(da.view5()), dW=cpu(dw).view(cout, c), dgamma=cpu(dgamma), dbeta=cpu(dbeta), dz=cpu(dz.view5()),
                m=cpu(m), bias=cpu(bias))


def make_case(n, t, h, w, c, cout, seed=0, mean_shift=0.0):
    g = torch.Generator().manual_seed(seed)
    a = torch.relu(torch.randn(n, t, h, w, c, generator=g, dtype=torch.float64) + mean_shift)    # post-ReLU activations
    W = torch.randn(cout, c, generator=g, dtype=torch.float64) * (2.0 / c) ** 0.5
    gamma = torch.rand(cout, generator=g, dtype=torch.float64) + 0.5
    beta = torch.randn(cout, generator=g, dtype=torch.float64) * 0.2
    res = torch.relu(torch.randn(n, t, h, w, cout, generator=g, dtype=torch.float64))
    gout = torch.randn(n, t, h, w, cout, generator=g, dtype=torch.float64)
    return a, W, gamma, beta, res, gout


@pytest.mark.parametrize("c,cout", [(8, 32), (16, 64), (64, 256)])
def test_tail_algebra_equals_autograd_fp32(c, cout):
    case = make_case(2, 2, 5, 6, c, cout, seed=c)
    ref = tail_reference(*case)
    got = run_tail(EmuBackend(), "cpu", torch.float32, *case)
    assert rel_err(got["out"], ref["out"].float()) < 1e-5
    assert rel_err(got["mean"], ref["mean"].float()) < 1e-5 and rel_err(got["var"], ref["var"].float()) < 1e-4
    assert rel_err(got["rm"], 0.1 * ref["mean"].float()) < 1e-5
    assert rel_err(got["rv"], 0.9 + 0.1 * ref["unb"].float()) < 1e-4 and got["nbt"] == 1
    assert rel_err(got["dz"], ref["dres"].float()) < 1e-6             # the shortcut's gradient is dz itself
    for k in ("dgamma", "dbeta", "dW", "da"):
        assert rel_err(got[k], ref[k].float()) < 2e-4, k


def test_tail_algebra_with_a_large_channel_mean():
    """var = E[y^2] - E[y]^2 from the Gram matrix: still right when the mean dominates (post-ReLU inputs shifted up)"""
    case = make_case(2, 2, 6, 6, 16, 32, seed=3, mean_shift=3.0)
    ref = tail_reference(*case)
    got = run_tail(EmuBackend(), "cpu", torch.float32, *case)
    assert rel_err(got["var"], ref["var"].float()) < 1e-3
    assert rel_err(got["out"], ref["out"].float()) < 1e-4
    for k in ("dgamma", "dbeta", "dW", "da"):
        assert rel_err(got[k], ref[k].float()) < 1e-3, k


def test_tail_projection_shortcut_affine():
    """first block of a stage: shortcut = branch1_norm(branch1_conv(x)) enters as res * res_scale + res_shift"""
    case = make_case(1, 2, 4, 4, 8, 32, seed=9)
    a, W, gamma, beta, res, gout = case
    rs = torch.rand(32, dtype=torch.float64) + 0.5
    rh = torch.randn(32, dtype=torch.float64) * 0.1
    ref = tail_reference(a, W, gamma, beta, res * rs + rh, gout)
    got = run_tail(EmuBackend(), "cpu", torch.float32, *case, res_affine=(rs.float(), rh.float()))
    assert rel_err(got["out"], ref["out"].float()) < 1e-5
    for k in ("dgamma", "dbeta", "dW", "da"):
        assert rel_err(got[k], ref[k].float()) < 2e-4, k


@pytest.mark.parametrize("c,cout", [(8, 32), (16, 64)])
def test_dual_pass_equals_the_two_data_gradient_passes(c, cout):
    """sfk_conv_pw_dual's contract: y = x1 w1^T + x2 w2^T + bias == the plain pass followed by the += pass with a bias, up to
    the one bf16 rounding the two-pass form does in between (tests/emu_backend.py states the contract; the GPU kernel is
    checked against it in tests/test_gpu_kernels.py)"""
    be = EmuBackend()
    g = torch.Generator().manual_seed(c)
    bf = torch.bfloat16
    n, t, h, w = 2, 2, 5, 7
    dz = FMap((torch.randn(n * t * h * w * cout, generator=g)).to(bf), n, t, h, w, cout)
    a = FMap(torch.relu(torch.randn(n * t * h * w * c, generator=g)).to(bf), n, t, h, w, c)
    wd = (torch.randn(c * cout, generator=g) * cout ** -0.5).to(bf)
    m = (torch.randn(c * c, generator=g) * c ** -0.5).to(bf)
    bias = torch.randn(c, generator=g)
    assert be.conv_pw_dual_supported(dz, a, FMap(torch.zeros(n * t * h * w * c, dtype=bf), n, t, h, w, c))
    y1 = FMap(torch.zeros(n * t * h * w * c, dtype=bf), n, t, h, w, c)
    be.conv_pw_dual(dz, wd, a, m, bias, y1)(0)
    y2 = FMap(torch.zeros(n * t * h * w * c, dtype=bf), n, t, h, w, c)
    be.conv_igemm(ConvPass(dz, y2, (t, h, w), ONE, ONE, (0, 0, 0), TAP0, wd, 1, cout, c))(0)
    be.conv_igemm(ConvPass(a, y2, (t, h, w), ONE, ONE, (0, 0, 0), TAP0, m, 1, c, c, accumulate=True, ep=ConvEpilogue(shift=bias)))(0)
    assert rel_err(y1.buf.float(), y2.buf.float()) < 1e-2          # (bf16: the two-pass form rounds the first pass's result)
    ref = dz.view5().double() @ wd.view(c, cout).double().t() + a.view5().double() @ m.view(c, c).double().t() + bias.double()
    assert rel_err(y1.view5().float(), ref.float()) < 4e-3

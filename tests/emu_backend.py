"""TEST INFRASTRUCTURE: a torch-CPU restatement of the C-ABI contract of include/sfk.h.

Same method surface as video_classification_amd._lib.HipBackend, so
  * on CPU the planner (plan.py) and the whole engine schedule (engine.py) are checked against the oracle's
    autograd without a GPU, and
  * on the GPU box every HIP entry point is compared with this restatement on identical inputs.
It is never imported by the product package.
"""
from __future__ import annotations

import numpy as np
import torch

from video_classification_amd._lib import ConvPass, FMap, StemSrc, WgradPass, stem_kp


def _tile_bm(cout: int) -> int:
    return 128 if cout > 64 else 256


def _gather(X: torch.Tensor, rows, gs, tap):
    """X (N,T,H,W,C) float -> (N,rt,rh,rw,C) gathered at r*gs + d with zero padding."""
    idx = []
    mask = None
    for axis, (r, g, d, ext) in enumerate(zip(rows, gs, tap[:3], X.shape[1:4])):
        i = torch.arange(r) * g + d
        ok = (i >= 0) & (i < ext)
        idx.append(i.clamp(0, ext - 1))
        shape = [1, 1, 1, 1, 1]
        shape[axis + 1] = r
        m = ok.view(shape)
        mask = m if mask is None else (mask & m)
    out = X[:, idx[0]][:, :, idx[1]][:, :, :, idx[2]]
    return out * mask.to(out.dtype)


_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(z: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (z + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def keep_mask(seed: int, n: int, c: int, f_off: int, positions: int, rate: float) -> np.ndarray:
    """keep[n][c][p] exactly as csrc/pool_head.hip::keep_of."""
    if rate <= 0:
        return np.ones((n, c, positions), dtype=bool)
    ni = np.arange(n, dtype=np.uint64).reshape(n, 1, 1)
    fi = (np.arange(c, dtype=np.uint64) + np.uint64(f_off)).reshape(1, c, 1)
    pi = np.arange(positions, dtype=np.uint64).reshape(1, 1, positions)
    key = (ni << np.uint64(40)) ^ (fi << np.uint64(20)) ^ pi
    r = _splitmix64(np.uint64(seed & 0xFFFFFFFFFFFFFFFF) ^ _splitmix64(key)) >> np.uint64(40)
    u = r.astype(np.float32) * np.float32(1.0 / 16777216.0)
    return u >= np.float32(rate)


class EmuBackend:
    name = "emu"

    # ------------------------------------------------------------------ convolution
    def conv_igemm_mtiles(self, p: ConvPass) -> int:
        m = p.x.n * p.rows[0] * p.rows[1] * p.rows[2]
        bm = _tile_bm(p.cout)
        return (m + bm - 1) // bm

    def conv_igemm(self, p: ConvPass):
        def run(stream):
            X = p.x.view5().float()
            Wf = p.w[: p.cout * p.wtaps * p.cin].view(p.cout, p.wtaps, p.cin).float()
            acc = torch.zeros(p.x.n, *p.rows, p.cout)
            for tap in p.taps:
                acc += torch.einsum("nthwc,oc->nthwo", _gather(X, p.rows, p.gs, tap), Wf[:, tap[3], :])
            Y = p.y.view5()
            sl = tuple(slice(o, o + (r - 1) * s + 1, s) for o, s, r in zip(p.oo, p.os, p.rows))
            dest = Y[:, sl[0], sl[1], sl[2]]
            ep = getattr(p, "ep", None)
            if ep is not None:                    # sfk_conv_epilogue: scale / shift, (+ old), shortcut, ReLU (+ bitmap)
                v = acc
                if ep.scale is not None:
                    v = v * ep.scale[: p.cout].float()
                if ep.shift is not None:
                    v = v + ep.shift[: p.cout].float()
                if p.accumulate:
                    v = v + dest.float()
                if ep.res is not None:
                    r = ep.res.view5().float()
                    if ep.res_scale is not None:
                        r = r * ep.res_scale[: p.cout].float()
                    if ep.res_shift is not None:
                        r = r + ep.res_shift[: p.cout].float()
                    v = v + r
                if ep.relu:
                    if ep.relu_bits is not None:
                        vec = 8 if Y.dtype == torch.bfloat16 else 4
                        m = (v > 0).reshape(-1, p.cout // vec, vec).to(torch.int32)
                        packed = (m << torch.arange(vec, dtype=torch.int32)).sum(-1).to(torch.uint8)
                        ep.relu_bits[: packed.numel()].copy_(packed.reshape(-1))
                    v = torch.relu(v)
                dest.copy_(v.to(Y.dtype))
                return
            res = dest.float() + acc if p.accumulate else acc
            if p.relu_out_bits is not None:       # Y is the gradient w.r.t. a ReLU output: store result * mask
                vec = 8 if Y.dtype == torch.bfloat16 else 4
                b = p.relu_out_bits[: p.y.pixels * (p.cout // vec)].to(torch.int32).reshape(-1, p.cout // vec, 1)
                m = ((b >> torch.arange(vec, dtype=torch.int32, device=b.device)) & 1).reshape(res.shape)
                res = res * m.float()
            dest.copy_(res.to(Y.dtype))
            if p.bnb is not None and p.bnb.y_bn is None:     # bitmap mask + per-tile partial sums of the STORED dz
                flat = dest.float().reshape(-1, p.cout)
                bm = _tile_bm(p.cout)
                mt = (flat.shape[0] + bm - 1) // bm
                st = p.bnb.partials[: mt * p.cout * 2].view(mt, p.cout, 2)
                for i in range(mt):
                    st[i, :, 0] = flat[i * bm:(i + 1) * bm].sum(0)
                    st[i, :, 1] = 0
            if p.stats is not None:
                flat = acc.reshape(-1, p.cout)
                bm = _tile_bm(p.cout)
                mt = (flat.shape[0] + bm - 1) // bm
                st = p.stats[: mt * p.cout * 2].view(mt, p.cout, 2)
                for i in range(mt):
                    blk = flat[i * bm:(i + 1) * bm]
                    st[i, :, 0] = blk.sum(0)
                    st[i, :, 1] = (blk * blk).sum(0)
        return run

    def conv_wgrad_workspace_bytes(self, p) -> int:
        return 0

    def conv_bnb_supported(self, p) -> bool:
        return False          # the emulation keeps the stand-alone BatchNorm-backward reduce

    def conv_relu_out_supported(self, p) -> bool:
        return tuple(p.os) == (1, 1, 1) and tuple(p.oo) == (0, 0, 0) and tuple(p.rows) == (p.y.t, p.y.h, p.y.w)

    def conv_epilogue_supported(self, p) -> bool:
        return tuple(p.os) == (1, 1, 1) and tuple(p.oo) == (0, 0, 0) and tuple(p.rows) == (p.y.t, p.y.h, p.y.w)

    # ------------------------------------------------------------------ bottleneck tail (sfk_bn_tail_*, sfk_relu_bits_mask)
    def bn_tail_fwd(self, gram, a_sums, a_nparts, count, g, c, w, cout, gamma, beta, eps, momentum, rm, rv, nbt, mean, invstd,
                    scale, shift, t, wd=None):
        def run(stream):
            G = gram[: c * c].view(c, c).double()
            gs = a_sums[: a_nparts * c * 2].view(a_nparts, c, 2)[:, :, 0].double().sum(0)
            g[:c].copy_(gs.float())
            g_, n = gs, float(count)
            W = w[: cout * c].view(cout, c).double()
            T = W @ G
            t[: cout * c].copy_(T.float().reshape(-1))
            mu = (W @ g_) / n
            var = ((T * W).sum(1) / n - mu * mu).clamp_min(0.0)
            is_ = 1.0 / torch.sqrt(var + eps)
            mean[:cout].copy_(mu.float())
            invstd[:cout].copy_(is_.float())
            sc = gamma[:cout].double() * is_
            scale[:cout].copy_(sc.float())
            shift[:cout].copy_((beta[:cout].double() - mu * sc).float())
            if wd is not None:
                wd[: cout * c].copy_((sc[:, None] * W).t().contiguous().reshape(-1).to(wd.dtype))
            if rm is not None:
                unb = var * n / (n - 1.0) if n > 1 else var
                rm[:cout].mul_(1 - momentum).add_(momentum * mu.float())
                rv[:cout].mul_(1 - momentum).add_(momentum * unb.float())
            if nbt is not None:
                nbt.add_(1)
        return run

    def bn_tail_bwd(self, r, dz_partials, nparts, g_in, count, t, c, w, cout, gamma, mean, invstd, dgamma, dbeta, dw, m,
                    bias, coef):
        def run(stream):
            R = r[: cout * c].view(cout, c).double()
            s_ = dz_partials[: nparts * cout * 2].view(nparts, cout, 2)[:, :, 0].double().sum(0)
            g, n = g_in[:c].double(), float(count)
            W = w[: cout * c].view(cout, c).double()
            T = t[: cout * c].view(cout, c).double()
            is_, mu = invstd[:cout].double(), mean[:cout].double()
            sxh = is_ * ((W * R).sum(1) - mu * s_)
            dgamma[:cout].add_(sxh.float())
            dbeta[:cout].add_(s_.float())
            A = gamma[:cout].double() * is_
            c1, c2 = s_ / n, sxh / n
            B = -A * c2 * is_
            Cc = A * (c2 * is_ * mu - c1)
            dw[: cout * c].add_((A[:, None] * R + B[:, None] * T + Cc[:, None] * g[None, :]).float().reshape(-1))
            m[: c * c].copy_((W.t() @ (B[:, None] * W)).reshape(-1).to(m.dtype))
            bias[:c].copy_((Cc @ W).float())
        return run

    @staticmethod
    def conv_pw_dual_supported(x1, x2, y) -> bool:
        return y.buf.dtype == torch.bfloat16 and (x1.c, x2.c, y.c) in ((32, 8, 8), (64, 16, 16))

    def conv_pw_dual(self, x1, w1, x2, w2, bias, y):
        """y = x1 w1^T + x2 w2^T + bias (include/sfk.h sfk_conv_pw_dual): fp32 accumulation, one rounding"""
        def run(stream):
            W1 = w1[: y.c * x1.c].view(y.c, x1.c).float()
            W2 = w2[: y.c * x2.c].view(y.c, x2.c).float()
            out = x1.view5().float() @ W1.t() + x2.view5().float() @ W2.t()
            if bias is not None:
                out = out + bias[: y.c].float()
            y.view5().copy_(out.to(y.buf.dtype))
        return run

    def conv_wgrad(self, p: WgradPass):
        def run(stream):
            X = p.x.view5().float()
            dY = p.dy.view5().float()
            rows = (p.dy.t, p.dy.h, p.dy.w)
            dw = p.dw[: p.cout * p.wtaps * p.cin].view(p.cout, p.wtaps, p.cin)
            for tap in p.taps:
                dw[:, tap[3], :] += torch.einsum("nthwo,nthwc->oc", dY, _gather(X, rows, p.gs, tap))
            if p.dg_w is not None:          # fused data gradient of the same dY (include/sfk.h sfk_wgrad_desc.dg_w / dg_y)
                Wd = p.dg_w[: p.cin * p.cout].view(p.cin, p.cout).float()
                p.dg_y.view5().copy_((dY @ Wd.t()).to(p.dg_y.dtype))
        return run

    def conv_wgrad_dg_supported(self, p: WgradPass) -> bool:
        return (p.dg_w is not None and p.x.dtype == torch.bfloat16 and p.cout == 256 and p.cin == 64 and len(p.taps) == 1
                and tuple(p.taps[0][:3]) == (0, 0, 0) and tuple(p.gs) == (1, 1, 1))

    @staticmethod
    def _stem_x(p: StemSrc, dtype):
        x = p.src.to(dtype).float()          # the clip is rounded to the compute precision when the patch is staged
        if p.t_index is not None:
            x = x.index_select(2, p.t_index.long())
        return x

    @staticmethod
    def stem_weight_from_layout(w, cout, cin, kt):
        """[co][((f*cin+ci)*7+kh)*8+kw] -> (co, ci, kt, 7, 7)"""
        kp = stem_kp(cin, kt)
        w = w[: cout * kp].view(cout, kp)[:, : kt * cin * 7 * 8].view(cout, kt, cin, 7, 8)[..., :7]
        return w.permute(0, 2, 1, 3, 4).float()

    def stem_conv_tiles(self, p: StemSrc, y: FMap) -> int:
        return y.n * y.t * ((y.h + 15) // 16) * ((y.w + 15) // 16)

    def stem_conv_fwd(self, p: StemSrc, w, y: FMap, stats):
        def run(stream):
            x = self._stem_x(p, y.dtype)
            wt = self.stem_weight_from_layout(w, y.c, x.shape[1], p.kt)
            out = torch.nn.functional.conv3d(x, wt, None, (1, 2, 2), (p.kt // 2, 3, 3))   # n co t ho wo
            y.view5().copy_(out.permute(0, 2, 3, 4, 1).to(y.dtype))
            if stats is not None:
                mt = self.stem_conv_tiles(p, y)
                st = stats[: mt * y.c * 2].view(mt, y.c, 2)
                st.zero_()
                st[0, :, 0] = out.sum((0, 2, 3, 4))
                st[0, :, 1] = (out * out).sum((0, 2, 3, 4))
        return run

    def stem_conv_wgrad(self, p: StemSrc, dy: FMap, dw):
        def run(stream):
            x = self._stem_x(p, dy.dtype)
            cin, cout, kt = x.shape[1], dy.c, p.kt
            with torch.enable_grad():        # the engine's backward runs inside autograd's no_grad region
                wt = torch.zeros(cout, cin, kt, 7, 7, requires_grad=True)
                out = torch.nn.functional.conv3d(x.detach(), wt, None, (1, 2, 2), (kt // 2, 3, 3))
                out.backward(dy.view5().float().permute(0, 4, 1, 2, 3))
            kp = stem_kp(cin, kt)
            g = torch.nn.functional.pad(wt.grad.permute(0, 2, 1, 3, 4), (0, 1)).reshape(cout, kt * cin * 7 * 8)
            dw[: cout * kp].view(cout, kp)[:, : g.shape[1]] += g
        return run

    # ------------------------------------------------------------------ batch norm
    def bn_finalize(self, partials, nparts, c, count, gamma, beta, eps, momentum, rm, rv, nbt, mean, invstd, scale,
                    shift, workspace=None):
        def run(stream):
            pt = partials[: nparts * c * 2].view(nparts, c, 2).double().sum(0)
            mu = pt[:, 0] / count
            var = (pt[:, 1] / count - mu * mu).clamp_min(0)
            is_ = (1.0 / torch.sqrt(var + eps)).float()
            mean[:c] = mu.float()
            invstd[:c] = is_
            scale[:c] = gamma[:c] * is_
            shift[:c] = beta[:c] - mu.float() * scale[:c]
            if rm is not None:
                unb = var * count / (count - 1) if count > 1 else var
                rm[:c] = (1 - momentum) * rm[:c] + momentum * mu.float()
                rv[:c] = (1 - momentum) * rv[:c] + momentum * unb.float()
            if nbt is not None:
                nbt.add_(1)
        return run

    def bn_finalize_apply(self, partials, nparts, count, gamma, beta, eps, momentum, rm, rv, nbt, mean, invstd, workspace, sync, y,
                          scale, shift, res, res_scale, res_shift, relu, out, relu_bits=None):
        """the fused launch = the two stand-alone steps (sfk_bn_finalize_apply)"""
        fin = self.bn_finalize(partials, nparts, y.c, count, gamma, beta, eps, momentum, rm, rv, nbt, mean, invstd, scale, shift, workspace)
        app = self.bn_apply(y, scale, shift, res, res_scale, res_shift, relu, out, relu_bits=relu_bits)

        def run(stream):
            fin(stream)
            app(stream)
        return run

    def bn_bwd_finalize_apply(self, partials, nparts, count, gamma, dgamma, dbeta, coef, workspace, sync, da, y, mask_src, mean, invstd,
                              scale, shift, relu, dy):
        fin = self.bn_bwd_finalize(partials, nparts, y.c, count, gamma, invstd, dgamma, dbeta, coef, workspace)
        app = self.bn_bwd_apply(da, y, mask_src, mean, invstd, scale, shift, relu, coef, dy)

        def run(stream):
            fin(stream)
            app(stream)
        return run

    def bn_eval_coeffs(self, gamma, beta, rm, rv, eps, c, scale, shift):
        def run(stream):
            is_ = 1.0 / torch.sqrt(rv[:c] + eps)
            scale[:c] = gamma[:c] * is_
            shift[:c] = beta[:c] - rm[:c] * scale[:c]
        return run

    def bn_stats(self, y: FMap, partials, max_parts):
        def run(stream):
            v = y.view5().float().reshape(-1, y.c)
            pt = partials[: y.c * 2].view(1, y.c, 2)
            pt[0, :, 0] = v.sum(0)
            pt[0, :, 1] = (v * v).sum(0)
        return run, 1

    @staticmethod
    def _vec(y):
        return 8 if y.dtype == torch.bfloat16 else 4

    def bn_apply(self, y, scale, shift, res, res_scale, res_shift, relu, out, relu_bits=None, out_sums=None, max_parts=0):
        def run(stream):
            c = y.c
            v = y.view5().float() * scale[:c] + shift[:c]
            if res is not None:
                r = res.view5().float()
                if res_scale is not None:
                    r = r * res_scale[:c] + res_shift[:c]
                v = v + r
            if relu:
                if relu_bits is not None:                       # byte [pixel][group] = sign bits of the group's channels
                    vec = self._vec(y)
                    pos = (v > 0).reshape(-1, c // vec, vec).to(torch.int32)
                    wts = (1 << torch.arange(vec, dtype=torch.int32, device=pos.device))
                    relu_bits[: pos.shape[0] * (c // vec)] = (pos * wts).sum(-1).to(torch.uint8).reshape(-1)
                v = v.clamp_min(0)
            out.view5().copy_(v.to(out.dtype))
            if out_sums is not None:                            # column sums of the output AS STORED: ONE partial row
                out_sums[: 2 * c].view(c, 2).zero_()
                out_sums[: 2 * c].view(c, 2)[:, 0] = out.view5().float().reshape(-1, c).sum(0)
        if out_sums is not None:
            assert res is None and relu and max_parts > 0
            return run, 1
        return run

    @staticmethod
    def _dz(da, y, mask_src, scale, shift, relu):
        c = y.c
        dz = da.view5().float()
        yv = y.view5().float()
        if mask_src is not None:
            dz = dz * (mask_src.view5().float() > 0)
        elif relu:
            dz = dz * ((yv * scale[:c] + shift[:c]) > 0)
        return dz, yv

    def bn_bwd_reduce(self, da, y, mask_src, mean, invstd, scale, shift, relu, dz_out, partials, max_parts,
                      relu_bits=None):
        def run(stream):
            c = da.c
            if relu_bits is not None:
                assert mask_src is None
                vec = self._vec(da)
                b = relu_bits[: da.pixels * (c // vec)].to(torch.int32).reshape(-1, c // vec, 1)
                m = ((b >> torch.arange(vec, dtype=torch.int32, device=b.device)) & 1).reshape(-1, c)
                dav = da.view5().float()
                dz = dav * m.reshape(dav.shape).float()
                yv = y.view5().float() if y is not None else None
            else:
                dz, yv = self._dz(da, y, mask_src, scale, shift, relu)
            pt = partials[: c * 2].view(1, c, 2)
            pt[0, :, 0] = dz.reshape(-1, c).sum(0)
            if yv is None:                      # y == NULL: the fused block tail's variant, (sum dz, 0)
                pt[0, :, 1] = 0
            else:
                xhat = (yv - mean[:c]) * invstd[:c]
                pt[0, :, 1] = (dz * xhat).reshape(-1, c).sum(0)
            if dz_out is not None:
                dz_out.view5().copy_(dz.to(dz_out.dtype))
        return run, 1

    def bn_bwd_finalize(self, partials, nparts, c, count, gamma, invstd, dgamma, dbeta, coef, workspace=None):
        def run(stream):
            pt = partials[: nparts * c * 2].view(nparts, c, 2).double().sum(0)
            if dgamma is not None:
                dgamma[:c] += pt[:, 1].float()
            if dbeta is not None:
                dbeta[:c] += pt[:, 0].float()
            cf = coef[: c * 3].view(c, 3)
            cf[:, 0] = gamma[:c] * invstd[:c]
            cf[:, 1] = (pt[:, 0] / count).float()
            cf[:, 2] = (pt[:, 1] / count).float()
        return run

    def bn_bwd_apply(self, da, y, mask_src, mean, invstd, scale, shift, relu, coef, dy):
        def run(stream):
            c = y.c
            dz, yv = self._dz(da, y, mask_src, scale, shift, relu)
            cf = coef[: c * 3].view(c, 3)
            xhat = (yv - mean[:c]) * invstd[:c]
            dy.view5().copy_((cf[:, 0] * (dz - cf[:, 1] - xhat * cf[:, 2])).to(dy.dtype))
        return run

    # ------------------------------------------------------------------ pooling / head / loss
    def maxpool_fwd(self, x, y, argmax, k, s, p):
        def run(stream):
            X = x.view5().float()
            best = torch.full((x.n, x.t, y.h, y.w, x.c), -float("inf"))
            arg = torch.zeros((x.n, x.t, y.h, y.w, x.c), dtype=torch.uint8)
            seen = torch.zeros((1, 1, y.h, y.w, 1), dtype=torch.bool)
            for kh in range(k):
                for kw in range(k):
                    tap = (0, kh - p, kw - p, 0)
                    hi = torch.arange(y.h) * s - p + kh
                    wi = torch.arange(y.w) * s - p + kw
                    ok = ((hi >= 0) & (hi < x.h)).view(1, 1, -1, 1, 1) & ((wi >= 0) & (wi < x.w)).view(1, 1, 1, -1, 1)
                    v = _gather(X, (x.t, y.h, y.w), (1, s, s), tap)
                    take = ok & ((~seen) | (v > best) | torch.isnan(v))
                    best = torch.where(take, v, best)
                    arg = torch.where(take, torch.tensor(kh * k + kw, dtype=torch.uint8), arg)
                    seen = seen | ok
            y.view5().copy_(best.to(y.dtype))
            argmax[: arg.numel()].view_as(arg).copy_(arg)
        return run

    # ---- the stem's BatchNorm -> ReLU -> MaxPool without the activation map: by definition the composition of the
    # stand-alone calls (include/sfk.h sfk_bn_maxpool_*)
    @staticmethod
    def bn_maxpool_supported(k, s, p):
        return (k, s, p) == (3, 2, 1)

    @staticmethod
    def _like(m: FMap) -> FMap:
        return FMap(torch.zeros(m.pixels * m.c, dtype=m.dtype), m.n, m.t, m.h, m.w, m.c)

    def bn_maxpool_fwd(self, y, scale, shift, out, argmax, k, s, p):
        def run(stream):
            a = self._like(y)
            self.bn_apply(y, scale, shift, None, None, None, True, a)(stream)
            self.maxpool_fwd(a, out, argmax, k, s, p)(stream)
        return run

    def bn_maxpool_bwd_reduce(self, d_out, argmax, y, mean, invstd, scale, shift, partials, max_parts):
        def run(stream):
            da = self._like(y)
            self.maxpool_bwd(d_out, argmax, da, 3, 2, 1)(stream)
            self.bn_bwd_reduce(da, y, None, mean, invstd, scale, shift, True, None, partials, max_parts)[0](stream)
        return run, 1

    def bn_maxpool_bwd_apply(self, d_out, argmax, y, mean, invstd, scale, shift, coef, dy):
        def run(stream):
            da = self._like(y)
            self.maxpool_bwd(d_out, argmax, da, 3, 2, 1)(stream)
            self.bn_bwd_apply(da, y, None, mean, invstd, scale, shift, True, coef, dy)(stream)
        return run

    def maxpool_bwd(self, dy, argmax, dx, k, s, p):
        def run(stream):
            G = dy.view5().float()
            arg = argmax[: G.numel()].view(G.shape)
            out = torch.zeros(dx.n, dx.t, dx.h, dx.w, dx.c)
            for kh in range(k):
                for kw in range(k):
                    sel = (arg == kh * k + kw)
                    for ho in range(dy.h):
                        hi = ho * s - p + kh
                        if not (0 <= hi < dx.h):
                            continue
                        for wo in range(dy.w):
                            wi = wo * s - p + kw
                            if 0 <= wi < dx.w:
                                out[:, :, hi, wi] += G[:, :, ho, wo] * sel[:, :, ho, wo]
            dx.view5().copy_(out.to(dx.dtype))
        return run

    @staticmethod
    def _positions(x, k):
        return (x.t - k[0] + 1, x.h - k[1] + 1, x.w - k[2] + 1)

    def head_pool_fwd(self, x, k, rate, seed, feat, feat_ld, f_off):
        def run(stream):
            X = x.view5().float().permute(0, 4, 1, 2, 3)                      # n c t h w
            pooled = torch.nn.functional.avg_pool3d(X, tuple(k), stride=1)   # n c pt ph pw
            n, c = pooled.shape[:2]
            P = pooled[0, 0].numel()
            pooled = pooled.reshape(n, c, P)
            if rate > 0:
                keep = torch.from_numpy(keep_mask(int(seed[0]), n, c, f_off, P, rate))
                pooled = pooled * keep / (1.0 - rate)
            feat[: n * feat_ld].view(n, feat_ld)[:, f_off:f_off + c] = pooled.mean(2)
        return run

    def head_pool_bwd(self, dfeat, feat_ld, f_off, k, rate, seed, dx):
        def run(stream):
            n, c = dx.n, dx.c
            pt, ph, pw = self._positions(dx, k)
            P = pt * ph * pw
            g = dfeat[: n * feat_ld].view(n, feat_ld)[:, f_off:f_off + c]   # n c
            w = torch.ones(n, c, P)
            if rate > 0:
                w = torch.from_numpy(keep_mask(int(seed[0]), n, c, f_off, P, rate)).float() / (1.0 - rate)
            w = (w * g.unsqueeze(2) / P).view(n, c, pt, ph, pw)
            # adjoint of avg_pool3d(stride 1)
            ones = torch.ones(c, 1, *k) / float(k[0] * k[1] * k[2])
            full = torch.nn.functional.conv_transpose3d(w, ones, groups=c)   # n c t h w
            dx.view5().copy_(full.permute(0, 2, 3, 4, 1).to(dx.dtype))
        return run

    def head_dropout_mask(self, n, c, f_off, positions, rate, seed, mask):
        def run(stream):
            m = keep_mask(int(seed[0]), n, c, f_off, positions, rate)
            mask[: m.size].view(n, c, positions).copy_(torch.from_numpy(m.astype(np.uint8)))
        return run

    def fc_fwd(self, feat, w, b, logits, n, f, k):
        def run(stream):
            out = feat[: n * f].view(n, f) @ w[: k * f].view(k, f).t()
            if b is not None:
                out = out + b[:k]
            logits.view(-1)[: n * k].copy_(out.reshape(-1))
        return run

    def fc_bwd(self, dlogits, feat, w, dfeat, dw, db, n, f, k):
        def run(stream):
            dl = dlogits.reshape(-1)[: n * k].view(n, k)
            if dfeat is not None:
                dfeat[: n * f].view(n, f).copy_(dl @ w[: k * f].view(k, f))
            if dw is not None:
                dw[: k * f].view(k, f).add_(dl.t() @ feat[: n * f].view(n, f))
                if db is not None:
                    db[:k].add_(dl.sum(0))
        return run

    def softmax_ce(self, logits, labels, n, k, gscale, dlogits, loss_out, loss_sum, correct):
        def run(stream):
            lg = logits.reshape(-1)[: n * k].view(n, k)
            lsm = torch.log_softmax(lg, 1)
            loss = -lsm[torch.arange(n), labels[:n]].sum() / n
            if dlogits is not None:
                d = torch.softmax(lg, 1)
                d[torch.arange(n), labels[:n]] -= 1
                dlogits.reshape(-1)[: n * k].copy_((d / n * gscale).reshape(-1))
            if loss_out is not None:
                loss_out[0] += loss
            if loss_sum is not None:
                loss_sum[0] += loss
            if correct is not None:
                correct[0] += int((lg.argmax(1) == labels[:n]).sum())
        return run

    # ------------------------------------------------------------------ optimiser / misc
    def adam(self, p, g, m, v, count, lr, b1, b2, eps, gscale, step, shadow=None):
        def run(stream):
            step.add_(1)
            t = int(step[0])
            gg = g[:count] * gscale
            m[:count].mul_(b1).add_(gg, alpha=1 - b1)
            v[:count].mul_(b2).addcmul_(gg, gg, value=1 - b2)
            bc1, bc2 = 1 - b1 ** t, 1 - b2 ** t
            denom = v[:count].sqrt() / (bc2 ** 0.5) + eps
            p[:count].addcdiv_(m[:count], denom, value=-lr / bc1)
            if shadow is not None:
                shadow[:count].copy_(p[:count])
        return run

    def filter_transpose(self, src, dst, cout, wtaps, cin):
        def run(stream):
            n = cout * wtaps * cin
            dst[:n].view(cin, wtaps, cout).copy_(src[:n].view(cout, wtaps, cin).permute(2, 1, 0))
        return run

    def filter_refresh(self, master, s, st, layers):
        def run(stream):
            for off, cout, wtaps, cin, tr in layers:
                n = cout * wtaps * cin
                if s is not None:
                    s[off:off + n].copy_(master[off:off + n])
                if st is not None and tr:
                    st[off:off + n].view(cin, wtaps, cout).copy_(master[off:off + n].view(cout, wtaps, cin).permute(2, 1, 0))
        return run

    def cast(self, src, dst, count):
        def run(stream):
            dst[:count].copy_(src[:count])
        return run

    # ------------------------------------------------------------------ rows either side of the hot path
    def u8_normalize_crop(self, src_u8, lut, crop, pad, out):
        def run(stream):
            n, t, h, w, c = src_u8.shape
            x = lut[src_u8.long()].permute(0, 1, 4, 2, 3)                         # (n,t,c,h,w) fp32
            if crop is None:
                out.copy_(x)
                return
            xp = torch.nn.functional.pad(x, (pad, pad, pad, pad))
            for i in range(n):
                oy, ox = int(crop[i, 0]), int(crop[i, 1])
                out[i] = xp[i, :, :, oy:oy + h, ox:ox + w]
        return run

    def eval_aggregate(self, logits, labels, seg_off, nvideos, softmax, ps_out, pred, correct):
        def run(stream):
            ps = torch.softmax(logits, dim=1) if softmax else logits
            if ps_out is not None:
                ps_out.copy_(ps)
            for v in range(nvideos):
                r0, r1 = int(seg_off[v]), int(seg_off[v + 1])
                if r1 <= r0:
                    pred[v] = -1
                    continue
                k = int(torch.argmax(ps[r0:r1].sum(0) / (r1 - r0)))
                pred[v] = k
                if correct is not None and k == int(labels[r0]):
                    correct.add_(1)
        return run

    def sparse_fusion_fwd(self, x, w, b, y, n, p, c):
        def run(stream):
            y.copy_(torch.einsum("npc,cp->nc", x, w.view(c, p)) + b)
        return run

    def sparse_fusion_bwd(self, x, dy, dw, db, n, p, c):
        def run(stream):
            dw.view(c, p).add_(torch.einsum("nc,npc->cp", dy, x))
            db.add_(dy.sum(0))
        return run

    def fill_zero(self, t):
        def run(stream):
            t.zero_()
        return run

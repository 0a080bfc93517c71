"""CPU emulation of the kernel interface `lcgan_amd.kernels.HipKernels`  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Each method restates, with plain torch fp32 ops (and torch autograd for the backward kernels), what the HIP kernel of
the same name computes on the same NHWC operands.  Uses:
  * CPU tests install it with `tests/helpers.py:install_backend()` to check the host-side autograd wiring (ops.py,
    custom_layers.py, cnn.py, worker.py) against the oracle without a GPU;
  * GPU tests call the HIP kernel and this emulation on identical inputs (per-kernel parity).
Only tests may import this module.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn.functional as F

ACT_NONE, ACT_LRELU, ACT_TANH = 0, 1, 2
SLOPE = 0.2


def ceil8(n):
    return (n + 7) // 8 * 8


def nchw(x):      # NHWC any dtype -> NCHW f32
    return x.float().permute(0, 3, 1, 2)


def nhwc(y, dtype, calloc=None):   # NCHW f32 -> NHWC dtype (zero-padded to calloc channels)
    if calloc is not None and y.shape[1] < calloc:
        y = F.pad(y, (0, 0, 0, 0, 0, calloc - y.shape[1]))
    return y.permute(0, 2, 3, 1).contiguous().to(dtype)


def act_fwd(v, act):
    if act == ACT_LRELU:
        return F.leaky_relu(v, SLOPE)
    if act == ACT_TANH:
        return torch.tanh(v)
    return v


def act_grad_from_out(y, act, gain):
    if act == ACT_LRELU:
        return torch.where(y > 0, torch.ones_like(y), torch.full_like(y, SLOPE)) * gain
    if act == ACT_TANH:
        t = y / gain
        return (1 - t * t) * gain
    return torch.full_like(y, gain)


def rb(x, like_dtype):
    """round to the feature dtype the HIP kernel stages operands in (bf16 mode rounds, f32 mode keeps ~fp32)"""
    return x.to(torch.bfloat16).float() if like_dtype == torch.bfloat16 else x


def _hue_shift(r, g, b, shift):
    """RGB -> HSV, h += shift (mod 1), -> RGB; the same sector arithmetic as csrc/views.hip:hue_shift"""
    mx, mn = torch.maximum(r, torch.maximum(g, b)), torch.minimum(r, torch.minimum(g, b))
    d = mx - mn
    safe = torch.where(d > 0, d, torch.ones_like(d))
    h = torch.where(mx == r, (g - b) / safe, torch.where(mx == g, 2 + (b - r) / safe, 4 + (r - g) / safe)) / 6
    h = torch.where(d > 0, h - torch.floor(h), torch.zeros_like(h))
    s = torch.where(mx > 0, d / torch.where(mx > 0, mx, torch.ones_like(mx)), torch.zeros_like(mx))
    v = mx
    h = h + shift
    h = h - torch.floor(h)
    h6 = h * 6
    fi = torch.floor(h6)
    f = h6 - fi
    i = fi.long() % 6
    p, q, t = v * (1 - s), v * (1 - s * f), v * (1 - s * (1 - f))
    sel = lambda opts: sum(torch.where(i == k, o, torch.zeros_like(o)) for k, o in enumerate(opts))
    return sel([v, q, p, p, t, v]), sel([t, v, v, q, p, p]), sel([p, p, t, v, v, q])


class EmuWeight:
    def __init__(self, P4, N, Kpad, k, need_lo):
        self.P4, self.N, self.Kpad, self.k, self.need_lo = P4, N, Kpad, k, need_lo
        self.hi, self.lo = P4, (P4 if need_lo else None)


class EmuTable:
    def __init__(self, entries):
        self.entries = entries


class EmulatedKernels:
    name = "cpu-emulation"

    # ---- conv family ------------------------------------------------------------------------------------
    def prep_weight(self, w, scale, transpose, need_lo, want_wsq=False):
        ws = w.detach().float() * scale
        wsq = ws.square().sum(dim=(2, 3)) if want_wsq else None
        P4 = ws.transpose(0, 1).contiguous() if transpose else ws
        if not need_lo:
            P4 = P4.to(torch.bfloat16).float()
        Kc = P4.shape[1]
        return EmuWeight(P4, P4.shape[0], (Kc + 31) // 32 * 32, w.shape[-1], need_lo), wsq

    def unprep_wgrad(self, gwp, A, Bc, k, scale, transposed=False, w=None, gwsq=None):
        g = gwp.view(k, k, Bc, A).permute(3, 2, 0, 1) if transposed else gwp.view(k, k, A, Bc).permute(2, 3, 0, 1)
        gw = g * scale
        if gwsq is not None:
            gw = gw + 2 * scale * scale * w.detach() * gwsq[:, :, None, None]
        return gw.contiguous()

    def conv_wgrad_unprep(self, x, g, A, Bc, k, stride, scale, transposed=False, pre_x=None, pre_g=None, w=None, gwsq=None):
        """lcgan_conv_wgrad_fused: the two steps above as one call"""
        gwp = self.conv_wgrad(x, g, A, Bc, k, stride, pre_x=pre_x, pre_g=pre_g)
        wA, wBc = (Bc, A) if transposed else (A, Bc)
        return self.unprep_wgrad(gwp, wA, wBc, k, scale, transposed=transposed, w=w, gwsq=gwsq)

    def _epilogue(self, v, N, post, bias, bias_scale, act, gain, residual, dtype):
        if post is not None:
            v = v * post[:, :N, None, None]
        if bias is not None:
            v = v + (bias.detach() * bias_scale).view(1, -1, 1, 1)
        v = act_fwd(v, act) * gain
        y = nhwc(v, torch.float32, ceil8(N))
        if residual is not None:
            y = y + residual.float()
        return y.to(dtype)

    def prep_weight_group(self, jobs):
        return [self.prep_weight(w, sc, tr, lo, ws) for w, sc, tr, lo, ws in jobs]

    @staticmethod
    def _res(residual, half):
        if residual is None or not half:
            return residual
        return nhwc(F.interpolate(nchw(residual), scale_factor=2, mode="nearest") * 0.25, torch.float32)

    def _sr(self, u, post, xs, dtype, residual=None):
        """fused style-gradient reduction: (post * u [+ residual], sum_pixels xs * u) from the UNSCALED fp32 result u (NHWC f32)"""
        gs = (u * xs.float()).sum(dim=(1, 2))
        y = (u * post[:, None, None, :u.shape[-1]]).to(dtype)
        if residual is not None:                              # joins the stored (rounded) value, like the add_ pass it replaces
            y = (y.float() + residual.float()).to(dtype)
        return y, gs

    def conv_fwd(self, x, pw, N, k, stride, pre=None, post=None, bias=None, bias_scale=1.0, act=ACT_NONE, gain=1.0, residual=None,
                 residual_half=False, xs=None, pool=False, want_mask=False):
        if want_mask:                                         # (y, sign mask of the pre-activation | None): see kernels.HipKernels.conv_fwd
            y = self.conv_fwd(x, pw, N, k, stride, pre, post, bias, bias_scale, act, gain, residual, residual_half)
            return y, ((y > 0) if (act == ACT_LRELU and residual is None) else None)
        if pool:                                              # by-product avg_pool2d(y, 2) of the values as stored
            y = self.conv_fwd(x, pw, N, k, stride, pre, post, bias, bias_scale, act, gain, residual, residual_half, xs)
            return y, self.avgpool2(y)
        residual = self._res(residual, residual_half)
        Kc = pw.P4.shape[1]
        xin = nchw(x)[:, :Kc]
        if pre is not None:
            xin = rb(xin * pre[:, :Kc, None, None], x.dtype)
        v = F.conv2d(xin, pw.P4, stride=stride, padding=k // 2)
        if xs is not None:
            return self._sr(nhwc(v, torch.float32, ceil8(N)), post, xs, x.dtype, residual)
        return self._epilogue(v, N, post, bias, bias_scale, act, gain, residual, x.dtype)

    def conv_bwd_data(self, g, pw, N, k, stride, pre=None, post=None, bias=None, bias_scale=1.0, act=ACT_NONE, gain=1.0, residual=None,
                      residual_half=False, xs=None):
        residual = self._res(residual, residual_half)
        Kc = pw.P4.shape[1]
        gs = nchw(g)[:, :Kc]
        if pre is not None:
            gs = rb(gs * pre[:, :Kc, None, None], g.dtype)
        v = F.conv_transpose2d(gs, pw.P4.permute(1, 0, 2, 3), stride=stride, padding=k // 2, output_padding=stride - 1)
        if xs is not None:
            return self._sr(nhwc(v, torch.float32, ceil8(N)), post, xs, g.dtype, residual)
        return self._epilogue(v, N, post, bias, bias_scale, act, gain, residual, g.dtype)

    # ---- MX-fp8 convolution path (csrc/conv_fp8.hip): the same quantisation rule, then exact fp32 products --------------------
    @staticmethod
    def mx_quant(t, dim):
        """Quantise `t` along `dim` in blocks of 32 like the kernels: power-of-two block scale 2^ex with max|.| / 448 = f 2^ex,
        f in [0.5, 1) (so the block maximum lands in [224, 448)), e4m3 round-to-nearest-even; returns the DEQUANTISED tensor."""
        t = t.float().movedim(dim, -1)
        n = t.shape[-1]
        pad = (-n) % 32
        tp = F.pad(t, (0, pad)).reshape(*t.shape[:-1], (n + pad) // 32, 32)
        m = tp.abs().amax(dim=-1, keepdim=True)
        _, ex = torch.frexp(m * (1.0 / 448.0))
        ex = torch.where(m > 0, ex, torch.zeros_like(ex)).clamp(-126, 127)
        scale = torch.ldexp(torch.ones_like(m), ex)
        q = (tp / scale).to(torch.float8_e4m3fn).float() * scale
        return q.reshape(*t.shape[:-1], n + pad)[..., :n].movedim(-1, dim)

    def prep_weight_fp8(self, w, scale, transpose):
        ws = w.detach().float() * scale
        P4 = ws.transpose(0, 1).contiguous() if transpose else ws            # [N][Kc][k][k]
        return EmuWeight(self.mx_quant(P4, 1), P4.shape[0], (P4.shape[1] + 63) // 64 * 64, w.shape[-1], False)

    def conv_fwd_fp8(self, x, pw, N, k, stride, pre=None, post=None, bias=None, bias_scale=1.0, act=ACT_NONE, gain=1.0, residual=None,
                     residual_half=False):
        residual = self._res(residual, residual_half)
        Kc = pw.P4.shape[1]
        xin = nchw(x)[:, :Kc]
        if pre is not None:
            xin = xin * pre[:, :Kc, None, None]
        v = F.conv2d(self.mx_quant(xin, 1), pw.P4, stride=stride, padding=k // 2)
        return self._epilogue(v, N, post, bias, bias_scale, act, gain, residual, x.dtype)

    def conv_bwd_data_fp8(self, g, pw, N, k, stride, pre=None, post=None, bias=None, bias_scale=1.0, act=ACT_NONE, gain=1.0, residual=None,
                          residual_half=False):
        residual = self._res(residual, residual_half)
        Kc = pw.P4.shape[1]
        gs = nchw(g)[:, :Kc]
        if pre is not None:
            gs = gs * pre[:, :Kc, None, None]
        v = F.conv_transpose2d(self.mx_quant(gs, 1), pw.P4.permute(1, 0, 2, 3), stride=stride, padding=k // 2, output_padding=stride - 1)
        return self._epilogue(v, N, post, bias, bias_scale, act, gain, residual, g.dtype)

    def conv_wgrad(self, x, g, A, Bc, k, stride, pre_x=None, pre_g=None):
        # exact math: the row-segment kernel applies the per-sample scales in fp32 on the accumulator; the generic kernel
        # scales the staged operands (one extra bf16 rounding in bf16 mode, covered by the test tolerance)
        xs, gs = nchw(x)[:, :Bc], nchw(g)[:, :A]
        if pre_x is not None:
            xs = xs * pre_x[:, :Bc, None, None]
        if pre_g is not None:
            gs = gs * pre_g[:, :A, None, None]
        gw = torch.nn.grad.conv2d_weight(xs, (A, Bc, k, k), gs, stride=stride, padding=k // 2)
        return gw.permute(2, 3, 0, 1).reshape(k * k, A, Bc).contiguous()

    # ---- stencils ---------------------------------------------------------------------------------------
    def box3_act(self, x, act, gain):
        return nhwc(act_fwd(F.avg_pool2d(nchw(x), 3, 1, 1), act) * gain, x.dtype)

    def box3_actbwd(self, gy, y, act, gain, clog, want_gbias, mask=None):
        ag = act_grad_from_out(nchw(y), act, gain) if mask is None else torch.where(nchw(mask) > 0, 1.0, SLOPE) * gain
        gz = F.avg_pool2d(nchw(gy), 3, 1, 1) * ag
        gb = gz[:, :clog].sum(dim=(0, 2, 3)) if want_gbias else None
        return nhwc(gz, gy.dtype), gb

    def box3_act_bwd(self, gy, y, act, gain):
        gz = nchw(gy) * (act_grad_from_out(nchw(y), act, gain) if act != ACT_NONE else gain)
        return nhwc(F.avg_pool2d(gz, 3, 1, 1), gy.dtype)

    def up2box(self, x, residual):
        y = F.avg_pool2d(F.interpolate(nchw(x), scale_factor=2, mode="nearest"), 3, 1, 1)
        y = nhwc(y, torch.float32)
        if residual is not None:
            y = y + residual.float()
        return y.to(x.dtype)

    def up2box_bwd(self, gy):
        B, H2, W2, C = gy.shape
        x = torch.zeros(B, C, H2 // 2, W2 // 2, requires_grad=True)
        with torch.enable_grad():
            y = F.avg_pool2d(F.interpolate(x, scale_factor=2, mode="nearest"), 3, 1, 1)
            (gx,) = torch.autograd.grad(y, x, nchw(gy))
        return nhwc(gx, gy.dtype)

    def avgpool2(self, x):
        return nhwc(F.avg_pool2d(nchw(x), 2, 2), x.dtype)

    def avgpool2_bwd(self, gy):
        return nhwc(F.interpolate(nchw(gy), scale_factor=2, mode="nearest") * 0.25, gy.dtype)

    def act_bwd_reduce(self, gy, y, act, gain, clog, want_gz=True, bias=None, bias_scale=1.0, want_gbias=False, want_gdq=False, mask=None,
                       out_scale=None):
        g = gy.float()
        yo = y.float() if y is not None else None
        if mask is not None and not want_gdq:
            assert act == ACT_LRELU
            z = g * torch.where(mask, 1.0, SLOPE) * gain
        else:
            z = g * act_grad_from_out(yo, act, gain) if act != ACT_NONE else g * gain
        zs = z * out_scale[:, None, None, :] if out_scale is not None else z        # the stored gradient carries out_scale, the reductions do not
        gz = zs.to(gy.dtype) if want_gz else None
        gbias = z.sum(dim=(0, 1, 2))[:clog].contiguous() if want_gbias else None
        gdq = None
        if want_gdq:
            t = yo / gain
            if act == ACT_LRELU:
                t = torch.where(t < 0, t / SLOPE, t)
            bv = torch.zeros(gy.shape[-1])
            if bias is not None:
                bv[:clog] = bias.detach() * bias_scale
            gdq = (z * (t - bv)).sum(dim=(1, 2)).contiguous()
        return gz, gbias, gdq

    def scale_reduce(self, u, x, s):
        uf = u.float()
        gs = (uf * x.float()).sum(dim=(1, 2)).contiguous()
        u.copy_((uf * s[:, None, None, :]).to(u.dtype))
        return u, gs

    @staticmethod
    def _grid(flow, H, W, scale):
        gy_, gx_ = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
        base = torch.stack((2 * gx_ / (W - 1) - 1, 2 * gy_ / (H - 1) - 1), dim=-1)[None]
        return base + flow[..., :2] * scale

    def warp_fwd(self, x, flow, scale):
        B, H, W, C = x.shape
        y = F.grid_sample(nchw(x), self._grid(flow.float(), H, W, scale), mode="bicubic", padding_mode="zeros", align_corners=False)
        return nhwc(y, x.dtype)

    def warp_bwd(self, gy, x, flow, scale):
        B, H, W, C = x.shape
        xf = nchw(x).detach().requires_grad_(True)
        ff = flow.float().detach().requires_grad_(True)
        with torch.enable_grad():
            y = F.grid_sample(xf, self._grid(ff, H, W, scale), mode="bicubic", padding_mode="zeros", align_corners=False)
            gx, gf = torch.autograd.grad(y, [xf, ff], nchw(gy))
        gf = gf.clone()
        gf[..., 2:] = 0
        return nhwc(gx, x.dtype), gf.to(flow.dtype)

    @staticmethod
    def _mbstd_stat(x, G):        # x: [N,H,W,C] f32 -> stat broadcast [N,H,W,1]
        N, H, W, C = x.shape
        y = x.reshape(G, -1, H, W, C)
        y = y - y.mean(dim=0)
        y = (y.square().mean(dim=0) + 1e-8).sqrt().mean(dim=[1, 2, 3])          # [M]
        return y.reshape(1, -1, 1, 1, 1).expand(G, -1, H, W, 1).reshape(N, H, W, 1)

    def _mbstd(self, x, G, Cy):
        N, H, W, C = x.shape
        return torch.cat([x, self._mbstd_stat(x, G), x.new_zeros(N, H, W, Cy - C - 1)], dim=-1)

    def mbstd_fwd(self, x, G, Cy):
        return self._mbstd(x.float(), G, Cy).to(x.dtype)

    def mbstd_bwd(self, gy, x, G):
        xf = x.float().detach().requires_grad_(True)
        with torch.enable_grad():
            (gx,) = torch.autograd.grad(self._mbstd(xf, G, gy.shape[-1]), xf, gy.float())
        return gx.to(x.dtype)

    def mbstd_bwd2(self, v, gy, x, G):
        xf = x.float().detach().requires_grad_(True)
        gf = gy.float().detach().requires_grad_(True)
        with torch.enable_grad():
            (gx,) = torch.autograd.grad(self._mbstd(xf, G, gy.shape[-1]), xf, gf, create_graph=True)
            ggy, gx2 = torch.autograd.grad(gx, [gf, xf], v.float())
        return ggy.to(gy.dtype), gx2.to(x.dtype)

    # ---- RGB ----------------------------------------------------------------------------------------------
    def rgb_expand(self, img, w, bias, bias_scale, clog, act, gain, dtype, pool=False):
        if pool:
            y = self.rgb_expand(img, w, bias, bias_scale, clog, act, gain, dtype)
            return y, self.avgpool2(y)
        B = img.shape[0]
        wb = w.detach() if w.shape[0] > 1 else w.detach().expand(B, -1, -1)
        v = torch.einsum("bohw,boc->bhwc", img.float(), wb)
        if bias is not None:
            v = v + F.pad(bias.detach() * bias_scale, (0, v.shape[-1] - clog))
        v = act_fwd(v, act) * gain
        v[..., clog:] = 0
        return v.to(dtype).contiguous()

    def rgb_reduce(self, x, w, bias, bias_scale):
        B = x.shape[0]
        wb = w.detach() if w.shape[0] > 1 else w.detach().expand(B, -1, -1)
        img = torch.einsum("bhwc,boc->bohw", x.float(), wb)
        if bias is not None:
            img = img + (bias.detach() * bias_scale).view(1, 3, 1, 1)
        return img.contiguous()

    def rgb_wgrad(self, img, feat, per_sample):
        gw = torch.einsum("bohw,bhwc->boc", img.float(), feat.float())
        return gw.contiguous() if per_sample else gw.sum(0, keepdim=True).contiguous()

    def rgb_expand_bwd(self, gy, y, img, w, act, gain, clog, want_gimg, want_gw, want_gbias, fbias=None, fbias_scale=1.0, recompute=False):
        """lcgan_rgb_expand_bwd: gz = gy * act'(y) stays fp32 (never rounded to the feature dtype)"""
        B = gy.shape[0]
        z = gy.float() * (act_grad_from_out(y.float(), act, gain) if act != ACT_NONE else gain)
        z[..., clog:] = 0
        wb = w.detach() if w.shape[0] > 1 else w.detach().expand(B, -1, -1)
        gimg = torch.einsum("bhwc,boc->bohw", z, wb).contiguous() if want_gimg else None
        gw = None
        if want_gw:
            gw = torch.einsum("bohw,bhwc->boc", img.float(), z)
            gw = gw.contiguous() if w.shape[0] > 1 else gw.sum(0, keepdim=True).contiguous()
        gbias = z.sum(dim=(0, 1, 2))[:clog].contiguous() if want_gbias else None
        return gimg, gw, gbias

    def rgb_reduce_bwd_act(self, gimg, y, wm, bias, bias_scale, act, gain, clog, want_gbias=True, want_gdq=True, out_scale=None):
        """lcgan_rgb_reduce_bwd_act"""
        B, H, W, Cc = y.shape
        per_sample = wm.shape[0] > 1
        wb = (wm.detach() if per_sample else wm.detach().expand(B, -1, -1)).clone()
        wb[..., clog:] = 0
        yo = y.float()
        gf = torch.einsum("bohw,boc->bhwc", gimg.float(), wb)
        z = gf * act_grad_from_out(yo, act, gain)
        gbias = z.sum(dim=(0, 1, 2))[:clog].contiguous() if want_gbias else None
        gdq = None
        if want_gdq:
            t = yo / gain
            if act == ACT_LRELU:
                t = torch.where(t < 0, t / SLOPE, t)
            bv = torch.zeros(Cc)
            if bias is not None:
                bv[:clog] = bias.detach() * bias_scale
            gdq = (z * (t - bv)).sum(dim=(1, 2)).contiguous()
        gwm = torch.einsum("bohw,bhwc->boc", gimg.float(), yo)
        gwm[..., clog:] = 0
        gwm = gwm.contiguous() if per_sample else gwm.sum(0, keepdim=True).contiguous()
        if out_scale is not None:
            z = z * out_scale[:, None, None, :]
        return z.to(y.dtype), gbias, gdq, gwm

    # ---- flow layer as 1x1 GEMM + scatter (csrc/stencil.hip: flow_col2im / flow_im2col) ---------------------------------
    @staticmethod
    def _flow_onehot():
        w = torch.zeros(18, 2, 3, 3)                       # [(ky*3+kx)*2+o][o'][ky'][kx'] = 1 where all three match
        for ky in range(3):
            for kx in range(3):
                for o in range(2):
                    w[(ky * 3 + kx) * 2 + o, o, ky, kx] = 1.0
        return w

    def flow_col2im(self, t, d, bias):
        up = F.conv_transpose2d(nchw(t)[:, :18], self._flow_onehot(), stride=2, padding=1, output_padding=1)     # out[2i-1+ky] += in * w[ky]
        u = up * d[:, :2, None, None]
        if bias is not None:
            u = u + bias.detach().view(1, 2, 1, 1)
        return nhwc(u, t.dtype, 8)

    def flow_im2col(self, gu, d):
        g = nchw(gu)[:, :2] * d[:, :2, None, None]
        gt = F.conv2d(g, self._flow_onehot(), stride=2, padding=1)                                                # gt[(ky,kx,o)][i] = g[o][2i-1+ky]
        return nhwc(gt, gu.dtype, 24)

    # ---- layout ---------------------------------------------------------------------------------------------
    def nchw_to_nhwc(self, src, B, calloc, dtype):
        s = src.float()
        if s.shape[0] == 1 and B > 1:
            s = s.expand(B, -1, -1, -1)
        return nhwc(s, dtype, calloc)

    def nhwc_to_nchw(self, src, clog, reduce):
        y = nchw(src)[:, :clog]
        return (y.sum(0, keepdim=True) if reduce else y).contiguous()

    # ---- linears ----------------------------------------------------------------------------------------------
    def linear_fwd(self, x, w, bias, scale, bias_scale, act, gain):
        v = x @ w.detach().t() * scale
        if bias is not None:
            v = v + bias.detach() * bias_scale
        return act_fwd(v, act) * gain

    def linear_bwd_data(self, gy, w, scale):
        return gy @ w.detach() * scale

    def linear_wgrad(self, gy, x, scale):
        return gy.t() @ x * scale

    def colsum(self, gy, scale):
        return gy.sum(0) * scale

    def linear_wgrad_bias(self, gy, x, scale, bias_scale):
        return self.linear_wgrad(gy, x, scale), self.colsum(gy, bias_scale)

    def linear_group_fwd(self, x, ws, biases, scales, bias_scales, act=0, gain=1.0):
        return [self.linear_fwd(x, w, b, sc, bs, act, gain) for w, b, sc, bs in zip(ws, biases, scales, bias_scales)]

    def linear_multi_fwd(self, xs, ws, biases, scales, bias_scales, act=0, gain=1.0):
        return [self.linear_fwd(x, w, b, sc, bs, act, gain) for x, w, b, sc, bs in zip(xs, ws, biases, scales, bias_scales)]

    def linear_multi_bwd(self, gys, xs, ws, scales, bias_scales, want_gx=True):
        gxs = [self.linear_bwd_data(g, w, sc) for g, w, sc in zip(gys, ws, scales)] if want_gx else None
        return gxs, [self.linear_wgrad(g, x, sc) for g, x, sc in zip(gys, xs, scales)], [self.colsum(g, bs) for g, bs in zip(gys, bias_scales)]

    def linear_group_bwd(self, gys, x, ws, scales, bias_scales, want_gx=True):
        gx = None
        if want_gx:
            gx = torch.zeros_like(x)
            for gy, w, sc in zip(gys, ws, scales):
                gx = gx + self.linear_bwd_data(gy, w, sc)
        gws = [self.linear_wgrad(gy, x, sc) for gy, sc in zip(gys, scales)]
        gbs = [self.colsum(gy, bs) for gy, bs in zip(gys, bias_scales)]
        return gx, gws, gbs

    def act_bwd_f32(self, gy, y, act, gain):
        return gy * act_grad_from_out(y, act, gain)

    def demod_fwd(self, s, wsq, ostride, eps=1e-8):
        d = torch.zeros(s.shape[0], ostride)
        d[:, :wsq.shape[0]] = torch.rsqrt(s.square() @ wsq.t() + eps)
        return d

    def demod_group(self, ss, wsqs, ostrides, eps=1e-8):
        return [self.demod_fwd(s_, w_, int(o), eps) for s_, w_, o in zip(ss, wsqs, ostrides)]

    def demod_bwd(self, gdq, d, s, wsq, gs):
        O = wsq.shape[0]
        gq = -0.5 * gdq[:, :O] * d[:, :O] ** 2
        gs += 2 * s * (gq @ wsq)
        return gq.t() @ s.square()

    # ---- losses ---------------------------------------------------------------------------------------------------
    def bce_fwd(self, logit, target_one):
        return F.softplus(-logit if target_one else logit).mean()

    def bce_bwd(self, logit, target_one, gout):
        sgn = -1.0 if target_one else 1.0
        return gout * sgn * torch.sigmoid(sgn * logit) / logit.numel()

    def contrastive_fwd(self, a, p, n, tau):
        t = ((a * n).sum(1) - (a * p).sum(1)) / tau
        return F.softplus(t).mean(), t

    def contrastive_bwd(self, a, p, n, t, gout, tau):
        k = (gout * torch.sigmoid(t) / (a.shape[0] * tau))[:, None]
        return k * (n - p), -k * a, k * a

    def l2norm_fwd(self, x, eps=1e-12):
        ns = x.norm(dim=1).clamp_min(eps)
        return x / ns[:, None], ns

    def l2norm_bwd(self, gy, y, ns):
        return (gy - y * (gy * y).sum(1, keepdim=True)) / ns[:, None]

    def powsum(self, x, pw, coef):
        return (x.abs().sum() if pw == 1 else x.square().sum()) * coef

    def powsum_bwd(self, x, pw, coef, gout):
        return gout * coef * (torch.sign(x) if pw == 1 else 2 * x)

    def qr(self, A):
        return torch.linalg.qr(A, mode="reduced")

    def avg_latent(self, w, avg, beta):
        m = w.mean(0)
        avg.copy_(m + beta * (avg - m))

    # ---- training views (csrc/views.hip; custom_dataset.py:59-88) ---------------------------------------------------------
    def make_views(self, src, params):
        B, _, R, _ = src.shape
        x01 = (src.float() + 1) * 0.5
        outs = [torch.empty_like(src) for _ in range(3)]
        ys, xs = torch.meshgrid(torch.arange(R, dtype=torch.float32), torch.arange(R, dtype=torch.float32), indexing="ij")
        for b in range(B):
            P = params[b].float()
            img = x01[b].flip(-1) if P[0] != 0 else x01[b]
            outs[0][b] = (img * 2 - 1).clamp(-1, 1)
            w = P[7] * xs + P[8] * ys + P[9]
            iw = torch.where(w.abs() > 1e-12, 1.0 / w, torch.zeros_like(w))
            sx, sy = (P[1] * xs + P[2] * ys + P[3]) * iw, (P[4] * xs + P[5] * ys + P[6]) * iw
            far = ~((sx > -2) & (sy > -2) & (sx < R + 1) & (sy < R + 1))
            fx, fy = torch.floor(sx), torch.floor(sy)
            ax, ay = sx - fx, sy - fy
            x0 = torch.where(far, torch.full_like(fx, -4), fx).long()
            y0 = torch.where(far, torch.full_like(fy, -4), fy).long()

            def fetch(yy, xx):
                ok = (yy >= 0) & (yy < R) & (xx >= 0) & (xx < R)
                v = img[:, yy.clamp(0, R - 1), xx.clamp(0, R - 1)]
                return torch.where(ok.unsqueeze(0), v, torch.zeros_like(v))
            g = (1 - ay) * ((1 - ax) * fetch(y0, x0) + ax * fetch(y0, x0 + 1)) + ay * ((1 - ax) * fetch(y0 + 1, x0) + ax * fetch(y0 + 1, x0 + 1))
            quant = P[25] != 0                                      # uint8 round trip of the augmented views (custom_dataset.py:76-79)
            if quant:
                g = torch.round(g.clamp(0, 1) * 255) / 255
            outs[1][b] = (g * 2 - 1).clamp(-1, 1)
            r, gch, bl = img[0].clone(), img[1].clone(), img[2].clone()
            if P[10] == 0:
                hole = (xs >= int(P[11])) & (xs < int(P[13])) & (ys >= int(P[12])) & (ys < int(P[14]))
                r, gch, bl = (torch.where(hole, torch.zeros_like(c), c) for c in (r, gch, bl))
            else:
                for k in range(4):
                    op = int(P[19 + k])
                    if op == 0:
                        r, gch, bl = ((c * P[15]).clamp(0, 1) for c in (r, gch, bl))
                    elif op == 1:
                        # pivot < 0: the mean luma of the image as it reaches the contrast op (views_pivot_kernel)
                        m = P[23] if P[23] >= 0 else (0.299 * r + 0.587 * gch + 0.114 * bl).mean()
                        r, gch, bl = (((c - m) * P[16] + m).clamp(0, 1) for c in (r, gch, bl))
                    elif op == 2:
                        gr = 0.299 * r + 0.587 * gch + 0.114 * bl
                        r, gch, bl = ((gr + (c - gr) * P[17]).clamp(0, 1) for c in (r, gch, bl))
                    else:
                        r, gch, bl = _hue_shift(r, gch, bl, float(P[18]))
            app = torch.stack([r, gch, bl])
            if quant:
                app = torch.round(app * 255) / 255
            outs[2][b] = app * 2 - 1
        return tuple(outs)

    # ---- multi-tensor ---------------------------------------------------------------------------------------------------
    def multi_tensor(self, table, op, a0, a1=0.0, a2=0.0):
        for e in table.entries:
            if op == 0:
                p, g, m, v, f0, f1 = e
                m.mul_(a0).add_(g, alpha=1 - a0)
                v.mul_(a1).addcmul_(g, g, value=1 - a1)
                p.sub_(f0 * m / (v.sqrt() * f1 + a2))
            elif op == 1:
                pe, ps = e[0], e[1]
                pe.copy_(ps + a0 * (pe - ps))
            else:
                e[0].copy_(e[1] * a0)

    def prof_enable(self, on):
        pass

    def prof_collect(self):
        return {}

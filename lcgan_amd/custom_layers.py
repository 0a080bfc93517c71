"""Layer library of LC-GAN on the MI355X HIP kernels.

Mirrors the reference's `custom_layers.py` class-for-class (same class names, constructor signatures, parameter /
buffer names and shapes, hence the same `state_dict()` layout), but every `forward` runs hand-written gfx950
kernels through `lcgan_amd.ops`.  Between layers, feature maps are NHWC `[B,H,W,C]` tensors in the configured
feature dtype (`lcgan_amd.config`); NCHW fp32 only appears at the Generator / Discriminator boundary.

reference: custom_layers.py:7-306
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn as nn

from . import config
from . import kernels as KM
from . import ops
from .kernels import ACT_LRELU, ACT_NONE, ACT_TANH, ceil8

SQRT2 = math.sqrt(2.0)
SQRT_HALF = math.sqrt(0.5)


class EqualizedWeight(nn.Module):
    """reference custom_layers.py:7-14 -- stores W / lr_mul; the run-time scale c = lr_mul / sqrt(fan_in) is folded
    into the kernels (weight prep / linear scale) instead of materialising W*c."""

    def __init__(self, shape, lr_mul=1.0):
        super().__init__()
        self.c = float(1 / np.sqrt(np.prod(shape[1:])) * lr_mul)
        self.weight = nn.Parameter(torch.randn(shape).div_(lr_mul))

    def forward(self):
        return self.weight * self.c


class EqualizedLinear(nn.Module):
    """reference custom_layers.py:17-25.  x: f32 [M, in_features]."""

    def __init__(self, in_features, out_features, bias=0.0, lr_mul=1.0):
        super().__init__()
        self.weight = EqualizedWeight([out_features, in_features], lr_mul)
        self.bias = nn.Parameter(torch.ones(out_features) * bias)
        self.lr_mul = lr_mul

    def forward(self, x, act=ACT_NONE):
        return ops.LinearFn.apply(x.contiguous(), self.weight.weight, self.bias, self.weight.c, self.lr_mul, act, 1.0)


class EqualizedConv2d(nn.Module):
    """reference custom_layers.py:28-44.  `forward` takes an NHWC feature map; `forward_rgb` takes the fp32 NCHW image
    (the discriminator's first 1x1 conv, cnn.py:20).  act / gain / residual are fused into the conv epilogue."""

    def __init__(self, in_features, out_features, kernel_size, stride=1, no_bias=False, lr_mul=1.0):
        super().__init__()
        self.padding = kernel_size // 2
        self.kernel_size = kernel_size
        self.weight = EqualizedWeight([out_features, in_features, kernel_size, kernel_size], lr_mul)
        self.no_bias = no_bias
        if not self.no_bias:
            self.bias = nn.Parameter(torch.zeros([out_features]))
        self.stride = stride
        self.lr_mul = lr_mul

    def forward(self, x, act=ACT_NONE, gain=1.0, residual=None, pool=False):
        """pool: returns (y, avg_pool2d(y, 2)); the pooled copy is a non-differentiable by-product (ops.Conv2dPoolFn)"""
        bias = None if self.no_bias else self.bias
        wscale = self.weight.c
        if act == ACT_NONE and gain != 1.0:          # a linear conv: fold the gain into the weight scale
            assert bias is None
            wscale, gain = wscale * gain, 1.0
        fn = ops.Conv2dPoolFn if pool else ops.Conv2dFn
        return fn.apply(x, self.weight.weight, bias, residual, self.kernel_size, self.stride, act, gain, wscale, self.lr_mul)

    def forward_with_pool(self, x, act=ACT_NONE, gain=1.0, box=False, pooled_hint=None):
        """(box3?(self(x, act, gain)), avg_pool2d(x, 2)) as ONE autograd node (see ops.ConvPoolFn); pooled_hint: avg_pool2d(x, 2) where
        the producer of x already wrote it"""
        assert self.stride == 1 and act != ACT_NONE
        bias = None if self.no_bias else self.bias
        return ops.ConvPoolFn.apply(x, self.weight.weight, bias, self.kernel_size, act, gain, self.weight.c, self.lr_mul, box, pooled_hint)

    def forward_rgb(self, img, act=ACT_NONE, gain=1.0, pool=False):
        w = self.weight.weight
        C = w.shape[0]
        assert self.kernel_size == 1 and w.shape[1] == 3 and C % 8 == 0
        wt = (w.view(C, 3).t() * self.weight.c).unsqueeze(0).contiguous()            # [1,3,C] torch glue (384 values)
        bias = None if self.no_bias else self.bias
        return ops.RGBExpandFn.apply(img.contiguous(), wt, bias, self.lr_mul, C, act, gain, config.feature_dtype(), pool)


class ModulatedConv2d(nn.Module):
    """reference custom_layers.py:47-86.  x: NHWC feature map, s: f32 [B, in_features] style."""

    def __init__(self, in_features, out_features, kernel_size, up=1, eps=1e-8, lr_mul=1.0):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.kernel_size = kernel_size
        self.padding = (kernel_size - 1) // 2
        self.up = up
        self.weight = EqualizedWeight([out_features, in_features, kernel_size, kernel_size], lr_mul)
        self.bias = nn.Parameter(torch.zeros([out_features]))
        self.lr_mul = lr_mul
        self.eps = eps

    def forward(self, x, s, act=ACT_NONE, gain=1.0):
        assert self.kernel_size == 3 and self.lr_mul == 1.0
        if self.up == 2 and self.out_features == 2 and act == ACT_NONE and gain == 1.0 and config.flow_gemm():
            return ops.FlowConvFn.apply(x, self.weight.weight, self.bias, s)     # the 2-channel flow layer: 1x1 GEMM + scatter
        return ops.ModConvFn.apply(x, self.weight.weight, self.bias, s, self.up, act, gain)

    def forward_to_rgb(self, x, s):
        """kernel_size 1, out_features 3 (ToRGBBlock.modulated_conv1, custom_layers.py:175,181): per-sample 3xC weights
        (modulate + demodulate: a few hundred values of torch glue) feed the HIP reduce kernel; output is the f32 NCHW image."""
        return ops.RGBReduceFn.apply(x, self.rgb_weights(s), self.bias, self.lr_mul)

    def rgb_weights(self, s):
        """[B,3,C] modulated + demodulated 1x1 weights of the to-RGB layer (custom_layers.py:62-68)"""
        assert self.kernel_size == 1 and self.out_features == 3
        w = self.weight.weight.view(3, self.in_features) * self.weight.c                # [3,C]
        wm = w.unsqueeze(0) * s.unsqueeze(1)                                              # [B,3,C]   custom_layers.py:62-64
        wm = wm * torch.rsqrt(wm.square().sum(dim=2, keepdim=True) + self.eps)            # custom_layers.py:67-68
        return wm.contiguous()


class SynthesisLayer(nn.Module):
    """reference custom_layers.py:89-111 (use_noise is always False in the reference, cnn.py:83,87)."""

    def __init__(self, in_features, out_features, latent_dim, resolution, kernel_size=3, up=1, lr_mul=1.0, use_noise=False):
        super().__init__()
        assert not use_noise, "the reference never enables use_noise (cnn.py:83,87)"
        self.latent_dim = latent_dim
        self.up = up
        self.resolution = resolution
        self.use_noise = use_noise
        self.linear = EqualizedLinear(self.latent_dim, in_features, bias=1.0, lr_mul=1.0)
        self.modulated_conv = ModulatedConv2d(in_features, out_features, kernel_size, up=self.up, lr_mul=1.0)

    def forward(self, x, latent, act=ACT_NONE, gain=1.0, style=None):
        s = self.linear(latent) if style is None else style      # style: this layer's affine, precomputed by ops.grouped_linear
        if self.modulated_conv.kernel_size == 1:
            return self.modulated_conv.forward_to_rgb(x, s)
        return self.modulated_conv(x, s, act, gain)


def _split_latents(lat, n):
    if isinstance(lat, (tuple, list)):
        assert len(lat) == n
        return list(lat)
    if lat.dim() == 2:
        return [lat] * n
    return [lat[:, i].contiguous() for i in range(n)]      # reference layout [B, n, D] (custom_layers.py:141-142)


class SynthesisBlock(nn.Module):
    """reference custom_layers.py:114-166."""

    def __init__(self, in_features, out_features, g_latent_dim, a_latent_dim, resolution, max_flow_scale, use_noise=False):
        super().__init__()
        self.resolution = resolution
        self.use_noise = use_noise
        self.max_flow_scale = max_flow_scale
        self.modulated_conv0 = SynthesisLayer(in_features, out_features, a_latent_dim, resolution, up=2, use_noise=self.use_noise)
        self.modulated_conv1 = SynthesisLayer(out_features, out_features, a_latent_dim, resolution, up=1, use_noise=self.use_noise)
        self.skip_layer = EqualizedConv2d(in_features, out_features, kernel_size=1, no_bias=True, lr_mul=1.0)
        self.flow_layer = SynthesisLayer(in_features, 2, g_latent_dim, resolution, up=2, use_noise=False)
        self.gain = np.sqrt(2)
        self.skip_gain = np.sqrt(0.5)

    def style_layers(self):
        return (self.flow_layer, self.modulated_conv0, self.modulated_conv1)

    def forward(self, x, g_latent, a_latent, styles=(None, None, None)):
        (g_lat,) = _split_latents(g_latent, 1)
        a0, a1 = _split_latents(a_latent, 2)
        fl, c0, sk = self.flow_layer, self.modulated_conv0, self.skip_layer
        if (config.synth_fork() and config.flow_gemm() and config.conv_operands() == "bf16" and fl.modulated_conv.out_features == 2
                and sk.no_bias and sk.kernel_size == 1 and c0.modulated_conv.kernel_size == 3):
            # the three consumers of x as ONE autograd node: their data gradients chain through the convolution epilogues instead of
            # being summed by two add passes of autograd's (ops.SynthForkFn)
            sf = fl.linear(g_lat) if styles[0] is None else styles[0]
            s0 = c0.linear(a0) if styles[1] is None else styles[1]
            u, y0, skip = ops.SynthForkFn.apply(x, fl.modulated_conv.weight.weight, fl.modulated_conv.bias, sf,
                                                c0.modulated_conv.weight.weight, c0.modulated_conv.bias, s0,
                                                sk.weight.weight, sk.weight.c * SQRT_HALF)
            flow = ops.Box3ActFn.apply(u, ACT_TANH, 1.0)                             # :150-151
            h = ops.Box3ActFn.apply(y0, ACT_LRELU, SQRT2)                            # :154-155
        else:
            # flow field: up-conv (2 channels, padded to 8) -> box filter -> tanh                 :149-151
            flow = ops.Box3ActFn.apply(fl(x, g_lat, style=styles[0]), ACT_TANH, 1.0)
            # main branch: up-conv -> box filter -> lrelu*sqrt2                                      :153-155
            h = ops.Box3ActFn.apply(c0(x, a0, style=styles[1]), ACT_LRELU, SQRT2)
            # skip branch: 1x1 conv * sqrt(.5) at low res                                            :145
            skip = sk(x, ACT_NONE, SQRT_HALF)
        h = self.modulated_conv1(h, a1, ACT_LRELU, 1.0, style=styles[2])           # conv -> lrelu  :157-158
        # nearest x2 + box filter of the skip branch fused with the add                              :146-147,159
        y = ops.Up2BoxAddFn.apply(skip, h)
        # bicubic feature warp                                                                   :162-165
        return ops.WarpFn.apply(y, flow, float(self.max_flow_scale))


class ToRGBBlock(nn.Module):
    """reference custom_layers.py:169-182; returns the fp32 NCHW image."""

    def __init__(self, in_features, out_features, a_latent_dim, resolution, use_noise=False):
        super().__init__()
        self.resolution = resolution
        self.use_noise = use_noise
        self.modulated_conv0 = SynthesisLayer(in_features, in_features, a_latent_dim, resolution, use_noise=self.use_noise)
        self.modulated_conv1 = SynthesisLayer(in_features, out_features, a_latent_dim, resolution, kernel_size=1, use_noise=False)

    def forward(self, x, a_latent, styles=(None, None)):
        a0, a1 = _split_latents(a_latent, 2)
        l0, l1 = self.modulated_conv0, self.modulated_conv1
        s0 = l0.linear(a0) if styles[0] is None else styles[0]
        s1 = l1.linear(a1) if styles[1] is None else styles[1]
        m0, m1 = l0.modulated_conv, l1.modulated_conv
        assert m0.kernel_size == 3 and m0.up == 1 and m1.kernel_size == 1 and m1.out_features == 3
        # both layers as one autograd node (ops.ModConvRGBFn): the backward needs one pass over the 3x3 conv's activation
        return ops.ModConvRGBFn.apply(x, m0.weight.weight, m0.bias, s0, m1.rgb_weights(s1), m1.bias, m1.lr_mul, ACT_LRELU, 1.0)


class DiscriminatorBlock(nn.Module):
    """reference custom_layers.py:185-217 (the reference always builds it with skip=True, cnn.py:25)."""

    def __init__(self, in_features, out_features, skip=False):
        super().__init__()
        self.conv0 = EqualizedConv2d(in_features, in_features, kernel_size=3, lr_mul=1.0)
        self.conv1 = EqualizedConv2d(in_features, out_features, kernel_size=3, stride=2, lr_mul=1.0)
        self.skip = skip
        if self.skip:
            self.skip_layer = EqualizedConv2d(in_features, out_features, kernel_size=1, no_bias=True, lr_mul=1.0)
            self.gain = np.sqrt(2)
            self.skip_gain = np.sqrt(0.5)

    def forward(self, x, pooled=None, want_pool=False):
        """pooled: avg_pool2d(x, 2) when the producer of x left it as a by-product; want_pool: return (out, avg_pool2d(out, 2)) for the
        next block (the pooled copy then comes out of the closing convolution's epilogue instead of a pooling pass over `out`)"""
        if self.skip:
            h, pooled = self.conv0.forward_with_pool(x, ACT_LRELU, SQRT2, box=True, pooled_hint=pooled)   # :202, :204-206 (one node: see ops.ConvPoolFn)
        else:
            h = ops.Box3Fn.apply(self.conv0(x, ACT_LRELU, 1.0))                 # :212-214
        h = self.conv1(h, ACT_LRELU, 1.0)                                       # :207-208
        if not self.skip:
            return (h, None) if want_pool else h
        return self.skip_layer(pooled, ACT_NONE, SQRT_HALF, residual=h, pool=want_pool)   # :203, :209 (add fused in the epilogue)


class MinibatchStdLayer(nn.Module):
    """reference custom_layers.py:237-256 (num_channels = 1)."""

    def __init__(self, group_size, num_channels=1):
        super().__init__()
        assert num_channels == 1
        self.group_size = group_size
        self.num_channels = num_channels

    def forward(self, x, n_sub=1):
        """n_sub > 1: the batch holds n_sub independent passes of the reference (worker.py:163-165, 198-200 evaluate the discriminator
        once per view) laid end to end; the statistic is taken per pass, exactly as n_sub separate calls would (custom_layers.py:243-256)"""
        if n_sub > 1:
            assert x.shape[0] % n_sub == 0
            return torch.cat([self.forward(c) for c in x.chunk(n_sub, dim=0)], dim=0)
        N = x.shape[0]
        G = min(self.group_size, N) if self.group_size is not None else N
        return ops.MbstdFn.apply(x.contiguous(), G)


class DiscriminatorEpilogue(nn.Module):
    """reference custom_layers.py:220-234; returns f32 [B, in_features]."""

    def __init__(self, in_features, resolution, mbstd_group_size=4):
        super().__init__()
        self.resolution = resolution
        self.mb_std = MinibatchStdLayer(group_size=mbstd_group_size)
        self.conv = EqualizedConv2d(in_features + 1, in_features, kernel_size=3, lr_mul=1.0)
        self.linear = EqualizedLinear(in_features * (resolution ** 2), in_features, lr_mul=0.01)

    def forward(self, x, n_sub=1):
        x = self.mb_std(x, n_sub)                                               # [B,4,4,C+1 -> padded to a multiple of 8]
        x = self.conv(x, ACT_LRELU, 1.0)
        flat = ops.ToNCHWFn.apply(x, self.conv.weight.weight.shape[0]).flatten(1)   # NCHW order, as x.flatten(1) at :232
        return self.linear(flat, ACT_LRELU)


def _qr_q(matrix: torch.Tensor) -> torch.Tensor:
    """Q of the reduced Householder QR (torch.qr(...)[0], custom_layers.py:274-276).  Matrices up to 64 x 64 (the reference's
    noise dimensions) run in ONE HIP workgroup (rocSOLVER needs ~200 micro-launches for the same factorisation)."""
    if matrix.shape[0] <= 64:
        return ops.QrQFn.apply(matrix)
    return torch.linalg.qr(matrix, mode="reduced")[0]


class MappingNetwork(nn.Module):
    """reference custom_layers.py:259-287: x = (Q(tanh(basis)) diag(|d|+eps)) z, then affine layers without activation."""

    def __init__(self, channels_list, lr_mul=0.01):
        super().__init__()
        self.eps = 1e-6
        self.matrix_size = channels_list[0]
        self.diagonal_params = nn.Parameter(torch.randn([self.matrix_size]))
        self.basis_params = nn.Parameter(torch.randn([self.matrix_size, self.matrix_size]))
        self.num_layers = len(channels_list) - 1
        self.mlp = nn.Sequential(*[EqualizedLinear(channels_list[i], channels_list[i + 1], lr_mul=lr_mul)
                                   for i in range(self.num_layers)])

    def orthogonalize(self, matrix):
        return _qr_q(matrix)

    def mapping_matrix(self):
        return self.orthogonalize(torch.tanh(self.basis_params)) * (self.diagonal_params.abs() + self.eps).unsqueeze(0)

    def forward(self, z, L=None):
        """L: this network's mapping matrix when the caller computed it together with another network's (mapping_matrices)"""
        if L is None:
            L = self.mapping_matrix()
        L = L.contiguous()                                                      # [m,m]; 64x64 torch glue
        x = ops.LinearFn.apply(z.contiguous(), L, None, 1.0, 0.0, ACT_NONE, 1.0)   # x_b = L z_b  (:283-285)
        for layer in self.mlp:
            x = layer(x)
        return x


def mapping_matrices(nets):
    """[net.mapping_matrix() for net in nets] with ONE QR launch when the matrices have the same size (the geometry and the
    appearance network: two 64 x 64 factorisations, one workgroup each)."""
    nets = list(nets)
    if len({n.matrix_size for n in nets}) != 1 or nets[0].matrix_size > 64:
        return [n.mapping_matrix() for n in nets]
    Q = ops.QrQFn.apply(torch.tanh(torch.stack([n.basis_params for n in nets])))
    d = torch.stack([n.diagonal_params for n in nets]).abs() + nets[0].eps
    return list((Q * d.unsqueeze(1)).unbind(0))


def mapping_forward(nets, zs, Ls=None):
    """[net(z, L) for net, z, L in zip(nets, zs, Ls)] with the layers the networks have at equal depth evaluated together (ops.MultiLinearFn:
    the reference's geometry and appearance mapping networks are twelve layers each, cnn.py:66-72 -- 12 launches per pass instead of 24)."""
    nets = list(nets)
    if Ls is None:
        Ls = mapping_matrices(nets)
    xs = [ops.LinearFn.apply(z.contiguous(), L.contiguous(), None, 1.0, 0.0, ACT_NONE, 1.0) for z, L in zip(zs, Ls)]   # x_b = L z_b (:283-285)
    depth = max(n.num_layers for n in nets)
    for i in range(depth):
        live = [k for k, n in enumerate(nets) if i < n.num_layers]
        layers = [nets[k].mlp[i] for k in live]
        if len(live) > 1 and len({xs[k].shape[0] for k in live}) == 1:
            for k, y in zip(live, ops.multi_linear([xs[k] for k in live], layers)):
                xs[k] = y
        else:
            for k, l in zip(live, layers):
                xs[k] = l(xs[k])
    return xs


class ProjectionHead(nn.Module):
    """reference custom_layers.py:290-306.  The LeakyReLU modules stay in the Sequential (state_dict indices 0,2,4) but
    are fused into the preceding linear kernel."""

    def __init__(self, channels_list, lr_mul=0.01):
        super().__init__()
        self.num_layers = len(channels_list) - 1
        if self.num_layers > 0:
            mlp = []
            for idx in range(self.num_layers):
                mlp += [EqualizedLinear(channels_list[idx], channels_list[idx + 1], lr_mul=lr_mul)]
                if idx < self.num_layers - 1:
                    mlp += [nn.LeakyReLU(0.2)]
            self.mlp = nn.Sequential(*mlp)

    def forward(self, z):
        mods = list(self.mlp)
        x, i = z, 0
        while i < len(mods):
            fused = i + 1 < len(mods) and isinstance(mods[i + 1], nn.LeakyReLU)
            x = mods[i](x, ACT_LRELU if fused else ACT_NONE)
            i += 2 if fused else 1
        return x

"""Generator / Discriminator of LC-GAN on the MI355X HIP kernels -- drop-in for the reference's `cnn.py`:
same constructor (`args` namespace), forward signatures, attribute names and state_dict layout (cnn.py:7-115).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from . import config, ops
from . import kernels as KM
from .custom_layers import (DiscriminatorBlock, DiscriminatorEpilogue, EqualizedConv2d, MappingNetwork, ProjectionHead, mapping_forward, mapping_matrices,
                            SynthesisBlock, ToRGBBlock)
from .kernels import ACT_LRELU


def _base_nf(res: int) -> int:
    return 32 if res == 1024 else 64 if res == 512 else 128       # cnn.py:17, :54


class Discriminator(torch.nn.Module):
    """reference cnn.py:7-43.  forward(image [B,3,R,R] f32, get_embedding_features) -> (logit [B,1], geo [B,P] | None, app | None)"""

    def __init__(self, args):
        super().__init__()
        self.img_resolution = args.img_resolution
        self.last_block_resolution = 4
        self.log_last_block_resolution = int(np.log2(self.last_block_resolution))
        self.num_blocks = int(np.log2(self.img_resolution)) - self.log_last_block_resolution
        self.geo_projection_dim = args.geo_projection_dim
        self.app_projection_dim = args.app_projection_dim
        self.max_nf = 512
        self.base_nf = _base_nf(self.img_resolution)

        blocks = [EqualizedConv2d(3, self.base_nf, kernel_size=1), nn.LeakyReLU(0.2)]
        out_features = self.base_nf
        for i in range(self.num_blocks):
            in_features = min(self.base_nf * (2 ** i), self.max_nf)
            out_features = min(self.base_nf * (2 ** (i + 1)), self.max_nf)
            blocks += [DiscriminatorBlock(in_features, out_features, skip=True)]
        self.shared_model = nn.Sequential(*blocks)       # children order [conv1x1, LeakyReLU, block...] is relied on by freezeD (worker.py:128-131)
        self.discriminator_epilogue = DiscriminatorEpilogue(out_features, resolution=self.last_block_resolution, mbstd_group_size=8)
        self.logit_mapper = ProjectionHead([out_features, 1])
        self.projection_header1 = ProjectionHead([out_features * 16, out_features * 4, out_features, self.geo_projection_dim])
        self.projection_header2 = ProjectionHead([out_features * 16, out_features * 4, out_features, self.app_projection_dim])

    def forward(self, image, get_embedding_features=False, n_sub=1):
        """n_sub (not in the reference): `image` holds n_sub passes of the reference end to end (e.g. the three views of an even
        iteration, worker.py:163-165); everything is per sample except the minibatch-stddev statistic, which is taken per pass"""
        mods = list(self.shared_model)
        blocks = mods[2:]
        # every block's input arrives with its 2 x 2 average (the skip branch's input, custom_layers.py:202) already made by the kernel
        # that produced it: fromRGB here, then each block's closing convolution
        h, pooled = mods[0].forward_rgb(image.float(), ACT_LRELU, 1.0, pool=True)          # 1x1 conv + LeakyReLU fused (cnn.py:20-21)
        for i, blk in enumerate(blocks):
            if i + 1 < len(blocks):
                h, pooled = blk(h, pooled, want_pool=True)
            else:
                h = blk(h, pooled)
        logit = self.logit_mapper(self.discriminator_epilogue(h, n_sub))
        geometry_embedding = None
        appearance_embedding = None
        if get_embedding_features:
            x = ops.ToNCHWFn.apply(h, h.shape[-1]).flatten(1)            # h.flatten(1) in NCHW order (cnn.py:39)
            geometry_embedding = ops.L2NormalizeFn.apply(self.projection_header1(x))
            appearance_embedding = ops.L2NormalizeFn.apply(self.projection_header2(x))
        return logit, geometry_embedding, appearance_embedding


class Generator(torch.nn.Module):
    """reference cnn.py:46-115.  forward(rand_noise1 [B,geo_noise], rand_noise2 [B,app_noise], w_psi=-1) -> [B,3,R,R] f32"""

    def __init__(self, args):
        super().__init__()
        self.img_resolution = args.img_resolution
        self.first_block_resolution = 4
        self.log_first_block_resolution = int(np.log2(self.first_block_resolution))
        self.num_blocks = int(np.log2(self.img_resolution)) - self.log_first_block_resolution
        self.max_nf = 512
        self.base_nf = _base_nf(self.img_resolution)

        self.geo_latent_dim = args.geo_latent_dim
        self.app_latent_dim = args.app_latent_dim
        self.geo_noise_dim = args.geo_noise_dim
        self.app_noise_dim = args.app_noise_dim
        self.max_flow_scale = args.max_flow_scale

        self.w_avg_beta = 0.998
        self.register_buffer("avg_latent1", torch.zeros([self.geo_latent_dim]))
        self.register_buffer("avg_latent2", torch.zeros([self.app_latent_dim]))

        geometry_channels = [self.geo_noise_dim] + [self.geo_latent_dim] * 12                                    # cnn.py:66-68
        appearance_channels = [self.app_noise_dim, self.app_latent_dim // 4, self.app_latent_dim // 2] + [self.app_latent_dim] * 10   # :70-72
        self.geometry_mapping = MappingNetwork(geometry_channels)
        self.appearance_mapping = MappingNetwork(appearance_channels)
        self.const = torch.nn.Parameter(torch.randn([self.max_nf, self.first_block_resolution, self.first_block_resolution]))
        blocks = []
        in_features = self.max_nf
        out_features, out_resolution = in_features, self.first_block_resolution
        for i in range(self.num_blocks):
            out_features = min(self.base_nf * 2 ** (self.num_blocks - i - 1), self.max_nf)
            out_resolution = 2 ** (self.log_first_block_resolution + 1 + i)
            blocks += [SynthesisBlock(in_features, out_features, self.geo_latent_dim, self.app_latent_dim, out_resolution,
                                      self.max_flow_scale, use_noise=False)]
            in_features = out_features
        self.model = nn.Sequential(*blocks)
        self.rgb_layer = ToRGBBlock(out_features, 3, self.app_latent_dim, out_resolution, use_noise=False)

    def forward(self, rand_noise1, rand_noise2, w_psi=-1.0, n_sub=1):
        """n_sub (not in the reference): the noise batches hold n_sub passes of the reference end to end (the three generator calls of an
        even iteration, worker.py:194-196); the running latent means are then updated once per pass, in order, as n_sub calls would"""
        batch_size = rand_noise1.size(0)
        Lg, La = mapping_matrices((self.geometry_mapping, self.appearance_mapping))      # both QR factorisations in one launch
        # the two mapping chains side by side: their layers of equal shape (the fourth to the twelfth) share launches
        geometry_code, appearance_code = mapping_forward((self.geometry_mapping, self.appearance_mapping),
                                                         (rand_noise1.float(), rand_noise2.float()), (Lg, La))

        if w_psi <= 0:                                   # running latent means (cnn.py:95-97), one tiny kernel each
            assert batch_size % n_sub == 0
            for gc, ac in zip(geometry_code.detach().chunk(n_sub, dim=0), appearance_code.detach().chunk(n_sub, dim=0)):
                KM.K.avg_latent(gc.contiguous(), self.avg_latent1, self.w_avg_beta)
                KM.K.avg_latent(ac.contiguous(), self.avg_latent2, self.w_avg_beta)
        if w_psi > 0.0:                                  # truncation trick (cnn.py:99-101), inference only
            geometry_code = self.avg_latent1.lerp(geometry_code, w_psi)
            appearance_code = self.avg_latent2.lerp(appearance_code, w_psi)

        # every block receives the SAME latents (the reference repeats them, cnn.py:103-104), so the style affines of all
        # layers (custom_layers.py:100,108) are two grouped launches: 6 flow layers on the geometry code, 14 on the appearance code
        blocks = list(self.model)
        g_styles = ops.grouped_linear(geometry_code, [b.flow_layer.linear for b in blocks])
        a_styles = ops.grouped_linear(appearance_code, [l.linear for b in blocks for l in (b.modulated_conv0, b.modulated_conv1)]
                                      + [self.rgb_layer.modulated_conv0.linear, self.rgb_layer.modulated_conv1.linear])
        x = ops.ConstInputFn.apply(self.const, batch_size, config.feature_dtype())     # cnn.py:106
        # ... and the demodulation vectors of all 19 modulated 3x3 layers (custom_layers.py:67) one launch (ops.precompute_demod)
        ops.precompute_demod([(b.flow_layer.modulated_conv.weight.weight, g_styles[i], config.flow_gemm()) for i, b in enumerate(blocks)]
                             + [(l.modulated_conv.weight.weight, a_styles[2 * i + j], False)
                                for i, b in enumerate(blocks) for j, l in enumerate((b.modulated_conv0, b.modulated_conv1))]
                             + [(self.rgb_layer.modulated_conv0.modulated_conv.weight.weight, a_styles[-2], False)],
                             config.feature_dtype() == torch.float32)
        try:
            for i, block in enumerate(blocks):
                x = block(x, (geometry_code,), (appearance_code, appearance_code),
                          styles=(g_styles[i], a_styles[2 * i], a_styles[2 * i + 1]))
            return self.rgb_layer(x, (appearance_code, appearance_code), styles=(a_styles[-2], a_styles[-1]))
        finally:
            ops.clear_demod()

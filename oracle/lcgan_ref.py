"""CPU oracle for the LC-GAN G+D training step  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A functional, fp32, torch-CPU restatement of the arithmetic of rakutentech/lcgan's hot path.  Only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module; the
product package `lcgan_amd` never does (it fails loudly when its HIP library is missing).

Parity status: PINNED.  `oracle/make_golden.py` imports the reference's own Python modules in the build
container, loads numpy-seeded weights into them and stores inputs/outputs under `tests/golden/`;
`tests/test_oracle_golden.py` checks every function here against those vectors.

All parameters live in a flat dict keyed exactly like the reference's `state_dict()` (no `module.` prefix),
e.g. `model.3.modulated_conv0.modulated_conv.weight.weight`.  Every function cites the reference lines it
restates (paths relative to /root/reference).  The contractions themselves execute in PyTorch ATen, the
third-party library the reference calls (custom_layers.py:25,41,43,78,83,137,146,165,275).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]

SQRT2 = math.sqrt(2.0)
SQRT_HALF = math.sqrt(0.5)
W_AVG_BETA = 0.998  # cnn.py:62


# --------------------------------------------------------------------------------------------------
# architecture description (cnn.py:10-31, 49-87)
# --------------------------------------------------------------------------------------------------
def base_nf(res: int) -> int:
    """cnn.py:17,54 -- channel base: 32 @1024, 64 @512, else 128."""
    return 32 if res == 1024 else 64 if res == 512 else 128


def g_channels(res: int):
    """(in, out) channels of every SynthesisBlock, cnn.py:78-85."""
    nb = int(math.log2(res)) - 2
    chans, cin = [], 512
    for i in range(nb):
        cout = min(base_nf(res) * 2 ** (nb - i - 1), 512)
        chans.append((cin, cout))
        cin = cout
    return chans


def d_channels(res: int):
    """(in, out) channels of every DiscriminatorBlock, cnn.py:22-25."""
    nb = int(math.log2(res)) - 2
    return [(min(base_nf(res) * 2 ** i, 512), min(base_nf(res) * 2 ** (i + 1), 512)) for i in range(nb)]


def g_param_shapes(res=256, geo_noise=64, app_noise=64, geo_lat=64, app_lat=512):
    """Every Generator state_dict entry (params + 2 buffers) with its shape, cnn.py:46-87."""
    s = {"const": (512, 4, 4), "avg_latent1": (geo_lat,), "avg_latent2": (app_lat,)}
    geo = [geo_noise] + [geo_lat] * 12                                   # cnn.py:66-68
    app = [app_noise, app_lat // 4, app_lat // 2] + [app_lat] * 10        # cnn.py:70-72
    for name, ch in (("geometry_mapping", geo), ("appearance_mapping", app)):
        s[f"{name}.diagonal_params"] = (ch[0],)
        s[f"{name}.basis_params"] = (ch[0], ch[0])
        for i in range(12):
            s[f"{name}.mlp.{i}.weight.weight"] = (ch[i + 1], ch[i])
            s[f"{name}.mlp.{i}.bias"] = (ch[i + 1],)

    def synth(prefix, cin, cout, lat, k):
        s[f"{prefix}.linear.weight.weight"] = (cin, lat)
        s[f"{prefix}.linear.bias"] = (cin,)
        s[f"{prefix}.modulated_conv.weight.weight"] = (cout, cin, k, k)
        s[f"{prefix}.modulated_conv.bias"] = (cout,)

    for b, (cin, cout) in enumerate(g_channels(res)):
        synth(f"model.{b}.modulated_conv0", cin, cout, app_lat, 3)
        synth(f"model.{b}.modulated_conv1", cout, cout, app_lat, 3)
        s[f"model.{b}.skip_layer.weight.weight"] = (cout, cin, 1, 1)
        synth(f"model.{b}.flow_layer", cin, 2, geo_lat, 3)
    c_last = g_channels(res)[-1][1]
    synth("rgb_layer.modulated_conv0", c_last, c_last, app_lat, 3)
    synth("rgb_layer.modulated_conv1", c_last, 3, app_lat, 1)
    return s


def d_param_shapes(res=256, geo_proj=256, app_proj=256):
    """Every Discriminator state_dict entry with its shape, cnn.py:7-31."""
    s = {"shared_model.0.weight.weight": (base_nf(res), 3, 1, 1), "shared_model.0.bias": (base_nf(res),)}
    for b, (cin, cout) in enumerate(d_channels(res)):
        p = f"shared_model.{b + 2}"
        s[f"{p}.conv0.weight.weight"] = (cin, cin, 3, 3)
        s[f"{p}.conv0.bias"] = (cin,)
        s[f"{p}.conv1.weight.weight"] = (cout, cin, 3, 3)
        s[f"{p}.conv1.bias"] = (cout,)
        s[f"{p}.skip_layer.weight.weight"] = (cout, cin, 1, 1)
    c = d_channels(res)[-1][1]
    s["discriminator_epilogue.conv.weight.weight"] = (c, c + 1, 3, 3)
    s["discriminator_epilogue.conv.bias"] = (c,)
    s["discriminator_epilogue.linear.weight.weight"] = (c, c * 16)
    s["discriminator_epilogue.linear.bias"] = (c,)
    s["logit_mapper.mlp.0.weight.weight"] = (1, c)
    s["logit_mapper.mlp.0.bias"] = (1,)
    for h, proj in (("projection_header1", geo_proj), ("projection_header2", app_proj)):
        dims = [c * 16, c * 4, c, proj]
        for j in range(3):
            s[f"{h}.mlp.{2 * j}.weight.weight"] = (dims[j + 1], dims[j])
            s[f"{h}.mlp.{2 * j}.bias"] = (dims[j + 1],)
    return s


G_BUFFERS = ("avg_latent1", "avg_latent2")

# keys whose EqualizedLinear uses lr_mul = 0.01 (custom_layers.py:226, :260, :291)
_LR001 = ("geometry_mapping.mlp", "appearance_mapping.mlp", "logit_mapper.mlp", "projection_header1.mlp",
          "projection_header2.mlp", "discriminator_epilogue.linear")


def lr_mul_of(key: str) -> float:
    return 0.01 if key.startswith(_LR001) else 1.0


# --------------------------------------------------------------------------------------------------
# layers
# --------------------------------------------------------------------------------------------------
def eq_scale(w: Tensor, lr_mul: float = 1.0) -> Tensor:
    """EqualizedWeight.forward, custom_layers.py:10,14: W * lr_mul / sqrt(fan_in)."""
    return w * (lr_mul / math.sqrt(w[0].numel()))


def eq_linear(P: Params, prefix: str, x: Tensor) -> Tensor:
    """EqualizedLinear.forward, custom_layers.py:24-25."""
    lm = lr_mul_of(prefix)
    return F.linear(x, eq_scale(P[prefix + ".weight.weight"], lm), P[prefix + ".bias"] * lm)


def eq_conv(P: Params, prefix: str, x: Tensor, stride: int = 1) -> Tensor:
    """EqualizedConv2d.forward, custom_layers.py:39-44 (bias optional, padding k//2, lr_mul 1)."""
    w = P[prefix + ".weight.weight"]
    b = P.get(prefix + ".bias")
    return F.conv2d(x, eq_scale(w), b, stride=stride, padding=w.shape[-1] // 2)


def box3(x: Tensor) -> Tensor:
    """box_filter, custom_layers.py:136-138 / :196-198 (count_include_pad=True: borders divide by 9)."""
    return F.avg_pool2d(x, 3, 1, 1)


def modulated_conv(x: Tensor, w: Tensor, bias: Tensor, s: Tensor, up: int, eps: float = 1e-8) -> Tensor:
    """ModulatedConv2d.forward, custom_layers.py:60-86: modulate (:62-64), demodulate (:67-68), grouped
    conv (:83) or grouped stride-2 transposed conv with padding 1 / output_padding 1 (:73-80), + bias (:85)."""
    B, C, H, W = x.shape
    O, _, k, _ = w.shape
    wm = eq_scale(w)[None] * s[:, None, :, None, None]                       # [B,O,C,k,k]
    wm = wm * torch.rsqrt(wm.square().sum(dim=(2, 3, 4), keepdim=True) + eps)
    xg = x.reshape(1, B * C, H, W)
    if up > 1:
        wt = wm.transpose(1, 2).reshape(B * C, O, k, k)
        y = F.conv_transpose2d(xg, wt, stride=up, padding=(k - 1) // 2, output_padding=1, groups=B)
    else:
        y = F.conv2d(xg, wm.reshape(B * O, C, k, k), padding=(k - 1) // 2, groups=B)
    return y.reshape(B, O, y.shape[-2], y.shape[-1]) + bias.view(1, -1, 1, 1)


def synthesis_layer(P: Params, prefix: str, x: Tensor, latent: Tensor, up: int) -> Tensor:
    """SynthesisLayer.forward, custom_layers.py:103-106 (use_noise is always False, cnn.py:83,87)."""
    s = F.linear(latent, eq_scale(P[prefix + ".linear.weight.weight"]), P[prefix + ".linear.bias"])
    return modulated_conv(x, P[prefix + ".modulated_conv.weight.weight"], P[prefix + ".modulated_conv.bias"], s, up)


def base_grid(h: int, w: int) -> Tensor:
    """get_coordinates, custom_layers.py:127-134: [2,h,w] = (x, y) in [-1,1] with the align_corners=True formula."""
    gy, gx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    return torch.stack((2 * gx / (w - 1) - 1, 2 * gy / (h - 1) - 1))


def warp_bicubic(x: Tensor, flow: Tensor, max_flow_scale: float) -> Tensor:
    """custom_layers.py:162-165: grid = coords + flow*scale; bicubic grid_sample with the DEFAULT
    align_corners=False and zero padding."""
    grid = base_grid(x.shape[-2], x.shape[-1])[None] + flow * max_flow_scale
    return F.grid_sample(x, grid.permute(0, 2, 3, 1), mode="bicubic", padding_mode="zeros", align_corners=False)


def synthesis_block(P: Params, prefix: str, x: Tensor, g_lat: Tensor, a_lat0: Tensor, a_lat1: Tensor,
                    max_flow_scale: float) -> Tensor:
    """SynthesisBlock.forward, custom_layers.py:140-166."""
    skip = eq_conv(P, prefix + ".skip_layer", x) * SQRT_HALF                       # :145
    skip = box3(F.interpolate(skip, scale_factor=2, mode="nearest"))             # :146-147
    flow = torch.tanh(box3(synthesis_layer(P, prefix + ".flow_layer", x, g_lat, 2)))  # :149-151
    h = synthesis_layer(P, prefix + ".modulated_conv0", x, a_lat0, 2)            # :153
    h = F.leaky_relu(box3(h), 0.2) * SQRT2                                        # :154-155
    h = F.leaky_relu(synthesis_layer(P, prefix + ".modulated_conv1", h, a_lat1, 1), 0.2)  # :157-158
    return warp_bicubic(skip + h, flow, max_flow_scale)                           # :159-165


def mapping_matrix(P: Params, prefix: str) -> Tensor:
    """MappingNetwork.forward, custom_layers.py:280-282: L = Q(tanh(basis)) @ diag(|d| + 1e-6)."""
    q = torch.linalg.qr(torch.tanh(P[prefix + ".basis_params"]), mode="reduced")[0]   # torch.qr(...)[0], :275
    return q * (P[prefix + ".diagonal_params"].abs() + 1e-6)[None, :]


def mapping_network(P: Params, prefix: str, z: Tensor) -> Tensor:
    """MappingNetwork.forward, custom_layers.py:278-287: x = L z, then 12 EqualizedLinear(lr_mul .01), no activations."""
    x = z @ mapping_matrix(P, prefix).t()
    i = 0
    while f"{prefix}.mlp.{i}.bias" in P:      # 12 layers in the Generator (cnn.py:66-72)
        x = eq_linear(P, f"{prefix}.mlp.{i}", x)
        i += 1
    return x


def generator_forward(P: Params, z1: Tensor, z2: Tensor, res: int, w_psi: float = -1.0,
                      max_flow_scale: float = 0.1, update_avg: bool = True) -> Tensor:
    """Generator.forward, cnn.py:89-115.  Mutates P['avg_latent1/2'] when w_psi <= 0 (cnn.py:95-97)."""
    w_geo = mapping_network(P, "geometry_mapping", z1)
    w_app = mapping_network(P, "appearance_mapping", z2)
    if w_psi <= 0 and update_avg:
        with torch.no_grad():
            P["avg_latent1"] = w_geo.detach().mean(0).lerp(P["avg_latent1"], W_AVG_BETA)
            P["avg_latent2"] = w_app.detach().mean(0).lerp(P["avg_latent2"], W_AVG_BETA)
    if w_psi > 0:
        w_geo = P["avg_latent1"].lerp(w_geo, w_psi)                                # cnn.py:99-101
        w_app = P["avg_latent2"].lerp(w_app, w_psi)
    x = P["const"][None].expand(z1.shape[0], -1, -1, -1)                           # cnn.py:106
    nb = int(math.log2(res)) - 2
    for b in range(nb):
        x = synthesis_block(P, f"model.{b}", x, w_geo, w_app, w_app, max_flow_scale)   # same latent for every layer (cnn.py:103-104)
    h = F.leaky_relu(synthesis_layer(P, "rgb_layer.modulated_conv0", x, w_app, 1), 0.2)   # custom_layers.py:179-180
    return synthesis_layer(P, "rgb_layer.modulated_conv1", h, w_app, 1)           # :181 (1x1, still demodulated)


def minibatch_std(x: Tensor, group_size: int = 8) -> Tensor:
    """MinibatchStdLayer.forward, custom_layers.py:243-256 (num_channels=1; strided grouping: reshape(G,-1,...))."""
    N, C, H, W = x.shape
    G = min(group_size, N)
    y = x.reshape(G, -1, 1, C, H, W)
    y = y - y.mean(dim=0)
    y = (y.square().mean(dim=0) + 1e-8).sqrt()
    y = y.mean(dim=[2, 3, 4]).reshape(-1, 1, 1, 1).repeat(G, 1, H, W)
    return torch.cat([x, y], dim=1)


def discriminator_block(P: Params, prefix: str, x: Tensor) -> Tensor:
    """DiscriminatorBlock.forward with skip=True, custom_layers.py:200-209."""
    skip = eq_conv(P, prefix + ".skip_layer", F.avg_pool2d(x, 2, 2)) * SQRT_HALF
    h = F.leaky_relu(eq_conv(P, prefix + ".conv0", x), 0.2) * SQRT2
    h = F.leaky_relu(eq_conv(P, prefix + ".conv1", box3(h), stride=2), 0.2)
    return skip + h


def projection_head(P: Params, prefix: str, x: Tensor, n_layers: int) -> Tensor:
    """ProjectionHead.forward, custom_layers.py:290-306: linear (+LeakyReLU between, not after the last)."""
    for j in range(n_layers):
        x = eq_linear(P, f"{prefix}.mlp.{2 * j}", x)
        if j < n_layers - 1:
            x = F.leaky_relu(x, 0.2)
    return x


def discriminator_forward(P: Params, img: Tensor, res: int, heads: bool = False
                          ) -> Tuple[Tensor, Optional[Tensor], Optional[Tensor]]:
    """Discriminator.forward, cnn.py:33-43."""
    h = F.leaky_relu(eq_conv(P, "shared_model.0", img), 0.2)                        # cnn.py:20-21
    for b in range(int(math.log2(res)) - 2):
        h = discriminator_block(P, f"shared_model.{b + 2}", h)
    e = F.leaky_relu(eq_conv(P, "discriminator_epilogue.conv", minibatch_std(h, 8)), 0.2)   # custom_layers.py:229-231
    e = F.leaky_relu(eq_linear(P, "discriminator_epilogue.linear", e.flatten(1)), 0.2)    # :232-233
    logit = projection_head(P, "logit_mapper", e, 1)
    if not heads:
        return logit, None, None
    flat = h.flatten(1)
    return (logit, F.normalize(projection_head(P, "projection_header1", flat, 3)),   # cnn.py:40-41
            F.normalize(projection_head(P, "projection_header2", flat, 3)))


# --------------------------------------------------------------------------------------------------
# losses (loss.py) and EMA (ema.py)
# --------------------------------------------------------------------------------------------------
def contrastive_loss(anchor: Tensor, pos: Tensor, neg: Tensor, tau: float) -> Tensor:
    """loss.py:9-15: -log(e^{a.p/tau} / (e^{a.p/tau} + e^{a.n/tau})), mean over the batch."""
    ep = torch.exp((anchor * pos).sum(1) / tau)
    en = torch.exp((anchor * neg).sum(1) / tau)
    return (-torch.log(ep / (ep + en))).mean()


def r1_penalty(real_logit: Tensor, images: Tensor) -> Tensor:
    """loss.py:18-34: 0.5 * mean_b sum (d sum(logit) / d image)^2, differentiable (create_graph=True)."""
    g = torch.autograd.grad(real_logit.sum(), images, create_graph=True, retain_graph=True)[0]
    return 0.5 * g.square().flatten(1).sum(1).mean(0) + images[:, 0, 0, 0].mean() * 0


def bce_logits(logit: Tensor, target_one: bool) -> Tensor:
    """F.binary_cross_entropy_with_logits against all-ones / all-zeros labels, worker.py:156-157,191."""
    return F.softplus(-logit).mean() if target_one else F.softplus(logit).mean()


def ema_update(src: Params, tgt: Params, decay: float, it: int, start_iter: int = 0) -> None:
    """Ema.update, ema.py:19-32: p_ema <- lerp(p, p_ema, decay) for parameters AND buffers."""
    d = 0.0 if (0 <= it < start_iter) else decay
    with torch.no_grad():
        for k in tgt:
            tgt[k] = src[k].detach().lerp(tgt[k], d)


# --------------------------------------------------------------------------------------------------
# training step (worker.py:137-214) as pure functions returning losses and gradients
# --------------------------------------------------------------------------------------------------
class Hyper:
    """The flags of main.py:19-38 that the step reads."""
    tau = 0.05
    l_aux = 0.5
    l_r1 = 10.0
    l_s = 1e-7
    max_flow_scale = 0.1


def _leaves(P: Params, skip=()) -> Params:
    return {k: (v.detach().clone().requires_grad_(True) if k not in skip else v.detach().clone()) for k, v in P.items()}


def g_step(GP: Params, DP: Params, res: int, epoch: int, z: Tuple[Tensor, Tensor, Tensor, Tensor], hp=Hyper):
    """train_generator, worker.py:179-214.  z = (rand1, rand2, resample1, resample2).  Returns
    (g_loss, grads-of-G dict, updated avg-latent buffers)."""
    G = _leaves(GP, skip=G_BUFFERS)
    D = {k: v.detach() for k, v in DP.items()}
    r1, r2, s1, s2 = z
    gen = lambda a, b: generator_forward(G, a, b, res, max_flow_scale=hp.max_flow_scale)
    if epoch % 2 == 1:
        logit, _, _ = discriminator_forward(D, gen(r1, r2), res, False)
        loss = bce_logits(logit, True)
        parts = {"adv": loss.detach()}
    else:
        imgs = (gen(r1, r2), gen(s1, r2), gen(r1, s2))                                # worker.py:194-196
        logit, gf, af = discriminator_forward(D, imgs[0], res, True)
        _, gp, an = discriminator_forward(D, imgs[1], res, True)
        _, gn, ap = discriminator_forward(D, imgs[2], res, True)
        adv = bce_logits(logit, True)
        aux = (contrastive_loss(gf, gp, gn, hp.tau) + contrastive_loss(af, ap, an, hp.tau)) * hp.l_aux
        spars = torch.cat([G["geometry_mapping.diagonal_params"], G["appearance_mapping.diagonal_params"]]).abs().sum() * hp.l_s
        loss = adv + aux + spars
        parts = {"adv": adv.detach(), "aux": aux.detach(), "sparsity": spars.detach()}
    loss.backward()
    grads = {k: v.grad for k, v in G.items() if k not in G_BUFFERS and v.grad is not None}
    bufs = {k: G[k].detach() for k in G_BUFFERS}
    return loss.detach(), grads, bufs, parts


def d_step(GP: Params, DP: Params, res: int, epoch: int, z: Tuple[Tensor, Tensor], real: Tuple[Tensor, Tensor, Tensor],
           hp=Hyper, frozen=()):
    """train_discriminator, worker.py:137-177.  real = (image, geometry_change, appearance_change).
    `frozen` = key prefixes excluded by freeze_discriminator (worker.py:127-131).  Returns
    (d_loss, grads-of-D dict (absent = grad None), updated avg-latent buffers, parts)."""
    G = {k: v.detach().clone() for k, v in GP.items()}
    D = {k: (v.detach().clone().requires_grad_(not k.startswith(tuple(frozen)) if frozen else True)) for k, v in DP.items()}
    fake = generator_forward(G, z[0], z[1], res, max_flow_scale=hp.max_flow_scale).detach()
    fake_logit, _, _ = discriminator_forward(D, fake, res, False)
    image, geo_c, app_c = real
    parts = {}
    if epoch % 2 == 1:
        image = image.detach().clone().requires_grad_(True)                           # worker.py:152
        real_logit, _, _ = discriminator_forward(D, image, res, False)
        loss = bce_logits(real_logit, True) + bce_logits(fake_logit, False)
        parts["adv"] = loss.detach()
        if epoch % 8 == 1:
            r1 = r1_penalty(real_logit, image)
            parts["r1"] = r1.detach()
            loss = loss + r1 * hp.l_r1
    else:
        real_logit, gf, af = discriminator_forward(D, image, res, True)
        _, gp, an = discriminator_forward(D, geo_c, res, True)
        _, gn, ap = discriminator_forward(D, app_c, res, True)
        adv = bce_logits(real_logit, True) + bce_logits(fake_logit, False)
        aux = (contrastive_loss(gf, gp, gn, hp.tau) + contrastive_loss(af, ap, an, hp.tau)) * hp.l_aux
        loss = adv + aux
        parts.update(adv=adv.detach(), aux=aux.detach())
    loss.backward()
    grads = {k: v.grad for k, v in D.items() if v.grad is not None}
    bufs = {k: G[k].detach() for k in G_BUFFERS}
    return loss.detach(), grads, bufs, parts


def adam_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float, b1=0.0, b2=0.99, eps=1e-8):
    """torch.optim.Adam single-tensor update as configured at worker.py:98-110 (no weight decay, no amsgrad).
    `step` is the 1-based count AFTER this update.  Returns (p, m, v)."""
    m = m * b1 + g * (1 - b1)
    v = v * b2 + g * g * (1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    return p - (lr / bc1) * m / denom, m, v

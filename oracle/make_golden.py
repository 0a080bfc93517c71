"""Golden-vector generator (TEST INFRASTRUCTURE; runs ONLY in the build container).

Imports the reference's own Python modules from /root/reference (cnn, custom_layers, loss, ema -- the
hot-path modules; worker/loader/main need torchvision/albumentations/av which are absent, so the few
lines of worker.py:137-214 that sequence a step are driven from here on the reference modules),
loads numpy-seeded weights (oracle/weights.py) with load_state_dict, and writes small .npz fixtures to
tests/golden/.  Nothing of the reference travels: fixtures hold seeds, inputs and expected outputs only.

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden            # from the repo root
"""
from __future__ import annotations

import os
import sys
import types
import warnings

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"

from . import lcgan_ref as O          # noqa: E402  (shapes only; the numbers below come from the reference)
from .weights import grad_stats, seeded_state, seeded_tensor   # noqa: E402


def _import_reference():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    warnings.filterwarnings("ignore")
    import cnn, custom_layers, loss, ema   # noqa: E401
    return cnn, custom_layers, loss, ema


def sample(t: torch.Tensor, n: int = 257) -> np.ndarray:
    """Strided sample (<= n values) of a tensor -- what the fixtures store for big outputs."""
    f = t.detach().reshape(-1)
    step = max(1, f.numel() // n)
    return f[::step][:n].contiguous().numpy().copy()


def summarize(prefix: str, t: torch.Tensor, d: dict):
    d[prefix + "/sample"] = sample(t)
    d[prefix + "/sum"] = np.float64(t.detach().double().sum().item())
    st = grad_stats(t, prefix.split("/grad/")[-1].split("/r1grad/")[-1])        # projections are keyed by the PARAMETER name
    d[prefix + "/abssum"] = np.float64(st["abssum"])
    d[prefix + "/l2"] = np.float64(st["l2"])
    d[prefix + "/proj"] = st["proj"]


def load(module, state):
    sd = module.state_dict()
    assert set(sd) == set(state), (sorted(set(sd) ^ set(state)))
    for k in sd:
        assert tuple(sd[k].shape) == tuple(state[k].shape), k
    module.load_state_dict({k: v.clone() for k, v in state.items()})
    return module


def module_state(module, seed):
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    return seeded_state(shapes, seed)


def args_for(res):
    a = types.SimpleNamespace()
    a.img_resolution = res
    a.geo_latent_dim, a.app_latent_dim = 64, 512
    a.geo_noise_dim, a.app_noise_dim = 64, 64
    a.max_flow_scale = 0.1
    a.geo_projection_dim, a.app_projection_dim = 256, 256
    return a


# ---------------------------------------------------------------------------------------------
def layer_fixtures(CL, LOSS, EMA):
    d = {}
    torch.manual_seed(0)

    # ModulatedConv2d: k=3 up=1, k=3 up=2, k=1 up=1          (custom_layers.py:47-86)
    for name, (ci, co, k, up, hw) in {"modconv_k3": (8, 16, 3, 1, 6), "modconv_up": (8, 16, 3, 2, 5),
                                      "modconv_k1": (16, 3, 1, 1, 6)}.items():
        m = CL.ModulatedConv2d(ci, co, k, up=up)
        load(m, module_state(m, 11))
        x = seeded_tensor((3, ci, hw, hw), 12).requires_grad_(True)
        s = (seeded_tensor((3, ci), 13) * 0.5 + 1).requires_grad_(True)
        y = m(x, s)
        go = seeded_tensor(tuple(y.shape), 14)
        gx, gs, gw, gb = torch.autograd.grad((y * go).sum(), [x, s, m.weight.weight, m.bias])
        for n, t in (("y", y), ("gx", gx), ("gs", gs), ("gw", gw), ("gb", gb)):
            d[f"{name}/{n}"] = t.detach().numpy()

    # SynthesisBlock                                          (custom_layers.py:114-166)
    m = CL.SynthesisBlock(16, 8, 6, 10, 10, 0.1)
    st = module_state(m, 21)
    load(m, st)
    x = seeded_tensor((2, 16, 5, 5), 22).requires_grad_(True)
    gl = seeded_tensor((2, 1, 6), 23).requires_grad_(True)
    al = seeded_tensor((2, 2, 10), 24).requires_grad_(True)
    y = m(x, gl, al)
    go = seeded_tensor(tuple(y.shape), 25)
    params = dict(m.named_parameters())
    grads = torch.autograd.grad((y * go).sum(), [x, gl, al] + list(params.values()))
    d["synblock/y"] = y.detach().numpy()
    d["synblock/gx"], d["synblock/ggl"], d["synblock/gal"] = (g.numpy() for g in grads[:3])
    for (k, _), g in zip(params.items(), grads[3:]):
        d[f"synblock/grad/{k}"] = g.numpy()

    # ToRGBBlock                                              (custom_layers.py:169-182)
    m = CL.ToRGBBlock(8, 3, 10, 8)
    load(m, module_state(m, 31))
    x = seeded_tensor((2, 8, 6, 6), 32).requires_grad_(True)
    al = seeded_tensor((2, 2, 10), 33).requires_grad_(True)
    y = m(x, al)
    go = seeded_tensor(tuple(y.shape), 34)
    params = dict(m.named_parameters())
    grads = torch.autograd.grad((y * go).sum(), [x, al] + list(params.values()))
    d["torgb/y"] = y.detach().numpy()
    d["torgb/gx"], d["torgb/gal"] = grads[0].numpy(), grads[1].numpy()
    for (k, _), g in zip(params.items(), grads[2:]):
        d[f"torgb/grad/{k}"] = g.numpy()

    # DiscriminatorBlock incl. R1-style double backward       (custom_layers.py:185-217, loss.py:18-34)
    m = CL.DiscriminatorBlock(8, 16, skip=True)
    load(m, module_state(m, 41))
    x = seeded_tensor((2, 8, 8, 8), 42).requires_grad_(True)
    y = m(x)
    go = seeded_tensor(tuple(y.shape), 43)
    gx = torch.autograd.grad((y * go).sum(), x, create_graph=True)[0]
    params = dict(m.named_parameters())
    g2 = torch.autograd.grad(gx.square().sum(), list(params.values()), retain_graph=True, allow_unused=True)
    g1 = torch.autograd.grad((y * go).sum(), list(params.values()), allow_unused=True)
    d["dblock/y"], d["dblock/gx"] = y.detach().numpy(), gx.detach().numpy()
    for (k, _), a, b in zip(params.items(), g1, g2):
        d[f"dblock/grad1/{k}"] = a.numpy()
        d[f"dblock/grad2/{k}"] = (torch.zeros_like(params[k]) if b is None else b).numpy()

    # MinibatchStdLayer, three batch sizes, first + second order (custom_layers.py:237-256)
    for N in (4, 8, 16):
        m = CL.MinibatchStdLayer(group_size=8)
        x = seeded_tensor((N, 6, 4, 4), 50 + N).requires_grad_(True)
        y = m(x)
        go = seeded_tensor(tuple(y.shape), 51 + N)
        gx = torch.autograd.grad((y * go).sum(), x, create_graph=True)[0]
        v = seeded_tensor(tuple(x.shape), 52 + N)
        ggx = torch.autograd.grad((gx * v).sum(), x)[0]
        d[f"mbstd{N}/y"], d[f"mbstd{N}/gx"], d[f"mbstd{N}/ggx"] = y.detach().numpy(), gx.detach().numpy(), ggx.numpy()

    # DiscriminatorEpilogue                                   (custom_layers.py:220-234)
    m = CL.DiscriminatorEpilogue(8, resolution=4, mbstd_group_size=8)
    load(m, module_state(m, 61))
    x = seeded_tensor((8, 8, 4, 4), 62).requires_grad_(True)
    y = m(x)
    go = seeded_tensor(tuple(y.shape), 63)
    params = dict(m.named_parameters())
    grads = torch.autograd.grad((y * go).sum(), [x] + list(params.values()))
    d["depi/y"], d["depi/gx"] = y.detach().numpy(), grads[0].numpy()
    for (k, _), g in zip(params.items(), grads[1:]):
        d[f"depi/grad/{k}"] = g.numpy()

    # MappingNetwork                                          (custom_layers.py:259-287)
    m = CL.MappingNetwork([6, 8, 8, 12])
    load(m, module_state(m, 71))
    z = seeded_tensor((5, 6), 72).requires_grad_(True)
    y = m(z)
    go = seeded_tensor(tuple(y.shape), 73)
    params = dict(m.named_parameters())
    grads = torch.autograd.grad((y * go).sum(), [z] + list(params.values()))
    d["mapping/y"], d["mapping/gz"] = y.detach().numpy(), grads[0].numpy()
    for (k, _), g in zip(params.items(), grads[1:]):
        d[f"mapping/grad/{k}"] = g.numpy()

    # ProjectionHead                                          (custom_layers.py:290-306)
    m = CL.ProjectionHead([16, 12, 8, 4])
    load(m, module_state(m, 81))
    z = seeded_tensor((5, 16), 82)
    d["phead/y"] = m(z).detach().numpy()

    # losses                                                  (loss.py:9-34)
    a, p, n = (F.normalize(seeded_tensor((6, 16), 90 + i)).requires_grad_(True) for i in range(3))
    l = LOSS.contrastive_loss(a, p, n, 0.05)
    ga, gp, gn = torch.autograd.grad(l, [a, p, n])
    d["closs/l"], d["closs/ga"], d["closs/gp"], d["closs/gn"] = l.detach().numpy(), ga.numpy(), gp.numpy(), gn.numpy()

    # Ema                                                     (ema.py:4-32)
    src, tgt = CL.MappingNetwork([4, 4, 4]), CL.MappingNetwork([4, 4, 4])
    load(src, module_state(src, 95)), load(tgt, module_state(tgt, 96))
    src.register_buffer("buf", seeded_tensor((3,), 97)), tgt.register_buffer("buf", seeded_tensor((3,), 98))
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        e = EMA.Ema(src, tgt, decay=0.9, start_iter=2)
    load(src, {**module_state(tgt, 99), "buf": seeded_tensor((3,), 100)})   # move the source after the ctor copy
    e.update(1)   # < start_iter -> decay 0
    d["ema/after_it1"] = torch.cat([v.reshape(-1) for v in tgt.state_dict().values()]).numpy()
    load(src, {**module_state(tgt, 101), "buf": seeded_tensor((3,), 102)})
    e.update(5)
    d["ema/after_it5"] = torch.cat([v.reshape(-1) for v in tgt.state_dict().values()]).numpy()
    np.savez_compressed(os.path.join(OUT, "layers.npz"), **d)
    print("layers.npz", len(d), "arrays")


def narrow_fixtures(CL):
    """Backward fixtures of the NARROW octaves (C = 32 / 64: the high-resolution layers of the 512 x 512 / 1024 x 1024 networks, cnn.py:17,54):
    a DiscriminatorBlock(32, 64) at 64 x 64 with first-order gradients and the R1-style double backward (custom_layers.py:185-217,
    loss.py:18-34), and a SynthesisBlock(64 -> 32) at 32 x 32 -> 64 x 64 with every gradient (custom_layers.py:114-166).  Big tensors are
    stored as strided slices / the kink-robust statistics of oracle/weights.py:grad_stats."""
    d = {}

    def stat(prefix, t, key):
        st = grad_stats(t, key)
        d[prefix + "/abssum"], d[prefix + "/l2"], d[prefix + "/proj"] = np.float64(st["abssum"]), np.float64(st["l2"]), st["proj"]

    B, C, R = 2, 32, 64
    m = CL.DiscriminatorBlock(C, 2 * C, skip=True)
    load(m, module_state(m, 141))
    x = seeded_tensor((B, C, R, R), 142).requires_grad_(True)
    y = m(x)
    go = seeded_tensor(tuple(y.shape), 143)
    gx = torch.autograd.grad((y * go).sum(), x, create_graph=True)[0]
    params = dict(m.named_parameters())
    g2 = torch.autograd.grad(gx.square().sum(), list(params.values()), retain_graph=True, allow_unused=True)
    g1 = torch.autograd.grad((y * go).sum(), list(params.values()), allow_unused=True)
    d["dblock/B"], d["dblock/C"], d["dblock/R"], d["dblock/g2_scale"] = np.int64(B), np.int64(C), np.int64(R), np.float64(1.0)
    d["dblock/y_slice"] = y.detach()[:, :, ::4, ::4].numpy().copy()
    stat("dblock/gx", gx.detach(), "dblock/gx")
    for (k, p), a, b in zip(params.items(), g1, g2):
        stat(f"dblock/grad1/{k}", a, k)
        stat(f"dblock/grad2/{k}", torch.zeros_like(p) if b is None else b, k)

    B, Ci, Co, R = 2, 64, 32, 32
    m = CL.SynthesisBlock(Ci, Co, 64, 512, 2 * R, 0.1)
    load(m, module_state(m, 121))
    x = seeded_tensor((B, Ci, R, R), 122).requires_grad_(True)
    gl = seeded_tensor((B, 1, 64), 123).requires_grad_(True)
    al = seeded_tensor((B, 2, 512), 124).requires_grad_(True)
    y = m(x, gl, al)
    go = seeded_tensor(tuple(y.shape), 125)
    params = dict(m.named_parameters())
    grads = torch.autograd.grad((y * go).sum(), [x, gl, al] + list(params.values()))
    d["synblock/B"], d["synblock/Ci"], d["synblock/Co"], d["synblock/R"] = np.int64(B), np.int64(Ci), np.int64(Co), np.int64(R)
    d["synblock/y_slice"] = y.detach()[:, :, ::4, ::4].numpy().copy()
    stat("synblock/gx", grads[0], "synblock/gx")
    d["synblock/ggl"], d["synblock/gal"] = grads[1].numpy(), grads[2].numpy()
    for (k, _), g in zip(params.items(), grads[3:]):
        stat(f"synblock/grad/{k}", g, k)
    np.savez_compressed(os.path.join(OUT, "narrow.npz"), **d)
    print("narrow.npz", len(d), "arrays")


# ---------------------------------------------------------------------------------------------
HP = dict(tau=0.05, l_aux=0.5, l_r1=10.0, l_s=1e-7)


def step_fixtures(CNN, LOSS, res=32, B=8):
    """One G step and one D step of every iteration type (worker.py:137-214 sequencing, restated here on the
    reference's modules because worker.py itself cannot be imported)."""
    d = {"res": np.int64(res), "B": np.int64(B)}
    a = args_for(res)
    G, D = CNN.Generator(a), CNN.Discriminator(a)
    gstate, dstate = module_state(G, 1001), module_state(D, 1002)
    d["g_keys"] = np.array(sorted(gstate))
    d["d_keys"] = np.array(sorted(dstate))
    d["g_shapes"] = np.array([str(tuple(gstate[k].shape)) for k in sorted(gstate)])
    d["d_shapes"] = np.array([str(tuple(dstate[k].shape)) for k in sorted(dstate)])
    ones = lambda: torch.ones(B, 1)
    zeros = lambda: torch.zeros(B, 1)
    bce = F.binary_cross_entropy_with_logits

    def fresh():
        load(G, gstate), load(D, dstate)
        for p in list(G.parameters()) + list(D.parameters()):
            p.grad = None
            p.requires_grad_(True)

    zs = [seeded_tensor((B, 64), 2000 + i) for i in range(4)]
    real = [seeded_tensor((B, 3, res, res), 2100 + i, "uniform_pm1") for i in range(3)]

    for epoch in (0, 1):                                         # train_generator, worker.py:179-214
        fresh()
        for p in D.parameters():
            p.requires_grad_(False)
        r1, r2, s1, s2 = zs
        if epoch % 2 == 1:
            logit, _, _ = D(G(r1, r2), False)
            loss = bce(logit, ones())
        else:
            i0, i1, i2 = G(r1, r2), G(s1, r2), G(r1, s2)
            logit, gf, af = D(i0, True)
            _, gp, an = D(i1, True)
            _, gn, ap = D(i2, True)
            aux = (LOSS.contrastive_loss(gf, gp, gn, HP["tau"]) + LOSS.contrastive_loss(af, ap, an, HP["tau"])) * HP["l_aux"]
            sp = torch.norm(torch.cat([G.geometry_mapping.diagonal_params.view(-1),
                                       G.appearance_mapping.diagonal_params.view(-1)]), p=1) * HP["l_s"]
            loss = bce(logit, ones()) + aux + sp
            d[f"g{epoch}/aux"], d[f"g{epoch}/sparsity"] = aux.detach().numpy(), sp.detach().numpy()
        loss.backward()
        d[f"g{epoch}/loss"] = loss.detach().numpy()
        for k, p in G.named_parameters():
            summarize(f"g{epoch}/grad/{k}", p.grad, d)
        d[f"g{epoch}/avg_latent1"], d[f"g{epoch}/avg_latent2"] = G.avg_latent1.numpy().copy(), G.avg_latent2.numpy().copy()

    for epoch, frozen in ((0, 0), (1, 0), (3, 0), (1, 2)):       # train_discriminator, worker.py:137-177
        tag = f"d{epoch}" + (f"_freeze{frozen}" if frozen else "")
        fresh()
        for p in G.parameters():
            p.requires_grad_(False)
        if frozen:                                               # freeze_discriminator, worker.py:127-131
            for i, (_, layer) in enumerate(D.shared_model.named_children()):
                if i < frozen + 2:
                    for p in layer.parameters():
                        p.requires_grad_(False)
        fake = G(zs[0], zs[1])
        fake_logit, _, _ = D(fake, False)
        image = real[0].clone()
        if epoch % 2 == 1:
            image.requires_grad_(True)
            real_logit, _, _ = D(image, False)
            loss = bce(real_logit, ones()) + bce(fake_logit, zeros())
            if epoch % 8 == 1:
                r1 = LOSS.cal_r1_reg(real_logit, image, "cpu")
                d[f"{tag}/r1"] = r1.detach().numpy()
                if not frozen:
                    # the gradient of the R1 term ALONE (loss.py:18-34: the double backward through D): it is < 1 % of the D gradient,
                    # so the full-step statistics below cannot pin it
                    params = dict(D.named_parameters())
                    gr1 = torch.autograd.grad(r1 * HP["l_r1"], list(params.values()), retain_graph=True, allow_unused=True)
                    for (k, _), g in zip(params.items(), gr1):
                        if g is not None:
                            summarize(f"{tag}/r1grad/{k}", g, d)
                    d[f"{tag}/r1grad_none"] = np.array([k for (k, _), g in zip(params.items(), gr1) if g is None] or [""])
                loss = loss + r1 * HP["l_r1"]
        else:
            real_logit, gf, af = D(image, True)
            _, gp, an = D(real[1], True)
            _, gn, ap = D(real[2], True)
            aux = (LOSS.contrastive_loss(gf, gp, gn, HP["tau"]) + LOSS.contrastive_loss(af, ap, an, HP["tau"])) * HP["l_aux"]
            d[f"{tag}/aux"] = aux.detach().numpy()
            loss = bce(real_logit, ones()) + bce(fake_logit, zeros()) + aux
        loss.backward()
        d[f"{tag}/loss"] = loss.detach().numpy()
        d[f"{tag}/fake_sample"] = sample(fake)
        d[f"{tag}/real_logit"] = real_logit.detach().numpy()
        d[f"{tag}/fake_logit"] = fake_logit.detach().numpy()
        none = []
        for k, p in D.named_parameters():
            if p.grad is None:
                none.append(k)
            else:
                summarize(f"{tag}/grad/{k}", p.grad, d)
        d[f"{tag}/grad_none"] = np.array(none if none else [""])
    np.savez_compressed(os.path.join(OUT, f"step_r{res}.npz"), **d)
    print(f"step_r{res}.npz", len(d), "arrays")


def forward_fixtures(CNN, res=256, B=1, stride=16):
    """Whole-network forward at the benchmark resolutions (cnn.py:33-43, 89-115): 256 (config 2), 512 (config 3), 1024 (config 4)."""
    d = {"res": np.int64(res), "B": np.int64(B), "stride": np.int64(stride)}
    a = args_for(res)
    G, D = CNN.Generator(a), CNN.Discriminator(a)
    load(G, module_state(G, 1001)), load(D, module_state(D, 1002))
    z1, z2 = seeded_tensor((B, 64), 3000), seeded_tensor((B, 64), 3001)
    with torch.no_grad():
        img = G(z1, z2)
        d["img/slice"] = img[:, :, ::stride, ::stride].numpy().copy()
        d["img/sum"], d["img/abssum"] = np.float64(img.double().sum()), np.float64(img.double().abs().sum())
        d["avg_latent1"], d["avg_latent2"] = G.avg_latent1.numpy().copy(), G.avg_latent2.numpy().copy()
        img_t = G(z1, z2, 0.7)                                   # truncation branch, cnn.py:99-101
        d["img_trunc/slice"] = img_t[:, :, ::stride, ::stride].numpy().copy()
        real = seeded_tensor((B, 3, res, res), 3002, "uniform_pm1")
        logit, ge, ae = D(real, True)
        d["logit"], d["geo_emb"], d["app_emb"] = logit.numpy(), ge.numpy(), ae.numpy()
        logit_f, _, _ = D(img, False)
        d["logit_fake"] = logit_f.numpy()
    np.savez_compressed(os.path.join(OUT, f"forward_r{res}.npz"), **d)
    print(f"forward_r{res}.npz", len(d), "arrays")


def freeze_fixtures(CNN, LOSS):
    """Which discriminator parameters end an odd+R1 D step WITHOUT a gradient under the README's freezeD recipes (worker.py:127-131,
    cnn.py:17,54; README.md:29,37,49): frozen layers + the projection heads that odd iterations do not evaluate (cnn.py:38).
    One real-image pass at batch 1 is enough to see the set; the values are not stored."""
    d = {}
    for res, layer in ((256, 3), (512, 4), (1024, 5)):
        D = CNN.Discriminator(args_for(res))
        load(D, module_state(D, 1002))
        for i, (_, m) in enumerate(D.shared_model.named_children()):
            if i < layer + 2:
                for p in m.parameters():
                    p.requires_grad_(False)
        image = seeded_tensor((1, 3, res, res), 2100, "uniform_pm1").requires_grad_(True)
        logit, _, _ = D(image, False)
        loss = F.binary_cross_entropy_with_logits(logit, torch.ones(1, 1)) + LOSS.cal_r1_reg(logit, image, "cpu") * HP["l_r1"]
        loss.backward()
        d[f"r{res}_layer{layer}/grad_none"] = np.array([k for k, p in D.named_parameters() if p.grad is None])
        d[f"r{res}_layer{layer}/n_params"] = np.int64(len(list(D.parameters())))
        d[f"r{res}_layer{layer}/frozen_numel"] = np.int64(sum(p.numel() for p in D.parameters() if not p.requires_grad))
        print(f"freeze r{res} layer {layer}:", len(d[f"r{res}_layer{layer}/grad_none"]), "params without grad")
    np.savez_compressed(os.path.join(OUT, "freeze_sets.npz"), **d)


def fid_fixture():
    """eval/fid.py:4-27 on seeded feature sets (incl. a rank-deficient one that takes the eps branch or a complex root)."""
    sys.path.insert(0, REF)
    from eval import fid as RF
    d = {}
    for name, (n, f, shift) in {"a": (400, 24, 0.3), "b": (64, 24, 1.0), "c": (12, 24, 0.0)}.items():
        x = seeded_tensor((n, f), 4000 + n).double().numpy()
        y = (seeded_tensor((n, f), 4100 + n).double() * 1.3 + shift).numpy()
        try:
            v = float(RF.calc_fid(x.mean(0), np.cov(x, rowvar=False), y.mean(0), np.cov(y, rowvar=False)))
        except ValueError:
            v = float("nan")
        d[f"{name}/n"], d[f"{name}/f"], d[f"{name}/shift"], d[f"{name}/fid"] = np.int64(n), np.int64(f), np.float64(shift), np.float64(v)
        print("fid", name, v)
    np.savez_compressed(os.path.join(OUT, "fid.npz"), **d)


def main():
    assert os.path.isdir(REF), "the reference only exists in the build container"
    os.makedirs(OUT, exist_ok=True)
    CNN, CL, LOSS, EMA = _import_reference()
    torch.set_num_threads(8)
    only = set(sys.argv[1:])                                     # e.g. `python -m oracle.make_golden narrow`: just that file

    def want(name):
        return not only or name in only
    if want("layers"):
        layer_fixtures(CL, LOSS, EMA)
    if want("narrow"):
        narrow_fixtures(CL)
    if want("step"):
        step_fixtures(CNN, LOSS, res=32, B=8)
    if want("forward"):
        forward_fixtures(CNN, res=256, B=1)
        forward_fixtures(CNN, res=512, B=1, stride=32)
        forward_fixtures(CNN, res=1024, B=1, stride=64)
    if want("freeze"):
        freeze_fixtures(CNN, LOSS)
    if want("fid"):
        fid_fixture()


if __name__ == "__main__":
    main()

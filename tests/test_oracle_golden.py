"""Pins the CPU oracle (oracle/lcgan_ref.py) against golden vectors captured from the reference's own
modules (oracle/make_golden.py -> tests/golden/*.npz).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import lcgan_ref as O
from oracle.weights import seeded_state, seeded_tensor

GOLD = os.path.join(os.path.dirname(__file__), "golden")
RTOL = 1e-3   # north_star: outputs within 1e-3 relative (fp32) of the reference


def rel(a, b):
    a = torch.as_tensor(np.asarray(a), dtype=torch.float64)
    b = torch.as_tensor(np.asarray(b), dtype=torch.float64)
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def close(a, b, tol=RTOL, what=""):
    r = rel(a.detach() if isinstance(a, torch.Tensor) else a, b)
    assert r <= tol, f"{what}: rel err {r:.3e} > {tol}"


@pytest.fixture(scope="module")
def L():
    return np.load(os.path.join(GOLD, "layers.npz"))


def _state(shapes, seed, prefix=""):
    return {prefix + k: v for k, v in seeded_state(shapes, seed).items()}


@pytest.mark.parametrize("name,ci,co,k,up,hw", [("modconv_k3", 8, 16, 3, 1, 6), ("modconv_up", 8, 16, 3, 2, 5),
                                                 ("modconv_k1", 16, 3, 1, 1, 6)])
def test_modulated_conv(L, name, ci, co, k, up, hw):
    st = _state({"weight.weight": (co, ci, k, k), "bias": (co,)}, 11)
    w, b = st["weight.weight"].requires_grad_(True), st["bias"].requires_grad_(True)
    x = seeded_tensor((3, ci, hw, hw), 12).requires_grad_(True)
    s = (seeded_tensor((3, ci), 13) * 0.5 + 1).requires_grad_(True)
    y = O.modulated_conv(x, w, b, s, up)
    go = seeded_tensor(tuple(y.shape), 14)
    gx, gs, gw, gb = torch.autograd.grad((y * go).sum(), [x, s, w, b])
    for n, t in (("y", y), ("gx", gx), ("gs", gs), ("gw", gw), ("gb", gb)):
        close(t, L[f"{name}/{n}"], what=f"{name}/{n}")


def _synth_shapes(prefix, cin, cout, lat, k):
    return {f"{prefix}.linear.weight.weight": (cin, lat), f"{prefix}.linear.bias": (cin,),
            f"{prefix}.modulated_conv.weight.weight": (cout, cin, k, k), f"{prefix}.modulated_conv.bias": (cout,)}


def test_synthesis_block(L):
    sh = {}
    sh.update(_synth_shapes("modulated_conv0", 16, 8, 10, 3))
    sh.update(_synth_shapes("modulated_conv1", 8, 8, 10, 3))
    sh["skip_layer.weight.weight"] = (8, 16, 1, 1)
    sh.update(_synth_shapes("flow_layer", 16, 2, 6, 3))
    P = {k: v.requires_grad_(True) for k, v in _state(sh, 21, "b.").items()}
    x = seeded_tensor((2, 16, 5, 5), 22).requires_grad_(True)
    gl = seeded_tensor((2, 1, 6), 23).requires_grad_(True)
    al = seeded_tensor((2, 2, 10), 24).requires_grad_(True)
    y = O.synthesis_block(P, "b", x, gl[:, 0], al[:, 0], al[:, 1], 0.1)
    close(y, L["synblock/y"], what="y")
    go = seeded_tensor(tuple(y.shape), 25)
    grads = torch.autograd.grad((y * go).sum(), [x, gl, al] + list(P.values()))
    close(grads[0], L["synblock/gx"], what="gx")
    close(grads[1], L["synblock/ggl"], what="ggl")
    close(grads[2], L["synblock/gal"], what="gal")
    for k, g in zip(P, grads[3:]):
        close(g, L["synblock/grad/" + k[2:]], what=k)


def test_to_rgb(L):
    sh = {}
    sh.update(_synth_shapes("modulated_conv0", 8, 8, 10, 3))
    sh.update(_synth_shapes("modulated_conv1", 8, 3, 10, 1))
    P = {k: v.requires_grad_(True) for k, v in _state(sh, 31, "rgb_layer.").items()}
    x = seeded_tensor((2, 8, 6, 6), 32).requires_grad_(True)
    al = seeded_tensor((2, 2, 10), 33).requires_grad_(True)
    import torch.nn.functional as F
    h = F.leaky_relu(O.synthesis_layer(P, "rgb_layer.modulated_conv0", x, al[:, 0], 1), 0.2)
    y = O.synthesis_layer(P, "rgb_layer.modulated_conv1", h, al[:, 1], 1)
    close(y, L["torgb/y"], what="y")
    go = seeded_tensor(tuple(y.shape), 34)
    grads = torch.autograd.grad((y * go).sum(), [x, al] + list(P.values()))
    close(grads[0], L["torgb/gx"], what="gx")
    close(grads[1], L["torgb/gal"], what="gal")
    for k, g in zip(P, grads[2:]):
        close(g, L["torgb/grad/" + k[len("rgb_layer."):]], what=k)


def test_discriminator_block_double_backward(L):
    sh = {"conv0.weight.weight": (8, 8, 3, 3), "conv0.bias": (8,), "conv1.weight.weight": (16, 8, 3, 3),
          "conv1.bias": (16,), "skip_layer.weight.weight": (16, 8, 1, 1)}
    P = {k: v.requires_grad_(True) for k, v in _state(sh, 41, "b.").items()}
    x = seeded_tensor((2, 8, 8, 8), 42).requires_grad_(True)
    y = O.discriminator_block(P, "b", x)
    go = seeded_tensor(tuple(y.shape), 43)
    gx = torch.autograd.grad((y * go).sum(), x, create_graph=True)[0]
    close(y, L["dblock/y"], what="y")
    close(gx, L["dblock/gx"], what="gx")
    g2 = torch.autograd.grad(gx.square().sum(), list(P.values()), retain_graph=True, allow_unused=True)
    g1 = torch.autograd.grad((y * go).sum(), list(P.values()), allow_unused=True)
    for k, a, b in zip(P, g1, g2):
        close(a, L["dblock/grad1/" + k[2:]], what="grad1 " + k)
        ref2 = L["dblock/grad2/" + k[2:]]
        if b is None:
            assert np.abs(ref2).max() == 0
        else:
            close(b, ref2, what="grad2 " + k)


@pytest.mark.parametrize("N", [4, 8, 16])
def test_minibatch_std(L, N):
    x = seeded_tensor((N, 6, 4, 4), 50 + N).requires_grad_(True)
    y = O.minibatch_std(x, 8)
    go = seeded_tensor(tuple(y.shape), 51 + N)
    gx = torch.autograd.grad((y * go).sum(), x, create_graph=True)[0]
    v = seeded_tensor(tuple(x.shape), 52 + N)
    ggx = torch.autograd.grad((gx * v).sum(), x)[0]
    close(y, L[f"mbstd{N}/y"]), close(gx, L[f"mbstd{N}/gx"]), close(ggx, L[f"mbstd{N}/ggx"])


def test_mapping_network(L):
    sh = {"diagonal_params": (6,), "basis_params": (6, 6)}
    dims = [6, 8, 8, 12]
    for i in range(3):
        sh[f"mlp.{i}.weight.weight"] = (dims[i + 1], dims[i])
        sh[f"mlp.{i}.bias"] = (dims[i + 1],)
    P = {k: v.requires_grad_(True) for k, v in _state(sh, 71, "geometry_mapping.").items()}
    z = seeded_tensor((5, 6), 72).requires_grad_(True)
    y = O.mapping_network(P, "geometry_mapping", z)
    close(y, L["mapping/y"], what="y")
    go = seeded_tensor(tuple(y.shape), 73)
    grads = torch.autograd.grad((y * go).sum(), [z] + list(P.values()))
    close(grads[0], L["mapping/gz"], what="gz")
    for k, g in zip(P, grads[1:]):
        close(g, L["mapping/grad/" + k[len("geometry_mapping."):]], what=k)


def test_projection_head(L):
    dims = [16, 12, 8, 4]
    sh = {}
    for j in range(3):
        sh[f"mlp.{2 * j}.weight.weight"] = (dims[j + 1], dims[j])
        sh[f"mlp.{2 * j}.bias"] = (dims[j + 1],)
    P = _state(sh, 81, "projection_header1.")
    close(O.projection_head(P, "projection_header1", seeded_tensor((5, 16), 82), 3), L["phead/y"])


def test_discriminator_epilogue(L):
    import torch.nn.functional as F
    sh = {"conv.weight.weight": (8, 9, 3, 3), "conv.bias": (8,), "linear.weight.weight": (8, 128), "linear.bias": (8,)}
    P = {k: v.requires_grad_(True) for k, v in _state(sh, 61, "discriminator_epilogue.").items()}
    x = seeded_tensor((8, 8, 4, 4), 62).requires_grad_(True)
    e = F.leaky_relu(O.eq_conv(P, "discriminator_epilogue.conv", O.minibatch_std(x, 8)), 0.2)
    y = F.leaky_relu(O.eq_linear(P, "discriminator_epilogue.linear", e.flatten(1)), 0.2)
    close(y, L["depi/y"], what="y")
    go = seeded_tensor(tuple(y.shape), 63)
    grads = torch.autograd.grad((y * go).sum(), [x] + list(P.values()))
    close(grads[0], L["depi/gx"], what="gx")
    for k, g in zip(P, grads[1:]):
        close(g, L["depi/grad/" + k[len("discriminator_epilogue."):]], what=k)


def test_contrastive_loss(L):
    import torch.nn.functional as F
    a, p, n = (F.normalize(seeded_tensor((6, 16), 90 + i)).requires_grad_(True) for i in range(3))
    l = O.contrastive_loss(a, p, n, 0.05)
    ga, gp, gn = torch.autograd.grad(l, [a, p, n])
    close(l, L["closs/l"]), close(ga, L["closs/ga"]), close(gp, L["closs/gp"]), close(gn, L["closs/gn"])


def test_ema(L):
    sh = {"diagonal_params": (4,), "basis_params": (4, 4), "mlp.0.weight.weight": (4, 4), "mlp.0.bias": (4,),
          "mlp.1.weight.weight": (4, 4), "mlp.1.bias": (4,)}
    # state_dict order of the reference module: own params, own buffers, then children
    order = ["diagonal_params", "basis_params", "buf", "mlp.0.bias", "mlp.0.weight.weight", "mlp.1.bias", "mlp.1.weight.weight"]
    tgt = {**seeded_state(sh, 95), "buf": seeded_tensor((3,), 97)}     # the Ema ctor copies source -> target (ema.py:13-17)
    shb = {**sh, "buf": (3,)}   # make_golden draws the later source states with the buffer already registered
    src = {**seeded_state(shb, 99), "buf": seeded_tensor((3,), 100)}
    O.ema_update(src, tgt, 0.9, 1, start_iter=2)
    close(torch.cat([tgt[k].reshape(-1) for k in order]), L["ema/after_it1"])
    src = {**seeded_state(shb, 101), "buf": seeded_tensor((3,), 102)}
    O.ema_update(src, tgt, 0.9, 5, start_iter=2)
    close(torch.cat([tgt[k].reshape(-1) for k in order]), L["ema/after_it5"])


# ------------------------------------------------------------------------------------------------
def _sample(t, n=257):
    f = t.detach().reshape(-1)
    step = max(1, f.numel() // n)
    return f[::step][:n]


def _check_grads(S, tag, grads, tol=RTOL):
    from oracle.weights import grad_stats
    worst = 0.0
    for k, g in grads.items():
        st = grad_stats(g, k)
        ref_l1, ref_l2, ref_proj = float(S[f"{tag}/grad/{k}/abssum"]), float(S[f"{tag}/grad/{k}/l2"]), S[f"{tag}/grad/{k}/proj"]
        errs = (abs(st["abssum"] - ref_l1) / max(ref_l1, 1e-30), abs(st["l2"] - ref_l2) / max(ref_l2, 1e-30),
                float(np.abs(st["proj"] - ref_proj).max()) / max(ref_l2, 1e-30))
        worst = max(worst, *errs)
        assert max(errs) <= tol, (tag, k, errs)
    return worst


@pytest.fixture(scope="module")
def S():
    return np.load(os.path.join(GOLD, "step_r32.npz"))


def test_state_dict_layout(S):
    """The oracle's key/shape inventory equals the reference modules' state_dict (cnn.py:7-115)."""
    g, d = O.g_param_shapes(32), O.d_param_shapes(32)
    assert sorted(g) == list(S["g_keys"]) and sorted(d) == list(S["d_keys"])
    assert [str(tuple(g[k])) for k in sorted(g)] == list(S["g_shapes"])
    assert [str(tuple(d[k])) for k in sorted(d)] == list(S["d_shapes"])
    # parameter counts quoted in SURVEY.md section 8 (256x256)
    n_g = sum(int(np.prod(v)) for k, v in O.g_param_shapes(256).items() if k not in O.G_BUFFERS)
    n_d = sum(int(np.prod(v)) for v in O.d_param_shapes(256).values())
    assert (n_g, n_d) == (28112015, 64783489)


@pytest.mark.parametrize("epoch", [0, 1])
def test_generator_step(S, epoch):
    res, B = int(S["res"]), int(S["B"])
    GP, DP = seeded_state(O.g_param_shapes(res), 1001), seeded_state(O.d_param_shapes(res), 1002)
    zs = tuple(seeded_tensor((B, 64), 2000 + i) for i in range(4))
    loss, grads, bufs, parts = O.g_step(GP, DP, res, epoch, zs)
    close(loss, S[f"g{epoch}/loss"], what="g_loss")
    if epoch % 2 == 0:
        close(parts["aux"], S[f"g{epoch}/aux"]), close(parts["sparsity"], S[f"g{epoch}/sparsity"])
    close(bufs["avg_latent1"], S[f"g{epoch}/avg_latent1"]), close(bufs["avg_latent2"], S[f"g{epoch}/avg_latent2"])
    assert len(grads) == len(O.g_param_shapes(res)) - 2
    _check_grads(S, f"g{epoch}", grads)


@pytest.mark.parametrize("epoch,frozen", [(0, 0), (1, 0), (3, 0), (1, 2)])
def test_discriminator_step(S, epoch, frozen):
    res, B = int(S["res"]), int(S["B"])
    tag = f"d{epoch}" + (f"_freeze{frozen}" if frozen else "")
    GP, DP = seeded_state(O.g_param_shapes(res), 1001), seeded_state(O.d_param_shapes(res), 1002)
    zs = tuple(seeded_tensor((B, 64), 2000 + i) for i in range(2))
    real = tuple(seeded_tensor((B, 3, res, res), 2100 + i, "uniform_pm1") for i in range(3))
    frozen_prefixes = tuple(f"shared_model.{i}." for i in range(frozen + 2)) if frozen else ()
    loss, grads, bufs, parts = O.d_step(GP, DP, res, epoch, zs, real, frozen=frozen_prefixes)
    close(loss, S[f"{tag}/loss"], what="d_loss")
    if "r1" in parts:
        close(parts["r1"], S[f"{tag}/r1"], what="r1")
    none = sorted(k for k in DP if k not in grads)
    assert none == sorted(k for k in S[f"{tag}/grad_none"] if k), "set of grad=None parameters differs"
    _check_grads(S, tag, grads)


def test_r1_gradient_alone(S):
    """The oracle's double backward (r1_penalty: loss.py:18-34) against the reference's gradient of l_r1 * r1 ALONE."""
    from oracle.weights import grad_stats
    res, B = int(S["res"]), int(S["B"])
    DP = seeded_state(O.d_param_shapes(res), 1002)
    D = {k: v.detach().clone().requires_grad_(True) for k, v in DP.items()}
    image = seeded_tensor((B, 3, res, res), 2100, "uniform_pm1").requires_grad_(True)
    logit, _, _ = O.discriminator_forward(D, image, res, False)
    r1 = O.r1_penalty(logit, image)
    close(r1, S["d1/r1"], what="r1")
    (r1 * O.Hyper.l_r1).backward()
    none_ref = set(S["d1/r1grad_none"]) - {""}
    for k, v in D.items():
        if k in none_ref:
            assert v.grad is None, k
            continue
        ref_l2 = float(S[f"d1/r1grad/{k}/l2"])
        if v.grad is None:
            assert ref_l2 == 0.0, k
            continue
        st = grad_stats(v.grad, k)
        err = max(abs(st["l2"] - ref_l2), float(np.abs(st["proj"] - S[f"d1/r1grad/{k}/proj"]).max())) / max(ref_l2, 1e-30)
        assert (ref_l2 == 0.0 and st["l2"] == 0.0) or err <= 1e-3, (k, err)


def test_freeze_sets_follow_the_recipes():
    """tests/golden/freeze_sets.npz (captured from the reference under the README's freezeD recipes) against the oracle's key
    inventory: frozen = shared_model.0 .. shared_model.<layer+1>, plus the projection heads an odd iteration never evaluates."""
    FS = np.load(os.path.join(GOLD, "freeze_sets.npz"))
    for res, layer in ((256, 3), (512, 4), (1024, 5)):
        keys = list(O.d_param_shapes(res))
        frozen = [k for k in keys if k.startswith(tuple(f"shared_model.{i}." for i in range(layer + 2)))]
        heads = [k for k in keys if k.startswith("projection_header")]
        assert sorted(FS[f"r{res}_layer{layer}/grad_none"]) == sorted(frozen + heads)
        assert int(FS[f"r{res}_layer{layer}/frozen_numel"]) == sum(int(np.prod(O.d_param_shapes(res)[k])) for k in frozen)


@pytest.mark.parametrize("res_", [256, 512])
def test_forward_full_resolution(res_):
    """(the 1024 x 1024 fixture is checked against the HIP path on the GPU only: ~1 min of CPU here)"""
    Fw = np.load(os.path.join(GOLD, f"forward_r{res_}.npz"))
    res, B, st = int(Fw["res"]), int(Fw["B"]), int(Fw["stride"])
    GP, DP = seeded_state(O.g_param_shapes(res), 1001), seeded_state(O.d_param_shapes(res), 1002)
    z1, z2 = seeded_tensor((B, 64), 3000), seeded_tensor((B, 64), 3001)
    with torch.no_grad():
        img = O.generator_forward(GP, z1, z2, res)
        close(img[:, :, ::st, ::st], Fw["img/slice"], what="img")
        close(GP["avg_latent1"], Fw["avg_latent1"]), close(GP["avg_latent2"], Fw["avg_latent2"])
        img_t = O.generator_forward(GP, z1, z2, res, w_psi=0.7)
        close(img_t[:, :, ::st, ::st], Fw["img_trunc/slice"], what="img_trunc")
        real = seeded_tensor((B, 3, res, res), 3002, "uniform_pm1")
        logit, ge, ae = O.discriminator_forward(DP, real, res, True)
        close(logit, Fw["logit"]), close(ge, Fw["geo_emb"]), close(ae, Fw["app_emb"])
        close(O.discriminator_forward(DP, img, res, False)[0], Fw["logit_fake"], what="logit_fake")


# ---- narrow octaves (tests/golden/narrow.npz: C = 32 / 64 blocks at 64 x 64, captured from the reference) ----------------------------
def _stat_close(N, prefix, t, key, tol=RTOL):
    from oracle.weights import grad_stats
    st = grad_stats(t, key)
    l1, l2, proj = float(N[prefix + "/abssum"]), float(N[prefix + "/l2"]), N[prefix + "/proj"]
    if l2 == 0.0:
        assert float(t.abs().max()) == 0.0, prefix
        return
    err = max(abs(st["abssum"] - l1) / l1, abs(st["l2"] - l2) / l2, float(np.abs(st["proj"] - proj).max()) / l2)
    assert err <= tol, f"{prefix}: {err:.3e}"


def test_narrow_discriminator_block_double_backward():
    N = np.load(os.path.join(GOLD, "narrow.npz"))
    B, C, R = int(N["dblock/B"]), int(N["dblock/C"]), int(N["dblock/R"])
    sh = {"conv0.weight.weight": (C, C, 3, 3), "conv0.bias": (C,), "conv1.weight.weight": (2 * C, C, 3, 3),
          "conv1.bias": (2 * C,), "skip_layer.weight.weight": (2 * C, C, 1, 1)}
    P = {k: v.requires_grad_(True) for k, v in _state(sh, 141, "b.").items()}
    x = seeded_tensor((B, C, R, R), 142).requires_grad_(True)
    y = O.discriminator_block(P, "b", x)
    close(y[:, :, ::4, ::4], N["dblock/y_slice"], what="y")
    go = seeded_tensor(tuple(y.shape), 143)
    gx = torch.autograd.grad((y * go).sum(), x, create_graph=True)[0]
    _stat_close(N, "dblock/gx", gx.detach(), "dblock/gx")
    g2 = torch.autograd.grad(gx.square().sum(), list(P.values()), retain_graph=True, allow_unused=True)
    g1 = torch.autograd.grad((y * go).sum(), list(P.values()), allow_unused=True)
    for k, a, b in zip(P, g1, g2):
        _stat_close(N, "dblock/grad1/" + k[2:], a, k[2:])
        _stat_close(N, "dblock/grad2/" + k[2:], torch.zeros_like(P[k]) if b is None else b, k[2:])


def test_narrow_synthesis_block():
    N = np.load(os.path.join(GOLD, "narrow.npz"))
    B, Ci, Co, R = int(N["synblock/B"]), int(N["synblock/Ci"]), int(N["synblock/Co"]), int(N["synblock/R"])
    sh = {}
    sh.update(_synth_shapes("modulated_conv0", Ci, Co, 512, 3))
    sh.update(_synth_shapes("modulated_conv1", Co, Co, 512, 3))
    sh["skip_layer.weight.weight"] = (Co, Ci, 1, 1)
    sh.update(_synth_shapes("flow_layer", Ci, 2, 64, 3))
    P = {k: v.requires_grad_(True) for k, v in _state(sh, 121, "b.").items()}
    x = seeded_tensor((B, Ci, R, R), 122).requires_grad_(True)
    gl = seeded_tensor((B, 1, 64), 123).requires_grad_(True)
    al = seeded_tensor((B, 2, 512), 124).requires_grad_(True)
    y = O.synthesis_block(P, "b", x, gl[:, 0], al[:, 0], al[:, 1], 0.1)
    close(y[:, :, ::4, ::4], N["synblock/y_slice"], what="y")
    go = seeded_tensor(tuple(y.shape), 125)
    grads = torch.autograd.grad((y * go).sum(), [x, gl, al] + list(P.values()))
    _stat_close(N, "synblock/gx", grads[0], "synblock/gx")
    close(grads[1], N["synblock/ggl"], what="ggl")
    close(grads[2], N["synblock/gal"], what="gal")
    for k, g in zip(P, grads[3:]):
        _stat_close(N, "synblock/grad/" + k[2:], g, k[2:])

"""Layer-level fixtures captured from the reference's own modules (tests/golden/layers.npz, tests/golden/narrow.npz:
oracle/make_golden.py) fed through the PRODUCT's modules (lcgan_amd.custom_layers) in f32 parity mode.

`-m gpu`: the modules run on the HIP kernels through the C ABI (incl. the R1-style double backward of a DiscriminatorBlock).
Without a GPU the same bodies run on the CPU emulation of the kernel interface (host wiring only), which is also how the test
file itself is kept honest in the build container.

Tolerance: 1e-3 relative max-abs (BASELINE.json north_star) on every output and gradient; narrow-octave gradients (65 536-element
reductions) through the kink-robust statistics of tests/helpers.py."""
import os

import numpy as np
import pytest
import torch

from oracle.weights import grad_stats, seeded_state, seeded_tensor
from tests.helpers import GOLD, install_backend

TOL = 1e-3


def rel(a, b):
    a = torch.as_tensor(np.asarray(a.detach().cpu() if isinstance(a, torch.Tensor) else a), dtype=torch.float64)
    b = torch.as_tensor(np.asarray(b), dtype=torch.float64)
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def close(a, b, what, tol=TOL):
    r = rel(a, b)
    assert r <= tol, f"{what}: rel err {r:.3e} > {tol}"
    return r


def load_seeded(m, seed, dev):
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    m.load_state_dict(seeded_state(shapes, seed))
    return m.to(dev)


def to_feat(x_nchw):
    """fp32 NCHW leaf -> NHWC feature map (differentiable, so gradients come back in NCHW)"""
    from lcgan_amd import config, ops
    from lcgan_amd.kernels import ceil8
    return ops.ToNHWCFn.apply(x_nchw, ceil8(x_nchw.shape[1]), config.feature_dtype())


def to_nchw(feat, clog):
    from lcgan_amd import ops
    return ops.ToNCHWFn.apply(feat, clog)


@pytest.fixture(scope="module")
def L():
    return np.load(os.path.join(GOLD, "layers.npz"))


def _backend(dev):
    """HIP kernels on the GPU, the CPU emulation of the kernel interface otherwise; f32 parity mode either way"""
    import contextlib
    from lcgan_amd import config

    @contextlib.contextmanager
    def ctx():
        if dev == "cpu":
            from oracle.hip_emulation import EmulatedKernels
            install_backend(EmulatedKernels())
        else:
            from lcgan_amd import kernels
            install_backend(None)
            assert kernels.backend_name() == "hip"
        try:
            with config.feature_dtype_as(torch.float32):
                yield
        finally:
            install_backend(None)
    return ctx()


DEVICES = [pytest.param("cuda:0", marks=pytest.mark.gpu), "cpu"]


@pytest.mark.parametrize("dev", DEVICES)
@pytest.mark.parametrize("name,ci,co,k,up,hw", [("modconv_k3", 8, 16, 3, 1, 6), ("modconv_up", 8, 16, 3, 2, 5), ("modconv_k1", 16, 3, 1, 1, 6)])
def test_modulated_conv_module(L, dev, name, ci, co, k, up, hw):
    """ModulatedConv2d (custom_layers.py:47-86): k3, k3 + transposed x2, and the demodulated 1x1 to RGB"""
    from lcgan_amd import custom_layers as CL
    with _backend(dev):
        m = load_seeded(CL.ModulatedConv2d(ci, co, k, up=up), 11, dev)
        x = seeded_tensor((3, ci, hw, hw), 12).to(dev).requires_grad_(True)
        s = (seeded_tensor((3, ci), 13) * 0.5 + 1).to(dev).requires_grad_(True)
        y = m.forward_to_rgb(to_feat(x), s) if k == 1 else to_nchw(m(to_feat(x), s), co)
        go = seeded_tensor(tuple(y.shape), 14).to(dev)
        gx, gs, gw, gb = torch.autograd.grad((y * go).sum(), [x, s, m.weight.weight, m.bias])
        for n, t in (("y", y), ("gx", gx), ("gs", gs), ("gw", gw), ("gb", gb)):
            close(t, L[f"{name}/{n}"], f"{name}/{n}")


@pytest.mark.parametrize("dev", DEVICES)
def test_synthesis_block_module(L, dev):
    """SynthesisBlock (custom_layers.py:114-166): skip / flow / up-conv / conv / add / bicubic warp, all gradients"""
    from lcgan_amd import custom_layers as CL
    with _backend(dev):
        m = load_seeded(CL.SynthesisBlock(16, 8, 6, 10, 10, 0.1), 21, dev)
        x = seeded_tensor((2, 16, 5, 5), 22).to(dev).requires_grad_(True)
        gl = seeded_tensor((2, 1, 6), 23).to(dev).requires_grad_(True)
        al = seeded_tensor((2, 2, 10), 24).to(dev).requires_grad_(True)
        y = to_nchw(m(to_feat(x), gl, al), 8)
        close(y, L["synblock/y"], "synblock/y")
        go = seeded_tensor(tuple(y.shape), 25).to(dev)
        params = dict(m.named_parameters())
        grads = torch.autograd.grad((y * go).sum(), [x, gl, al] + list(params.values()))
        close(grads[0], L["synblock/gx"], "synblock/gx")
        close(grads[1], L["synblock/ggl"], "synblock/ggl")
        close(grads[2], L["synblock/gal"], "synblock/gal")
        for (k, _), g in zip(params.items(), grads[3:]):
            close(g, L[f"synblock/grad/{k}"], f"synblock/grad/{k}")


@pytest.mark.parametrize("dev", DEVICES)
def test_to_rgb_module(L, dev):
    """ToRGBBlock (custom_layers.py:169-182)"""
    from lcgan_amd import custom_layers as CL
    with _backend(dev):
        m = load_seeded(CL.ToRGBBlock(8, 3, 10, 8), 31, dev)
        x = seeded_tensor((2, 8, 6, 6), 32).to(dev).requires_grad_(True)
        al = seeded_tensor((2, 2, 10), 33).to(dev).requires_grad_(True)
        y = m(to_feat(x), al)
        close(y, L["torgb/y"], "torgb/y")
        go = seeded_tensor(tuple(y.shape), 34).to(dev)
        params = dict(m.named_parameters())
        grads = torch.autograd.grad((y * go).sum(), [x, al] + list(params.values()))
        close(grads[0], L["torgb/gx"], "torgb/gx")
        close(grads[1], L["torgb/gal"], "torgb/gal")
        for (k, _), g in zip(params.items(), grads[2:]):
            close(g, L[f"torgb/grad/{k}"], f"torgb/grad/{k}")


@pytest.mark.parametrize("dev", DEVICES)
def test_discriminator_block_module_double_backward(L, dev):
    """DiscriminatorBlock (custom_layers.py:185-217) incl. the R1-style double backward (loss.py:18-34): gradient of |d y.go / d x|^2
    with respect to every parameter"""
    from lcgan_amd import custom_layers as CL
    with _backend(dev):
        m = load_seeded(CL.DiscriminatorBlock(8, 16, skip=True), 41, dev)
        x = seeded_tensor((2, 8, 8, 8), 42).to(dev).requires_grad_(True)
        y = to_nchw(m(to_feat(x)), 16)
        close(y, L["dblock/y"], "dblock/y")
        go = seeded_tensor(tuple(y.shape), 43).to(dev)
        gx = torch.autograd.grad((y * go).sum(), x, create_graph=True)[0]
        close(gx, L["dblock/gx"], "dblock/gx")
        params = dict(m.named_parameters())
        g2 = torch.autograd.grad(gx.square().sum(), list(params.values()), retain_graph=True, allow_unused=True)
        g1 = torch.autograd.grad((y * go).sum(), list(params.values()), allow_unused=True)
        for (k, p), a, b in zip(params.items(), g1, g2):
            close(a, L[f"dblock/grad1/{k}"], f"dblock/grad1/{k}")
            ref2 = L[f"dblock/grad2/{k}"]
            if b is None:                                    # biases: the double backward reaches them only through the masks -> exactly 0
                assert float(np.abs(ref2).max()) == 0.0, k
            elif float(np.abs(ref2).max()) == 0.0:
                assert float(b.abs().max()) == 0.0, k
            else:
                close(b, ref2, f"dblock/grad2/{k}")


@pytest.mark.parametrize("dev", DEVICES)
def test_discriminator_epilogue_module(L, dev):
    """DiscriminatorEpilogue (custom_layers.py:220-234): minibatch-stddev (group 8) + 3x3 conv on C+1 channels + the 16C-wide linear"""
    from lcgan_amd import custom_layers as CL
    with _backend(dev):
        m = load_seeded(CL.DiscriminatorEpilogue(8, resolution=4, mbstd_group_size=8), 61, dev)
        x = seeded_tensor((8, 8, 4, 4), 62).to(dev).requires_grad_(True)
        y = m(to_feat(x))
        close(y, L["depi/y"], "depi/y")
        go = seeded_tensor(tuple(y.shape), 63).to(dev)
        params = dict(m.named_parameters())
        grads = torch.autograd.grad((y * go).sum(), [x] + list(params.values()))
        close(grads[0], L["depi/gx"], "depi/gx")
        for (k, _), g in zip(params.items(), grads[1:]):
            close(g, L[f"depi/grad/{k}"], f"depi/grad/{k}")


@pytest.mark.parametrize("dev", DEVICES)
def test_mapping_network_module(L, dev):
    """MappingNetwork (custom_layers.py:259-287): Householder-QR basis x |diagonal|, then affine layers"""
    from lcgan_amd import custom_layers as CL
    with _backend(dev):
        m = load_seeded(CL.MappingNetwork([6, 8, 8, 12]), 71, dev)
        z = seeded_tensor((5, 6), 72).to(dev).requires_grad_(True)
        y = m(z)
        close(y, L["mapping/y"], "mapping/y")
        go = seeded_tensor(tuple(y.shape), 73).to(dev)
        params = dict(m.named_parameters())
        grads = torch.autograd.grad((y * go).sum(), [z] + list(params.values()))
        close(grads[0], L["mapping/gz"], "mapping/gz")
        for (k, _), g in zip(params.items(), grads[1:]):
            close(g, L[f"mapping/grad/{k}"], f"mapping/grad/{k}")


# ---- narrow octaves (C = 32 / 64: the 512 x 512 / 1024 x 1024 networks' high-resolution layers) -------------------------------------------
# tests/golden/narrow.npz holds a DiscriminatorBlock(32, 64) at 64 x 64 (first-order gradients AND the R1-style double backward) and a
# SynthesisBlock(64 -> 32) at 32 x 32 -> 64 x 64 with every gradient, captured from the reference.  On the GPU the bf16-only narrow
# kernels (conv_halo_narrow_kernel, packed-channel-group weight gradient) are additionally compared in bf16 against the same fixtures
# at the benchmark dtype's stated tolerance.
def _stat_err(t, key, N, prefix):
    st = grad_stats(t, key)
    l1, l2, proj = float(N[f"{prefix}/abssum"]), float(N[f"{prefix}/l2"]), N[f"{prefix}/proj"]
    if l2 == 0.0:
        return float(t.abs().max())
    return max(abs(st["abssum"] - l1) / l1, abs(st["l2"] - l2) / l2, float(np.abs(st["proj"] - proj).max()) / l2)


def _narrow_dblock(dev, N, tol, dtype):
    from lcgan_amd import config
    from lcgan_amd import custom_layers as CL
    B, C, R = int(N["dblock/B"]), int(N["dblock/C"]), int(N["dblock/R"])
    with config.feature_dtype_as(dtype):
        m = load_seeded(CL.DiscriminatorBlock(C, 2 * C, skip=True), 141, dev)
        x = seeded_tensor((B, C, R, R), 142).to(dev).requires_grad_(True)
        y = to_nchw(m(to_feat(x)), 2 * C)
        go = seeded_tensor(tuple(y.shape), 143).to(dev)
        errs = {"y": rel(y[:, :, ::4, ::4], N["dblock/y_slice"])}
        gx = torch.autograd.grad((y * go).sum(), x, create_graph=True)[0]
        errs["gx"] = _stat_err(gx, "dblock/gx", N, "dblock/gx")
        params = dict(m.named_parameters())
        g2 = torch.autograd.grad(gx.square().sum() * float(N["dblock/g2_scale"]), list(params.values()), retain_graph=True, allow_unused=True)
        g1 = torch.autograd.grad((y * go).sum(), list(params.values()), allow_unused=True)
        for (k, p), a, b in zip(params.items(), g1, g2):
            errs[f"grad1/{k}"] = _stat_err(a, k, N, f"dblock/grad1/{k}")
            if float(N[f"dblock/grad2/{k}/l2"]) == 0.0:
                assert b is None or float(b.abs().max()) == 0.0, k
            else:
                errs[f"grad2/{k}"] = _stat_err(b, k, N, f"dblock/grad2/{k}")
    bad = {k: v for k, v in errs.items() if v > tol}
    assert not bad, f"narrow DiscriminatorBlock ({dtype}): {sorted(bad.items(), key=lambda kv: -kv[1])[:6]}"
    return max(errs.values())


def _narrow_synblock(dev, N, tol, dtype):
    from lcgan_amd import config
    from lcgan_amd import custom_layers as CL
    B, Ci, Co, R = int(N["synblock/B"]), int(N["synblock/Ci"]), int(N["synblock/Co"]), int(N["synblock/R"])
    with config.feature_dtype_as(dtype):
        m = load_seeded(CL.SynthesisBlock(Ci, Co, 64, 512, 2 * R, 0.1), 121, dev)
        x = seeded_tensor((B, Ci, R, R), 122).to(dev).requires_grad_(True)
        gl = seeded_tensor((B, 1, 64), 123).to(dev).requires_grad_(True)
        al = seeded_tensor((B, 2, 512), 124).to(dev).requires_grad_(True)
        y = to_nchw(m(to_feat(x), gl, al), Co)
        go = seeded_tensor(tuple(y.shape), 125).to(dev)
        errs = {"y": rel(y[:, :, ::4, ::4], N["synblock/y_slice"])}
        params = dict(m.named_parameters())
        grads = torch.autograd.grad((y * go).sum(), [x, gl, al] + list(params.values()))
        errs["gx"] = _stat_err(grads[0], "synblock/gx", N, "synblock/gx")
        errs["ggl"] = rel(grads[1], N["synblock/ggl"])
        errs["gal"] = rel(grads[2], N["synblock/gal"])
        for (k, _), g in zip(params.items(), grads[3:]):
            errs[f"grad/{k}"] = _stat_err(g, k, N, f"synblock/grad/{k}")
    bad = {k: v for k, v in errs.items() if v > tol}
    assert not bad, f"narrow SynthesisBlock ({dtype}): {sorted(bad.items(), key=lambda kv: -kv[1])[:6]}"
    return max(errs.values())


@pytest.fixture(scope="module")
def N():
    return np.load(os.path.join(GOLD, "narrow.npz"))


@pytest.mark.parametrize("dev", DEVICES)
def test_narrow_discriminator_block_f32(N, dev):
    with _backend(dev):
        _narrow_dblock(dev, N, 2e-3, torch.float32)


@pytest.mark.parametrize("dev", DEVICES)
def test_narrow_synthesis_block_f32(N, dev):
    with _backend(dev):
        _narrow_synblock(dev, N, 2e-3, torch.float32)


@pytest.mark.gpu
@pytest.mark.parametrize("narrow_min", [1, 256])
def test_narrow_blocks_bf16_narrow_kernels(N, narrow_min):
    """The same fixtures in bf16, where the narrow-layer kernels exist (conv_halo_narrow_kernel: option 7 = smallest grid that takes
    it; 1 forces it for these 64 x 64 grids, 256 is the shipped routing) and the packed-channel-group weight gradient (option 9).
    Tolerance of the benchmark dtype: 0.10 on the gradient statistics (bf16 activations, 8 mantissa bits, through a double backward)."""
    from lcgan_amd import kernels
    install_backend(None)
    lib = kernels.K.lib
    old = lib.lcgan_set_option(7, narrow_min)
    try:
        a = _narrow_dblock("cuda:0", N, 0.10, torch.bfloat16)
        b = _narrow_synblock("cuda:0", N, 0.10, torch.bfloat16)
        print(f"narrow bf16 (option 7 = {narrow_min}): worst statistic error dblock {a:.3e} synblock {b:.3e}")
    finally:
        lib.lcgan_set_option(7, old)

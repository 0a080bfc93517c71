"""Per-kernel parity on the MI355X: every C-ABI kernel (called through lcgan_amd.kernels.HipKernels -> liblcgan_hip.so)
against the CPU emulation of the same kernel on identical seeded inputs, in both feature dtypes.

Tolerances (relative to the max |reference| of the tensor):
  * f32 features ("parity mode", bf16x3 split MFMA):  2e-4  (well inside BASELINE.json's 1e-3)
  * bf16 features: the emulation rounds operands at the same points, so only accumulation order and the final bf16
    rounding differ: 1.2e-2 max-abs-relative (1.5 bf16 ulp of the largest element), 2e-3 relative L2.
"""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.hip_emulation import EmulatedKernels, ceil8   # noqa: E402

E = EmulatedKernels()
DTYPES = [torch.float32, torch.bfloat16]


@pytest.fixture(scope="module")
def H():
    from lcgan_amd.kernels import HipKernels
    return HipKernels()


def dev(t):
    return None if t is None else t.cuda()


def tol(dtype):
    return (2e-4, 1e-4) if dtype == torch.float32 else (1.2e-2, 2e-3)


def check(got, ref, dtype, what="", l2_scale=1.0):
    got, ref = got.detach().float().cpu(), ref.detach().float()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    mx, l2 = tol(dtype)
    l2 *= l2_scale
    scale = ref.abs().max().clamp_min(1e-20)
    e_max = float((got - ref).abs().max() / scale)
    e_l2 = float((got - ref).norm() / ref.norm().clamp_min(1e-20))
    assert math.isfinite(e_max) and e_max <= mx and e_l2 <= l2, f"{what}: max-rel {e_max:.3e} (tol {mx}), l2-rel {e_l2:.3e} (tol {l2})"


def feat(shape, dtype, seed, clog=None, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(shape, generator=g) * scale
    if clog is not None:
        x[..., clog:] = 0
    return x.to(dtype)


def vec(shape, seed, lo=0.5, hi=1.5):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(shape, generator=g) * (hi - lo) + lo


# ------------------------------------------------------------------------------------------------------------
CONV_CASES = [
    # B, H, W, Cin(log), Cout(log), k, stride
    (2, 8, 8, 32, 32, 3, 1),
    (3, 16, 16, 64, 128, 3, 1),
    (2, 12, 20, 40, 24, 3, 1),       # ragged: M, N, K all off-tile
    (2, 16, 16, 64, 96, 3, 2),
    (2, 8, 8, 128, 256, 1, 1),
    (4, 4, 4, 513, 512, 3, 1),       # discriminator epilogue conv (513 -> padded 520 / 544)
    (1, 64, 64, 128, 128, 3, 1),     # several M tiles per sample
    # shapes that take the bf16 halo-tile kernel (M grid >= 16 x 16): stride 2, ragged tiles, 1x1, padded channels
    (2, 32, 32, 64, 96, 3, 2),
    (1, 20, 40, 40, 24, 3, 1),
    (2, 32, 32, 128, 256, 1, 1),
    (2, 64, 32, 72, 136, 3, 2),
    # few tiles x deep reduction: the halo kernel splits its channel chunks over blockIdx.z (fp32 partials + finalize)
    (1, 32, 32, 256, 128, 3, 1),
    (2, 32, 32, 160, 64, 3, 2),
]


def test_prep_weight_group(H):
    """all preparations of a network in one launch == the single-weight kernel: layouts and parts bit for bit, wsq to the last bit or two"""
    g = torch.Generator().manual_seed(5)
    ws = [torch.randn(s, generator=g).cuda() for s in [(64, 40, 3, 3), (8, 128, 3, 3), (136, 72, 1, 1), (256, 256, 3, 3),
                                                          (24, 40, 4, 4), (513, 512, 3, 3)]]
    jobs = [(ws[0], 0.1, False, False, True), (ws[0], 0.1, True, False, False), (ws[1], 0.2, False, True, False),
            (ws[2], 0.3, True, True, True), (ws[3], 0.05, False, False, True), (ws[3], 0.05, True, False, False),
            (ws[4], 0.15, False, True, True), (ws[4], 0.15, True, False, False),     # 16 taps: staged in two passes of <= 9
            (ws[5], 0.02, False, False, False), (ws[5], 0.02, True, False, False)]   # ragged last tile in both directions
    for rep in range(2):                                   # second round takes the cached job table
        got = H.prep_weight_group(jobs)
        for (w, sc, tr, lo, wq), (pw, wsq) in zip(jobs, got):
            if w.shape[2] not in (1, 3):                   # the single-weight entry only takes k = 1, 3: restate the layout in torch
                v = (w * sc).permute(1, 0, 2, 3) if tr else w * sc
                N, Kc, kk = v.shape[0], v.shape[1], v.shape[2] * v.shape[3]
                v = torch.nn.functional.pad(v.reshape(N, Kc, kk).permute(2, 0, 1), (0, pw.Kpad - Kc)).contiguous()
                for part in range(pw.parts):
                    h = v.bfloat16()
                    assert torch.equal(pw.buf[part], h), (tuple(w.shape), tr, part)
                    v = v - h.float()
                if wq:
                    torch.testing.assert_close(wsq, ((w * sc) ** 2).sum((2, 3)), rtol=1e-6, atol=1e-7)
                continue
            ref_pw, ref_wsq = H.prep_weight(w, sc, tr, lo, wq)
            assert (pw.parts, pw.N, pw.Kpad, pw.k) == (ref_pw.parts, ref_pw.N, ref_pw.Kpad, ref_pw.k)
            assert torch.equal(pw.buf, ref_pw.buf)
            assert (wsq is None) == (ref_wsq is None)
            if wsq is not None:                            # (summed from the staged tile: same order, but the compiler may fuse the
                torch.testing.assert_close(wsq, ref_wsq, rtol=1e-6, atol=0)   # multiply-adds differently: last-bit differences)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [(2, 32, 32, 128, 256, 1), (3, 16, 16, 64, 128, 1), (2, 64, 64, 64, 64, 1), (4, 8, 8, 128, 256, 1), (2, 32, 32, 64, 96, 3),
                                  (32, 16, 16, 512, 512, 1)])
def test_conv_fwd_pooled_byproduct(H, dtype, case):
    """lcgan_conv_fwd(pool_out): avg_pool2d(y, 2) out of the closing convolution of a DiscriminatorBlock (1x1 + residual; the epilogue
    of the halo kernel on the fast path, the pooling kernel elsewhere) == the pooling kernel on the stored y, bit for bit."""
    B, Hh, W, Ci, Co, k = case
    x = feat((B, Hh, W, ceil8(Ci)), dtype, 1, Ci)
    w = torch.randn(Co, Ci, k, k, generator=torch.Generator().manual_seed(2))
    pw_h, _ = H.prep_weight(w.cuda(), 0.7 / math.sqrt(Ci * k * k), False, dtype == torch.float32)
    res = feat((B, Hh, W, ceil8(Co)), dtype, 6, Co)
    for residual in (res.cuda(), None):
        y, pooled = H.conv_fwd(x.cuda(), pw_h, Co, k, 1, residual=residual, pool=True)
        y0 = H.conv_fwd(x.cuda(), pw_h, Co, k, 1, residual=residual)
        check(y, y0.cpu(), dtype, "y")                      # (split-K partials of small grids meet through atomics: not bit-reproducible)
        assert torch.equal(pooled, H.avgpool2(y)), "pooled by-product differs from avg_pool2d of the stored output"


@pytest.mark.parametrize("dtype", DTYPES)
def test_rgb_expand_pooled_byproduct(H, dtype):
    B, Hh, W, C = 3, 16, 24, 128
    img = torch.randn(B, 3, Hh, W, generator=torch.Generator().manual_seed(71))
    w = torch.randn(1, 3, C, generator=torch.Generator().manual_seed(72)) * 0.3
    bias = torch.randn(C, generator=torch.Generator().manual_seed(73))
    y, pooled = H.rgb_expand(img.cuda(), w.cuda(), bias.cuda(), 0.5, C, 1, 1.2, dtype, pool=True)
    assert torch.equal(y, H.rgb_expand(img.cuda(), w.cuda(), bias.cuda(), 0.5, C, 1, 1.2, dtype))
    assert torch.equal(pooled, H.avgpool2(y))


@pytest.mark.parametrize("case", [(8, 32, 32, 64, 64, 3, 1), (8, 16, 16, 128, 96, 3, 1), (4, 32, 32, 64, 128, 3, 2), (8, 8, 8, 128, 128, 3, 1)])
def test_conv_wgrad_prescaled_path(H, case):
    """weight gradient with per-sample scales on small operands: scales applied by one elementwise pass + batch-wide reduction (option 17)
    against the per-sample-range path (option 17 = 0) and the emulation"""
    B, Hh, W, Cx, A, k, stride = case
    dtype = torch.bfloat16
    x = feat((B, Hh, W, ceil8(Cx)), dtype, 11, Cx)
    g = feat((B, Hh // stride, W // stride, ceil8(A)), dtype, 12, A)
    px, pg = vec((B, ceil8(Cx)), 13), vec((B, ceil8(A)), 14)
    ref = E.conv_wgrad(x, g, A, Cx, k, stride, pre_x=px, pre_g=pg)
    new = H.conv_wgrad(x.cuda(), g.cuda(), A, Cx, k, stride, pre_x=px.cuda(), pre_g=pg.cuda())
    old_opt = H.lib.lcgan_set_option(17, 0)
    try:
        old = H.conv_wgrad(x.cuda(), g.cuda(), A, Cx, k, stride, pre_x=px.cuda(), pre_g=pg.cuda())
    finally:
        H.lib.lcgan_set_option(17, old_opt)
    check(new, ref, dtype, "prescaled vs emulation", l2_scale=2.0)
    check(old, ref, dtype, "per-sample ranges vs emulation", l2_scale=2.0)
    check(new, old.cpu(), dtype, "prescaled vs per-sample ranges", l2_scale=2.0)


@pytest.mark.parametrize("case", [(2, 32, 32, 128, 128, "fwd"), (4, 16, 16, 512, 512, "fwd"), (3, 32, 32, 256, 136, "dgrad"), (2, 16, 16, 128, 96, "up"),
                                  (1, 48, 32, 160, 128, "fwd")])
def test_conv_halo_channel_split(H, case):
    """halo-tile launches too small to fill the chip can split their input-channel range over blockIdx.y (option 22, off by default; stride-1 LDS-DMA
    structure, partial tiles through slabs, the last split to arrive finishes): plain / bias + activation + residual / half-resolution
    residual / per-sample scales (folded into weight copies) against the emulation and the generic split-K kernel (option 22 = 0);
    the slab sum runs in split order, so two launches agree bit for bit"""
    B, Hh, W, Ci, Co, kind = case
    dtype, k = torch.bfloat16, 3
    st = 2 if kind == "up" else 1
    tr = kind in ("dgrad", "up")
    x = feat((B, Hh, W, ceil8(Ci)), dtype, 1, Ci)
    w = torch.randn(Ci, Co, k, k, generator=torch.Generator().manual_seed(2)) if tr else torch.randn(Co, Ci, k, k, generator=torch.Generator().manual_seed(2))
    scale = 1 / math.sqrt(Ci * k * k)
    pw_e, _ = E.prep_weight(w, scale, tr, False)
    pw_h, _ = H.prep_weight(w.cuda(), scale, tr, False)
    fn_h = H.conv_bwd_data if tr else H.conv_fwd
    fn_e = E.conv_bwd_data if tr else E.conv_fwd
    ref = fn_e(x, pw_e, Co, k, st)
    bias = vec((Co,), 5)
    res = feat(tuple(ref.shape), dtype, 7, Co)
    pre, post = vec((B, ceil8(Ci)), 8), vec((B, ceil8(Co)), 9)
    variants = {"plain": dict(), "epilogue": dict(bias=bias, act=1, gain=1.4, residual=res), "scaled": dict(pre=pre, post=post)}
    if kind != "up":
        variants["half residual"] = dict(residual=feat((B, ref.shape[1] // 2, ref.shape[2] // 2, ref.shape[3]), dtype, 10, Co), residual_half=True)
    cu = lambda kw: {key: (v.cuda() if torch.is_tensor(v) else v) for key, v in kw.items()}
    for name, kw in variants.items():
        want = fn_e(x, pw_e, Co, k, st, **kw)
        check(fn_h(x.cuda(), pw_h, Co, k, st, **cu(kw)), want, dtype, name + " (generic kernel)", l2_scale=2.0 if name == "scaled" else 1.0)
        old = H.lib.lcgan_set_option(22, 256)                      # (off by default: see the switch's comment in conv_igemm.hip)
        try:
            got = fn_h(x.cuda(), pw_h, Co, k, st, **cu(kw))
            check(got, want, dtype, name, l2_scale=2.0 if name == "scaled" else 1.0)
            assert torch.equal(got, fn_h(x.cuda(), pw_h, Co, k, st, **cu(kw))), name
        finally:
            H.lib.lcgan_set_option(22, old)


@pytest.mark.parametrize("case", [(4, 32, 32, 64, 128, 1, "fwd"), (4, 32, 32, 128, 64, 1, "dgrad"), (4, 16, 16, 64, 96, 2, "up"), (4, 32, 32, 96, 64, 2, "down"),
                                  (8, 64, 64, 128, 128, 1, "fwd"), (8, 64, 64, 128, 128, 1, "dgrad")])
def test_conv_per_sample_weights(H, case):
    """convolutions with per-sample input scales through per-sample weight copies (option 18; the scales folded into the weights once,
    then the unscaled LDS-DMA kernels) against the in-LDS scaling path (option 18 = 0) and the emulation: plain, + residual, and with the
    fused style-gradient reduction; forward, data gradient, x2 transposed (`up`) and stride-2 forward (`down`) geometries.  Option 6 = 1
    lets these small grids take the halo kernels at all."""
    B, Hh, W, Ci, Co, stride, kind = case
    dtype, k = torch.bfloat16, 3
    x = feat((B, Hh, W, ceil8(Ci)), dtype, 1, Ci)
    w = torch.randn(Co, Ci, k, k, generator=torch.Generator().manual_seed(2)) if kind in ("fwd", "down") else torch.randn(Ci, Co, k, k, generator=torch.Generator().manual_seed(2))
    scale = 1 / math.sqrt(Ci * k * k)
    tr = kind in ("dgrad", "up")
    pw_e, _ = E.prep_weight(w, scale, tr, False)
    pw_h, _ = H.prep_weight(w.cuda(), scale, tr, False)
    pre, post = vec((B, ceil8(Ci)), 4), vec((B, ceil8(Co)), 5)
    fn_h = H.conv_bwd_data if tr else H.conv_fwd
    fn_e = E.conv_bwd_data if tr else E.conv_fwd
    st = stride
    ref = fn_e(x, pw_e, Co, k, st, pre=pre, post=post)
    xs = feat(tuple(ref.shape), dtype, 6, Co)
    old6 = H.lib.lcgan_set_option(6, 1)
    try:
        outs = {}
        for mb in (80, 0):
            old18 = H.lib.lcgan_set_option(18, mb)
            try:
                y = fn_h(x.cuda(), pw_h, Co, k, st, pre=pre.cuda(), post=post.cuda())
                yr = fn_h(x.cuda(), pw_h, Co, k, st, pre=pre.cuda(), post=post.cuda(), residual=xs.cuda()) if kind != "up" else None
                yg, gs = fn_h(x.cuda(), pw_h, Co, k, st, pre=pre.cuda(), post=post.cuda(), xs=xs.cuda())
                ygr, _ = fn_h(x.cuda(), pw_h, Co, k, st, pre=pre.cuda(), post=post.cuda(), xs=xs.cuda(), residual=xs.cuda())
                assert torch.equal(ygr, (yg.float() + xs.cuda().float()).to(dtype)), f"option 18 = {mb}: fused y + residual != add pass"
                outs[mb] = (y, yr, yg, gs)
            finally:
                H.lib.lcgan_set_option(18, old18)
    finally:
        H.lib.lcgan_set_option(6, old6)
    yg_e, gs_e = fn_e(x, pw_e, Co, k, st, pre=pre, post=post, xs=xs)
    for mb, (y, yr, yg, gs) in outs.items():
        check(y, ref, dtype, f"option 18 = {mb}: plain", l2_scale=2.0)            # (w * pre rounded instead of pre * x: another bf16 rounding pattern)
        if yr is not None:
            check(yr, fn_e(x, pw_e, Co, k, st, pre=pre, post=post, residual=xs), dtype, f"option 18 = {mb}: residual", l2_scale=2.0)
        check(yg, yg_e, dtype, f"option 18 = {mb}: fused y", l2_scale=2.0)
        check(gs, gs_e, dtype, f"option 18 = {mb}: fused gs", l2_scale=3.0)
    check(outs[80][0], outs[0][0].cpu(), dtype, "per-sample weights vs in-LDS scaling", l2_scale=2.0)


def _same_as_add_pass(y_r, y_plain, r, dtype, what):
    """the residual of the style-gradient epilogue must give what a separate add pass over the stored result gives: bit for bit where
    the launch is deterministic (halo kernels, slab split-K); launches that meet their split-K partials through float atomics differ from
    run to run in the last bit of u, i.e. by one rounding step of the stored value"""
    want = (y_plain.float() + r.float()).to(dtype)
    if torch.equal(y_r, want):
        return
    ulp = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -20       # (f32 parity mode: atomics over up to 36 splits move u by a few units in the last place)
    err = float((y_r.float() - want.float()).abs().max()) / float(want.float().abs().max())
    assert err <= ulp, f"{what}: differs from the add pass by {err:.2e} (> one rounding step {ulp:.1e})"


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd(H, dtype, case):
    B, Hh, W, Ci, Co, k, stride = case
    x = feat((B, Hh, W, ceil8(Ci)), dtype, 1, Ci)
    w = torch.randn(Co, Ci, k, k, generator=torch.Generator().manual_seed(2))
    bias = torch.randn(Co, generator=torch.Generator().manual_seed(3))
    scale = 1 / math.sqrt(Ci * k * k)
    need_lo = dtype == torch.float32
    pw_e, _ = E.prep_weight(w, scale, False, need_lo)
    pw_h, _ = H.prep_weight(w.cuda(), scale, False, need_lo)
    # plain
    check(H.conv_fwd(x.cuda(), pw_h, Co, k, stride), E.conv_fwd(x, pw_e, Co, k, stride), dtype, "plain")
    # bias + lrelu*gain
    check(H.conv_fwd(x.cuda(), pw_h, Co, k, stride, bias=bias.cuda(), bias_scale=0.5, act=1, gain=1.4),
          E.conv_fwd(x, pw_e, Co, k, stride, bias=bias, bias_scale=0.5, act=1, gain=1.4), dtype, "bias+lrelu")
    # modulated: pre/post scales + residual
    pre, post = vec((B, ceil8(Ci)), 4), vec((B, ceil8(Co)), 5)
    ref0 = E.conv_fwd(x, pw_e, Co, k, stride)
    res = feat(tuple(ref0.shape), dtype, 6, Co)
    # (bf16: where the launch folds the input scales into per-sample weight copies -- option 18 -- w * pre is rounded instead of pre * x)
    check(H.conv_fwd(x.cuda(), pw_h, Co, k, stride, pre=pre.cuda(), post=post.cuda(), bias=bias.cuda(), residual=res.cuda()),
          E.conv_fwd(x, pw_e, Co, k, stride, pre=pre, post=post, bias=bias, residual=res), dtype, "mod+residual", l2_scale=2.0)
    y_h, gs_h = H.conv_fwd(x.cuda(), pw_h, Co, k, stride, pre=pre.cuda(), post=post.cuda(), xs=res.cuda())
    y_e, gs_e = E.conv_fwd(x, pw_e, Co, k, stride, pre=pre, post=post, xs=res)
    check(y_h, y_e, dtype, "fused y", l2_scale=2.0)        # generic path: u is rounded to bf16 BEFORE the style scale (two roundings)
    check(gs_h, gs_e, dtype, "fused gs", l2_scale=3.0)     # generic path reduces the bf16-rounded u; few values, cancelling sums
    # ... and with a residual joining the same epilogue (ops.SynthForkFn): exactly the value a separate add pass would store
    r2 = feat(tuple(ref0.shape), dtype, 7, Co)
    y_r, gs_r = H.conv_fwd(x.cuda(), pw_h, Co, k, stride, pre=pre.cuda(), post=post.cuda(), xs=res.cuda(), residual=r2.cuda())
    _same_as_add_pass(y_r, y_h, r2.cuda(), dtype, "fused y + residual")
    check(gs_r, gs_h.cpu(), dtype, "fused gs with residual", l2_scale=3.0)     # (same reduction; split-K atomics differ from run to run in u's last bit)
    check(y_r, E.conv_fwd(x, pw_e, Co, k, stride, pre=pre, post=post, xs=res, residual=r2)[0], dtype, "fused y + residual", l2_scale=2.0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES + [(2, 8, 8, 64, 2, 3, 2), (2, 16, 16, 2, 64, 3, 1), (2, 32, 32, 64, 2, 3, 2),
                                  (1, 64, 64, 2, 64, 3, 1), (1, 48, 40, 32, 48, 3, 2)])
def test_conv_bwd_data(H, dtype, case):
    B, Hh, W, Ci, Co, k, stride = case          # g has Co channels on the (Hh/stride) grid; output has Ci channels
    if stride == 2 and k != 3:
        pytest.skip("no strided 1x1")
    Hg, Wg = Hh // stride, W // stride
    g = feat((B, Hg, Wg, ceil8(Co)), dtype, 11, Co)
    w = torch.randn(Co, Ci, k, k, generator=torch.Generator().manual_seed(12))
    scale = 1 / math.sqrt(Ci * k * k)
    need_lo = dtype == torch.float32
    pw_e, _ = E.prep_weight(w, scale, True, need_lo)
    pw_h, _ = H.prep_weight(w.cuda(), scale, True, need_lo)
    check(H.conv_bwd_data(g.cuda(), pw_h, Ci, k, stride), E.conv_bwd_data(g, pw_e, Ci, k, stride), dtype, "plain")
    pre, post = vec((B, ceil8(Co)), 13), vec((B, ceil8(Ci)), 14)
    bias = torch.randn(Ci, generator=torch.Generator().manual_seed(15))
    check(H.conv_bwd_data(g.cuda(), pw_h, Ci, k, stride, pre=pre.cuda(), post=post.cuda(), bias=bias.cuda(), act=1, gain=1.2),
          E.conv_bwd_data(g, pw_e, Ci, k, stride, pre=pre, post=post, bias=bias, act=1, gain=1.2), dtype, "mod+bias+act", l2_scale=2.0)   # (see test_conv_fwd)
    # style-gradient reduction fused into the launch: gx = post * u, gs = sum_pixels xs * u
    xs = feat((B, Hh, W, ceil8(Ci)), dtype, 17, Ci)
    gx_h, gs_h = H.conv_bwd_data(g.cuda(), pw_h, Ci, k, stride, pre=pre.cuda(), post=post.cuda(), xs=xs.cuda())
    gx_e, gs_e = E.conv_bwd_data(g, pw_e, Ci, k, stride, pre=pre, post=post, xs=xs)
    check(gx_h, gx_e, dtype, "fused gx", l2_scale=2.0)     # generic path: u is rounded to bf16 BEFORE the style scale (two roundings)
    check(gs_h, gs_e, dtype, "fused gs", l2_scale=3.0)     # generic path reduces the bf16-rounded u; few values, cancelling sums
    r2 = feat((B, Hh, W, ceil8(Ci)), dtype, 18, Ci)        # a residual joining the style-gradient epilogue (ops.SynthForkFn)
    gx_r, gs_r = H.conv_bwd_data(g.cuda(), pw_h, Ci, k, stride, pre=pre.cuda(), post=post.cuda(), xs=xs.cuda(), residual=r2.cuda())
    _same_as_add_pass(gx_r, gx_h, r2.cuda(), dtype, "fused gx + residual")
    check(gs_r, gs_h.cpu(), dtype, "fused gs with residual", l2_scale=3.0)
    if Hh % 2 == 0 and W % 2 == 0:         # pooled-branch gradient folded into the epilogue (0.25 * nearest-x2 of a half-res tensor)
        rh = feat((B, Hh // 2, W // 2, ceil8(Ci)), dtype, 16, Ci)
        check(H.conv_bwd_data(g.cuda(), pw_h, Ci, k, stride, residual=rh.cuda(), residual_half=True),
              E.conv_bwd_data(g, pw_e, Ci, k, stride, residual=rh, residual_half=True), dtype, "half-res residual")


# ---- staging variants of the halo-tile kernel and the row-segment weight-gradient kernel ------------------------------------------
# (lcgan_set_option: 6 = 0 sends every eligible launch to the halo kernel whatever its grid size; 10 = LDS-DMA staging of the
#  unmodulated stride-1 / transposed geometries (0 off, 1 / 2 taps per barrier); 11 = the same records with the halo through
#  registers for convolutions with per-sample input scales; 12 = LDS-DMA staging of the weight-gradient kernel)
HALO_VARIANT_CASES = [
    # B, H, W, Cin, Cout, k, stride -- Cin a multiple of 32 (the DMA variants' domain), whole and ragged 16 x 16 tiles
    (2, 32, 32, 64, 128, 3, 1),
    (1, 48, 32, 128, 96, 3, 1),
    (2, 32, 32, 96, 160, 1, 1),
    (1, 32, 64, 64, 64, 3, 2),
    (2, 64, 64, 96, 160, 3, 2),
    (1, 40, 24, 32, 136, 3, 1),
]


@pytest.mark.parametrize("variant", [(0, 0), (1, 0), (2, 1), (1, 1), (2, 2)], ids=lambda v: f"dma{v[0]}-mod{v[1]}")
@pytest.mark.parametrize("case", HALO_VARIANT_CASES)
def test_conv_halo_staging_variants(H, case, variant):
    B, Hh, W, Ci, Co, k, stride = case
    dtype = torch.bfloat16
    old = [H.lib.lcgan_set_option(6, 0), H.lib.lcgan_set_option(10, variant[0]), H.lib.lcgan_set_option(11, variant[1]), H.lib.lcgan_set_option(13, 1)]
    old18 = H.lib.lcgan_set_option(18, 0)          # the in-kernel input scaling is what these variants are (per-sample weight copies: test_conv_per_sample_weights)
    try:
        scale = 1 / math.sqrt(Ci * k * k)
        w = torch.randn(Co, Ci, k, k, generator=torch.Generator().manual_seed(2))
        bias = torch.randn(Co, generator=torch.Generator().manual_seed(3))
        for s2dma in ((0, 1, 2, 4) if stride == 2 else (1,)):                     # forward: plain, epilogue, modulated, residual
            H.lib.lcgan_set_option(13, s2dma)                                # (stride 2: both the register-staged and the parity-plane structure)
            x = feat((B, Hh, W, Ci), dtype, 1)
            pw_e, _ = E.prep_weight(w, scale, False, False)
            pw_h, _ = H.prep_weight(w.cuda(), scale, False, False)
            check(H.conv_fwd(x.cuda(), pw_h, Co, k, stride), E.conv_fwd(x, pw_e, Co, k, stride), dtype, "plain")
            check(H.conv_fwd(x.cuda(), pw_h, Co, k, stride, bias=bias.cuda(), bias_scale=0.5, act=1, gain=1.4),
                  E.conv_fwd(x, pw_e, Co, k, stride, bias=bias, bias_scale=0.5, act=1, gain=1.4), dtype, "bias+lrelu")
            pre, post = vec((B, Ci), 4), vec((B, ceil8(Co)), 5)
            res = feat((B, Hh // stride, W // stride, ceil8(Co)), dtype, 6, Co)
            check(H.conv_fwd(x.cuda(), pw_h, Co, k, stride, residual=res.cuda()), E.conv_fwd(x, pw_e, Co, k, stride, residual=res), dtype, "residual")
            check(H.conv_fwd(x.cuda(), pw_h, Co, k, stride, pre=pre.cuda(), post=post.cuda(), bias=bias.cuda(), residual=res.cuda()),
                  E.conv_fwd(x, pw_e, Co, k, stride, pre=pre, post=post, bias=bias, residual=res), dtype, "mod+residual")
        # data gradient of the same layer (stride 2: the 4-phase transposed convolution): g has Co channels, the output Ci
        g = feat((B, Hh // stride, W // stride, ceil8(Co)), dtype, 11, Co)
        pw_e, _ = E.prep_weight(w, scale, True, False)
        pw_h, _ = H.prep_weight(w.cuda(), scale, True, False)
        if ceil8(Co) % 32 == 0:
            check(H.conv_bwd_data(g.cuda(), pw_h, Ci, k, stride), E.conv_bwd_data(g, pw_e, Ci, k, stride), dtype, "dgrad")
            rh = feat((B, Hh // 2, W // 2, Ci), dtype, 16)
            check(H.conv_bwd_data(g.cuda(), pw_h, Ci, k, stride, residual=rh.cuda(), residual_half=True),
                  E.conv_bwd_data(g, pw_e, Ci, k, stride, residual=rh, residual_half=True), dtype, "dgrad + half-res residual")
            pre, post = vec((B, ceil8(Co)), 13), vec((B, Ci), 14)
            xs = feat((B, Hh, W, Ci), dtype, 17)
            gx_h, gs_h = H.conv_bwd_data(g.cuda(), pw_h, Ci, k, stride, pre=pre.cuda(), post=post.cuda(), xs=xs.cuda())
            gx_e, gs_e = E.conv_bwd_data(g, pw_e, Ci, k, stride, pre=pre, post=post, xs=xs)
            check(gx_h, gx_e, dtype, "fused gx", l2_scale=2.0)
            check(gs_h, gs_e, dtype, "fused gs", l2_scale=3.0)
    finally:
        for o, v in zip((6, 10, 11, 13), old):
            H.lib.lcgan_set_option(o, v)
        H.lib.lcgan_set_option(18, old18)


@pytest.mark.parametrize("case", [(2, 64, 64, 128, 128, 3, 1), (1, 64, 128, 64, 256, 3, 1), (2, 64, 64, 96, 160, 1, 1)])
def test_conv_staging_variants_bit_identical(H, case):
    """register staging, LDS-DMA with one / two taps per barrier and the modulated in-place scaling accumulate in the SAME order:
    identical bits (the stride-2 parity-plane structure walks 16-channel half-chunks and may differ by one bf16 ulp, not tested here)"""
    B, Hh, W, Ci, Co, k, stride = case
    x = feat((B, Hh, W, Ci), torch.bfloat16, 1).cuda()
    w = torch.randn(Co, Ci, k, k, generator=torch.Generator().manual_seed(2)).cuda()
    bias = torch.randn(Co, generator=torch.Generator().manual_seed(3)).cuda()
    pre, post = vec((B, Ci), 4).cuda(), vec((B, ceil8(Co)), 5).cuda()
    pw, _ = H.prep_weight(w, 1 / math.sqrt(Ci * k * k), False, False)
    old = [H.lib.lcgan_set_option(6, 0), H.lib.lcgan_set_option(10, 0), H.lib.lcgan_set_option(11, 0), H.lib.lcgan_set_option(18, 0)]
    try:
        ref_p = H.conv_fwd(x, pw, Co, k, 1, bias=bias, act=1, gain=1.4)
        ref_m = H.conv_fwd(x, pw, Co, k, 1, pre=pre, post=post, bias=bias, act=1, gain=1.4)
        for dma, mod in ((1, 1), (2, 2)):
            H.lib.lcgan_set_option(10, dma); H.lib.lcgan_set_option(11, mod)
            assert torch.equal(H.conv_fwd(x, pw, Co, k, 1, bias=bias, act=1, gain=1.4), ref_p), (dma, "plain")
            assert torch.equal(H.conv_fwd(x, pw, Co, k, 1, pre=pre, post=post, bias=bias, act=1, gain=1.4), ref_m), (dma, mod, "modulated")
    finally:
        for o, v in zip((6, 10, 11, 18), old):
            H.lib.lcgan_set_option(o, v)


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
@pytest.mark.parametrize("case", [(1, 64, 64, 128, 256, 3, 1), (2, 64, 128, 64, 72, 3, 1), (1, 128, 128, 96, 128, 3, 2), (3, 64, 64, 256, 128, 3, 1)])
def test_conv_wgrad_staging_variants(H, case, mode):
    B, Hh, W, Ci, Co, k, stride = case
    dtype = torch.bfloat16
    x = feat((B, Hh, W, ceil8(Ci)), dtype, 21, Ci)
    g = feat((B, Hh // stride, W // stride, ceil8(Co)), dtype, 22, Co)
    old = H.lib.lcgan_set_option(12, mode)
    try:
        check(H.conv_wgrad(x.cuda(), g.cuda(), Co, Ci, k, stride), E.conv_wgrad(x, g, Co, Ci, k, stride), dtype, "plain")
        px, pg = vec((B, ceil8(Ci)), 23), vec((B, ceil8(Co)), 24)
        got, ref_p = H.conv_wgrad(x.cuda(), g.cuda(), Co, Ci, k, stride, pre_x=px.cuda(), pre_g=pg.cuda()), E.conv_wgrad(x, g, Co, Ci, k, stride, pre_x=px, pre_g=pg)
        e_l2 = float((got.float().cpu() - ref_p).norm() / ref_p.norm())
        assert e_l2 <= 4e-3, e_l2
    finally:
        H.lib.lcgan_set_option(12, old)


@pytest.mark.parametrize("case", [(2, 8, 8, 64, 128, 3, 1), (4, 4, 4, 512, 512, 3, 1), (2, 16, 16, 64, 96, 3, 2), (2, 8, 8, 128, 256, 1, 1)])
def test_conv_igemm_register_staging(H, case):
    """the generic implicit-GEMM kernel's register-staged main loop (option 16 = 0; LDS-DMA staging is the default)"""
    B, Hh, W, Ci, Co, k, stride = case
    dtype = torch.bfloat16
    old = H.lib.lcgan_set_option(16, 0)
    try:
        x = feat((B, Hh, W, Ci), dtype, 1)
        w = torch.randn(Co, Ci, k, k, generator=torch.Generator().manual_seed(2))
        scale = 1 / math.sqrt(Ci * k * k)
        pw_e, _ = E.prep_weight(w, scale, False, False)
        pw_h, _ = H.prep_weight(w.cuda(), scale, False, False)
        check(H.conv_fwd(x.cuda(), pw_h, Co, k, stride), E.conv_fwd(x, pw_e, Co, k, stride), dtype, "plain")
        g = feat((B, Hh // stride, W // stride, ceil8(Co)), dtype, 11, Co)
        pw_e, _ = E.prep_weight(w, scale, True, False)
        pw_h, _ = H.prep_weight(w.cuda(), scale, True, False)
        check(H.conv_bwd_data(g.cuda(), pw_h, Ci, k, stride), E.conv_bwd_data(g, pw_e, Ci, k, stride), dtype, "dgrad")
    finally:
        H.lib.lcgan_set_option(16, old)


@pytest.mark.parametrize("case", [(8, 64, 64, 128), (24, 32, 48, 256), (32, 32, 32, 512), (16, 64, 32, 64), (40, 40, 24, 32), (2, 16, 16, 64)])
def test_flow_wgrad_one_pass(H, case):
    """the flow layer's 1x1 weight gradient (A = 18 against Cin, per-sample style scale on the input side) as one pass over x
    (flow_wgrad_kernel, option 23) against the row-segment kernel (option 23 = 0) and the emulation"""
    B, Hh, W, Ci = case
    dtype = torch.bfloat16
    x = feat((B, Hh, W, Ci), dtype, 31)
    gt = feat((B, Hh, W, 24), dtype, 32, 18)
    s_ = vec((B, Ci), 33)
    want = E.conv_wgrad(x, gt, 18, Ci, 1, 1, pre_x=s_)
    got = H.conv_wgrad(x.cuda(), gt.cuda(), 18, Ci, 1, 1, pre_x=s_.cuda())
    old = H.lib.lcgan_set_option(23, 0)
    try:
        ref = H.conv_wgrad(x.cuda(), gt.cuda(), 18, Ci, 1, 1, pre_x=s_.cuda())
    finally:
        H.lib.lcgan_set_option(23, old)
    for name, t in (("one pass", got), ("row-segment kernel", ref)):
        e_l2 = float((t.float().cpu() - want).norm() / want.norm())
        assert e_l2 <= 4e-3, (name, e_l2)
    assert float((got - ref).norm() / ref.norm()) <= 4e-3
    # without the scale, and through the fused entry (which clears gwp itself)
    check(H.conv_wgrad(x.cuda(), gt.cuda(), 18, Ci, 1, 1), E.conv_wgrad(x, gt, 18, Ci, 1, 1), dtype, "unscaled", l2_scale=2.0)


@pytest.mark.parametrize("opts", [(2, 0), (3, 0), (3, 8), (4, 64)])
@pytest.mark.parametrize("case", [(2, 8, 8, 64, 128, 3, 1), (8, 8, 8, 512, 512, 3, 1), (4, 4, 4, 512, 512, 3, 1), (3, 16, 16, 64, 96, 3, 2), (2, 8, 8, 128, 256, 1, 1),
                                  (32, 16, 16, 128, 128, 3, 1), (32, 8, 8, 512, 512, 3, 1), (4, 32, 32, 512, 256, 3, 1), (16, 32, 32, 256, 256, 3, 2)])
def test_conv_igemm_dma_variants(H, case, opts):
    """the generic implicit-GEMM kernel's LDS-DMA loops: four waves (option 16 = 2), eight waves with three / four stages (3 / 4), split-K
    through atomics + finalize (option 19 = 0) or through per-split slabs finished by the last split to arrive (up to option 19 splits);
    the slab form sums in split order, so two launches agree bit for bit"""
    B, Hh, W, Ci, Co, k, stride = case
    dtype = torch.bfloat16
    old16, old19, old6 = H.lib.lcgan_set_option(16, opts[0]), H.lib.lcgan_set_option(19, opts[1]), H.lib.lcgan_set_option(6, 1 << 20)
    try:
        x = feat((B, Hh, W, Ci), dtype, 1)
        w = torch.randn(Co, Ci, k, k, generator=torch.Generator().manual_seed(2))
        scale = 1 / math.sqrt(Ci * k * k)
        pw_e, _ = E.prep_weight(w, scale, False, False)
        pw_h, _ = H.prep_weight(w.cuda(), scale, False, False)
        bias = vec((Co,), 5)
        res = feat((B, Hh // stride, W // stride, ceil8(Co)), dtype, 7, Co)
        got = H.conv_fwd(x.cuda(), pw_h, Co, k, stride, bias=bias.cuda(), act=1, gain=1.4, residual=res.cuda())
        check(got, E.conv_fwd(x, pw_e, Co, k, stride, bias=bias, act=1, gain=1.4, residual=res), dtype, "fwd")
        if opts[1] >= 64 and B * (Hh // stride) * (W // stride) >= 2048:      # (smaller launches keep the atomics: see dispatch_igemm)
            again = H.conv_fwd(x.cuda(), pw_h, Co, k, stride, bias=bias.cuda(), act=1, gain=1.4, residual=res.cuda())
            assert torch.equal(got, again)
        g = feat((B, Hh // stride, W // stride, ceil8(Co)), dtype, 11, Co)
        pw_e, _ = E.prep_weight(w, scale, True, False)
        pw_h, _ = H.prep_weight(w.cuda(), scale, True, False)
        check(H.conv_bwd_data(g.cuda(), pw_h, Ci, k, stride), E.conv_bwd_data(g, pw_e, Ci, k, stride), dtype, "dgrad")
    finally:
        H.lib.lcgan_set_option(16, old16); H.lib.lcgan_set_option(19, old19); H.lib.lcgan_set_option(6, old6)


@pytest.mark.parametrize("case", [(2, 64, 64, 128, 128, 3, 1, False), (2, 64, 64, 64, 96, 3, 2, True), (4, 16, 16, 128, 256, 3, 1, False),
                                  (1, 32, 32, 72, 40, 1, 1, False),
                                  # small grids, few chunks: ONE split whose epilogue writes the weight-layout gradient itself
                                  (2, 8, 8, 128, 256, 3, 1, False), (2, 8, 8, 256, 136, 3, 1, True), (3, 16, 16, 72, 40, 1, 1, False),
                                  (2, 16, 16, 128, 128, 3, 2, False), (2, 16, 16, 200, 264, 1, 1, True),
                                  # 4 x 4 grids: the generic kernel, one split, same direct epilogue
                                  (8, 4, 4, 128, 256, 3, 1, False), (8, 4, 4, 256, 136, 3, 1, True), (3, 4, 4, 72, 40, 1, 1, False), (32, 4, 4, 64, 64, 3, 1, False)])
def test_conv_wgrad_unprep_fused(H, case):
    """lcgan_conv_wgrad_fused == lcgan_conv_wgrad + lcgan_conv_wgrad_unprep (slab, atomic and single-split routes, both orientations, demod term)"""
    B, Hh, W, Ci, Co, k, stride, tr = case
    dtype = torch.bfloat16
    x = feat((B, Hh, W, ceil8(Ci)), dtype, 21, Ci).cuda()
    g = feat((B, Hh // stride, W // stride, ceil8(Co)), dtype, 22, Co).cuda()
    a, bc = (g, x) if tr else (x, g)                           # transposed: the gradient call runs with the roles swapped (up-sampling convs)
    A, Bc = (Ci, Co) if tr else (Co, Ci)
    wA, wBc = (Bc, A) if tr else (A, Bc)
    w = torch.randn(wA, wBc, k, k, generator=torch.Generator().manual_seed(25)).cuda()
    gwsq = torch.randn(wA, wBc, generator=torch.Generator().manual_seed(26)).cuda()
    px = vec((B, a.shape[-1]), 23).cuda()
    pg = vec((B, bc.shape[-1]), 24).cuda()
    for kw in (dict(), dict(pre_x=px, pre_g=pg)):
        ref = H.unprep_wgrad(H.conv_wgrad(a, bc, A, Bc, k, stride, **kw), wA, wBc, k, 0.3, tr, w, gwsq)
        got = H.conv_wgrad_unprep(a, bc, A, Bc, k, stride, 0.3, tr, w=w, gwsq=gwsq, **kw)
        assert got.shape == ref.shape
        err = float((got - ref).abs().max() / ref.abs().max())
        assert err < 2e-5, err                                  # (fp32 sums in a different order)
    # the fused entry's gwp is scratch whose contents must not matter: hand it a poisoned buffer through the C ABI
    from lcgan_amd.kernels import dt_code
    gwp = torch.full((k * k, A, Bc), float("nan"), device="cuda")
    gw = torch.empty((wA, wBc, k, k), device="cuda")
    H._call("lcgan_conv_wgrad_fused", a.data_ptr(), bc.data_ptr(), gwp.data_ptr(), B, a.shape[1], a.shape[2], a.shape[3],
            bc.shape[1], bc.shape[2], bc.shape[3], A, Bc, k, stride, px.data_ptr(), pg.data_ptr(), dt_code(dtype), 0.3, int(tr),
            w.data_ptr(), gwsq.data_ptr(), gw.data_ptr(), H._stream())
    assert float((gw - got).abs().max() / got.abs().max()) < 2e-5          # (the atomic route sums in arrival order)


NARROW_CASES = [
    # B, H, W, Cin, Cout, stride: layers with <= 64 output channels on grids of whole 32 x 32 tiles (the C = 32 / 64 octaves of the
    # 512 x 512 and 1024 x 1024 networks) -> conv_halo_narrow_kernel (forced here for small grids through option 7)
    (1, 64, 64, 32, 32, 1),
    (2, 32, 64, 64, 64, 1),
    (1, 32, 32, 128, 24, 1),
    (2, 64, 32, 40, 64, 1),
]


@pytest.mark.parametrize("case", NARROW_CASES)
def test_conv_narrow_layers(H, case):
    B, Hh, W, Ci, Co, _ = case
    dtype = torch.bfloat16
    old = H.lib.lcgan_set_option(7, 1)
    try:
        x = feat((B, Hh, W, ceil8(Ci)), dtype, 31, Ci)
        w = torch.randn(Co, Ci, 3, 3, generator=torch.Generator().manual_seed(32))
        bias = torch.randn(Co, generator=torch.Generator().manual_seed(33))
        scale = 1 / math.sqrt(Ci * 9)
        pw_e, _ = E.prep_weight(w, scale, False, False)
        pw_h, _ = H.prep_weight(w.cuda(), scale, False, False)
        check(H.conv_fwd(x.cuda(), pw_h, Co, 3, 1, bias=bias.cuda(), bias_scale=0.5, act=1, gain=1.4),
              E.conv_fwd(x, pw_e, Co, 3, 1, bias=bias, bias_scale=0.5, act=1, gain=1.4), dtype, "bias+lrelu")
        pre, post = vec((B, ceil8(Ci)), 34), vec((B, ceil8(Co)), 35)
        res = feat((B, Hh, W, ceil8(Co)), dtype, 36, Co)
        check(H.conv_fwd(x.cuda(), pw_h, Co, 3, 1, pre=pre.cuda(), post=post.cuda(), bias=bias.cuda(), residual=res.cuda()),
              E.conv_fwd(x, pw_e, Co, 3, 1, pre=pre, post=post, bias=bias, residual=res), dtype, "mod+residual")
        y_h, gs_h = H.conv_fwd(x.cuda(), pw_h, Co, 3, 1, pre=pre.cuda(), post=post.cuda(), xs=res.cuda())
        y_e, gs_e = E.conv_fwd(x, pw_e, Co, 3, 1, pre=pre, post=post, xs=res)
        check(y_h, y_e, dtype, "fused y")
        check(gs_h, gs_e, dtype, "fused gs", l2_scale=3.0)
        # data gradient (stride 1: Cin output channels must be <= 64 to stay on the narrow kernel) and the 4-phase transposed conv
        g = feat((B, Hh, W, ceil8(Co)), dtype, 37, Co)
        pwt_e, _ = E.prep_weight(w, scale, True, False)
        pwt_h, _ = H.prep_weight(w.cuda(), scale, True, False)
        rh = feat((B, Hh // 2, W // 2, ceil8(Ci)), dtype, 38, Ci)
        check(H.conv_bwd_data(g.cuda(), pwt_h, Ci, 3, 1, residual=rh.cuda(), residual_half=True),
              E.conv_bwd_data(g, pwt_e, Ci, 3, 1, residual=rh, residual_half=True), dtype, "dgrad + half-res residual")
        check(H.conv_bwd_data(g.cuda(), pwt_h, Ci, 3, 2, pre=post.cuda(), bias=None, act=0),
              E.conv_bwd_data(g, pwt_e, Ci, 3, 2, pre=post, bias=None, act=0), dtype, "transposed conv")
    finally:
        H.lib.lcgan_set_option(7, old)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", CONV_CASES + [(2, 16, 16, 64, 2, 3, 2),
                                  # row-segment bf16 kernel (k = 3, output-side width a multiple of 32)
                                  (1, 64, 64, 128, 256, 3, 2), (1, 32, 64, 72, 40, 3, 1), (2, 32, 64, 64, 2, 3, 2),
                                  (2, 32, 32, 256, 128, 3, 1),
                                  # narrow layers (several image rows per chunk) and 1x1 kernels on the row-segment kernel
                                  (3, 16, 16, 128, 96, 3, 1), (2, 8, 8, 64, 128, 3, 1), (2, 32, 32, 64, 40, 3, 2), (2, 16, 16, 72, 64, 3, 2),
                                  (2, 64, 64, 128, 256, 1, 1), (2, 16, 16, 256, 128, 1, 1), (1, 128, 128, 64, 32, 1, 1),
                                  # <= 64 channels on both sides, 64-wide rows: packed channel groups (4 or 2 image rows per chunk)
                                  (1, 64, 64, 32, 32, 3, 1), (2, 64, 128, 24, 16, 3, 1), (1, 64, 64, 64, 48, 3, 1), (1, 128, 128, 32, 32, 3, 2),
                                  (2, 128, 128, 40, 64, 3, 2)])
def test_conv_wgrad(H, dtype, case):
    B, Hh, W, Ci, Co, k, stride = case
    x = feat((B, Hh, W, ceil8(Ci)), dtype, 21, Ci)
    g = feat((B, Hh // stride, W // stride, ceil8(Co)), dtype, 22, Co)
    ref = E.conv_wgrad(x, g, Co, Ci, k, stride)
    check(H.conv_wgrad(x.cuda(), g.cuda(), Co, Ci, k, stride), ref, dtype, "plain")
    old = H.lib.lcgan_set_option(5, 1)          # also exercise the (default-off) narrow / 1x1 routing of the row-segment kernel
    try:
        check(H.conv_wgrad(x.cuda(), g.cuda(), Co, Ci, k, stride), ref, dtype, "plain, narrow routing")
    finally:
        H.lib.lcgan_set_option(5, old)
    px, pg = vec((B, ceil8(Ci)), 23), vec((B, ceil8(Co)), 24)
    got, ref_p = H.conv_wgrad(x.cuda(), g.cuda(), Co, Ci, k, stride, pre_x=px.cuda(), pre_g=pg.cuda()), E.conv_wgrad(x, g, Co, Ci, k, stride, pre_x=px, pre_g=pg)
    if dtype == torch.float32:
        check(got, ref_p, dtype, "prescaled")
    else:   # the generic bf16 kernel rounds the prescaled operands to bf16 (2^-9 per operand): L2 up to ~3e-3
        e_l2 = float((got.float().cpu() - ref_p).norm() / ref_p.norm())
        assert e_l2 <= 4e-3, e_l2
    # un-prep, both orientations, with the demod term
    w = torch.randn(Co, Ci, k, k, generator=torch.Generator().manual_seed(25))
    gwsq = torch.randn(Co, Ci, generator=torch.Generator().manual_seed(26))
    check(H.unprep_wgrad(ref.cuda(), Co, Ci, k, 0.3, False, w.cuda(), gwsq.cuda()), E.unprep_wgrad(ref, Co, Ci, k, 0.3, False, w, gwsq),
          torch.float32, "unprep")
    refT = ref.permute(0, 2, 1).contiguous()
    check(H.unprep_wgrad(refT.cuda(), Co, Ci, k, 0.3, True, w.cuda(), gwsq.cuda()), E.unprep_wgrad(refT, Co, Ci, k, 0.3, True, w, gwsq),
          torch.float32, "unprep-T")


def test_mfma_layout_asymmetric(H):
    """A = identity-like input against an asymmetric weight: catches a transposed C/D or operand map (guide: sec. 3)."""
    C = 32
    x = torch.zeros(1, 8, 8, C)
    for c in range(C):
        x[0, c % 8, c // 8, c] = 1.0 + c            # one-hot pixels/channels
    w = (torch.arange(C * C, dtype=torch.float32).reshape(C, C, 1, 1) % 17) - 8.0
    pw_e, _ = E.prep_weight(w, 1.0, False, True)
    pw_h, _ = H.prep_weight(w.cuda(), 1.0, False, True)
    got, ref = H.conv_fwd(x.cuda(), pw_h, C, 1, 1).cpu(), E.conv_fwd(x, pw_e, C, 1, 1)
    assert torch.allclose(got, ref, atol=1e-3), float((got - ref).abs().max())
    gw = H.conv_wgrad(x.cuda(), ref.cuda(), C, C, 1, 1).cpu()
    assert torch.allclose(gw, E.conv_wgrad(x, ref, C, C, 1, 1), rtol=1e-3, atol=1e-2)


# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2, 8, 8, 16), (3, 12, 20, 40), (1, 32, 32, 128),
                                   (1, 20, 12, 256), (1, 10, 6, 512), (1, 6, 8, 16)])   # 2 columns per wave; 1 (no lane sharing); a partial wave
def test_stencils(H, dtype, shape):
    x = feat(shape, dtype, 31)
    for act, gain in ((0, 1.0), (1, 1.4), (2, 1.0)):
        y_ref = E.box3_act(x, act, gain)
        check(H.box3_act(x.cuda(), act, gain), y_ref, dtype, f"box3 act{act}")
        gy = feat(shape, dtype, 32)
        check(H.box3_act_bwd(gy.cuda(), y_ref.cuda(), act, gain), E.box3_act_bwd(gy, y_ref, act, gain), dtype, f"box3 bwd act{act}")
    B, Hh, W, C = shape
    res = feat((B, 2 * Hh, 2 * W, C), dtype, 33)
    check(H.up2box(x.cuda(), res.cuda()), E.up2box(x, res), dtype, "up2box+res")
    check(H.up2box(x.cuda(), None), E.up2box(x, None), dtype, "up2box")
    check(H.up2box_bwd(res.cuda()), E.up2box_bwd(res), dtype, "up2box bwd")
    check(H.avgpool2(x.cuda()), E.avgpool2(x), dtype, "avgpool2")
    check(H.avgpool2_bwd(x.cuda()), E.avgpool2_bwd(x), dtype, "avgpool2 bwd")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape,clog", [((2, 8, 8, 16), 16), ((3, 10, 6, 8), 2), ((2, 16, 16, 520), 513), ((2, 64, 64, 128), 128)])
def test_act_bwd_reduce_and_scale_reduce(H, dtype, shape, clog):
    B, Hh, W, C = shape
    gy, y = feat(shape, dtype, 41, clog), feat(shape, dtype, 42, clog)
    bias = torch.randn(clog, generator=torch.Generator().manual_seed(43))
    for act, gain in ((1, 1.4), (0, 1.0)):
        r = E.act_bwd_reduce(gy, y, act, gain, clog, True, bias, 0.7, True, True)
        g = H.act_bwd_reduce(gy.cuda(), y.cuda(), act, gain, clog, True, bias.cuda(), 0.7, True, True)
        check(g[0], r[0], dtype, "gz")
        check(g[1], r[1], torch.float32, "gbias")
        check(g[2], r[2], torch.float32, "gdq")
        # the stored gradient with a per-(sample, channel) factor (a modulated convolution's demodulation): reductions unchanged bit for bit
        osc = vec((B, C), 47) + 1.5
        rs = E.act_bwd_reduce(gy, y, act, gain, clog, True, bias, 0.7, True, True, out_scale=osc)
        gs_ = H.act_bwd_reduce(gy.cuda(), y.cuda(), act, gain, clog, True, bias.cuda(), 0.7, True, True, out_scale=osc.cuda())
        check(gs_[0], rs[0], dtype, "gz * out_scale")
        check(gs_[1], r[1], torch.float32, "gbias (out_scale)")
        check(gs_[2], r[2], torch.float32, "gdq (out_scale)")
        one = H.act_bwd_reduce(gy.cuda(), y.cuda(), act, gain, clog, True, bias.cuda(), 0.7, True, True, out_scale=torch.ones(B, C).cuda())
        assert torch.equal(one[0], g[0]), "out_scale == 1 must store the plain gz"
    u, x = feat(shape, dtype, 44), feat(shape, dtype, 45)
    s = vec((B, C), 46)
    u_ref, gs_ref = E.scale_reduce(u.clone(), x, s)
    u_got, gs_got = H.scale_reduce(u.cuda(), x.cuda(), s.cuda())
    check(u_got, u_ref, dtype, "scaled u")
    check(gs_got, gs_ref, torch.float32, "gs")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2, 8, 8, 32), (2, 16, 16, 128), (1, 32, 32, 512)])
def test_warp(H, dtype, shape):
    B, Hh, W, C = shape
    x = feat(shape, dtype, 51)
    flow = torch.zeros(B, Hh, W, 8)
    flow[..., :2] = torch.tanh(torch.randn(B, Hh, W, 2, generator=torch.Generator().manual_seed(52)))
    flow = flow.to(dtype)
    check(H.warp_fwd(x.cuda(), flow.cuda(), 0.1), E.warp_fwd(x, flow, 0.1), dtype, "warp fwd")
    gy = feat(shape, dtype, 53)
    gx_r, gf_r = E.warp_bwd(gy, x, flow, 0.1)
    gx_g, gf_g = H.warp_bwd(gy.cuda(), x.cuda(), flow.cuda(), 0.1)
    check(gx_g, gx_r, dtype, "warp gx")
    check(gf_g, gf_r, dtype, "warp gflow")


def test_warp_bwd_collapsing_flow(H):
    """Every output pixel samples (almost) the same point: 16 * 256 taps land on a handful of input pixels.  The backward's
    per-input-pixel lists are exact CSR lists (count, scan, fill), so the gather must still be exact."""
    B, Hh, W, C = 2, 16, 16, 32
    dtype = torch.float32
    x, gy = feat((B, Hh, W, C), dtype, 54), feat((B, Hh, W, C), dtype, 55)
    gyy, gxx = torch.meshgrid(torch.arange(Hh, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    flow = torch.zeros(B, Hh, W, 8)
    flow[..., 0] = -(2 * gxx / (W - 1) - 1) + 0.013        # cancels the base grid: every pixel lands near the centre
    flow[..., 1] = -(2 * gyy / (Hh - 1) - 1) - 0.021
    gx_r, gf_r = E.warp_bwd(gy, x, flow, 1.0)
    gx_g, gf_g = H.warp_bwd(gy.cuda(), x.cuda(), flow.cuda(), 1.0)
    check(gx_g, gx_r, dtype, "collapsed gx")
    check(gf_g, gf_r, dtype, "collapsed gflow")


def test_warp_bwd_rough_flow_large(H):
    """A flow that compresses areas several-fold on a grid large enough for the multi-tile prefix scan (65 536 + 1 counts):
    the first implementation (fixed 32-entry lists + capped overflow list) dropped entries here."""
    B, Hh, W, C = 1, 256, 256, 16
    dtype = torch.float32
    x, gy = feat((B, Hh, W, C), dtype, 56), feat((B, Hh, W, C), dtype, 57)
    g = torch.Generator().manual_seed(58)
    coarse = torch.randn(B, 2, 8, 8, generator=g)
    flow = torch.zeros(B, Hh, W, 8)
    flow[..., :2] = torch.tanh(torch.nn.functional.interpolate(coarse, size=(Hh, W), mode="bilinear", align_corners=False).permute(0, 2, 3, 1)) * 4
    gx_r, gf_r = E.warp_bwd(gy, x, flow, 0.1)
    gx_g, gf_g = H.warp_bwd(gy.cuda(), x.cuda(), flow.cuda(), 0.1)
    check(gx_g, gx_r, dtype, "rough gx")
    check(gf_g, gf_r, dtype, "rough gflow")


def test_warp_pixel_count_beyond_2_24(H):
    """more than 2^24 pixels in one call (1024 x 1024 at batch 32 has 2^25): the forward and the backward of the LAST samples of a
    big batch -- whose pixel indices are the large ones -- must equal the same samples run as a small batch (the 24-bit index
    multiplies of the first round-3 version of the gather kernel refused such calls)"""
    dtype = torch.bfloat16
    B, Hh, W, C = 18, 1024, 1024, 8                              # 18.9 M pixels
    g = torch.Generator(device="cuda").manual_seed(59)
    x = torch.randn(B, Hh, W, C, device="cuda", generator=g).to(dtype)
    gy = torch.randn(B, Hh, W, C, device="cuda", generator=g).to(dtype)
    coarse = torch.randn(B, 2, 16, 16, device="cuda", generator=g)
    flow = torch.zeros(B, Hh, W, 8, device="cuda")
    flow[..., :2] = torch.tanh(torch.nn.functional.interpolate(coarse, size=(Hh, W), mode="bilinear", align_corners=False).permute(0, 2, 3, 1))
    flow = flow.to(dtype)
    y = H.warp_fwd(x, flow, 0.1)
    gx, gf = H.warp_bwd(gy, x, flow, 0.1)
    torch.cuda.synchronize()
    tail = slice(B - 2, B)
    y2 = H.warp_fwd(x[tail].contiguous(), flow[tail].contiguous(), 0.1)
    gx2, gf2 = H.warp_bwd(gy[tail].contiguous(), x[tail].contiguous(), flow[tail].contiguous(), 0.1)
    assert torch.equal(y[tail], y2)
    assert torch.equal(gf[tail], gf2)
    rel = float((gx[tail].float() - gx2.float()).norm() / gx2.float().norm())
    assert rel <= 2e-3, rel                                      # (list order = arrival order of the fill pass: fp32 sums in another order)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("N,G,C", [(4, 4, 64), (8, 8, 64), (16, 8, 64), (32, 8, 64), (32, 4, 512), (8, 4, 20)])
def test_mbstd(H, dtype, N, G, C):
    """C = 20: the element-per-thread path of channel counts that are not a multiple of 8; the others take 8 channels per thread"""
    x = feat((N, 4, 4, C), dtype, 61)
    Cy = ceil8(C + 1)
    check(H.mbstd_fwd(x.cuda(), G, Cy), E.mbstd_fwd(x, G, Cy), dtype, "fwd")
    gy = feat((N, 4, 4, Cy), dtype, 62, C + 1)
    check(H.mbstd_bwd(gy.cuda(), x.cuda(), G), E.mbstd_bwd(gy, x, G), dtype, "bwd")
    v = feat((N, 4, 4, C), dtype, 63)
    r = E.mbstd_bwd2(v, gy, x, G)
    g = H.mbstd_bwd2(v.cuda(), gy.cuda(), x.cuda(), G)
    check(g[0], r[0], dtype, "bwd2 ggy")
    check(g[1], r[1], dtype, "bwd2 gx")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("per_sample", [False, True])
def test_rgb(H, dtype, per_sample):
    B, Hh, W, C = 3, 16, 16, 128
    img = torch.randn(B, 3, Hh, W, generator=torch.Generator().manual_seed(71))
    w = torch.randn(B if per_sample else 1, 3, C, generator=torch.Generator().manual_seed(72)) * 0.3
    bias = torch.randn(C, generator=torch.Generator().manual_seed(73))
    check(H.rgb_expand(img.cuda(), w.cuda(), bias.cuda(), 0.5, C, 1, 1.2, dtype), E.rgb_expand(img, w, bias, 0.5, C, 1, 1.2, dtype), dtype, "expand")
    x = feat((B, Hh, W, C), dtype, 74)
    b3 = torch.randn(3, generator=torch.Generator().manual_seed(75))
    check(H.rgb_reduce(x.cuda(), w.cuda(), b3.cuda(), 1.0), E.rgb_reduce(x, w, b3, 1.0), torch.float32 if dtype == torch.float32 else dtype, "reduce")
    check(H.rgb_wgrad(img.cuda(), x.cuda(), per_sample), E.rgb_wgrad(img, x, per_sample), torch.float32 if dtype == torch.float32 else dtype, "wgrad")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("per_sample", [False, True])
@pytest.mark.parametrize("shape", [(3, 16, 16, 128), (2, 9, 7, 64), (1, 40, 40, 32)])
def test_rgb_fused_backward(H, dtype, per_sample, shape):
    """lcgan_rgb_expand_bwd / lcgan_rgb_reduce_bwd_act: the one-pass backward of the two layers that touch the image, against the
    emulation AND against the three-pass composition they replace (activation backward -> rgb_reduce / rgb_wgrad)."""
    B, Hh, W, C = shape
    clog = C
    f32ish = torch.float32 if dtype == torch.float32 else dtype
    img = torch.randn(B, 3, Hh, W, generator=torch.Generator().manual_seed(71))
    w = torch.randn(B if per_sample else 1, 3, C, generator=torch.Generator().manual_seed(72)) * 0.3
    gy, y = feat((B, Hh, W, C), dtype, 76), feat((B, Hh, W, C), dtype, 77)
    for flags in ((True, True, True), (True, False, False), (False, True, True)):
        got = H.rgb_expand_bwd(gy.cuda(), y.cuda(), img.cuda() if flags[1] else None, w.cuda(), 1, 1.2, clog, *flags)
        ref = E.rgb_expand_bwd(gy, y, img, w, 1, 1.2, clog, *flags)
        for name, g, r in zip(("gimg", "gw", "gbias"), got, ref):
            assert (g is None) == (r is None), name
            if g is not None:
                check(g, r, f32ish, f"expand_bwd {name} {flags}")
    # the composition: gz rounded to the feature dtype in between (hence the looser bf16 tolerance is the right one)
    gz, gb, _ = H.act_bwd_reduce(gy.cuda(), y.cuda(), 1, 1.2, clog, want_gz=True, want_gbias=True)
    gimg, gw, gbias = H.rgb_expand_bwd(gy.cuda(), y.cuda(), img.cuda(), w.cuda(), 1, 1.2, clog, True, True, True)
    check(gimg, H.rgb_reduce(gz, w.cuda(), None, 0.0).cpu(), f32ish, "expand_bwd vs composition: gimg", l2_scale=2.0)
    check(gw, H.rgb_wgrad(img.cuda(), gz, per_sample).cpu(), f32ish, "expand_bwd vs composition: gw", l2_scale=2.0)
    check(gbias, gb.cpu(), f32ish, "expand_bwd vs composition: gbias", l2_scale=2.0)

    # recompute form: the activation's sign from the image (w . img + bias) instead of the saved output -- with y the layer's own output both
    # forms see the same signs, so everything agrees to reduction order
    fb = torch.randn(C, generator=torch.Generator().manual_seed(80)) * 0.2
    y_own = H.rgb_expand(img.cuda(), w.cuda(), fb.cuda(), 0.7, clog, 1, 1.2, dtype)
    a = H.rgb_expand_bwd(gy.cuda(), y_own, img.cuda(), w.cuda(), 1, 1.2, clog, True, True, True)
    r = H.rgb_expand_bwd(gy.cuda(), None, img.cuda(), w.cuda(), 1, 1.2, clog, True, True, True, fbias=fb.cuda(), fbias_scale=0.7, recompute=True)
    assert torch.equal(a[0], r[0]), "recomputed sign: gimg differs"
    for name, u, v in zip(("gw", "gbias"), a[1:], r[1:]):
        assert torch.allclose(u, v, rtol=1e-4, atol=1e-5 * float(u.abs().max())), name           # (float atomics: order)

    gimg_in = torch.randn(B, 3, Hh, W, generator=torch.Generator().manual_seed(78))
    bias = torch.randn(C, generator=torch.Generator().manual_seed(79)) * 0.2
    got = H.rgb_reduce_bwd_act(gimg_in.cuda(), y.cuda(), w.cuda(), bias.cuda(), 1.0, 1, 1.3, clog)
    ref = E.rgb_reduce_bwd_act(gimg_in, y, w, bias, 1.0, 1, 1.3, clog)
    check(got[0], ref[0], dtype, "reduce_bwd_act gz")
    for name, g, r in zip(("gbias", "gdq", "gwm"), got[1:], ref[1:]):
        check(g, r, f32ish, f"reduce_bwd_act {name}")
    osc = vec((B, C), 80) + 1.5
    gots = H.rgb_reduce_bwd_act(gimg_in.cuda(), y.cuda(), w.cuda(), bias.cuda(), 1.0, 1, 1.3, clog, out_scale=osc.cuda())
    check(gots[0], E.rgb_reduce_bwd_act(gimg_in, y, w, bias, 1.0, 1, 1.3, clog, out_scale=osc)[0], dtype, "reduce_bwd_act gz * out_scale")
    for name, g, r in zip(("gbias", "gdq", "gwm"), gots[1:], ref[1:]):
        check(g, r, f32ish, f"reduce_bwd_act {name} (out_scale)")
    gfeat = H.rgb_expand(gimg_in.cuda(), w.cuda(), None, 0.0, clog, 0, 1.0, dtype)
    gz2, gb2, gdq2 = H.act_bwd_reduce(gfeat, y.cuda(), 1, 1.3, clog, want_gz=True, bias=bias.cuda(), bias_scale=1.0, want_gbias=True, want_gdq=True)
    check(got[0], gz2.cpu(), dtype, "reduce_bwd_act vs composition: gz", l2_scale=2.0)     # (the composition rounds gfeat to bf16 in between)
    check(got[1], gb2.cpu(), f32ish, "reduce_bwd_act vs composition: gbias", l2_scale=2.0)
    check(got[2], gdq2.cpu(), f32ish, "reduce_bwd_act vs composition: gdq", l2_scale=2.0)
    check(got[3], H.rgb_wgrad(gimg_in.cuda(), y.cuda(), per_sample).cpu(), f32ish, "reduce_bwd_act vs composition: gwm", l2_scale=2.0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2, 4, 4), (3, 16, 24), (1, 5, 7)])
def test_flow_col2im_im2col(H, dtype, shape):
    """the scatter / gather halves of the flow layer's x2 transposed convolution against the emulation, and their adjointness"""
    B, Hh, W = shape
    t = feat((B, Hh, W, 24), dtype, 31, 18)
    d = vec((B, 8), 32)
    bias = torch.randn(2, generator=torch.Generator().manual_seed(33))
    u_h, u_e = H.flow_col2im(t.cuda(), d.cuda(), bias.cuda()), E.flow_col2im(t, d, bias)
    check(u_h, u_e, dtype, "col2im")
    assert float(u_h[..., 2:].abs().max()) == 0.0
    gu = feat((B, 2 * Hh, 2 * W, 8), dtype, 34, 2)
    gt_h, gt_e = H.flow_im2col(gu.cuda(), d.cuda()), E.flow_im2col(gu, d)
    check(gt_h, gt_e, dtype, "im2col")
    assert float(gt_h[..., 18:].abs().max()) == 0.0
    # <col2im(t) - bias, gu> == <t, im2col(gu)>  (d enters both)
    u0 = E.flow_col2im(t, d, None).double()
    lhs, rhs = float((u0 * gu.double()).sum()), float((t.double() * gt_e.double()).sum())
    assert abs(lhs - rhs) <= (2e-2 if dtype == torch.bfloat16 else 1e-5) * max(abs(lhs), 1.0), (lhs, rhs)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2, 8, 8, 64), (2, 32, 32, 128), (3, 16, 32, 256), (8, 64, 64, 64)])
def test_flow_layer_gemm_vs_generic(dtype, shape):
    """ops.FlowConvFn (1x1 GEMM + scatter) against ops.ModConvFn (the generic x2 transposed convolution) on the flow layer's shapes:
    forward and every gradient, both on the HIP kernels."""
    from lcgan_amd import config, ops
    from tests.helpers import install_backend
    install_backend(None)
    B, Hh, W, Cin = shape
    g = torch.Generator().manual_seed(41)
    x = feat((B, Hh, W, Cin), dtype, 42).cuda()
    w = torch.nn.Parameter(torch.randn(2, Cin, 3, 3, generator=g).cuda())
    bias = torch.nn.Parameter((torch.randn(2, generator=g) * 0.1).cuda())
    s0 = (torch.rand(B, Cin, generator=g) + 0.5).cuda()
    go = feat((B, 2 * Hh, 2 * W, 8), dtype, 43, 2).cuda()
    outs = []
    with config.feature_dtype_as(dtype):
        for fn in (lambda xx, ss: ops.FlowConvFn.apply(xx, w, bias, ss), lambda xx, ss: ops.ModConvFn.apply(xx, w, bias, ss, 2, 0, 1.0)):
            xx, ss = x.clone().requires_grad_(True), s0.clone().requires_grad_(True)
            w.grad = bias.grad = None
            y = fn(xx, ss)
            (y.float() * go.float()).sum().backward()
            outs.append((y.detach(), xx.grad, ss.grad, w.grad.clone(), bias.grad.clone()))
    for name, a, b in zip(("u", "gx", "gs", "gw", "gb"), *outs):
        # two different roundings of the same sums in bf16 (t is stored per tap before the scatter): twice the single-kernel tolerance
        check(a, b.cpu(), dtype if name in ("u", "gx") else (torch.float32 if dtype == torch.float32 else dtype), f"flow {name}", l2_scale=3.0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2, 16, 16, 64, 64), (3, 32, 32, 128, 128), (2, 64, 64, 128, 72)])
def test_modconv_backward_with_demodulated_gradient(dtype, shape):
    """ops.ModConvFn's backward with the activation backward storing d * gz (config.gz_demod: the data- and weight-gradient launches then
    carry no per-sample scale on that operand) against the per-sample-scale form: every gradient, both on the HIP kernels"""
    from lcgan_amd import config, ops
    from tests.helpers import install_backend
    install_backend(None)
    B, Hh, W, Cin, O = shape
    g = torch.Generator().manual_seed(141)
    x = feat((B, Hh, W, Cin), dtype, 142).cuda()
    w = torch.nn.Parameter(torch.randn(O, Cin, 3, 3, generator=g).cuda())
    bias = torch.nn.Parameter((torch.randn(O, generator=g) * 0.1).cuda())
    s0 = (torch.rand(B, Cin, generator=g) + 0.5).cuda()
    go = feat((B, Hh, W, ceil8(O)), dtype, 143, O).cuda()
    outs = []
    was = config.gz_demod()
    try:
        with config.feature_dtype_as(dtype):
            for on in (True, False):
                config.set_gz_demod(on)
                xx, ss = x.clone().requires_grad_(True), s0.clone().requires_grad_(True)
                w.grad = bias.grad = None
                y = ops.ModConvFn.apply(xx, w, bias, ss, 1, 1, 1.4)
                (y.float() * go.float()).sum().backward()
                outs.append((xx.grad, ss.grad, w.grad.clone(), bias.grad.clone()))
    finally:
        config.set_gz_demod(was)
    # the bias gradient reduces the UNSCALED gz in both forms: equal up to the order of its float atomics
    assert torch.allclose(outs[0][3], outs[1][3], rtol=1e-4, atol=1e-5 * float(outs[1][3].abs().max())), "gbias"
    for name, a, b in zip(("gx", "gs", "gw"), *outs):
        # bf16: d enters once as a rounding of the fp32 product instead of on the staged bf16 operand -- two roundings of the same sums
        check(a, b.cpu(), dtype if name == "gx" else (torch.float32 if dtype == torch.float32 else dtype), f"modconv {name}", l2_scale=3.0)


def test_gradient_accumulation_over_two_backward_passes():
    """Two forward / backward passes of a modulated convolution WITHOUT zero_grad() in between: the second backward accumulates in place into
    gradients autograd took over from the first -- slices of the zero pool's slab, like the demodulation vector the second forward saved.
    (As views of one slab they shared a version counter and the second backward raised "modified by an inplace operation".)"""
    from lcgan_amd import config, ops
    from tests.helpers import install_backend
    install_backend(None)
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(151)
    x = feat((2, 16, 16, 64), dtype, 152).cuda()
    w = torch.nn.Parameter(torch.randn(64, 64, 3, 3, generator=g).cuda())
    bias = torch.nn.Parameter((torch.randn(64, generator=g) * 0.1).cuda())
    s0 = (torch.rand(2, 64, generator=g) + 0.5).cuda()
    with config.feature_dtype_as(dtype):
        grads = []
        for it in range(2):
            ss = s0.clone().requires_grad_(True)
            y = ops.ModConvFn.apply(x, w, bias, ss, 1, 1, 1.4)
            y.float().sum().backward()
            grads.append((w.grad.clone(), bias.grad.clone()))
    assert torch.allclose(grads[1][0], 2 * grads[0][0], rtol=1e-3, atol=1e-3 * float(grads[0][0].abs().max()))
    assert torch.allclose(grads[1][1], 2 * grads[0][1], rtol=1e-3, atol=1e-3 * float(grads[0][1].abs().max()))


@pytest.mark.parametrize("dtype", DTYPES)
def test_layout(H, dtype):
    src = torch.randn(3, 13, 4, 4, generator=torch.Generator().manual_seed(81))
    check(H.nchw_to_nhwc(src.cuda(), 3, 16, dtype), E.nchw_to_nhwc(src, 3, 16, dtype), dtype, "to nhwc")
    check(H.nchw_to_nhwc(src[:1].contiguous().cuda(), 5, 16, dtype), E.nchw_to_nhwc(src[:1], 5, 16, dtype), dtype, "broadcast")
    f = feat((3, 4, 4, 16), dtype, 82, 13)
    check(H.nhwc_to_nchw(f.cuda(), 13, False), E.nhwc_to_nchw(f, 13, False), torch.float32, "to nchw")
    check(H.nhwc_to_nchw(f.cuda(), 13, True), E.nhwc_to_nchw(f, 13, True), torch.float32, "reduce")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(2, 40, 24, 16), (1, 7, 9, 72), (3, 64, 64, 128)])
def test_box3_actbwd(H, dtype, shape):
    """backward of act -> blur in one pass: gz = box3(gy) * act'(y), gbias = sum gz"""
    B, Hh, W, C = shape
    gy, y = feat((B, Hh, W, ceil8(C)), dtype, 61, C), feat((B, Hh, W, ceil8(C)), dtype, 62, C)
    gz, gb = H.box3_actbwd(gy.cuda(), y.cuda(), 1, 1.4, C, True)
    ez, eb = E.box3_actbwd(gy, y, 1, 1.4, C, True)
    check(gz, ez, dtype, "gz")
    check(gb, eb, torch.float32, "gbias")
    assert H.box3_actbwd(gy.cuda(), y.cuda(), 1, 1.4, C, False)[1] is None


# ------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,I,O", [(8, 64, 64), (32, 8192, 512), (5, 6, 8), (96, 512, 1), (32, 512, 2048)])
def test_linear(H, M, I, O):
    g = torch.Generator().manual_seed(91)
    x, w, b = torch.randn(M, I, generator=g), torch.randn(O, I, generator=g), torch.randn(O, generator=g)
    sc = 1 / math.sqrt(I)
    f32 = torch.float32
    check(H.linear_fwd(x.cuda(), w.cuda(), b.cuda(), sc, 0.01, 1, 1.0), E.linear_fwd(x, w, b, sc, 0.01, 1, 1.0), f32, "fwd")
    gy = torch.randn(M, O, generator=g)
    check(H.linear_bwd_data(gy.cuda(), w.cuda(), sc), E.linear_bwd_data(gy, w, sc), f32, "bwd data")
    check(H.linear_wgrad(gy.cuda(), x.cuda(), sc), E.linear_wgrad(gy, x, sc), f32, "wgrad")
    check(H.colsum(gy.cuda(), 0.01), E.colsum(gy, 0.01), f32, "colsum")
    gw, gb = H.linear_wgrad_bias(gy.cuda(), x.cuda(), sc, 0.01)           # both from one launch
    check(gw, E.linear_wgrad(gy, x, sc), f32, "wgrad (fused)")
    check(gb, E.colsum(gy, 0.01), f32, "colsum (fused)")
    check(H.act_bwd_f32(gy.cuda(), x[:, :1].expand(M, O).contiguous().cuda(), 1, 1.0), E.act_bwd_f32(gy, x[:, :1].expand(M, O), 1, 1.0), f32, "act bwd")


@pytest.mark.parametrize("M,I,Os", [(32, 512, [512, 512, 256, 128, 8]), (4, 64, [64] * 24), (37, 100, [3, 130])])
def test_linear_group(H, M, I, Os):
    """grouped launches (the generator's style affines) against the per-layer emulation"""
    g = torch.Generator().manual_seed(92)
    x = torch.randn(M, I, generator=g)
    ws = [torch.randn(O, I, generator=g) for O in Os]
    bs = [torch.randn(O, generator=g) for O in Os]
    scs = [1 / math.sqrt(I) * (1 + 0.1 * l) for l in range(len(Os))]
    bss = [1.0 - 0.01 * l for l in range(len(Os))]
    f32 = torch.float32
    ys = H.linear_group_fwd(x.cuda(), [w.cuda() for w in ws], [b.cuda() for b in bs], scs, bss)
    for l, (y, e) in enumerate(zip(ys, E.linear_group_fwd(x, ws, bs, scs, bss))):
        check(y, e, f32, f"fwd {l}")
    gys = [torch.randn(M, O, generator=g) for O in Os]
    gx, gws, gbs = H.linear_group_bwd([t.cuda() for t in gys], x.cuda(), [w.cuda() for w in ws], scs, bss)
    ex, ews, ebs = E.linear_group_bwd(gys, x, ws, scs, bss)
    check(gx, ex, f32, "gx")
    for l in range(len(Os)):
        check(gws[l], ews[l], f32, f"gw {l}")
        check(gbs[l], ebs[l], f32, f"gb {l}")
    assert H.linear_group_bwd([t.cuda() for t in gys], x.cuda(), [w.cuda() for w in ws], scs, bss, want_gx=False)[0] is None


@pytest.mark.parametrize("M,shapes", [(32, [(64, 64), (512, 512)]), (4, [(64, 64), (128, 64)]), (5, [(64, 96), (100, 130), (8, 24)]), (37, [(512, 256), (64, 64)])])
def test_linear_multi(H, M, shapes):
    """linear layers with their own inputs and shapes (the two mapping networks at equal depth) from shared launches"""
    g = torch.Generator().manual_seed(93)
    L = len(shapes)
    xs = [torch.randn(M, I, generator=g) for I, _ in shapes]
    ws = [torch.randn(O, I, generator=g) for I, O in shapes]
    bs = [torch.randn(O, generator=g) for _, O in shapes]
    scs = [1 / math.sqrt(I) * (1 + 0.1 * l) for l, (I, _) in enumerate(shapes)]
    bss = [0.01 * (1 + l) for l in range(L)]
    f32 = torch.float32
    cu = lambda ts: [t.cuda() for t in ts]
    for l, (y, e) in enumerate(zip(H.linear_multi_fwd(cu(xs), cu(ws), cu(bs), scs, bss), E.linear_multi_fwd(xs, ws, bs, scs, bss))):
        check(y, e, f32, f"fwd {l}")
    gys = [torch.randn(M, O, generator=g) for _, O in shapes]
    gxs, gws, gbs = H.linear_multi_bwd(cu(gys), cu(xs), cu(ws), scs, bss)
    exs, ews, ebs = E.linear_multi_bwd(gys, xs, ws, scs, bss)
    for l in range(L):
        check(gxs[l], exs[l], f32, f"gx {l}")
        check(gws[l], ews[l], f32, f"gw {l}")
        check(gbs[l], ebs[l], f32, f"gb {l}")
    assert H.linear_multi_bwd(cu(gys), cu(xs), cu(ws), scs, bss, want_gx=False)[0] is None


def test_demod_group(H):
    """demodulation vectors of several layers from one launch == the per-layer kernel"""
    g = torch.Generator().manual_seed(102)
    B = 5
    shapes = [(64, 24, 24), (512, 512, 512), (128, 2, 8), (72, 40, 40)]          # (C, O, alloc stride)
    ss = [torch.randn(B, C, generator=g) + 1 for C, _, _ in shapes]
    wsqs = [torch.rand(O, C, generator=g) * 0.1 for C, O, _ in shapes]
    got = H.demod_group([t.cuda() for t in ss], [t.cuda() for t in wsqs], [o for _, _, o in shapes])
    for (C, O, ost), s_, w_, d in zip(shapes, ss, wsqs, got):
        check(d, E.demod_fwd(s_, w_, ost), torch.float32, f"C{C} O{O}")
        assert torch.equal(d, H.demod_fwd(s_.cuda(), w_.cuda(), ost))


def test_demod(H):
    g = torch.Generator().manual_seed(101)
    B, C, O = 4, 64, 24
    s, wsq = torch.randn(B, C, generator=g) + 1, torch.rand(O, C, generator=g) * 0.1
    d_ref = E.demod_fwd(s, wsq, ceil8(O))
    check(H.demod_fwd(s.cuda(), wsq.cuda(), ceil8(O)), d_ref, torch.float32, "demod")
    gdq = torch.zeros(B, ceil8(O))
    gdq[:, :O] = torch.randn(B, O, generator=g)
    gs0 = torch.randn(B, C, generator=g)
    gs_ref = gs0.clone()
    gwsq_ref = E.demod_bwd(gdq, d_ref, s, wsq, gs_ref)
    gs_got = gs0.clone().cuda()
    gwsq_got = H.demod_bwd(gdq.cuda(), d_ref.cuda(), s.cuda(), wsq.cuda(), gs_got)
    check(gs_got, gs_ref, torch.float32, "gs")
    check(gwsq_got, gwsq_ref, torch.float32, "gwsq")


def test_losses(H):
    g = torch.Generator().manual_seed(111)
    f32 = torch.float32
    logit = torch.randn(32, 1, generator=g) * 3
    gout = torch.tensor(0.7)
    for t1 in (True, False):
        check(H.bce_fwd(logit.cuda(), t1), E.bce_fwd(logit, t1), f32, "bce")
        check(H.bce_bwd(logit.cuda(), t1, gout.cuda()), E.bce_bwd(logit, t1, gout), f32, "bce bwd")
    a, p, n = (torch.nn.functional.normalize(torch.randn(32, 256, generator=g)) for _ in range(3))
    lr, tr = E.contrastive_fwd(a, p, n, 0.05)
    lg, tg = H.contrastive_fwd(a.cuda(), p.cuda(), n.cuda(), 0.05)
    check(lg, lr, f32, "contrastive"), check(tg, tr, f32, "t")
    for x, y in zip(H.contrastive_bwd(a.cuda(), p.cuda(), n.cuda(), tg, gout.cuda(), 0.05), E.contrastive_bwd(a, p, n, tr, gout, 0.05)):
        check(x, y, f32, "contrastive bwd")
    x = torch.randn(32, 256, generator=g)
    yr, nr = E.l2norm_fwd(x)
    yg, ng = H.l2norm_fwd(x.cuda())
    check(yg, yr, f32, "l2norm"), check(ng, nr, f32, "norm")
    gy = torch.randn(32, 256, generator=g)
    check(H.l2norm_bwd(gy.cuda(), yg, ng), E.l2norm_bwd(gy, yr, nr), f32, "l2norm bwd")
    big = torch.randn(3 * 256 * 256 + 5, generator=g)
    for pw in (1, 2):
        check(H.powsum(big.cuda(), pw, 0.25), E.powsum(big, pw, 0.25), f32, "powsum")
        check(H.powsum_bwd(big.cuda(), pw, 0.25, gout.cuda()), E.powsum_bwd(big, pw, 0.25, gout), f32, "powsum bwd")
    w, avg = torch.randn(8, 512, generator=g), torch.randn(512, generator=g)
    avg_ref, avg_got = avg.clone(), avg.clone().cuda()
    E.avg_latent(w, avg_ref, 0.998), H.avg_latent(w.cuda(), avg_got, 0.998)
    check(avg_got, avg_ref, f32, "avg latent")


@pytest.mark.parametrize("n", [6, 33, 64])
def test_qr_householder(H, n):
    """One-workgroup Householder QR against LAPACK (torch.linalg.qr on the CPU), incl. the sign convention."""
    A = torch.tanh(torch.randn(n, n, generator=torch.Generator().manual_seed(121)))
    Qr, Rr = torch.linalg.qr(A, mode="reduced")
    Qg, Rg = H.qr(A.cuda())
    check(Qg, Qr, torch.float32, "Q")
    check(Rg, Rr, torch.float32, "R")
    assert (torch.sign(torch.diagonal(Rg.cpu())) == torch.sign(torch.diagonal(Rr))).all()


def test_qr_householder_batched_and_degenerate(H):
    """[nb, n, n] input: one workgroup per matrix; a 1 x 1 matrix, and a zero column (xn2 == 0: tau = 0, the LAPACK H = I case)."""
    A = torch.tanh(torch.randn(5, 17, 17, generator=torch.Generator().manual_seed(122)))
    A[3, 4:, 4] = 0.0                                    # column 4 of matrix 3 is already upper-triangular
    Qr, Rr = torch.linalg.qr(A, mode="reduced")
    Qg, Rg = H.qr(A.cuda())
    check(Qg, Qr, torch.float32, "batched Q")
    check(Rg, Rr, torch.float32, "batched R")
    one = torch.tensor([[-0.75]])
    Q1, R1 = H.qr(one.cuda())
    Qc, Rc = torch.linalg.qr(one)
    assert torch.equal(Q1.cpu(), Qc) and torch.equal(R1.cpu(), Rc)


# ------------------------------------------------------------------------------------------------------------
# MX-fp8 convolution path (csrc/conv_fp8.hip, BASELINE configs[4]).  Two checks per case:
#  (1) against the CPU emulation that quantises with the SAME rule (oracle/hip_emulation.py:mx_quant): products are exact in fp32,
#      so only the summation order and the final bf16 rounding differ -- the bf16 tolerances of the other conv tests apply; this pins
#      the operand layout, the block scales and every tap / phase / epilogue variant;
#  (2) against the UNQUANTISED convolution: the stated precision of the path.  e4m3 keeps 3 mantissa bits (relative rounding error
#      <= 2^-4 per operand, uniform on average 2^-5), and a K-term dot product of independent errors comes out at ~3-4 % of the
#      output's RMS: bound 6e-2 relative L2.
FP8_CASES = [
    # B, H, W, Cin, Cout, k, stride
    (2, 32, 32, 64, 128, 3, 1),
    (1, 64, 64, 128, 128, 3, 1),
    (2, 32, 32, 192, 96, 3, 2),      # Cin not a multiple of 128, Cout ragged
    (2, 16, 32, 72, 24, 3, 1),       # Cin not a multiple of 64 (zero-padded MX block), ragged tile
    (2, 32, 32, 128, 256, 1, 1),
]


@pytest.mark.parametrize("case", FP8_CASES)
def test_conv_fp8(H, case):
    B, Hh, W, Ci, Co, k, stride = case
    dtype = torch.bfloat16
    x = feat((B, Hh, W, ceil8(Ci)), dtype, 31, Ci)
    w = torch.randn(Co, Ci, k, k, generator=torch.Generator().manual_seed(32))
    bias = torch.randn(Co, generator=torch.Generator().manual_seed(33))
    scale = 1 / math.sqrt(Ci * k * k)
    pw_e, pw_h = E.prep_weight_fp8(w, scale, False), H.prep_weight_fp8(w.cuda(), scale, False)
    pre, post = vec((B, ceil8(Ci)), 34), vec((B, ceil8(Co)), 35)
    res = feat((B, Hh // stride, W // stride, ceil8(Co)), dtype, 36, Co)
    variants = [dict(), dict(bias=bias, bias_scale=0.5, act=1, gain=1.4), dict(pre=pre, post=post, bias=bias, act=1), dict(residual=res)]
    if stride == 1:                                                          # (the stride-2 forward conv stays on the bf16 kernel)
        for kw in variants:
            got = H.conv_fwd_fp8(x.cuda(), pw_h, Co, k, 1, **{a: dev(b) if isinstance(b, torch.Tensor) else b for a, b in kw.items()})
            check(got, E.conv_fwd_fp8(x, pw_e, Co, k, 1, **kw), dtype, f"fp8 fwd {sorted(kw)}")
        # precision against the unquantised convolution
        pw_ref, _ = E.prep_weight(w, scale, False, True)
        ref = E.conv_fwd(x.float(), pw_ref, Co, k, 1).float()
        got = H.conv_fwd_fp8(x.cuda(), pw_h, Co, k, 1).float().cpu()
        err = float((got - ref).norm() / ref.norm())
        assert err <= 6e-2, err
    # data gradient / transposed convolution (g on the strided grid)
    if not (stride == 2 and k != 3):
        Hg, Wg = Hh // stride, W // stride
        g = feat((B, Hg, Wg, ceil8(Co)), dtype, 37, Co, scale=1e-3)            # gradient-sized values: the block scales carry them
        pwT_e, pwT_h = E.prep_weight_fp8(w, scale, True), H.prep_weight_fp8(w.cuda(), scale, True)
        rh = feat((B, Hg * stride // 2, Wg * stride // 2, ceil8(Ci)), dtype, 38, Ci, scale=1e-3)
        for kw in (dict(), dict(pre=post, post=pre), dict(residual=rh, residual_half=True)):
            got = H.conv_bwd_data_fp8(g.cuda(), pwT_h, Ci, k, stride, **{a: dev(b) if isinstance(b, torch.Tensor) else b for a, b in kw.items()})
            check(got, E.conv_bwd_data_fp8(g, pwT_e, Ci, k, stride, **kw), dtype, f"fp8 dgrad {sorted(kw)}")


@pytest.mark.parametrize("case", [(4, 64, 64, 128, 128, 1), (2, 64, 64, 128, 256, 2), (2, 32, 32, 512, 512, 1), (3, 48, 40, 64, 96, 1)])
def test_activation_sign_masks(H, case):
    """lcgan_conv_fwd_m leaves the leaky-ReLU sign bits of its pre-activations as a by-product (bit j of byte v of a pixel = channel 8 v + j);
    the activation-backward kernels reading the mask give bit for bit what they give reading y (custom_layers.py:205,208)."""
    B, Hh, W, Ci, Co, stride = case
    dtype, k = torch.bfloat16, 3
    x = feat((B, Hh, W, ceil8(Ci)), dtype, 1, Ci)
    w = torch.randn(Co, Ci, k, k, generator=torch.Generator().manual_seed(2))
    bias = torch.randn(Co, generator=torch.Generator().manual_seed(3))
    pw, _ = H.prep_weight(w.cuda(), 1 / math.sqrt(Ci * k * k), False, False)
    old6 = H.lib.lcgan_set_option(6, 1)                           # (small grids may take the halo kernels)
    try:
        y, mask = H.conv_fwd(x.cuda(), pw, Co, k, stride, bias=bias.cuda(), bias_scale=0.5, act=1, gain=1.4, want_mask=True)
        y0 = H.conv_fwd(x.cuda(), pw, Co, k, stride, bias=bias.cuda(), bias_scale=0.5, act=1, gain=1.4)
    finally:
        H.lib.lcgan_set_option(6, old6)
    assert torch.equal(y, y0)
    if Co % 32:
        assert mask is None                                       # (96 channels: no whole mask words -> the caller keeps using y)
        return
    assert mask is not None and mask.shape == (y.shape[0] * y.shape[1] * y.shape[2], Co // 8)
    bits = (y.reshape(-1, Co // 8, 8) > 0).to(torch.int32)
    want = (bits << torch.arange(8, device="cuda", dtype=torch.int32)).sum(-1).to(torch.uint8)
    assert torch.equal(mask, want)
    gy = feat(tuple(y.shape), dtype, 9, Co).cuda()
    a = H.act_bwd_reduce(gy, y, 1, 1.4, Co, want_gz=True, want_gbias=True)
    b = H.act_bwd_reduce(gy, y, 1, 1.4, Co, want_gz=True, want_gbias=True, mask=mask)
    assert torch.equal(a[0], b[0]) and torch.allclose(a[1], b[1], rtol=1e-5, atol=1e-5 * float(a[1].abs().max()))
    a = H.box3_actbwd(gy, y, 1, 1.4, Co, True)
    b = H.box3_actbwd(gy, y, 1, 1.4, Co, True, mask=mask)
    assert torch.equal(a[0], b[0]) and torch.allclose(a[1], b[1], rtol=1e-5, atol=1e-5 * float(a[1].abs().max()))


@pytest.mark.parametrize("case", [(2, 64, 64, 128, 256), (4, 32, 32, 256, 512), (2, 64, 32, 64, 128), (6, 32, 32, 512, 128)])
def test_conv_s2duo_kernel(H, case):
    """conv_s2duo_kernel (two anti-phased teams per 1024-thread workgroup, option 26) against the one-stage stride-2 kernel it replaces:
    the same operand images, tap order and accumulation order, hence bit-identical outputs and masks; plus the emulation."""
    B, Hh, W, Ci, Co = case
    dtype, k = torch.bfloat16, 3
    x = feat((B, Hh, W, ceil8(Ci)), dtype, 1, Ci)
    w = torch.randn(Co, Ci, k, k, generator=torch.Generator().manual_seed(2))
    bias = torch.randn(Co, generator=torch.Generator().manual_seed(3))
    scale = 1 / math.sqrt(Ci * k * k)
    pw, _ = H.prep_weight(w.cuda(), scale, False, False)
    pw_e, _ = E.prep_weight(w, scale, False, False)
    old6 = H.lib.lcgan_set_option(6, 1)
    outs = {}
    try:
        for duo in (0, 2):
            old26 = H.lib.lcgan_set_option(26, duo)
            try:
                outs[duo] = (H.conv_fwd(x.cuda(), pw, Co, k, 2, bias=bias.cuda(), bias_scale=0.5, act=1, gain=1.4, want_mask=True),
                             H.conv_fwd(x.cuda(), pw, Co, k, 2))
            finally:
                H.lib.lcgan_set_option(26, old26)
    finally:
        H.lib.lcgan_set_option(6, old6)
    (ya, ma), pa = outs[0]
    (yb, mb), pb = outs[2]
    assert torch.equal(ya, yb) and torch.equal(pa, pb)
    assert (ma is None) == (mb is None) and (ma is None or torch.equal(ma, mb))
    check(yb, E.conv_fwd(x, pw_e, Co, k, 2, bias=bias, bias_scale=0.5, act=1, gain=1.4), dtype, "duo vs emulation")


def test_warp_of_4_gb_and_more_goes_in_halves(H):
    """a feature map of 4 GB or more is beyond the warp kernels' 32-bit byte offsets: the entry points process the batch in halves (samples are
    independent) instead of refusing it; the result equals the per-sample calls"""
    B, R, C = 4, 1024, 512                                    # 4 x 1024 x 1024 x 512 bf16 = exactly 4 GiB (the backward takes at most 512 channels)
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(B, R, R, C, device="cuda", dtype=torch.bfloat16, generator=g)
    flow = torch.zeros(B, R, R, 8, device="cuda", dtype=torch.bfloat16)
    flow[..., :2] = torch.randn(B, R, R, 2, device="cuda", generator=g).clamp(-1, 1).bfloat16()
    y = H.warp_fwd(x, flow, 0.1)
    for b in range(B):
        assert torch.equal(y[b:b + 1], H.warp_fwd(x[b:b + 1].contiguous(), flow[b:b + 1].contiguous(), 0.1))
    gy = torch.randn(B, R, R, C, device="cuda", dtype=torch.bfloat16, generator=g)
    gx, gflow = H.warp_bwd(gy, x, flow, 0.1)
    for b in range(B):
        gx1, gf1 = H.warp_bwd(gy[b:b + 1].contiguous(), x[b:b + 1].contiguous(), flow[b:b + 1].contiguous(), 0.1)
        assert torch.equal(gflow[b:b + 1], gf1)
        # (the transposed lists of one sample are filled through integer atomics: their ORDER, hence the fp32 summation order, differs from run to run)
        assert float((gx[b:b + 1].float() - gx1.float()).abs().max()) <= 2.0 ** -6 * float(gx1.float().abs().max())

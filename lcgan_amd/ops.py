"""torch.autograd.Function wrappers around the HIP kernels.

Two families:
  * double-differentiable (discriminator path, needed by the R1 penalty, loss.py:18-34): every backward is itself
    composed of Functions of this module, so autograd.grad(..., create_graph=True) through the discriminator and a
    second backward through that graph both run on the HIP kernels.  The convolution triple
    (Conv2dFn, ConvTransposeFn, ConvWeightGradFn) is closed under differentiation because conv is bilinear.
  * first-order only (generator path: ModConvFn, Box3ActFn, Up2BoxAddFn, WarpFn) with fused backward kernels.

Internal feature maps are NHWC [B,H,W,C] (C multiple of 8) in `feature dtype` (bf16, or f32 = parity mode).
"""
from __future__ import annotations

import math

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import kernels as KM
from .kernels import ACT_LRELU, ACT_NONE, ACT_TANH, ceil8

Tensor = torch.Tensor


def _K():
    return KM.K


def _need_lo(x: Tensor) -> bool:
    return x.dtype == torch.float32


# ---- "inputs only" backward ---------------------------------------------------------------------------------------
# autograd.grad(outputs, inputs=images, only_inputs=True) (loss.py:27-34 of the reference) asks for the image gradient only; ATen's
# convolution_backward then skips the weight / bias gradients through its output_mask.  A Python autograd.Function only sees
# needs_input_grad, which is fixed at forward time, so loss.cal_derivative raises this flag around its autograd.grad call and the
# first-order backward of every parameterised node drops its (discarded) parameter gradients: one whole discriminator weight-gradient
# pass per R1 iteration.  The graph from the data gradient to the WEIGHT (ConvTransposeFn / LinearTFn / RGBReduceFn take w as an
# input) is untouched, so the second backward still reaches the parameters.  The autograd engine runs these backward()s on its own
# device thread, hence a plain module global (autograd.grad blocks until they are done).
_inputs_only = False
_inputs_only_task = None     # id of the autograd graph task that runs under the flag (torch._C._current_graph_task_id)


class inputs_only:
    def __enter__(self):
        global _inputs_only, _inputs_only_task
        self.prev, _inputs_only = (_inputs_only, _inputs_only_task), True
        _inputs_only_task = None

    def __exit__(self, *exc):
        global _inputs_only, _inputs_only_task
        _inputs_only, _inputs_only_task = self.prev


def _inputs_only_here() -> bool:
    """The flag, as seen from inside a backward function.  It is a process global read on the autograd engine's thread: correct while ONE
    backward pass runs at a time (autograd.grad blocks; a WORKER is driven by one thread).  Two passes that overlap -- a second
    thread calling backward() while loss.cal_derivative is inside autograd.grad -- would silently lose parameter gradients, so the
    first backward function that sees the flag pins the graph task it belongs to and any other graph task that sees it fails loudly."""
    global _inputs_only_task
    if not _inputs_only:
        return False
    tid = getattr(torch._C, "_current_graph_task_id", lambda: -1)()
    if tid >= 0:
        if _inputs_only_task is None:
            _inputs_only_task = tid
        elif tid != _inputs_only_task:
            raise RuntimeError("two backward passes overlap while ops.inputs_only is active (loss.cal_derivative): the parameter "
                               "gradients of one of them would be dropped; run them one at a time")
    return True


class _NoGraphCtx:
    """stand-in for a Function's ctx when no graph is recorded: forward()'s bookkeeping calls become no-ops"""
    __slots__ = ("cfg", "G", "clog", "per_sample", "scale", "t", "tau")

    def save_for_backward(self, *tensors):
        pass

    def mark_non_differentiable(self, *tensors):
        pass

    def set_materialize_grads(self, value):
        pass


def _ap(fn, *args):
    """fn.apply(*args) from inside a backward pass.  The nested Functions exist so that a backward pass is itself differentiable
    (R1: create_graph=True).  In an ordinary backward pass grad mode is off, no node would be recorded, and Function.apply is
    pure dispatch overhead (~5 us x ~600 nested calls per iteration; a batch-4 step is launch-bound): call forward() directly."""
    if torch.is_grad_enabled():
        return fn.apply(*args)
    return fn.forward(_NoGraphCtx(), *args)


def _wants(ctx, i: int) -> bool:
    """does this backward owe a gradient for PARAMETER input i?"""
    return ctx.needs_input_grad[i] and not _inputs_only_here()


# ---- prepared-weight cache -------------------------------------------------------------------------------------------
# A conv weight is used several times per iteration (D runs 2-3 forwards and as many backwards between two optimiser
# steps); its bf16 GEMM-layout copy only changes when the parameter does.  Valid for (parameter object, torch version counter,
# weight epoch); the epoch is bumped by everything that rewrites parameters through raw pointers (Adam / EMA kernels).
# Only nn.Parameters are cached (identified by object identity, guarded by a weakref): temporaries such as the cotangents of
# the double-backward, or parameters of a later-built module, may reuse a storage address.
_weight_epoch = 0
_prep_cache: dict = {}      # id(param) -> [weakref(param), version, epoch, {(scale, transpose, need_lo): (PreparedWeight, wsq)}]


def bump_weight_epoch() -> None:
    """Invalidate every cached prepared weight."""
    global _weight_epoch
    _weight_epoch += 1


_recipes: dict = {}         # id(param) -> [weakref(param), {(scale, transpose, need_lo): want_wsq}]   (survives invalidation)
_groups: dict = {}          # id(param) -> _Group of the invalidation that last hit it


class _Group:
    """The parameters rewritten by one optimiser / EMA step: their prepared copies are rebuilt together, in one launch, the
    first time any of them is needed again (96 prep + 25 wsq launches per iteration otherwise)."""
    __slots__ = ("refs", "done")

    def __init__(self, params):
        import weakref
        self.refs = [weakref.ref(p) for p in params]
        self.done = False


def invalidate_weights(params) -> None:
    """Invalidate the prepared copies of exactly these parameters (called by the Adam / EMA kernels' wrappers, which rewrite
    parameters through raw pointers without touching torch's version counters)."""
    params = list(params)
    grp = _Group(params)
    for p in params:
        _prep_cache.pop(id(p), None)
        _groups[id(p)] = grp


def _prep_group(grp) -> None:
    import weakref
    jobs, owners = [], []
    for r in grp.refs:
        p = r()
        rec = _recipes.get(id(p)) if p is not None else None
        if rec is None or rec[0]() is not p:
            continue
        for key, want_wsq in rec[1].items():
            if key[3]:
                continue                                             # MX-fp8 copies are rebuilt one by one (their own kernel)
            scale, transpose, need_lo, _ = key
            jobs.append((p, scale, transpose, need_lo, want_wsq))
            owners.append(p)
    if len(jobs) < 2:
        return
    for (p, scale, transpose, need_lo, _), hit in zip(jobs, _K().prep_weight_group(jobs)):
        ent = _prep_cache.get(id(p))
        if ent is None or ent[0]() is not p or ent[1] != p._version or ent[2] != _weight_epoch:
            ent = [weakref.ref(p), p._version, _weight_epoch, {}]
            _prep_cache[id(p)] = ent
        ent[3][(scale, transpose, need_lo, False)] = hit


def _prep(w: Tensor, scale: float, transpose: bool, need_lo: bool, want_wsq: bool = False, fp8: bool = False):
    """-> (prepared weight, wsq | None); fp8: the MX-fp8 copy (kernels.PreparedWeightFp8) instead of the bf16 one"""
    import weakref
    K = _K()
    build = (lambda ww: (K.prep_weight_fp8(w, scale, transpose), None)) if fp8 else (lambda ww: K.prep_weight(w, scale, transpose, need_lo, ww))
    if not isinstance(w, torch.nn.Parameter):
        return build(want_wsq)
    key = (float(scale), bool(transpose), bool(need_lo), bool(fp8))
    ent = _prep_cache.get(id(w))
    valid = ent is not None and ent[0]() is w and ent[1] == w._version and ent[2] == _weight_epoch
    hit = ent[3].get(key) if valid else None
    if hit is not None and not (want_wsq and hit[1] is None):
        return hit
    # miss: remember the recipe, then rebuild -- the whole group this parameter was invalidated with, if that is still due
    rec = _recipes.get(id(w))
    if rec is None or rec[0]() is not w:
        if len(_recipes) > 4096:
            _recipes.clear()
        rec = [weakref.ref(w), {}]
        _recipes[id(w)] = rec
    known = key in rec[1] and (rec[1][key] or not want_wsq)
    rec[1][key] = rec[1].get(key, False) or want_wsq
    grp = _groups.get(id(w))
    if known and grp is not None and not grp.done:
        grp.done = True
        _prep_group(grp)
        ent = _prep_cache.get(id(w))
        valid = ent is not None and ent[0]() is w and ent[1] == w._version and ent[2] == _weight_epoch
        hit = ent[3].get(key) if valid else None
        if hit is not None and not (want_wsq and hit[1] is None):
            return hit
    if not valid:
        if len(_prep_cache) > 4096:                       # parameters of dead modules: drop everything, it refills in one step
            _prep_cache.clear()
        ent = [weakref.ref(w), w._version, _weight_epoch, {}]
        _prep_cache[id(w)] = ent
    hit = build(want_wsq or (hit is not None and hit[1] is not None))
    ent[3][key] = hit
    return hit


def _gz_demod() -> bool:
    from . import config
    return config.gz_demod()


def _act_masks() -> bool:
    from . import config
    return config.act_masks()


def _use_fp8(x: Tensor, k: int, stride: int) -> bool:
    """MX-fp8 operands for this convolution launch?  (config.conv_operands() == "fp8": BASELINE configs[4]; the stride-1 3x3 / 1x1 forward
    and data-gradient launches on grids of at least 16 x 16 positions -- 60 % of the step's FLOPs; everything else stays bf16)"""
    from . import config
    if config.conv_operands() != "fp8" or x.dtype != torch.bfloat16 or stride != 1 or x.shape[1] < 16 or x.shape[2] < 16 or x.shape[3] < 64:
        return False
    # the R1 penalty (loss.py:18-34) squares data gradients: its create_graph pass stays bf16 (e4m3 carries ~4 % noise per layer)
    in_backward = getattr(torch._C, "_current_graph_task_id", lambda: -1)() >= 0
    return not (_inputs_only or (in_backward and torch.is_grad_enabled()))


# =====================================================================================================
# double-differentiable convolution triple (EqualizedConv2d, custom_layers.py:28-44)
# =====================================================================================================
class Conv2dFn(Function):
    """y = act(conv_{k,stride}(x, w*wscale) + bias*bias_scale) * gain (+ residual).  w: [A,Bc,k,k] f32 parameter."""

    @staticmethod
    def forward(ctx, x, w, bias, residual, k, stride, act, gain, wscale, bias_scale):
        assert not (act != ACT_NONE and residual is not None), "residual is only fused on linear (act-free) convs"
        assert act != ACT_NONE or gain == 1.0, "fold the gain of an act-free conv into wscale"
        K = _K()
        A = w.shape[0]
        mask = None
        if _use_fp8(x, k, stride):
            pw, _ = _prep(w, wscale, False, False, fp8=True)
            y = K.conv_fwd_fp8(x, pw, A, k, stride, bias=bias, bias_scale=bias_scale, act=act, gain=gain, residual=residual)
        else:
            pw, _ = _prep(w, wscale, False, _need_lo(x))
            if act == ACT_LRELU and residual is None and _act_masks():
                # the sign mask of the pre-activation leaves the epilogue as a by-product: the activation backward reads it (1/16 of y's bytes)
                y, mask = K.conv_fwd(x, pw, A, k, stride, bias=bias, bias_scale=bias_scale, act=act, gain=gain, want_mask=True)
            else:
                y = K.conv_fwd(x, pw, A, k, stride, bias=bias, bias_scale=bias_scale, act=act, gain=gain, residual=residual)
        ctx.save_for_backward(x, w, y if act != ACT_NONE else None, mask)
        ctx.cfg = (k, stride, act, gain, wscale, bias_scale, bias is not None, residual is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        if gy is None:                                     # (Conv2dPoolFn runs without materialised gradients)
            return (None,) * 10
        x, w, y, mask = ctx.saved_tensors
        k, stride, act, gain, wscale, bias_scale, has_bias, has_res = ctx.cfg
        gy = gy.contiguous()
        A = w.shape[0]
        want_gb = has_bias and _wants(ctx, 2)
        if act != ACT_NONE or want_gb:
            gz, gb = _ap(ActBwdFn, gy, y, act, gain, A, want_gb, bias_scale, mask)
        else:
            gz, gb = gy, None
        gx = _ap(ConvTransposeFn, gz, w, k, stride, wscale, x.shape[-1], None) if ctx.needs_input_grad[0] else None
        gw = _ap(ConvWeightGradFn, x, gz, k, stride, wscale, w.shape[0], w.shape[1]) if _wants(ctx, 1) else None
        gres = gy if (has_res and ctx.needs_input_grad[3]) else None
        return gx, gw, (gb if want_gb else None), gres, None, None, None, None, None, None


class Conv2dPoolFn(Function):
    """Conv2dFn that also returns avg_pool2d(y, 2) as a NON-DIFFERENTIABLE by-product: the closing 1x1 + residual convolution of a
    DiscriminatorBlock (custom_layers.py:203,209) leaves the pooled copy of its output that the next block's skip branch reads
    (custom_layers.py:202) while the output tile is still on chip.  The next block's ops.ConvPoolFn takes it as a hint in place of its own
    pooling pass; the autograd graph is the one without the hint (ConvPoolFn still owns d pooled / d x)."""

    @staticmethod
    def forward(ctx, x, w, bias, residual, k, stride, act, gain, wscale, bias_scale):
        assert not (act != ACT_NONE and residual is not None) and (act != ACT_NONE or gain == 1.0)
        K = _K()
        pw, _ = _prep(w, wscale, False, _need_lo(x))
        y, pooled = K.conv_fwd(x, pw, w.shape[0], k, stride, bias=bias, bias_scale=bias_scale, act=act, gain=gain, residual=residual, pool=True)
        ctx.save_for_backward(x, w, y if act != ACT_NONE else None, None)
        ctx.cfg = (k, stride, act, gain, wscale, bias_scale, bias is not None, residual is not None)
        ctx.mark_non_differentiable(pooled)
        ctx.set_materialize_grads(False)        # (autograd would otherwise FILL a zero gradient of pooled's size for every backward pass)
        return y, pooled

    @staticmethod
    def backward(ctx, gy, _gpooled):
        return Conv2dFn.backward(ctx, gy)


class ConvTransposeFn(Function):
    """gx = adjoint of conv_{k,stride}(., w*wscale) applied to g  (stride 2: the 4-phase transposed convolution)
    (+ 0.25 * nearest-x2(res_half): the adjoint of avg_pool2d, fused into the epilogue)."""

    @staticmethod
    def forward(ctx, g, w, k, stride, wscale, cin_alloc, res_half):
        K = _K()
        if _use_fp8(g, k, stride):
            pw, _ = _prep(w, wscale, True, False, fp8=True)
            gx = K.conv_bwd_data_fp8(g, pw, w.shape[1], k, stride, residual=res_half, residual_half=res_half is not None)
        else:
            pw, _ = _prep(w, wscale, True, _need_lo(g))
            gx = K.conv_bwd_data(g, pw, w.shape[1], k, stride, residual=res_half, residual_half=res_half is not None)
        assert gx.shape[-1] == cin_alloc
        ctx.save_for_backward(g, w)
        ctx.cfg = (k, stride, wscale)
        return gx

    @staticmethod
    def backward(ctx, ggx):
        g, w = ctx.saved_tensors
        k, stride, wscale = ctx.cfg
        ggx = ggx.contiguous()
        gg = _ap(Conv2dFn, ggx, w, None, None, k, stride, ACT_NONE, 1.0, wscale, 0.0) if ctx.needs_input_grad[0] else None
        gw = _ap(ConvWeightGradFn, ggx, g, k, stride, wscale, w.shape[0], w.shape[1]) if ctx.needs_input_grad[1] else None
        gres = _ap(AvgPool2Fn, ggx) if ctx.needs_input_grad[6] else None
        return gg, gw, None, None, None, None, gres


class ConvPoolFn(Function):
    """(h, pooled) = (box3?(act(conv_k(x, w*wscale) + bias*bias_scale) * gain), avg_pool2d(x, 2)): the two consumers of a
    DiscriminatorBlock's input (custom_layers.py:202,204-206).  Owning both lets the backward pass fold the pooled branch's
    gradient into the epilogue of the conv's data-gradient kernel instead of up-sampling it and adding two full-resolution
    tensors (3 ms per iteration at 256x256, batch 32); with box=True the blur after the activation joins the node, so its
    backward and the activation backward are one pass (BoxActBwdFn)."""

    @staticmethod
    def forward(ctx, x, w, bias, k, act, gain, wscale, bias_scale, box, pooled_hint=None):
        """pooled_hint: avg_pool2d(x, 2) where the producer of x already left it (Conv2dPoolFn, RGBExpandFn(pool=True)): the pooling pass is skipped"""
        K = _K()
        mask = None
        if _use_fp8(x, k, 1):
            pw, _ = _prep(w, wscale, False, False, fp8=True)
            y = K.conv_fwd_fp8(x, pw, w.shape[0], k, 1, bias=bias, bias_scale=bias_scale, act=act, gain=gain)
        else:
            pw, _ = _prep(w, wscale, False, _need_lo(x))
            if act == ACT_LRELU and _act_masks():
                y, mask = K.conv_fwd(x, pw, w.shape[0], k, 1, bias=bias, bias_scale=bias_scale, act=act, gain=gain, want_mask=True)
            else:
                y = K.conv_fwd(x, pw, w.shape[0], k, 1, bias=bias, bias_scale=bias_scale, act=act, gain=gain)
        ctx.save_for_backward(x, w, y if act != ACT_NONE else None, mask)
        ctx.cfg = (k, act, gain, wscale, bias_scale, bias is not None, box)
        pooled = K.avgpool2(x) if pooled_hint is None else pooled_hint.detach().view_as(pooled_hint)
        return (K.box3_act(y, ACT_NONE, 1.0) if box else y), pooled

    @staticmethod
    def backward(ctx, gy, gpooled):
        x, w, y, mask = ctx.saved_tensors
        k, act, gain, wscale, bias_scale, has_bias, box = ctx.cfg
        A = w.shape[0]
        want_gb = has_bias and _wants(ctx, 2)
        if gy is None:                                                   # only the pooled branch was used
            gx = _ap(AvgPool2TFn, gpooled.contiguous()) if ctx.needs_input_grad[0] else None
            return gx, None, None, None, None, None, None, None, None, None
        gy = gy.contiguous()
        if box and act != ACT_NONE:
            gz, gb = _ap(BoxActBwdFn, gy, y, act, gain, A, want_gb, bias_scale, mask)
        else:
            if box:
                gy = _ap(Box3Fn, gy)
            if act != ACT_NONE or want_gb:
                gz, gb = _ap(ActBwdFn, gy, y, act, gain, A, want_gb, bias_scale, mask)
            else:
                gz, gb = gy, None
        gx = None
        if ctx.needs_input_grad[0]:
            gx = _ap(ConvTransposeFn, gz, w, k, 1, wscale, x.shape[-1], None if gpooled is None else gpooled.contiguous())
        gw = _ap(ConvWeightGradFn, x, gz, k, 1, wscale, w.shape[0], w.shape[1]) if _wants(ctx, 1) else None
        return gx, gw, (gb if want_gb else None), None, None, None, None, None, None, None


class BoxActBwdFn(Function):
    """(gz, gbias) = (box3(gy) * act'(y), bias_scale * sum gz): backward of  act(.) -> box3  in one pass.  Linear in gy."""

    @staticmethod
    def forward(ctx, gy, y, act, gain, clog, want_gbias, bias_scale, mask=None):
        gz, gb = _K().box3_actbwd(gy, y, act, gain, clog, want_gbias, mask=mask)
        if gb is None:
            gb = gy.new_empty((0,), dtype=torch.float32)
        else:
            gb = gb * bias_scale if bias_scale != 1.0 else gb
        ctx.save_for_backward(y)
        ctx.cfg = (act, gain)
        ctx.mark_non_differentiable(gb)
        ctx.set_materialize_grads(False)
        return gz, gb

    @staticmethod
    @once_differentiable
    def backward(ctx, ggz, _ggb):
        (y,) = ctx.saved_tensors
        act, gain = ctx.cfg
        if ggz is None:
            return None, None, None, None, None, None, None, None
        return _K().box3_act_bwd(ggz.contiguous(), y, act, gain), None, None, None, None, None, None, None


class ConvWeightGradFn(Function):
    """gw[A,Bc,k,k] = wscale * sum_positions g (x) x_shifted  (x: conv input side, g: conv output side)."""

    @staticmethod
    def forward(ctx, x, g, k, stride, wscale, A, Bc):
        K = _K()
        gw = K.conv_wgrad_unprep(x, g, A, Bc, k, stride, wscale)
        ctx.save_for_backward(x, g)
        ctx.cfg = (k, stride, wscale)
        return gw

    @staticmethod
    def backward(ctx, ggw):
        x, g = ctx.saved_tensors
        k, stride, wscale = ctx.cfg
        ggw = ggw.contiguous()
        gx = _ap(ConvTransposeFn, g, ggw, k, stride, wscale, x.shape[-1], None) if ctx.needs_input_grad[0] else None
        gg = _ap(Conv2dFn, x, ggw, None, None, k, stride, ACT_NONE, 1.0, wscale, 0.0) if ctx.needs_input_grad[1] else None
        return gx, gg, None, None, None, None, None


class ActBwdFn(Function):
    """(gz, gbias) = (gy * act'(y), bias_scale * sum_{b,h,w} gz).  Linear in gy; y is the saved activation OUTPUT."""

    @staticmethod
    def forward(ctx, gy, y, act, gain, clog, want_gbias, bias_scale, mask=None):
        """mask: the activation's sign bits where the producing convolution left them (Conv2dFn / ConvPoolFn): read instead of y"""
        K = _K()
        if act == ACT_NONE:
            _, gb, _ = K.act_bwd_reduce(gy, None, ACT_NONE, 1.0, clog, want_gz=False, want_gbias=want_gbias)
            gz = gy
        else:
            gz, gb, _ = K.act_bwd_reduce(gy, y, act, gain, clog, want_gz=True, want_gbias=want_gbias, mask=mask)
        if gb is None:
            gb = gy.new_empty((0,), dtype=torch.float32)
        else:
            gb = gb * bias_scale if bias_scale != 1.0 else gb
        ctx.save_for_backward(y, mask)
        ctx.cfg = (act, gain, clog)
        ctx.mark_non_differentiable(gb)
        ctx.set_materialize_grads(False)
        return gz, gb

    @staticmethod
    def backward(ctx, ggz, _ggb):
        y, mask = ctx.saved_tensors
        act, gain, clog = ctx.cfg
        if act == ACT_NONE or ggz is None:
            return ggz, None, None, None, None, None, None, None
        g, _ = _ap(ActBwdFn, ggz.contiguous(), y, act, gain, clog, False, 1.0, mask)
        return g, None, None, None, None, None, None, None


class Box3Fn(Function):
    """3x3 box filter (zero padding, /9) -- linear and self-adjoint (custom_layers.py:196-198)."""

    @staticmethod
    def forward(ctx, x):
        return _K().box3_act(x, ACT_NONE, 1.0)

    @staticmethod
    def backward(ctx, gy):
        return _ap(Box3Fn, gy.contiguous())


class AvgPool2Fn(Function):
    """F.avg_pool2d(2,2), custom_layers.py:202."""

    @staticmethod
    def forward(ctx, x):
        return _K().avgpool2(x)

    @staticmethod
    def backward(ctx, gy):
        return _ap(AvgPool2TFn, gy.contiguous())


class AvgPool2TFn(Function):
    @staticmethod
    def forward(ctx, gy):
        return _K().avgpool2_bwd(gy)

    @staticmethod
    def backward(ctx, ggx):
        return _ap(AvgPool2Fn, ggx.contiguous())


# ---- 1x1 convs that touch the f32 NCHW image; wt is [Bw,3,C] f32 (built from the parameter with torch glue) ----
class RGBExpandFn(Function):
    """feat[b,p,c] = act(sum_o img[b,o,p] wt[bw,o,c] + bias[c]*bias_scale) * gain     (fromRGB: cnn.py:20-21)"""

    @staticmethod
    def forward(ctx, img, wt, bias, bias_scale, clog, act, gain, dtype, pool=False):
        """pool: also returns avg_pool2d(feat, 2) as a non-differentiable by-product (see Conv2dPoolFn)"""
        out = _K().rgb_expand(img, wt, bias, bias_scale, clog, act, gain, dtype, pool=pool)
        y = out[0] if pool else out
        ctx.save_for_backward(img, wt, y if act != ACT_NONE else None, bias)
        ctx.cfg = (bias_scale, clog, act, gain, bias is not None)
        if pool:
            ctx.mark_non_differentiable(out[1])
            ctx.set_materialize_grads(False)    # (no zero-filled gradient for the by-product)
            return y, out[1]
        return y

    @staticmethod
    def backward(ctx, gy, _gpooled=None):
        if gy is None:
            return (None,) * 9
        img, wt, y, fbias = ctx.saved_tensors
        bias_scale, clog, act, gain, has_bias = ctx.cfg
        gy = gy.contiguous()
        want_gb = has_bias and _wants(ctx, 2)
        if not torch.is_grad_enabled():
            # ordinary (first-order) backward: ONE pass forms gz = gy * act'(y) in registers and leaves the image gradient, the weight
            # gradient and the bias gradient (1.1 GB instead of 2.7 GB of traffic at 256 x 256, batch 32).  Under create_graph (R1,
            # loss.py:28-33) the composition below records the graph the second backward walks.
            want_gw = _wants(ctx, 1)
            # (leaky ReLU: the sign act' needs is recomputed from the image -- a 3-term dot product on operands the kernel holds -- instead of
            #  reading y back: half of the pass's bytes)
            recomp = act == ACT_LRELU and _act_masks()
            gimg, gwt, gb = _K().rgb_expand_bwd(gy, y, img if (want_gw or recomp) else None, wt, act, gain, clog, ctx.needs_input_grad[0], want_gw, want_gb,
                                                fbias=fbias.detach() if fbias is not None else None, fbias_scale=bias_scale, recompute=recomp)
            if gb is not None and bias_scale != 1.0:
                gb = gb * bias_scale
            return gimg, gwt, gb, None, None, None, None, None, None
        if act != ACT_NONE or want_gb:
            gz, gb = _ap(ActBwdFn, gy, y, act, gain, clog, want_gb, bias_scale)
        else:
            gz, gb = gy, None
        gimg = _ap(RGBReduceFn, gz, wt, None, 0.0) if ctx.needs_input_grad[0] else None
        gwt = _ap(RGBWeightGradFn, img, gz, wt.shape[0] > 1) if _wants(ctx, 1) else None
        return gimg, gwt, (gb if want_gb else None), None, None, None, None, None, None


class RGBReduceFn(Function):
    """img[b,o,p] = sum_c feat[b,p,c] wt[bw,o,c] + bias[o]*bias_scale                 (toRGB: custom_layers.py:181)"""

    @staticmethod
    def forward(ctx, feat, wt, bias, bias_scale):
        ctx.save_for_backward(feat, wt)
        ctx.cfg = (bias_scale, bias is not None)
        return _K().rgb_reduce(feat, wt, bias, bias_scale)

    @staticmethod
    def backward(ctx, gimg):
        feat, wt = ctx.saved_tensors
        bias_scale, has_bias = ctx.cfg
        gimg = gimg.contiguous()
        gfeat = (_ap(RGBExpandFn, gimg, wt, None, 0.0, wt.shape[-1], ACT_NONE, 1.0, feat.dtype)
                 if ctx.needs_input_grad[0] else None)
        gwt = _ap(RGBWeightGradFn, gimg, feat, wt.shape[0] > 1) if ctx.needs_input_grad[1] else None
        gb = gimg.sum(dim=(0, 2, 3)) * bias_scale if (has_bias and ctx.needs_input_grad[2]) else None   # 3 numbers
        return gfeat, gwt, gb, None


class RGBWeightGradFn(Function):
    """gwt[bw,o,c] = sum_p img[b,o,p] feat[b,p,c]"""

    @staticmethod
    def forward(ctx, img, feat, per_sample):
        ctx.save_for_backward(img, feat)
        ctx.per_sample = per_sample
        return _K().rgb_wgrad(img, feat, per_sample)

    @staticmethod
    def backward(ctx, ggw):
        img, feat = ctx.saved_tensors
        ggw = ggw.contiguous()
        gimg = _ap(RGBReduceFn, feat, ggw, None, 0.0) if ctx.needs_input_grad[0] else None
        gfeat = (_ap(RGBExpandFn, img, ggw, None, 0.0, ggw.shape[-1], ACT_NONE, 1.0, feat.dtype)
                 if ctx.needs_input_grad[1] else None)
        return gimg, gfeat, None


# ---- minibatch stddev (custom_layers.py:237-256) ---------------------------------------------------------
class MbstdFn(Function):
    @staticmethod
    def forward(ctx, x, G):
        ctx.save_for_backward(x)
        ctx.G = G
        return _K().mbstd_fwd(x, G, ceil8(x.shape[-1] + 1))

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        return _ap(MbstdBwdFn, gy.contiguous(), x, ctx.G), None


class MbstdBwdFn(Function):
    @staticmethod
    def forward(ctx, gy, x, G):
        ctx.save_for_backward(gy, x)
        ctx.G = G
        return _K().mbstd_bwd(gy, x, G)

    @staticmethod
    @once_differentiable
    def backward(ctx, v):
        gy, x = ctx.saved_tensors
        ggy, gx2 = _K().mbstd_bwd2(v.contiguous(), gy, x, ctx.G)
        return ggy, gx2, None


# ---- layout converts -----------------------------------------------------------------------------------------
class ToNCHWFn(Function):
    """feat [B,H,W,Calloc] -> f32 [B,Clog,H,W] (the order `flatten(1)` sees in the reference, custom_layers.py:232)."""

    @staticmethod
    def forward(ctx, feat, clog):
        ctx.cfg = (feat.shape[-1], feat.dtype)
        return _K().nhwc_to_nchw(feat, clog, False)

    @staticmethod
    def backward(ctx, g):
        calloc, dtype = ctx.cfg
        return _ap(ToNHWCFn, g.contiguous(), calloc, dtype), None


class ToNHWCFn(Function):
    @staticmethod
    def forward(ctx, src, calloc, dtype):
        ctx.clog = src.shape[1]
        return _K().nchw_to_nhwc(src, src.shape[0], calloc, dtype)

    @staticmethod
    def backward(ctx, g):
        return _ap(ToNCHWFn, g.contiguous(), ctx.clog), None, None


class ConstInputFn(Function):
    """const [C,4,4] f32 -> [B,4,4,C] feature map (cnn.py:106); backward sums over the batch."""

    @staticmethod
    def forward(ctx, const, B, dtype):
        return _K().nchw_to_nhwc(const.unsqueeze(0).contiguous(), B, ceil8(const.shape[0]), dtype)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        return _K().nhwc_to_nchw(g.contiguous(), g.shape[-1], True)[0], None, None


# ---- small f32 linear triple (EqualizedLinear, custom_layers.py:17-25) ---------------------------------------------
class LinearFn(Function):
    """y = act(scale * x @ w^T + bias*bias_scale) * gain ; x [M,I], w [O,I]"""

    @staticmethod
    def forward(ctx, x, w, bias, scale, bias_scale, act, gain):
        y = _K().linear_fwd(x, w, bias, scale, bias_scale, act, gain)
        ctx.save_for_backward(x, w, y if act != ACT_NONE else None)
        ctx.cfg = (scale, bias_scale, act, gain, bias is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        scale, bias_scale, act, gain, has_bias = ctx.cfg
        gy = gy.contiguous()
        gz = _ap(ActBwdF32Fn, gy, y, act, gain) if act != ACT_NONE else (gy * gain if gain != 1.0 else gy)
        gx = _ap(LinearTFn, gz, w, scale) if ctx.needs_input_grad[0] else None
        want_w, want_b = _wants(ctx, 1), has_bias and _wants(ctx, 2)
        if want_w and want_b and not torch.is_grad_enabled():    # (no graph through this backward: both gradients from one launch)
            gw, gb = _K().linear_wgrad_bias(gz.contiguous(), x, scale, bias_scale)
        else:
            gw = _ap(LinearWeightGradFn, gz, x, scale) if want_w else None
            gb = _K().colsum(gz.detach().contiguous(), bias_scale) if want_b else None
        return gx, gw, gb, None, None, None, None


class GroupedLinearFn(Function):
    """y_l = scale_l * x @ w_l^T + bias_l * bias_scale_l for L layers sharing x (the generator's style affines: every block gets
    the same latent, cnn.py:103-104).  One launch forward, three backward (gx, all gw_l, all gb_l); first order only -- the
    R1 double backward never reaches the generator."""

    @staticmethod
    def forward(ctx, x, scales, bias_scales, *wb):
        ws, bs = wb[0::2], wb[1::2]
        ctx.save_for_backward(x, *ws)
        ctx.cfg = (tuple(scales), tuple(bias_scales))
        return tuple(_K().linear_group_fwd(x, ws, bs, scales, bias_scales, ACT_NONE, 1.0))

    @staticmethod
    @once_differentiable
    def backward(ctx, *gys):
        x, *ws = ctx.saved_tensors
        scales, bias_scales = ctx.cfg
        gys = [torch.zeros((x.shape[0], w.shape[0]), dtype=torch.float32, device=x.device) if g is None else g.contiguous()
               for g, w in zip(gys, ws)]
        gx, gws, gbs = _K().linear_group_bwd(gys, x, ws, scales, bias_scales, want_gx=ctx.needs_input_grad[0])
        out = [gx, None, None]
        for gw, gb in zip(gws, gbs):
            out += [gw, gb]
        return tuple(out)


class MultiLinearFn(Function):
    """y_l = scale_l * x_l @ w_l^T + bias_l * bias_scale_l for L layers with their OWN inputs and shapes (the two mapping networks at equal
    depth): one launch forward, two backward; first order only (the R1 double backward never reaches the generator)."""

    @staticmethod
    def forward(ctx, scales, bias_scales, *xwb):
        L = len(scales)
        xs, ws, bs = xwb[:L], xwb[L:2 * L], xwb[2 * L:]
        ctx.save_for_backward(*xs, *ws)
        ctx.cfg = (tuple(scales), tuple(bias_scales))
        return tuple(_K().linear_multi_fwd([x.contiguous() for x in xs], ws, bs, scales, bias_scales, ACT_NONE, 1.0))

    @staticmethod
    @once_differentiable
    def backward(ctx, *gys):
        scales, bias_scales = ctx.cfg
        L = len(scales)
        saved = ctx.saved_tensors
        xs, ws = saved[:L], saved[L:]
        gys = [torch.zeros((x.shape[0], w.shape[0]), dtype=torch.float32, device=x.device) if g is None else g.contiguous()
               for g, x, w in zip(gys, xs, ws)]
        gxs, gws, gbs = _K().linear_multi_bwd(gys, [x.contiguous() for x in xs], ws, scales, bias_scales, want_gx=any(ctx.needs_input_grad[2:2 + L]))
        return (None, None, *(gxs if gxs is not None else [None] * L), *gws, *gbs)


def multi_linear(xs, linears):
    """[m(x) for m, x in zip(linears, xs)] for EqualizedLinear modules, through MultiLinearFn"""
    return list(MultiLinearFn.apply([m.weight.c for m in linears], [m.lr_mul for m in linears], *[x.contiguous() for x in xs],
                                    *[m.weight.weight for m in linears], *[m.bias for m in linears]))


def grouped_linear(x, linears):
    """[EqualizedLinear(x) for each module] through GroupedLinearFn (in slices of 24 layers)."""
    x = x.contiguous()
    ys = []
    for i in range(0, len(linears), 24):
        part = linears[i:i + 24]
        wb = []
        for m in part:
            wb += [m.weight.weight, m.bias]
        ys += list(GroupedLinearFn.apply(x, [m.weight.c for m in part], [m.lr_mul for m in part], *wb))
    return ys


class LinearTFn(Function):
    """gx = scale * g @ w"""

    @staticmethod
    def forward(ctx, g, w, scale):
        ctx.save_for_backward(g, w)
        ctx.scale = scale
        return _K().linear_bwd_data(g, w, scale)

    @staticmethod
    def backward(ctx, ggx):
        g, w = ctx.saved_tensors
        ggx = ggx.contiguous()
        gg = _ap(LinearFn, ggx, w, None, ctx.scale, 0.0, ACT_NONE, 1.0) if ctx.needs_input_grad[0] else None
        gw = _ap(LinearWeightGradFn, g, ggx, ctx.scale) if ctx.needs_input_grad[1] else None
        return gg, gw, None


class LinearWeightGradFn(Function):
    """gw[O,I] = scale * g^T @ x"""

    @staticmethod
    def forward(ctx, g, x, scale):
        ctx.save_for_backward(g, x)
        ctx.scale = scale
        return _K().linear_wgrad(g, x, scale)

    @staticmethod
    def backward(ctx, ggw):
        g, x = ctx.saved_tensors
        ggw = ggw.contiguous()
        gg = _ap(LinearFn, x, ggw, None, ctx.scale, 0.0, ACT_NONE, 1.0) if ctx.needs_input_grad[0] else None
        gx = _ap(LinearTFn, g, ggw, ctx.scale) if ctx.needs_input_grad[1] else None
        return gg, gx, None


class ActBwdF32Fn(Function):
    @staticmethod
    def forward(ctx, gy, y, act, gain):
        ctx.save_for_backward(y)
        ctx.cfg = (act, gain)
        return _K().act_bwd_f32(gy, y, act, gain)

    @staticmethod
    def backward(ctx, gg):
        (y,) = ctx.saved_tensors
        return _ap(ActBwdF32Fn, gg.contiguous(), y, *ctx.cfg), None, None, None


# =====================================================================================================
# generator path (first-order backward, fused kernels)
# =====================================================================================================
# ---- demodulation vectors of a whole generator pass in one launch ------------------------------------------------------------------
# Every modulated layer's style is known before the first convolution (the grouped affines), so Generator.forward has the d[b,o] of
# all its layers computed by ONE kernel (lcgan_demod_group) and the Functions below pick theirs up here instead of launching
# lcgan_demod_fwd each (19 launches per generator pass).  Keyed by (layer weight, style tensor's storage); filled and cleared by Generator.forward.
_demod_cache = {}


def precompute_demod(entries, need_lo: bool):
    """entries: [(w [O,Cin,k,k] parameter, s [B,Cin] style, flow: bool)] -> fills the cache the modulated Functions read"""
    K = _K()
    ss, wsqs, osts = [], [], []
    for w, s_, flow in entries:
        O, Cin, k, _ = w.shape
        c_eq = 1.0 / math.sqrt(Cin * k * k)
        wsq = _flow_prep(w, c_eq, need_lo)[2] if flow else _prep(w, c_eq, False, need_lo, want_wsq=True)[1]
        ss.append(s_.contiguous()); wsqs.append(wsq); osts.append(8 if flow else ceil8(O))
    ws = [e[0] for e in entries]
    for i in range(0, len(ss), 24):
        for w_, s_, d in zip(ws[i:i + 24], ss[i:i + 24], K.demod_group(ss[i:i + 24], wsqs[i:i + 24], osts[i:i + 24])):
            _demod_cache[(id(w_), s_.data_ptr(), s_.shape[0], s_.shape[1], d.shape[1])] = d


def clear_demod():
    _demod_cache.clear()


def _demod(K, w, s, wsq, ostride):
    """d[b,o] of layer `w` under style `s`: the vector Generator.forward precomputed for exactly this (weight, style) pair, else one launch"""
    hit = _demod_cache.get((id(w), s.data_ptr(), s.shape[0], s.shape[1], ostride))
    if hit is not None:
        assert hit.shape == (s.shape[0], ostride), (hit.shape, s.shape, ostride)
        return hit
    return K.demod_fwd(s, wsq, ostride)


class ModConvFn(Function):
    """Modulated + demodulated convolution (ModulatedConv2d, custom_layers.py:47-86) without per-sample weights:
         y = act(d[b,o] * conv(s[b,c] * x, W*c_eq) + bias) * gain ,   d = rsqrt(sum_c s^2 sum_k (W c_eq)^2 + eps)
       up == 2 runs the 4-phase transposed convolution (F.conv_transpose2d stride 2, pad 1, output_padding 1)."""

    @staticmethod
    def forward(ctx, x, w, bias, s, up, act, gain):
        K = _K()
        O, Cin, k, _ = w.shape
        c_eq = 1.0 / math.sqrt(Cin * k * k)
        s = s.contiguous()
        pw, wsq = _prep(w, c_eq, False, _need_lo(x), want_wsq=True)             # [t][O][Cin] serves conv AND up-conv forward
        d = _demod(K, w, s, wsq, ceil8(O))
        if up == 2:
            y = K.conv_bwd_data(x, pw, O, k, 2, pre=s, post=d, bias=bias, bias_scale=1.0, act=act, gain=gain)
        elif _use_fp8(x, k, 1):
            pw8, _ = _prep(w, c_eq, False, False, fp8=True)
            y = K.conv_fwd_fp8(x, pw8, O, k, 1, pre=s, post=d, bias=bias, bias_scale=1.0, act=act, gain=gain)
        else:
            y = K.conv_fwd(x, pw, O, k, 1, pre=s, post=d, bias=bias, bias_scale=1.0, act=act, gain=gain)
        ctx.save_for_backward(x, w, bias, s, d, wsq, y)
        ctx.cfg = (up, act, gain, c_eq)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, w, bias, s, d, wsq, y = ctx.saved_tensors
        up, act, gain, c_eq = ctx.cfg
        gx, gw, gb, gs = _modconv_backward(x, w, bias, s, d, wsq, y, up, act, gain, c_eq, gy)
        return gx, gw, gb, gs, None, None, None


def _modconv_backward(x, w, bias, s, d, wsq, y, up, act, gain, c_eq, gy, residual=None):
    """backward of ModConvFn: (gx [+ residual], gw, gbias, gs)"""
    # activation backward + bias gradient + demod statistic  gdq[b,o] = sum_p gz * (ypre - bias)
    gy = gy.contiguous()
    # where the pass writes gz anyway it writes d[b,o] * gz (one rounding of the fp32 product), the operand both launches of the tail want
    demod = act != ACT_NONE and _gz_demod()
    gz, gb, gdq = _K().act_bwd_reduce(gy, y, act, gain, w.shape[0], want_gz=(act != ACT_NONE), bias=bias, bias_scale=1.0,
                                      want_gbias=True, want_gdq=True, out_scale=d if demod else None)
    if gz is None:
        gz = gy
    gx, gw, gs = _modconv_bwd_tail(x, w, s, d, wsq, gz, gdq, up, c_eq, residual, gz_has_d=demod)
    return gx, gw, gb, gs


def _modconv_bwd_tail(x, w, s, d, wsq, gz, gdq, up, c_eq, residual=None, gz_has_d=False):
    """from the pre-activation gradient gz of a modulated convolution to (gx [+ residual], gw, gs); residual: the gradient another
    consumer of x already produced -- it joins in the data-gradient launch's epilogue instead of an add pass of autograd's;
    gz_has_d: gz is already d[b,o] * gz (lcgan_act_bwd_reduce_s / lcgan_rgb_reduce_bwd_act_s): no per-sample scale on that operand"""
    K = _K()
    O, Cin, k, _ = w.shape
    pwT, _ = _prep(w, c_eq, True, _need_lo(x))                               # [t][Cin][O]
    dz = None if gz_has_d else d
    # data gradient u = conv^T(d * gz); gx = s * u and gs = sum_p x * u leave the same launch (epilogue of the conv kernel)
    if up == 2:
        gx, gs = K.conv_fwd(gz, pwT, Cin, k, 2, pre=dz, post=s, xs=x, residual=residual)         # adjoint of the transposed conv
    else:
        gx, gs = K.conv_bwd_data(gz, pwT, Cin, k, 1, pre=dz, post=s, xs=x, residual=residual)
    gwsq = K.demod_bwd(gdq, d, s, wsq, gs)                                   # gs += demod path
    if up == 2:
        gw = K.conv_wgrad_unprep(gz, x, Cin, O, k, 2, c_eq, transposed=True, pre_x=dz, pre_g=s, w=w, gwsq=gwsq)    # gwp [t][Cin][O]
    else:
        gw = K.conv_wgrad_unprep(x, gz, O, Cin, k, 1, c_eq, transposed=False, pre_x=s, pre_g=dz, w=w, gwsq=gwsq)   # gwp [t][O][Cin]
    return gx, gw, gs


class ModConvRGBFn(Function):
    """ToRGBBlock as ONE node (custom_layers.py:177-182): img = rgb_reduce(lrelu(modconv3x3(x, s)), wm) + rgb_bias * rgb_bias_scale,
    wm [B,3,C] the modulated + demodulated 1x1 weights (torch glue on a few hundred values, custom_layers.py:62-68).  Owning both layers
    lets the backward go from the image gradient to the 3x3 conv's pre-activation gradient in one pass over its activation
    (lcgan_rgb_reduce_bwd_act) instead of rgb_expand + rgb_wgrad + act_bwd_reduce: 1.1 GB instead of 2.7 GB at 256 x 256, batch 32."""

    @staticmethod
    def forward(ctx, x, w, bias, s, wm, rgb_bias, rgb_bias_scale, act, gain):
        K = _K()
        O, Cin, k, _ = w.shape
        c_eq = 1.0 / math.sqrt(Cin * k * k)
        s = s.contiguous()
        pw, wsq = _prep(w, c_eq, False, _need_lo(x), want_wsq=True)
        d = _demod(K, w, s, wsq, ceil8(O))
        if _use_fp8(x, k, 1):
            pw8, _ = _prep(w, c_eq, False, False, fp8=True)
            y = K.conv_fwd_fp8(x, pw8, O, k, 1, pre=s, post=d, bias=bias, bias_scale=1.0, act=act, gain=gain)
        else:
            y = K.conv_fwd(x, pw, O, k, 1, pre=s, post=d, bias=bias, bias_scale=1.0, act=act, gain=gain)
        wm = wm.contiguous()
        img = K.rgb_reduce(y, wm, rgb_bias, rgb_bias_scale)
        ctx.save_for_backward(x, w, bias, s, d, wsq, y, wm)
        ctx.cfg = (act, gain, c_eq, rgb_bias_scale, rgb_bias is not None)
        return img

    @staticmethod
    @once_differentiable
    def backward(ctx, gimg):
        K = _K()
        x, w, bias, s, d, wsq, y, wm = ctx.saved_tensors
        act, gain, c_eq, rgb_bias_scale, has_rgb_bias = ctx.cfg
        gimg = gimg.contiguous()
        demod = _gz_demod()
        gz, gb, gdq, gwm = K.rgb_reduce_bwd_act(gimg, y, wm, bias, 1.0, act, gain, w.shape[0], out_scale=d if demod else None)
        gx, gw, gs = _modconv_bwd_tail(x, w, s, d, wsq, gz, gdq, 1, c_eq, gz_has_d=demod)
        grb = gimg.sum(dim=(0, 2, 3)) * rgb_bias_scale if (has_rgb_bias and ctx.needs_input_grad[5]) else None     # 3 numbers
        return gx, gw, gb, gs, gwm, grb, None, None, None


_flow_cache: dict = {}      # id(param) -> [weakref(param), version, epoch, (pw18, pw18T, wsq) per need_lo]


def _flow_prep(w: Tensor, c_eq: float, need_lo: bool):
    """Prepared copies of a flow layer's weight [2, Cin, 3, 3] for FlowConvFn: the [18][Cin] 1x1 weight (row (ky*3+kx)*2+o) in forward and
    transposed GEMM layout, and the demodulation statistic wsq [2, Cin]; valid for (parameter object, version counter, weight epoch)."""
    import weakref
    ent = _flow_cache.get(id(w)) if isinstance(w, torch.nn.Parameter) else None
    if ent is not None and ent[0]() is w and ent[1] == w._version and ent[2] == _weight_epoch and need_lo in ent[3]:
        return ent[3][need_lo]
    K = _K()
    wd = w.detach()
    w18 = wd.permute(2, 3, 0, 1).reshape(18, wd.shape[1], 1, 1).contiguous()          # torch glue on 9 K values
    out = (K.prep_weight(w18, c_eq, False, need_lo)[0], K.prep_weight(w18, c_eq, True, need_lo)[0], (wd * c_eq).square().sum(dim=(2, 3)))
    if isinstance(w, torch.nn.Parameter):
        if ent is None or ent[0]() is not w or ent[1] != w._version or ent[2] != _weight_epoch:
            if len(_flow_cache) > 256:
                _flow_cache.clear()
            ent = [weakref.ref(w), w._version, _weight_epoch, {}]
            _flow_cache[id(w)] = ent
        ent[3][need_lo] = out
    return out


class FlowConvFn(Function):
    """The flow layer of a SynthesisBlock -- ModulatedConv2d(Cin -> 2, k 3, up 2) with demodulation and bias, no activation
    (custom_layers.py:123,149; :62-80,85) -- as a 1x1 convolution Cin -> 18 on the LOW-resolution grid followed by the col2im scatter of
    the transposed convolution (csrc/stencil.hip: flow_col2im): x is read once at the HBM rate; the generic path ran four
    128-output-channel MFMA phases for 2 useful columns (0.5 ms per generator forward, 0.6 + 0.7 ms in its backward).  First order only."""

    @staticmethod
    def forward(ctx, x, w, bias, s):
        K = _K()
        O, Cin, k, _ = w.shape
        assert O == 2 and k == 3
        c_eq = 1.0 / math.sqrt(Cin * k * k)
        s = s.contiguous()
        pw18, _, wsq = _flow_prep(w, c_eq, _need_lo(x))
        d = _demod(K, w, s, wsq, 8)
        t = K.conv_fwd(x, pw18, 18, 1, 1, pre=s)                                 # [B,H,W,24]
        u = K.flow_col2im(t, d, bias)                                            # [B,2H,2W,8]
        ctx.save_for_backward(x, w, bias, s, d, wsq, u)
        ctx.c_eq = c_eq
        return u

    @staticmethod
    @once_differentiable
    def backward(ctx, gu):
        x, w, bias, s, d, wsq, u = ctx.saved_tensors
        return _flow_backward(x, w, bias, s, d, wsq, u, ctx.c_eq, gu)


def _flow_backward(x, w, bias, s, d, wsq, u, c_eq, gu, residual=None):
    """backward of FlowConvFn: (gx [+ residual], gw, gbias, gs)"""
    K = _K()
    Cin = w.shape[1]
    gu = gu.contiguous()
    _, gb, gdq = K.act_bwd_reduce(gu, u, ACT_NONE, 1.0, 2, want_gz=False, bias=bias, bias_scale=1.0, want_gbias=True, want_gdq=True)
    _, pw18T, _ = _flow_prep(w, c_eq, _need_lo(x))
    gt = K.flow_im2col(gu, d)                                                # [B,H,W,24], demodulation folded in
    gx, gs = K.conv_bwd_data(gt, pw18T, Cin, 1, 1, post=s, xs=x, residual=residual)   # gx = s * (gt @ W18) [+ residual], gs = sum_p x * (gt @ W18)
    gwsq = K.demod_bwd(gdq, d, s, wsq, gs)                                   # gs += demod path
    gw18 = K.conv_wgrad(x, gt, 18, Cin, 1, 1, pre_x=s)                       # [1][18][Cin], row (ky*3+kx)*2+o == the prepared layout [9][2][Cin]
    gw = K.unprep_wgrad(gw18.view(9, 2, Cin), 2, Cin, 3, c_eq, False, w=w.detach(), gwsq=gwsq)      # + the demodulation term, one launch
    return gx, gw, gb, gs


class SynthForkFn(Function):
    """The three consumers of a SynthesisBlock's input x as ONE autograd node (custom_layers.py:145-153):
         skip = conv1x1(x, Wk * wscale)                      skip_layer (the sqrt(.5) gain folded into wscale)   :145
         u    = FlowConvFn(x, Wf, bf, sf)                    flow_layer: ModulatedConv2d(C -> 2, k 3, up 2)      :149
         y0   = ModConvFn(x, W0, b0, s0, up 2, no act)       modulated_conv0                                     :153
    As three nodes autograd sums their three data gradients with two full-size add passes per block and backward pass
    (0.66 ms per iteration at 256 x 256, batch 32).  Owning the fork, the backward CHAINS them: the 1x1 data gradient first, then
    the flow layer's and the up-convolution's data-gradient launches each take the running sum as the residual of their epilogue
    (lcgan_conv_fwd / lcgan_conv_bwd_data with xs AND residual) -- the same bf16 values the add passes produced, no extra pass.
    First order only (the R1 double backward never reaches the generator)."""

    @staticmethod
    def forward(ctx, x, wf, bf, sf, w0, b0, s0, wk, wscale):
        K = _K()
        lo = _need_lo(x)
        Cin = x.shape[-1]
        # flow layer (FlowConvFn.forward)
        cf = 1.0 / math.sqrt(wf.shape[1] * 9)
        sf = sf.contiguous()
        pw18, _, wsqf = _flow_prep(wf, cf, lo)
        df = _demod(K, wf, sf, wsqf, 8)
        u = K.flow_col2im(K.conv_fwd(x, pw18, 18, 1, 1, pre=sf), df, bf)
        # up-convolution (ModConvFn.forward, up = 2, no activation)
        O = w0.shape[0]
        c0 = 1.0 / math.sqrt(w0.shape[1] * 9)
        s0 = s0.contiguous()
        pw0, wsq0 = _prep(w0, c0, False, lo, want_wsq=True)
        d0 = _demod(K, w0, s0, wsq0, ceil8(O))
        y0 = K.conv_bwd_data(x, pw0, O, 3, 2, pre=s0, post=d0, bias=b0, bias_scale=1.0, act=ACT_NONE, gain=1.0)
        # skip branch (Conv2dFn.forward, k = 1, no bias, no activation)
        pwk, _ = _prep(wk, wscale, False, lo)
        skip = K.conv_fwd(x, pwk, wk.shape[0], 1, 1)
        ctx.save_for_backward(x, wf, bf, sf, df, wsqf, u, w0, b0, s0, d0, wsq0, y0, wk)
        ctx.cfg = (cf, c0, wscale)
        return u, y0, skip

    @staticmethod
    @once_differentiable
    def backward(ctx, gu, gy0, gskip):
        K = _K()
        x, wf, bf, sf, df, wsqf, u, w0, b0, s0, d0, wsq0, y0, wk = ctx.saved_tensors
        cf, c0, wscale = ctx.cfg
        gx = gwk = gwf = gbf = gsf = gw0 = gb0 = gs0 = None
        if gskip is not None:
            gskip = gskip.contiguous()
            pwkT, _ = _prep(wk, wscale, True, _need_lo(x))
            gx = K.conv_bwd_data(gskip, pwkT, wk.shape[1], 1, 1)
            gwk = K.conv_wgrad_unprep(x, gskip, wk.shape[0], wk.shape[1], 1, 1, wscale)
        if gu is not None:
            gx, gwf, gbf, gsf = _flow_backward(x, wf, bf, sf, df, wsqf, u, cf, gu, residual=gx)
        if gy0 is not None:
            gx, gw0, gb0, gs0 = _modconv_backward(x, w0, b0, s0, d0, wsq0, y0, 2, ACT_NONE, 1.0, c0, gy0, residual=gx)
        return gx, gwf, gbf, gsf, gw0, gb0, gs0, gwk, None


class Box3ActFn(Function):
    """y = act(box3(x)) * gain  (custom_layers.py:150-151, 154-155)"""

    @staticmethod
    def forward(ctx, x, act, gain):
        y = _K().box3_act(x, act, gain)
        ctx.save_for_backward(y)
        ctx.cfg = (act, gain)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        return _K().box3_act_bwd(gy.contiguous(), y, *ctx.cfg), None, None


class Up2BoxAddFn(Function):
    """y = box3(nearest_x2(x)) + residual   (custom_layers.py:146-147, 159)"""

    @staticmethod
    def forward(ctx, x, residual):
        return _K().up2box(x, residual)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        gy = gy.contiguous()
        return _K().up2box_bwd(gy), gy


class WarpFn(Function):
    """bicubic grid_sample with the base grid + flow*scale built in the kernel (custom_layers.py:127-134, 162-165)"""

    @staticmethod
    def forward(ctx, x, flow, scale):
        ctx.save_for_backward(x, flow)
        ctx.scale = scale
        return _K().warp_fwd(x, flow, scale)

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        x, flow = ctx.saved_tensors
        gx, gflow = _K().warp_bwd(gy.contiguous(), x, flow, ctx.scale)
        return gx, gflow, None


class QrQFn(Function):
    """Q of the reduced Householder QR of square matrices [n,n] or [nb,n,n] (torch.qr(.)[0], custom_layers.py:274-276), one HIP
    workgroup per matrix.  Backward (Q only, A = QR square and invertible): with G = Q^T gQ,  gA = Q tril(G - G^T, -1) R^{-T}
    (from Q^T dA R^{-1} = Q^T dQ + dR R^{-1}: antisymmetric + upper-triangular); a 64 x 64 triangular solve of torch glue."""

    @staticmethod
    def forward(ctx, A):
        Q, R = _K().qr(A.contiguous())
        ctx.save_for_backward(Q, R)
        return Q

    @staticmethod
    @once_differentiable
    def backward(ctx, gQ):
        Q, R = ctx.saved_tensors
        G = Q.transpose(-1, -2) @ gQ
        return torch.linalg.solve_triangular(R.transpose(-1, -2), Q @ torch.tril(G - G.transpose(-1, -2), -1), upper=False, left=False)


# =====================================================================================================
# losses (first-order)
# =====================================================================================================
class BCELogitsFn(Function):
    """mean softplus(-+logit): F.binary_cross_entropy_with_logits against all-ones / all-zeros (worker.py:156-157,191)"""

    @staticmethod
    def forward(ctx, logit, target_one):
        logit = logit.contiguous()
        ctx.save_for_backward(logit)
        ctx.t = target_one
        return _K().bce_fwd(logit, target_one)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (logit,) = ctx.saved_tensors
        return _K().bce_bwd(logit, ctx.t, g.contiguous()), None


class ContrastiveFn(Function):
    """loss.py:9-15"""

    @staticmethod
    def forward(ctx, a, p, n, tau):
        a, p, n = a.contiguous(), p.contiguous(), n.contiguous()
        out, t = _K().contrastive_fwd(a, p, n, tau)
        ctx.save_for_backward(a, p, n, t)
        ctx.tau = tau
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        a, p, n, t = ctx.saved_tensors
        ga, gp, gn = _K().contrastive_bwd(a, p, n, t, g.contiguous(), ctx.tau)
        return ga, gp, gn, None


class L2NormalizeFn(Function):
    """F.normalize(x) (cnn.py:40-41)"""

    @staticmethod
    def forward(ctx, x):
        y, ns = _K().l2norm_fwd(x.contiguous())
        ctx.save_for_backward(y, ns)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, gy):
        y, ns = ctx.saved_tensors
        return _K().l2norm_bwd(gy.contiguous(), y, ns)


class PowSumFn(Function):
    """coef * sum |x|^pw  -- pw=1: L1 sparsity (worker.py:207-209); pw=2: R1 square sum (loss.py:20-23)."""

    @staticmethod
    def forward(ctx, x, pw, coef):
        x = x.contiguous()
        ctx.save_for_backward(x)
        ctx.cfg = (pw, coef)
        return _K().powsum(x.view(-1), pw, coef)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return _K().powsum_bwd(x.view(-1), ctx.cfg[0], ctx.cfg[1], g.contiguous()).view_as(x), None, None

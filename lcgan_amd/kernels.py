"""Tensor-level view of the C ABI (include/lcgan_hip.h): every method allocates its outputs with torch (device memory
is plumbing), passes raw device pointers + the current HIP stream to liblcgan_hip.so and returns the outputs.

`K` resolves to `HipKernels` on first use; there is no other backend in the product (tests/helpers.py:install_backend swaps the
object behind `K` for a CPU emulation of the same interface to exercise the host-side autograd wiring without a GPU).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib

Tensor = torch.Tensor
ACT_NONE, ACT_LRELU, ACT_TANH = 0, 1, 2
DT_F32, DT_BF16 = 0, 1


def ceil8(n: int) -> int:
    return (n + 7) // 8 * 8


def dt_code(dtype: torch.dtype) -> int:
    if dtype == torch.float32:
        return DT_F32
    if dtype == torch.bfloat16:
        return DT_BF16
    raise TypeError(f"feature maps must be float32 or bfloat16, got {dtype}")


def _p(t: Optional[Tensor]):
    return None if t is None else t.data_ptr()


class PreparedWeight:
    """bf16 GEMM-layout copy of a conv weight: [parts][k*k][N][Kpad]; parts = 1 (bf16 features) or 3 (f32 parity mode:
    scale*w = p0 + p1 + p2, 24 mantissa bits)."""
    __slots__ = ("buf", "parts", "N", "Kpad", "k")

    def __init__(self, buf, parts, N, Kpad, k):
        self.buf, self.parts, self.N, self.Kpad, self.k = buf, parts, N, Kpad, k


class PreparedWeightFp8:
    """MX-fp8 copy of a conv weight: e4m3 [k*k][N][K64][64] + E8M0 block scales [k*k][N][K64][2] (csrc/conv_fp8.hip)."""
    __slots__ = ("buf", "scales", "N", "K64", "k")

    def __init__(self, buf, scales, N, K64, k):
        self.buf, self.scales, self.N, self.K64, self.k = buf, scales, N, K64, k


class _ZeroPool:
    """Zero-initialised fp32 scratch carved out of one pre-cleared slab: the atomically accumulated outputs (weight / bias /
    style gradients) are many and small, and one fill per 64 MB replaces ~370 fill launches per iteration.  A slice keeps
    its slab alive through torch's storage refcount, so a slab is only recycled by the caching allocator once every
    tensor carved from it is gone."""
    SLAB = 1 << 24          # floats

    def __init__(self):
        self.buf, self.off = None, 0

    def take(self, shape, device) -> Tensor:
        n = 1
        for d in shape:
            n *= int(d)
        device = torch.device(device)
        if n > self.SLAB // 4:
            return torch.zeros(shape, dtype=torch.float32, device=device)
        if self.buf is None or self.off + n > self.SLAB or self.buf.device != device:
            self.buf, self.off = torch.zeros(self.SLAB, dtype=torch.float32, device=device), 0
        # A tensor of its own over the slab's storage, NOT a view of `buf`: views share one version counter, and the slices are both saved
        # for backward (demodulation vectors) and mutated in place by autograd (a stolen gradient that a second backward accumulates
        # into) -- as views, an in-place add on one slice made autograd reject every saved slice of the slab ("modified by an inplace
        # operation"): two backward passes without zero_grad() in between were enough.
        v = torch.empty(0, dtype=torch.float32, device=device).set_(self.buf.untyped_storage(), self.off, tuple(int(d) for d in shape))
        self.off += (n + 63) // 64 * 64                     # keep every slice 256-byte aligned
        return v


class HipKernels:
    name = "hip"

    def __init__(self):
        self.lib = _lib.load()
        self._zeros = _ZeroPool()
        self._prep_tables = {}                 # cached device job tables of prep_weight_group
        import os
        for kv in filter(None, os.environ.get("LCGAN_OPTIONS", "").split(",")):     # tuning switches, e.g. LCGAN_OPTIONS="4=1,5=0"
            k, v = kv.split("=")
            self.lib.lcgan_set_option(int(k), int(v))

    # ------------------------------------------------------------------------------------------------
    @staticmethod
    def _stream():
        # raw handle of torch's current stream on the current device; torch.cuda.current_stream() builds a Stream object through
        # several Python layers (9 us per call, ~1000 calls per iteration: 10 % of a batch-4 step, which is launch-bound)
        try:
            return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())
        except AttributeError:                                # (private torch entry points: fall back to the public, slower path)
            return torch.cuda.current_stream().cuda_stream

    def _call(self, name, *args):
        _lib.check(getattr(self.lib, name)(*args), name)

    @staticmethod
    def _chk(*ts):
        for t in ts:
            if t is not None:
                assert t.is_cuda and t.is_contiguous(), "HIP kernels need contiguous device tensors"

    # ---- conv family ------------------------------------------------------------------------------------
    def prep_weight(self, w: Tensor, scale: float, transpose: bool, need_lo: bool, want_wsq: bool = False):
        self._chk(w)
        A, Bc, k, _ = w.shape
        N, Kc = (Bc, A) if transpose else (A, Bc)
        Kpad = (Kc + 31) // 32 * 32
        parts = 3 if need_lo else 1
        buf = torch.empty((parts, k * k, N, Kpad), dtype=torch.bfloat16, device=w.device)
        wsq = torch.empty((A, Bc), dtype=torch.float32, device=w.device) if want_wsq else None
        self._call("lcgan_conv_weight_prep", w.data_ptr(), A, Bc, k, float(scale), int(transpose), buf.data_ptr(), parts, _p(wsq),
                   self._stream())
        return PreparedWeight(buf, parts, N, Kpad, k), wsq

    _PREP_DESC = None
    PREP_CHUNK = 16384

    def prep_weight_group(self, jobs):
        """jobs: [(w, scale, transpose, need_lo, want_wsq)] -> [(PreparedWeight, wsq | None)] from ONE launch.  The device job
        table only depends on the parameters' addresses / shapes and is cached."""
        import numpy as np
        if HipKernels._PREP_DESC is None:
            HipKernels._PREP_DESC = np.dtype([("w", "<u8"), ("out_off", "<i8"), ("wsq_off", "<i8"), ("A", "<i4"), ("Bc", "<i4"),
                                              ("kk", "<i4"), ("transpose", "<i4"), ("parts", "<i4"), ("N", "<i4"), ("Kc", "<i4"),
                                              ("Kpad", "<i4"), ("scale", "<f4"), ("pad", "<i4")])
            assert HipKernels._PREP_DESC.itemsize == 64
        dev = jobs[0][0].device
        sig = tuple((w.data_ptr(), tuple(w.shape), float(sc), bool(tr), bool(lo), bool(ws)) for w, sc, tr, lo, ws in jobs)
        tab = self._prep_tables.get(sig)
        if tab is None:
            if len(self._prep_tables) > 64:
                self._prep_tables.clear()
            arr = np.zeros(len(jobs), dtype=HipKernels._PREP_DESC)
            ce, ci, meta = [], [], []
            out_off = wsq_off = 0
            for i, (w, sc, tr, lo, ws) in enumerate(jobs):
                self._chk(w)
                A, Bc, k, _ = w.shape
                N, Kc = (Bc, A) if tr else (A, Bc)
                Kpad = (Kc + 31) // 32 * 32
                parts = 3 if lo else 1
                total = k * k * N * Kpad
                tile_wsq = bool(ws) and k * k <= 9                  # a tile stages every tap: it writes wsq on the way
                arr[i] = (w.data_ptr(), out_off, wsq_off, A, Bc, k * k, int(tr), parts, N, Kc, Kpad, sc, int(tile_wsq))
                n = ((N + 15) // 16) * ((Kpad + 63) // 64)          # tiles of 16 rows x 64 channels x all taps
                ce += [i] * n
                ci += list(range(n))
                if ws and not tile_wsq:
                    n = (A * Bc + self.PREP_CHUNK - 1) // self.PREP_CHUNK
                    ce += [i] * n
                    ci += [-(j + 1) for j in range(n)]
                meta.append((out_off, parts, total, N, Kpad, k, wsq_off if ws else -1, A, Bc))
                out_off += parts * total
                wsq_off += A * Bc if ws else 0
            tab = (torch.from_numpy(arr.view(np.uint8).copy()).to(dev), torch.tensor(ce, dtype=torch.int32).to(dev),
                   torch.tensor(ci, dtype=torch.int32).to(dev), meta, out_off, wsq_off)
            self._prep_tables[sig] = tab
        descs, ce, ci, meta, n_out, n_wsq = tab
        out = torch.empty((n_out,), dtype=torch.bfloat16, device=dev)
        wsqs = torch.empty((max(n_wsq, 1),), dtype=torch.float32, device=dev)
        self._call("lcgan_conv_weight_prep_group", descs.data_ptr(), ce.data_ptr(), ci.data_ptr(), ce.numel(), out.data_ptr(),
                   wsqs.data_ptr(), float(n_out), self._stream())
        res = []
        for (o, parts, total, N, Kpad, k, wo, A, Bc) in meta:
            buf = out[o:o + parts * total].view(parts, k * k, N, Kpad)
            res.append((PreparedWeight(buf, parts, N, Kpad, k), wsqs[wo:wo + A * Bc].view(A, Bc) if wo >= 0 else None))
        return res

    def unprep_wgrad(self, gwp: Tensor, A: int, Bc: int, k: int, scale: float, transposed: bool = False,
                     w: Optional[Tensor] = None, gwsq: Optional[Tensor] = None) -> Tensor:
        self._chk(gwp, w, gwsq)
        gw = torch.empty((A, Bc, k, k), dtype=torch.float32, device=gwp.device)
        self._call("lcgan_conv_wgrad_unprep", gwp.data_ptr(), A, Bc, k, float(scale), int(transposed), _p(w), _p(gwsq), gw.data_ptr(),
                   self._stream())
        return gw

    def conv_fwd(self, x: Tensor, pw: PreparedWeight, N: int, k: int, stride: int, pre=None, post=None, bias=None,
                 bias_scale: float = 1.0, act: int = ACT_NONE, gain: float = 1.0, residual=None, residual_half: bool = False,
                 xs: Optional[Tensor] = None, pool: bool = False, want_mask: bool = False):
        """xs given (needs post, no bias/act): returns (y = post * u [+ residual], gs[b,n] = sum_pixels xs * u), u = the unscaled result;
        pool: returns (y, avg_pool2d(y, 2)) -- the by-product a DiscriminatorBlock's closing convolution leaves for the next block"""
        self._chk(x, pre, post, bias, residual, xs)
        B, H, W, Cin = x.shape
        Cout = ceil8(N)
        Ho, Wo = (H + stride - 1) // stride, (W + stride - 1) // stride
        y = torch.empty((B, Ho, Wo, Cout), dtype=x.dtype, device=x.device)
        gs = self._zeros.take((B, Cout), x.device) if xs is not None else None
        pooled = torch.empty((B, Ho // 2, Wo // 2, Cout), dtype=x.dtype, device=x.device) if pool else None
        assert pw.parts == (3 if x.dtype == torch.float32 else 1) and not (pool and xs is not None)
        if want_mask:
            # want_mask: returns (y, mask | None) -- the activation's sign bits as a by-product of the epilogue ([B*Ho*Wo, Cout/8] bytes), where
            # the launch path can write them (lcgan_conv_fwd_m); None: the activation backward keeps reading y
            assert not pool and xs is None
            import ctypes as C
            mask, wrote = None, C.c_int(0)
            if act == ACT_LRELU and x.dtype == torch.bfloat16 and Cout % 32 == 0 and residual is None:
                mask = torch.empty((B * Ho * Wo, Cout // 8), dtype=torch.uint8, device=x.device)
            self._call("lcgan_conv_fwd_m", x.data_ptr(), pw.buf.data_ptr(), y.data_ptr(), B, H, W, Cin, Cout, N, k, stride,
                       _p(pre), _p(post), _p(bias), float(bias_scale), act, float(gain), _p(residual), int(residual_half), None, None,
                       None, _p(mask), C.addressof(wrote), dt_code(x.dtype), self._stream())
            return y, (mask if wrote.value else None)
        self._call("lcgan_conv_fwd", x.data_ptr(), pw.buf.data_ptr(), y.data_ptr(), B, H, W, Cin, Cout, N, k, stride,
                   _p(pre), _p(post), _p(bias), float(bias_scale), act, float(gain), _p(residual), int(residual_half), _p(xs), _p(gs),
                   _p(pooled), dt_code(x.dtype), self._stream())
        if pool:
            return y, pooled
        return y if xs is None else (y, gs)

    def conv_bwd_data(self, g: Tensor, pw: PreparedWeight, N: int, k: int, stride: int, pre=None, post=None, bias=None,
                      bias_scale: float = 1.0, act: int = ACT_NONE, gain: float = 1.0, residual=None, residual_half: bool = False,
                      xs: Optional[Tensor] = None):
        """residual_half: residual is [B,H/2,W/2,C] of gx's grid and enters as 0.25 * nearest-x2 (avg_pool2d adjoint);
        xs: as conv_fwd -> (gx, gs)"""
        self._chk(g, pre, post, bias, residual, xs)
        B, H, W, Cg = g.shape
        Cout = ceil8(N)
        gx = torch.empty((B, H * stride, W * stride, Cout), dtype=g.dtype, device=g.device)
        gs = self._zeros.take((B, Cout), g.device) if xs is not None else None
        assert pw.parts == (3 if g.dtype == torch.float32 else 1)
        self._call("lcgan_conv_bwd_data", g.data_ptr(), pw.buf.data_ptr(), gx.data_ptr(), B, H, W, Cg, Cout, N, k, stride,
                   _p(pre), _p(post), _p(bias), float(bias_scale), act, float(gain), _p(residual), int(residual_half), _p(xs), _p(gs),
                   dt_code(g.dtype), self._stream())
        return gx if xs is None else (gx, gs)

    def conv_wgrad(self, x: Tensor, g: Tensor, A: int, Bc: int, k: int, stride: int, pre_x=None, pre_g=None) -> Tensor:
        self._chk(x, g, pre_x, pre_g)
        B, Hx, Wx, Cx = x.shape
        _, Hg, Wg, Cg = g.shape
        gwp = self._zeros.take((k * k, A, Bc), x.device)
        self._call("lcgan_conv_wgrad", x.data_ptr(), g.data_ptr(), gwp.data_ptr(), B, Hx, Wx, Cx, Hg, Wg, Cg, A, Bc, k, stride,
                   _p(pre_x), _p(pre_g), dt_code(x.dtype), self._stream())
        return gwp

    def conv_wgrad_unprep(self, x: Tensor, g: Tensor, A: int, Bc: int, k: int, stride: int, scale: float, transposed: bool = False,
                          pre_x=None, pre_g=None, w: Optional[Tensor] = None, gwsq: Optional[Tensor] = None) -> Tensor:
        """conv_wgrad + unprep_wgrad as one call (lcgan_conv_wgrad_fused): returns the gradient in weight layout
        ([A][Bc][k][k], or [Bc][A][k][k] with transposed)."""
        self._chk(x, g, pre_x, pre_g, w, gwsq)
        B, Hx, Wx, Cx = x.shape
        _, Hg, Wg, Cg = g.shape
        gwp = torch.empty((k * k, A, Bc), dtype=torch.float32, device=x.device)     # scratch: the call clears it where it needs to
        wA, wBc = (Bc, A) if transposed else (A, Bc)
        gw = torch.empty((wA, wBc, k, k), dtype=torch.float32, device=x.device)
        self._call("lcgan_conv_wgrad_fused", x.data_ptr(), g.data_ptr(), gwp.data_ptr(), B, Hx, Wx, Cx, Hg, Wg, Cg, A, Bc, k, stride,
                   _p(pre_x), _p(pre_g), dt_code(x.dtype), float(scale), int(transposed), _p(w), _p(gwsq), gw.data_ptr(), self._stream())
        return gw

    # ---- MX-fp8 convolution path (BASELINE configs[4]) ------------------------------------------------------------------
    def prep_weight_fp8(self, w: Tensor, scale: float, transpose: bool):
        self._chk(w)
        A, Bc, k, _ = w.shape
        N, Kc = (Bc, A) if transpose else (A, Bc)
        K64 = (Kc + 63) // 64
        buf = torch.empty((k * k, N, K64, 64), dtype=torch.uint8, device=w.device)
        sc = torch.empty((k * k, N, K64, 2), dtype=torch.uint8, device=w.device)
        self._call("lcgan_conv_weight_prep_fp8", w.data_ptr(), A, Bc, k, float(scale), int(transpose), buf.data_ptr(), sc.data_ptr(),
                   self._stream())
        return PreparedWeightFp8(buf, sc, N, K64, k)

    def conv_fwd_fp8(self, x: Tensor, pw, N: int, k: int, stride: int, pre=None, post=None, bias=None, bias_scale: float = 1.0,
                     act: int = ACT_NONE, gain: float = 1.0, residual=None, residual_half: bool = False) -> Tensor:
        self._chk(x, pre, post, bias, residual)
        assert x.dtype == torch.bfloat16
        B, H, W, Cin = x.shape
        Cout = ceil8(N)
        y = torch.empty((B, (H + stride - 1) // stride, (W + stride - 1) // stride, Cout), dtype=x.dtype, device=x.device)
        self._call("lcgan_conv_fwd_fp8", x.data_ptr(), pw.buf.data_ptr(), pw.scales.data_ptr(), y.data_ptr(), B, H, W, Cin, Cout, N, k, stride,
                   _p(pre), _p(post), _p(bias), float(bias_scale), act, float(gain), _p(residual), int(residual_half), self._stream())
        return y

    def conv_bwd_data_fp8(self, g: Tensor, pw, N: int, k: int, stride: int, pre=None, post=None, bias=None, bias_scale: float = 1.0,
                          act: int = ACT_NONE, gain: float = 1.0, residual=None, residual_half: bool = False) -> Tensor:
        self._chk(g, pre, post, bias, residual)
        assert g.dtype == torch.bfloat16
        B, H, W, Cg = g.shape
        Cout = ceil8(N)
        gx = torch.empty((B, H * stride, W * stride, Cout), dtype=g.dtype, device=g.device)
        self._call("lcgan_conv_bwd_data_fp8", g.data_ptr(), pw.buf.data_ptr(), pw.scales.data_ptr(), gx.data_ptr(), B, H, W, Cg, Cout, N, k,
                   stride, _p(pre), _p(post), _p(bias), float(bias_scale), act, float(gain), _p(residual), int(residual_half), self._stream())
        return gx

    # ---- stencils ---------------------------------------------------------------------------------------
    def box3_act(self, x: Tensor, act: int, gain: float) -> Tensor:
        self._chk(x)
        B, H, W, Cc = x.shape
        y = torch.empty_like(x)
        self._call("lcgan_box3_act", x.data_ptr(), y.data_ptr(), B, H, W, Cc, act, float(gain), dt_code(x.dtype), self._stream())
        return y

    def box3_act_bwd(self, gy: Tensor, y: Optional[Tensor], act: int, gain: float) -> Tensor:
        self._chk(gy, y)
        B, H, W, Cc = gy.shape
        gx = torch.empty_like(gy)
        self._call("lcgan_box3_act_bwd", gy.data_ptr(), _p(y), gx.data_ptr(), B, H, W, Cc, act, float(gain), dt_code(gy.dtype),
                   self._stream())
        return gx

    def box3_actbwd(self, gy: Tensor, y: Tensor, act: int, gain: float, clog: int, want_gbias: bool, mask: Optional[Tensor] = None):
        """-> (gz = box3(gy) * act'(y), gbias [clog] | None); mask: the activation's sign bits (conv_fwd(want_mask=True)) read instead of y"""
        self._chk(gy, y, mask)
        B, H, W, Cc = gy.shape
        gz = torch.empty_like(gy)
        gbias = self._zeros.take((clog,), gy.device) if want_gbias else None
        self._call("lcgan_box3_actbwd_reduce_m", gy.data_ptr(), y.data_ptr(), _p(mask), gz.data_ptr(), _p(gbias), B, H, W, Cc, clog, act,
                   float(gain), dt_code(gy.dtype), self._stream())
        return gz, gbias

    def up2box(self, x: Tensor, residual: Optional[Tensor]) -> Tensor:
        self._chk(x, residual)
        B, H, W, Cc = x.shape
        y = torch.empty((B, 2 * H, 2 * W, Cc), dtype=x.dtype, device=x.device)
        self._call("lcgan_up2box", x.data_ptr(), _p(residual), y.data_ptr(), B, H, W, Cc, dt_code(x.dtype), self._stream())
        return y

    def up2box_bwd(self, gy: Tensor) -> Tensor:
        self._chk(gy)
        B, H2, W2, Cc = gy.shape
        gx = torch.empty((B, H2 // 2, W2 // 2, Cc), dtype=gy.dtype, device=gy.device)
        self._call("lcgan_up2box_bwd", gy.data_ptr(), gx.data_ptr(), B, H2 // 2, W2 // 2, Cc, dt_code(gy.dtype), self._stream())
        return gx

    def avgpool2(self, x: Tensor) -> Tensor:
        self._chk(x)
        B, H, W, Cc = x.shape
        y = torch.empty((B, H // 2, W // 2, Cc), dtype=x.dtype, device=x.device)
        self._call("lcgan_avgpool2", x.data_ptr(), y.data_ptr(), B, H, W, Cc, dt_code(x.dtype), self._stream())
        return y

    def avgpool2_bwd(self, gy: Tensor) -> Tensor:
        self._chk(gy)
        B, Ho, Wo, Cc = gy.shape
        gx = torch.empty((B, 2 * Ho, 2 * Wo, Cc), dtype=gy.dtype, device=gy.device)
        self._call("lcgan_avgpool2_bwd", gy.data_ptr(), gx.data_ptr(), B, 2 * Ho, 2 * Wo, Cc, dt_code(gy.dtype), self._stream())
        return gx

    def act_bwd_reduce(self, gy: Tensor, y: Optional[Tensor], act: int, gain: float, clog: int, want_gz: bool = True,
                       bias: Optional[Tensor] = None, bias_scale: float = 1.0, want_gbias: bool = False, want_gdq: bool = False,
                       mask: Optional[Tensor] = None, out_scale: Optional[Tensor] = None):
        """-> (gz | None, gbias [clog] | None, gdq [B, C] | None); mask: the activation's sign bits read instead of y (no gdq);
        out_scale [B, C] fp32 (needs want_gz and want_gdq): the returned gz is gz * out_scale, the reductions are of the unscaled gz"""
        self._chk(gy, y, bias, mask, out_scale)
        B, H, W, Cc = gy.shape
        gz = torch.empty_like(gy) if want_gz else None
        gbias = self._zeros.take((clog,), gy.device) if want_gbias else None
        gdq = self._zeros.take((B, Cc), gy.device) if want_gdq else None
        if out_scale is not None:
            assert want_gz and want_gdq and out_scale.shape == (B, Cc) and out_scale.dtype == torch.float32
        self._call("lcgan_act_bwd_reduce_s", gy.data_ptr(), _p(y), _p(mask if not want_gdq else None), _p(gz), _p(out_scale), _p(bias), float(bias_scale),
                   _p(gbias), _p(gdq), B, H * W, Cc, clog, act, float(gain), dt_code(gy.dtype), self._stream())
        return gz, gbias, gdq

    def scale_reduce(self, u: Tensor, x: Tensor, s: Tensor) -> Tuple[Tensor, Tensor]:
        """u <- s*u IN PLACE; returns (u, gs[b,c] = sum_p x*u_old)"""
        self._chk(u, x, s)
        B, H, W, Cc = u.shape
        gs = self._zeros.take((B, Cc), u.device)
        self._call("lcgan_scale_reduce", u.data_ptr(), x.data_ptr(), s.data_ptr(), gs.data_ptr(), B, H * W, Cc, dt_code(u.dtype),
                   self._stream())
        return u, gs

    def warp_fwd(self, x: Tensor, flow: Tensor, scale: float) -> Tensor:
        self._chk(x, flow)
        B, H, W, Cc = x.shape
        assert flow.shape == (B, H, W, 8)
        y = torch.empty_like(x)
        self._call("lcgan_warp_fwd", x.data_ptr(), flow.data_ptr(), y.data_ptr(), B, H, W, Cc, float(scale), dt_code(x.dtype),
                   self._stream())
        return y

    def warp_bwd(self, gy: Tensor, x: Tensor, flow: Tensor, scale: float) -> Tuple[Tensor, Tensor]:
        self._chk(gy, x, flow)
        B, H, W, Cc = x.shape
        npix = B * H * W
        ws_cnt = torch.empty((npix + 1,), dtype=torch.int32, device=x.device)
        ws_off = torch.empty((npix + 1,), dtype=torch.int32, device=x.device)
        ws_tiles = torch.empty(((npix + 1 + 1023) // 1024,), dtype=torch.int32, device=x.device)
        ws_ent = torch.empty((npix * 16, 2), dtype=torch.int32, device=x.device)
        gx, gflow = torch.empty_like(x), torch.empty_like(flow)
        self._call("lcgan_warp_bwd", gy.data_ptr(), x.data_ptr(), flow.data_ptr(), gx.data_ptr(), gflow.data_ptr(), ws_cnt.data_ptr(),
                   ws_off.data_ptr(), ws_tiles.data_ptr(), ws_ent.data_ptr(), B, H, W, Cc, float(scale), dt_code(x.dtype), self._stream())
        return gx, gflow

    def mbstd_fwd(self, x: Tensor, G: int, Cy: int) -> Tensor:
        self._chk(x)
        N, H, W, Cc = x.shape
        y = torch.empty((N, H, W, Cy), dtype=x.dtype, device=x.device)
        self._call("lcgan_mbstd_fwd", x.data_ptr(), y.data_ptr(), N, G, H * W, Cc, Cy, dt_code(x.dtype), self._stream())
        return y

    def mbstd_bwd(self, gy: Tensor, x: Tensor, G: int) -> Tensor:
        self._chk(gy, x)
        N, H, W, Cc = x.shape
        gx = torch.empty_like(x)
        self._call("lcgan_mbstd_bwd", gy.data_ptr(), x.data_ptr(), gx.data_ptr(), N, G, H * W, Cc, gy.shape[-1], dt_code(x.dtype),
                   self._stream())
        return gx

    def mbstd_bwd2(self, v: Tensor, gy: Tensor, x: Tensor, G: int) -> Tuple[Tensor, Tensor]:
        self._chk(v, gy, x)
        N, H, W, Cc = x.shape
        ggy, gx2 = torch.empty_like(gy), torch.empty_like(x)
        self._call("lcgan_mbstd_bwd2", v.data_ptr(), gy.data_ptr(), x.data_ptr(), ggy.data_ptr(), gx2.data_ptr(), N, G, H * W, Cc,
                   gy.shape[-1], dt_code(x.dtype), self._stream())
        return ggy, gx2

    # ---- RGB 1x1 convs (image is f32 NCHW, w is f32 [Bw,3,C]) ----------------------------------------------
    def rgb_expand(self, img: Tensor, w: Tensor, bias: Optional[Tensor], bias_scale: float, clog: int, act: int, gain: float,
                   dtype: torch.dtype, pool: bool = False):
        """pool: -> (y, avg_pool2d(y, 2)) from one pass"""
        self._chk(img, w, bias)
        B, _, H, W = img.shape
        Cc = w.shape[-1]
        y = torch.empty((B, H, W, Cc), dtype=dtype, device=img.device)
        pooled = torch.empty((B, H // 2, W // 2, Cc), dtype=dtype, device=img.device) if pool else None
        self._call("lcgan_rgb_expand", img.data_ptr(), w.data_ptr(), _p(bias), float(bias_scale), y.data_ptr(), B, H * W, Cc, clog,
                   int(w.shape[0] > 1), act, float(gain), _p(pooled), W, dt_code(dtype), self._stream())
        return (y, pooled) if pool else y

    def rgb_reduce(self, x: Tensor, w: Tensor, bias: Optional[Tensor], bias_scale: float) -> Tensor:
        self._chk(x, w, bias)
        B, H, W, Cc = x.shape
        img = torch.empty((B, 3, H, W), dtype=torch.float32, device=x.device)
        self._call("lcgan_rgb_reduce", x.data_ptr(), w.data_ptr(), _p(bias), float(bias_scale), img.data_ptr(), B, H * W, Cc,
                   int(w.shape[0] > 1), dt_code(x.dtype), self._stream())
        return img

    def rgb_wgrad(self, img: Tensor, feat: Tensor, per_sample: bool) -> Tensor:
        self._chk(img, feat)
        B, H, W, Cc = feat.shape
        gw = self._zeros.take((B if per_sample else 1, 3, Cc), feat.device)
        self._call("lcgan_rgb_wgrad", img.data_ptr(), feat.data_ptr(), gw.data_ptr(), B, H * W, Cc, int(per_sample),
                   dt_code(feat.dtype), self._stream())
        return gw

    def rgb_expand_bwd(self, gy: Tensor, y: Optional[Tensor], img: Optional[Tensor], w: Tensor, act: int, gain: float, clog: int,
                       want_gimg: bool, want_gw: bool, want_gbias: bool, fbias: Optional[Tensor] = None, fbias_scale: float = 1.0,
                       recompute: bool = False):
        """fused backward of rgb_expand -> (gimg [B,3,H,W] | None, gw [Bw,3,C] | None, gbias [clog] | None); gz = gy * act'(y) is never stored.
        recompute (leaky ReLU; img, fbias = the forward layer's bias given): the activation's sign from w . img + fbias instead of y"""
        self._chk(gy, y, img, w, fbias)
        recompute = bool(recompute and act == ACT_LRELU and img is not None)
        B, H, W, Cc = gy.shape
        per_sample = w.shape[0] > 1
        gimg = torch.empty((B, 3, H, W), dtype=torch.float32, device=gy.device) if want_gimg else None
        gw = self._zeros.take((B if per_sample else 1, 3, Cc), gy.device) if want_gw else None
        gbias = self._zeros.take((clog,), gy.device) if want_gbias else None
        self._call("lcgan_rgb_expand_bwd_r", gy.data_ptr(), _p(y), _p(img), w.data_ptr(), _p(fbias), float(fbias_scale), int(recompute),
                   _p(gimg), _p(gw), _p(gbias), B, H * W, Cc, clog, int(per_sample), act, float(gain), dt_code(gy.dtype), self._stream())
        return gimg, gw, gbias

    def rgb_reduce_bwd_act(self, gimg: Tensor, y: Tensor, wm: Tensor, bias: Optional[Tensor], bias_scale: float, act: int, gain: float,
                           clog: int, want_gbias: bool = True, want_gdq: bool = True, out_scale: Optional[Tensor] = None):
        """image gradient -> pre-activation gradient of the conv in front of rgb_reduce: (gz, gbias [clog] | None, gdq [B,C] | None, gwm [Bw,3,C]);
        out_scale [B, C] fp32: the returned gz is gz * out_scale (reductions unscaled), as in act_bwd_reduce"""
        self._chk(gimg, y, wm, bias, out_scale)
        B, H, W, Cc = y.shape
        per_sample = wm.shape[0] > 1
        gz = torch.empty_like(y)
        gbias = self._zeros.take((clog,), y.device) if want_gbias else None
        gdq = self._zeros.take((B, Cc), y.device) if want_gdq else None
        gwm = self._zeros.take((B if per_sample else 1, 3, Cc), y.device)
        assert out_scale is None or (out_scale.shape == (B, Cc) and out_scale.dtype == torch.float32)
        self._call("lcgan_rgb_reduce_bwd_act_s", gimg.data_ptr(), y.data_ptr(), wm.data_ptr(), _p(bias), float(bias_scale), gz.data_ptr(), _p(out_scale),
                   _p(gbias), _p(gdq), gwm.data_ptr(), B, H * W, Cc, clog, int(per_sample), act, float(gain), dt_code(y.dtype), self._stream())
        return gz, gbias, gdq, gwm

    # ---- flow layer (ModulatedConv2d(Cin -> 2, up 2)) as 1x1 GEMM + scatter ---------------------------------------------
    def flow_col2im(self, t: Tensor, d: Tensor, bias: Optional[Tensor]) -> Tensor:
        """t [B,H,W,24] (18 used) -> u [B,2H,2W,8] = d * col2im(t) + bias"""
        self._chk(t, d, bias)
        B, H, W, Ct = t.shape
        assert Ct == 24 and d.dtype == torch.float32
        u = torch.empty((B, 2 * H, 2 * W, 8), dtype=t.dtype, device=t.device)
        self._call("lcgan_flow_col2im", t.data_ptr(), d.data_ptr(), _p(bias), u.data_ptr(), B, H, W, d.shape[1], dt_code(t.dtype), self._stream())
        return u

    def flow_im2col(self, gu: Tensor, d: Tensor) -> Tensor:
        """gu [B,2H,2W,8] -> gt [B,H,W,24] = d * im2col(gu) (adjoint of flow_col2im)"""
        self._chk(gu, d)
        B, H2, W2, Cc = gu.shape
        assert Cc == 8 and d.dtype == torch.float32
        gt = torch.empty((B, H2 // 2, W2 // 2, 24), dtype=gu.dtype, device=gu.device)
        self._call("lcgan_flow_im2col", gu.data_ptr(), d.data_ptr(), gt.data_ptr(), B, H2 // 2, W2 // 2, d.shape[1], dt_code(gu.dtype), self._stream())
        return gt

    # ---- layout ---------------------------------------------------------------------------------------------
    def nchw_to_nhwc(self, src: Tensor, B: int, calloc: int, dtype: torch.dtype) -> Tensor:
        """src f32 [Bs,Clog,H,W] (Bs == B or 1 = broadcast) -> [B,H,W,calloc]"""
        self._chk(src)
        Bs, clog, H, W = src.shape
        dst = torch.empty((B, H, W, calloc), dtype=dtype, device=src.device)
        self._call("lcgan_nchw_to_nhwc", src.data_ptr(), dst.data_ptr(), B, H * W, calloc, clog, int(Bs == 1 and B > 1),
                   dt_code(dtype), self._stream())
        return dst

    def nhwc_to_nchw(self, src: Tensor, clog: int, reduce: bool) -> Tensor:
        self._chk(src)
        B, H, W, Cc = src.shape
        dst = torch.empty((1 if reduce else B, clog, H, W), dtype=torch.float32, device=src.device)
        self._call("lcgan_nhwc_to_nchw", src.data_ptr(), dst.data_ptr(), B, H * W, Cc, clog, int(reduce), dt_code(src.dtype),
                   self._stream())
        return dst

    # ---- small f32 linears ------------------------------------------------------------------------------------
    def linear_fwd(self, x: Tensor, w: Tensor, bias: Optional[Tensor], scale: float, bias_scale: float, act: int, gain: float):
        self._chk(x, w, bias)
        M, I = x.shape
        O = w.shape[0]
        y = torch.empty((M, O), dtype=torch.float32, device=x.device)
        self._call("lcgan_linear_fwd", x.data_ptr(), w.data_ptr(), _p(bias), y.data_ptr(), M, I, O, float(scale), float(bias_scale),
                   act, float(gain), self._stream())
        return y

    def linear_bwd_data(self, gy: Tensor, w: Tensor, scale: float) -> Tensor:
        self._chk(gy, w)
        M, O = gy.shape
        I = w.shape[1]
        gx = torch.empty((M, I), dtype=torch.float32, device=gy.device)
        self._call("lcgan_linear_bwd_data", gy.data_ptr(), w.data_ptr(), gx.data_ptr(), M, I, O, float(scale), self._stream())
        return gx

    def linear_wgrad(self, gy: Tensor, x: Tensor, scale: float) -> Tensor:
        self._chk(gy, x)
        M, O = gy.shape
        I = x.shape[1]
        gw = torch.empty((O, I), dtype=torch.float32, device=gy.device)
        self._call("lcgan_linear_wgrad", gy.data_ptr(), x.data_ptr(), gw.data_ptr(), M, I, O, float(scale), self._stream())
        return gw

    def linear_wgrad_bias(self, gy: Tensor, x: Tensor, scale: float, bias_scale: float):
        """-> (gw [O,I], gb [O]) from one launch"""
        self._chk(gy, x)
        M, O = gy.shape
        I = x.shape[1]
        gw = torch.empty((O, I), dtype=torch.float32, device=gy.device)
        gb = torch.empty((O,), dtype=torch.float32, device=gy.device)
        self._call("lcgan_linear_wgrad_bias", gy.data_ptr(), x.data_ptr(), gw.data_ptr(), gb.data_ptr(), M, I, O, float(scale), float(bias_scale),
                   self._stream())
        return gw, gb

    def colsum(self, gy: Tensor, scale: float) -> Tensor:
        self._chk(gy)
        M, O = gy.shape
        gb = torch.empty((O,), dtype=torch.float32, device=gy.device)
        self._call("lcgan_colsum", gy.data_ptr(), gb.data_ptr(), M, O, float(scale), self._stream())
        return gb

    @staticmethod
    def _ptr_array(ts):
        import ctypes as C
        return (C.c_void_p * len(ts))(*[None if t is None else t.data_ptr() for t in ts])

    def linear_group_fwd(self, x: Tensor, ws, biases, scales, bias_scales, act: int = ACT_NONE, gain: float = 1.0):
        """L linear layers sharing x [M,I]: -> [y_l [M,O_l]] from one launch"""
        import ctypes as C
        self._chk(x, *ws, *biases)
        M, I = x.shape
        L = len(ws)
        Os = [int(w.shape[0]) for w in ws]
        flat = torch.empty((M * sum(Os),), dtype=torch.float32, device=x.device)
        ys, off = [], 0
        for O in Os:
            ys.append(flat[off:off + M * O].view(M, O))
            off += M * O
        self._call("lcgan_linear_group_fwd", x.data_ptr(), self._ptr_array(ws), self._ptr_array(biases), self._ptr_array(ys),
                   (C.c_int * L)(*Os), (C.c_float * L)(*scales), (C.c_float * L)(*bias_scales), L, M, I, act, float(gain),
                   self._stream())
        return ys

    def linear_group_bwd(self, gys, x: Tensor, ws, scales, bias_scales, want_gx: bool = True):
        """-> (gx [M,I] | None, [gw_l [O_l,I]], [gb_l [O_l]])"""
        import ctypes as C
        self._chk(x, *gys, *ws)
        M, I = x.shape
        L = len(ws)
        Os = [int(w.shape[0]) for w in ws]
        gx = torch.empty((M, I), dtype=torch.float32, device=x.device) if want_gx else None
        flat = torch.empty((sum(Os) * (I + 1),), dtype=torch.float32, device=x.device)
        gws, gbs, off = [], [], 0
        for O in Os:
            gws.append(flat[off:off + O * I].view(O, I))
            off += O * I
        for O in Os:
            gbs.append(flat[off:off + O])
            off += O
        self._call("lcgan_linear_group_bwd", self._ptr_array(gys), x.data_ptr(), self._ptr_array(ws), (C.c_int * L)(*Os),
                   (C.c_float * L)(*scales), (C.c_float * L)(*bias_scales), L, M, I, _p(gx), self._ptr_array(gws),
                   self._ptr_array(gbs), self._stream())
        return gx, gws, gbs

    def linear_multi_fwd(self, xs, ws, biases, scales, bias_scales, act: int = ACT_NONE, gain: float = 1.0):
        """L linear layers with their own inputs: [y_l = act(scale_l x_l w_l^T + bias_l bias_scale_l) gain] from one launch"""
        import ctypes as C
        self._chk(*xs, *ws, *biases)
        M, L = xs[0].shape[0], len(ws)
        Is, Os = [int(w.shape[1]) for w in ws], [int(w.shape[0]) for w in ws]
        flat = torch.empty((M * sum(Os),), dtype=torch.float32, device=xs[0].device)
        ys, off = [], 0
        for O in Os:
            ys.append(flat[off:off + M * O].view(M, O))
            off += M * O
        self._call("lcgan_linear_multi_fwd", self._ptr_array(xs), self._ptr_array(ws), self._ptr_array(biases), self._ptr_array(ys),
                   (C.c_int * L)(*Is), (C.c_int * L)(*Os), (C.c_float * L)(*scales), (C.c_float * L)(*bias_scales), L, M, act, float(gain),
                   self._stream())
        return ys

    def linear_multi_bwd(self, gys, xs, ws, scales, bias_scales, want_gx=True):
        """-> ([gx_l [M,I_l]] | None, [gw_l [O_l,I_l]], [gb_l [O_l]]) from two launches"""
        import ctypes as C
        self._chk(*gys, *xs, *ws)
        M, L = xs[0].shape[0], len(ws)
        dev = xs[0].device
        Is, Os = [int(w.shape[1]) for w in ws], [int(w.shape[0]) for w in ws]
        flat = torch.empty((sum(M * I for I in Is) * int(want_gx) + sum(O * I + O for O, I in zip(Os, Is)),), dtype=torch.float32, device=dev)
        gxs, gws, gbs, off = ([] if want_gx else None), [], [], 0
        if want_gx:
            for I in Is:
                gxs.append(flat[off:off + M * I].view(M, I)); off += M * I
        for O, I in zip(Os, Is):
            gws.append(flat[off:off + O * I].view(O, I)); off += O * I
        for O in Os:
            gbs.append(flat[off:off + O]); off += O
        self._call("lcgan_linear_multi_bwd", self._ptr_array(gys), self._ptr_array(xs), self._ptr_array(ws), (C.c_int * L)(*Is), (C.c_int * L)(*Os),
                   (C.c_float * L)(*scales), (C.c_float * L)(*bias_scales), L, M, self._ptr_array(gxs) if want_gx else None,
                   self._ptr_array(gws), self._ptr_array(gbs), self._stream())
        return gxs, gws, gbs

    def act_bwd_f32(self, gy: Tensor, y: Tensor, act: int, gain: float) -> Tensor:
        self._chk(gy, y)
        gz = torch.empty_like(gy)
        self._call("lcgan_act_bwd_f32", gy.data_ptr(), y.data_ptr(), gz.data_ptr(), gy.numel(), act, float(gain), self._stream())
        return gz

    def demod_fwd(self, s: Tensor, wsq: Tensor, ostride: int, eps: float = 1e-8) -> Tensor:
        self._chk(s, wsq)
        B, Cc = s.shape
        O = wsq.shape[0]
        d = self._zeros.take((B, ostride), s.device)
        self._call("lcgan_demod_fwd", s.data_ptr(), wsq.data_ptr(), d.data_ptr(), B, Cc, O, ostride, float(eps), self._stream())
        return d

    def demod_group(self, ss, wsqs, ostrides, eps: float = 1e-8):
        """[demod_fwd(s_l, wsq_l, ostride_l)] for up to 24 layers from one launch"""
        import ctypes as C
        self._chk(*ss, *wsqs)
        B = ss[0].shape[0]
        L = len(ss)
        ds = [self._zeros.take((B, int(o)), ss[0].device) for o in ostrides]
        self._call("lcgan_demod_group", self._ptr_array(ss), self._ptr_array(wsqs), self._ptr_array(ds), (C.c_int * L)(*[int(t.shape[1]) for t in ss]),
                   (C.c_int * L)(*[int(w.shape[0]) for w in wsqs]), (C.c_int * L)(*[int(o) for o in ostrides]), L, B, float(eps), self._stream())
        return ds

    def demod_bwd(self, gdq: Tensor, d: Tensor, s: Tensor, wsq: Tensor, gs: Tensor) -> Tensor:
        """gs += demod path (in place); returns gwsq [O,C]"""
        self._chk(gdq, d, s, wsq, gs)
        B, Cc = s.shape
        O = wsq.shape[0]
        gwsq = torch.empty_like(wsq)
        self._call("lcgan_demod_bwd", gdq.data_ptr(), d.data_ptr(), s.data_ptr(), wsq.data_ptr(), gs.data_ptr(), gwsq.data_ptr(),
                   B, Cc, O, d.shape[1], self._stream())
        return gwsq

    # ---- losses -------------------------------------------------------------------------------------------------
    def bce_fwd(self, logit: Tensor, target_one: bool) -> Tensor:
        self._chk(logit)
        out = torch.empty((), dtype=torch.float32, device=logit.device)
        self._call("lcgan_bce_fwd", logit.data_ptr(), logit.numel(), int(target_one), out.data_ptr(), self._stream())
        return out

    def bce_bwd(self, logit: Tensor, target_one: bool, gout: Tensor) -> Tensor:
        self._chk(logit, gout)
        g = torch.empty_like(logit)
        self._call("lcgan_bce_bwd", logit.data_ptr(), logit.numel(), int(target_one), gout.data_ptr(), g.data_ptr(), self._stream())
        return g

    def contrastive_fwd(self, a: Tensor, p: Tensor, n: Tensor, tau: float):
        self._chk(a, p, n)
        B, Dd = a.shape
        t = torch.empty((B,), dtype=torch.float32, device=a.device)
        out = torch.empty((), dtype=torch.float32, device=a.device)
        self._call("lcgan_contrastive_fwd", a.data_ptr(), p.data_ptr(), n.data_ptr(), B, Dd, float(tau), t.data_ptr(), out.data_ptr(),
                   self._stream())
        return out, t

    def contrastive_bwd(self, a: Tensor, p: Tensor, n: Tensor, t: Tensor, gout: Tensor, tau: float):
        self._chk(a, p, n, t, gout)
        B, Dd = a.shape
        ga, gp, gn = torch.empty_like(a), torch.empty_like(a), torch.empty_like(a)
        self._call("lcgan_contrastive_bwd", a.data_ptr(), p.data_ptr(), n.data_ptr(), t.data_ptr(), gout.data_ptr(), B, Dd, float(tau),
                   ga.data_ptr(), gp.data_ptr(), gn.data_ptr(), self._stream())
        return ga, gp, gn

    def l2norm_fwd(self, x: Tensor, eps: float = 1e-12):
        self._chk(x)
        B, Dd = x.shape
        y = torch.empty_like(x)
        ns = torch.empty((B,), dtype=torch.float32, device=x.device)
        self._call("lcgan_l2norm_fwd", x.data_ptr(), y.data_ptr(), ns.data_ptr(), B, Dd, float(eps), self._stream())
        return y, ns

    def l2norm_bwd(self, gy: Tensor, y: Tensor, ns: Tensor) -> Tensor:
        self._chk(gy, y, ns)
        B, Dd = y.shape
        gx = torch.empty_like(y)
        self._call("lcgan_l2norm_bwd", gy.data_ptr(), y.data_ptr(), ns.data_ptr(), gx.data_ptr(), B, Dd, self._stream())
        return gx

    def powsum(self, x: Tensor, pw: int, coef: float) -> Tensor:
        self._chk(x)
        out = self._zeros.take((), x.device)
        self._call("lcgan_powsum", x.data_ptr(), x.numel(), pw, float(coef), out.data_ptr(), self._stream())
        return out

    def powsum_bwd(self, x: Tensor, pw: int, coef: float, gout: Tensor) -> Tensor:
        self._chk(x, gout)
        g = torch.empty_like(x)
        self._call("lcgan_powsum_bwd", x.data_ptr(), x.numel(), pw, float(coef), gout.data_ptr(), g.data_ptr(), self._stream())
        return g

    def qr(self, A: Tensor) -> Tuple[Tensor, Tensor]:
        """(Q, R) of the reduced Householder QR of square matrices [n,n] or [nb,n,n] (n <= 64), LAPACK sign convention"""
        self._chk(A)
        n = A.shape[-1]
        assert A.shape[-2] == n and n <= 64 and A.dtype == torch.float32 and A.dim() in (2, 3)
        Q, R = torch.empty_like(A), torch.empty_like(A)
        self._call("lcgan_qr_householder", A.data_ptr(), Q.data_ptr(), R.data_ptr(), A.numel() // (n * n), n, self._stream())
        return Q, R

    def avg_latent(self, w: Tensor, avg: Tensor, beta: float) -> None:
        self._chk(w, avg)
        B, Dd = w.shape
        self._call("lcgan_avg_latent", w.data_ptr(), avg.data_ptr(), B, Dd, float(beta), self._stream())

    # ---- training views ---------------------------------------------------------------------------------------------
    def make_views(self, src: Tensor, params: Tensor):
        """src f32 [B,3,R,R] in [-1,1], params f32 [B,32] (lcgan_amd.data.sample_view_params) -> (image, geometry_change, appearance_change)"""
        self._chk(src, params)
        B, _, R, _ = src.shape
        assert src.dtype == torch.float32 and params.dtype == torch.float32 and params.shape == (B, 32) and src.shape[1] == 3
        outs = [torch.empty_like(src) for _ in range(3)]
        self._call("lcgan_make_views", src.data_ptr(), params.data_ptr(), outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(),
                   B, R, self._stream())
        return tuple(outs)

    # ---- multi-tensor -----------------------------------------------------------------------------------------------
    def multi_tensor(self, table, op: int, a0: float, a1: float = 0.0, a2: float = 0.0) -> None:
        """table: lcgan_amd.optim.TensorTable (device descriptor + chunk arrays)."""
        self._call("lcgan_multi_tensor", table.descs.data_ptr(), table.chunk_tensor.data_ptr(), table.chunk_index.data_ptr(),
                   table.n_chunks, op, float(a0), float(a1), float(a2), float(table.total), self._stream())

    # ---- profiling --------------------------------------------------------------------------------------------------
    def prof_enable(self, on: bool) -> None:
        self.lib.lcgan_prof_enable(int(on))

    def prof_dump(self, path: str) -> None:
        """per-launch CSV (kid, ms, flops, bytes, tag) of everything recorded since prof_enable(True); clears the records"""
        _lib.check(self.lib.lcgan_prof_dump(path.encode()), "lcgan_prof_dump")

    def prof_collect(self):
        import ctypes as C
        n = 13
        ms, fl, by = (C.c_double * n)(), (C.c_double * n)(), (C.c_double * n)()
        cnt = (C.c_longlong * n)()
        k = self.lib.lcgan_prof_collect(ms, fl, by, cnt)
        names = ["conv_igemm", "conv_wgrad", "weight_prep", "stencil", "act_bwd", "warp_fwd", "warp_bwd", "rgb", "linear", "small",
                 "optim", "layout", "scale_reduce"]
        return {names[i]: {"ms": ms[i], "flops": fl[i], "bytes": by[i], "count": cnt[i]} for i in range(k)}


class _Lazy:
    """Resolves to HipKernels on first use so that importing lcgan_amd on a CPU box (build check, CPU tests) works,
    while any attempt to RUN the product without the HIP library raises."""
    _impl = None

    def __getattr__(self, item):
        if _Lazy._impl is None:
            _Lazy._impl = HipKernels()
        return getattr(_Lazy._impl, item)


K = _Lazy()


def backend_name() -> str:
    return getattr(K, "name")

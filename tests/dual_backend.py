"""A kernel backend that runs EVERY call on both the HIP library and the CPU emulation (oracle/hip_emulation.py) with the HIP
call's inputs, returns the HIP result and records, per call, (a) the output difference and (b) for every leaky-ReLU output the
positions where the two sides land on different sides of the kink -- TEST INFRASTRUCTURE (imports oracle/).

Why (b): the HIP path sums in another order than the reference, so a pre-activation within ~1e-7 of zero may come out with the
other sign; the activation mask of that element flips and single entries of downstream gradients move by percents although both
sides are fp32-correct.  The end-to-end gradient tests accept a tensor above the 1e-3 tolerance ONLY when this recorder shows such a
flip (count, |pre-activation| at the flip) -- see tests/helpers.py:check_grads_vs_golden_kink_tolerant.
"""
import torch

import lcgan_amd.kernels as KM
from lcgan_amd.kernels import ACT_LRELU, HipKernels, PreparedWeight
from oracle.hip_emulation import EmulatedKernels
from tests.helpers import install_backend

# position of the `act` argument of the calls that apply an activation to their output
_ACT_ARG = {"conv_fwd": ("act", None), "conv_bwd_data": ("act", None), "rgb_expand": ("act", 5), "linear_fwd": ("act", 5),
            "box3_act": ("act", 1), "linear_group_fwd": ("act", 5)}


class DualBackend:
    name = "hip"

    def __init__(self):
        self.H, self.E = HipKernels(), EmulatedKernels()
        self.wmap, self.keep, self.n = {}, [], 0
        self.diffs = []          # (max-rel, l2-rel, "#n call[i]", shape)
        self.flips = []          # (count, numel, max |y_emu| at a flip / max |y_emu|, "#n call")

    def _cpu(self, a):
        if isinstance(a, torch.Tensor):
            return a.detach().cpu().clone()
        if isinstance(a, PreparedWeight):
            return self.wmap[id(a)]
        if isinstance(a, (list, tuple)):
            return type(a)(self._cpu(v) for v in a)
        return a

    def _cmp(self, tag, h, e):
        hf, ef = h.detach().float().cpu(), e.detach().float()
        sc = ef.abs().max().clamp_min(1e-30)
        self.diffs.append((float((hf - ef).abs().max() / sc), float((hf - ef).norm() / ef.norm().clamp_min(1e-30)), tag, tuple(h.shape)))
        return hf, ef

    def __getattr__(self, item):
        hf, ef = getattr(self.H, item), getattr(self.E, item)
        if not callable(hf):
            return hf

        def call(*args, **kw):
            self.n += 1
            if item in ("multi_tensor", "prof_enable", "prof_collect"):      # in-place on device tables / no tensor result: HIP only
                return hf(*args, **kw)
            cargs, ckw = [self._cpu(a) for a in args], {k: self._cpu(v) for k, v in kw.items()}
            out_h, out_e = hf(*args, **kw), ef(*cargs, **ckw)
            if item in ("prep_weight", "prep_weight_group"):
                pairs = [(out_h, out_e)] if item == "prep_weight" else list(zip(out_h, out_e))
                for (ph, wh), (pe, we) in pairs:
                    self.wmap[id(ph)] = pe
                    self.keep.append(ph)
                    if wh is not None:
                        self._cmp(f"#{self.n} {item}.wsq", wh, we)
                return out_h
            outs_h = out_h if isinstance(out_h, (tuple, list)) else (out_h,)
            outs_e = out_e if isinstance(out_e, (tuple, list)) else (out_e,)
            act = None
            if item in _ACT_ARG:
                name, pos = _ACT_ARG[item]
                act = kw.get(name, args[pos] if (pos is not None and len(args) > pos) else 0)
            for i, (a, b) in enumerate(zip(outs_h, outs_e)):
                if not isinstance(a, torch.Tensor):
                    continue
                h32, e32 = self._cmp(f"#{self.n} {item}[{i}]", a, b)
                if act == ACT_LRELU and i == 0:
                    bad = (h32 > 0) != (e32 > 0)
                    if bool(bad.any()):
                        self.flips.append((int(bad.sum()), bad.numel(), float(e32[bad].abs().max() / e32.abs().max().clamp_min(1e-30)),
                                           f"#{self.n} {item}{tuple(a.shape)}"))
            if item == "demod_bwd":          # gs (args[4]) is updated in place
                self._cmp(f"#{self.n} {item}.gs", args[4], cargs[4])
            if item == "avg_latent":
                self._cmp(f"#{self.n} {item}.avg", args[1], cargs[1])
            return out_h
        return call

    # ---- reporting -------------------------------------------------------------------------------------------------
    def flip_report(self):
        return [f"{c}/{n} sign flips, |pre-act| <= {m:.1e} of the tensor's max, in {tag}" for c, n, m, tag in self.flips]

    def worst_calls(self, k=5):
        return [f"{mx:.2e} (l2 {l2:.2e}) {tag} {shape}" for mx, l2, tag, shape in sorted(self.diffs, reverse=True)[:k]]


class MaskRecorder:
    """Wraps ONE backend and keeps the sign pattern (and, for the CPU side, the values) of every leaky-ReLU output, keyed by the
    call's sequence number.  Two recordings of the same step -- the HIP chain on the GPU and the CPU emulation chain, each fed by its
    OWN earlier outputs -- are then compared call by call (compare_masks): unlike DualBackend's per-call comparison on identical
    inputs, this also sees a pre-activation that the accumulated upstream difference (~1e-6) carried across zero.  The emulation
    chain reproduces the reference within the strict 1e-3 criterion on every tensor (tests/test_wiring_cpu.py), so it stands in
    for the reference's masks."""
    name = "hip"

    def __init__(self, inner, keep_values=False):
        self.inner, self.keep_values, self.n, self.masks = inner, keep_values, 0, []

    def __getattr__(self, item):
        f = getattr(self.inner, item)
        if not callable(f) or item not in _ACT_ARG:
            return f

        def call(*args, **kw):
            out = f(*args, **kw)
            name, pos = _ACT_ARG[item]
            act = kw.get(name, args[pos] if (pos is not None and len(args) > pos) else 0)
            if act == ACT_LRELU:
                y = (out[0] if isinstance(out, (tuple, list)) else out).detach().float().cpu()
                self.masks.append((f"#{len(self.masks)} {item}{tuple(y.shape)}", y > 0, y if self.keep_values else None))
            return out
        return call


def compare_masks(hip: "MaskRecorder", cpu: "MaskRecorder"):
    """-> [(count, numel, max |y_cpu| at a flip / max |y_cpu|, tag)] over the leaky-ReLU outputs where the two chains disagree"""
    assert len(hip.masks) == len(cpu.masks), (len(hip.masks), len(cpu.masks))
    flips = []
    for (tag, mh, _), (tag_c, mc, yc) in zip(hip.masks, cpu.masks):
        assert mh.shape == mc.shape, (tag, tag_c)
        bad = mh != mc
        if bool(bad.any()):
            flips.append((int(bad.sum()), bad.numel(), float(yc[bad].abs().max() / yc.abs().max().clamp_min(1e-30)), tag))
    return flips


class dual_backend:
    """with dual_backend() as rec: ... run a step ...; rec.flips / rec.diffs"""

    def __enter__(self):
        self.rec = DualBackend()
        install_backend(self.rec)
        return self.rec

    def __exit__(self, *exc):
        install_backend(None)

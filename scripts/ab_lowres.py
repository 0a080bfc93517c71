"""A/B of two builds on the low-resolution convolution launches (the generic split-K kernel's territory): forward, data gradient with the
half-resolution residual, at batch 32 and at the local batch of 8 ranks.   python scripts/ab_lowres.py libA.so libB.so"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scripts.ab_conv import kernels_for
Ks = [kernels_for(p) for p in sys.argv[1:3]]


def t(fn, n=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tot = [0.0, 0.0]
for B in (32, 8):
    for Hh in (16, 8, 4):
        C = 512
        x = torch.randn(B, Hh, Hh, C, device="cuda").bfloat16()
        r = torch.randn(B, Hh // 2, Hh // 2, C, device="cuda").bfloat16()
        w = torch.randn(C, C, 3, 3, device="cuda")
        res = {}
        for i, K in enumerate(Ks):
            pw, _ = K.prep_weight(w, 1 / math.sqrt(C * 9), False, False)
            pwT, _ = K.prep_weight(w, 1 / math.sqrt(C * 9), True, False)
            res[i] = (K, pw, pwT, [], [], [])
        for rep in range(3):
            for i in (0, 1):
                K, pw, pwT, f, d, d2 = res[i]
                f.append(t(lambda: K.conv_fwd(x, pw, C, 3, 1, act=1, gain=1.4)))
                d.append(t(lambda: K.conv_bwd_data(x, pwT, C, 3, 1, residual=r, residual_half=True)))
                d2.append(t(lambda: K.conv_bwd_data(x, pwT, C, 3, 1)))
        same = torch.equal(res[0][0].conv_fwd(x, res[0][1], C, 3, 1, act=1, gain=1.4), res[1][0].conv_fwd(x, res[1][1], C, 3, 1, act=1, gain=1.4))
        line = f"B{B} {Hh}^2 C{C}:"
        for name, k in (("fwd", 3), ("dgrad+res/2", 4), ("dgrad", 5)):
            a, b = min(res[0][k]), min(res[1][k])
            tot[0] += a; tot[1] += b
            line += f"  {name} {a:6.1f} -> {b:6.1f} us ({(b / a - 1) * 100:+5.1f} %)"
        print(line + f"   fwd identical: {same}", flush=True)
print(f"sum: {tot[0]:.0f} -> {tot[1]:.0f} us ({(tot[1] / tot[0] - 1) * 100:+.1f} %)")

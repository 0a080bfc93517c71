"""A/B of two builds on the stencil / pooling kernels at the generator's and discriminator's big shapes.  python scripts/ab_stencil.py libA.so libB.so"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scripts.ab_conv import kernels_for
Ks = [kernels_for(p) for p in sys.argv[1:3]]
B = 32


def t(fn, n=8):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tot = [0.0, 0.0]
for (H, C) in ((256, 128), (128, 256), (64, 512)):
    x = torch.randn(B, H, H, C, device="cuda").bfloat16(); y = torch.randn(B, H, H, C, device="cuda").bfloat16()
    xl = torch.randn(B, H // 2, H // 2, C, device="cuda").bfloat16()
    cases = {
        "box3_act": lambda K: K.box3_act(x, 1, 1.4),
        "box3_act_bwd": lambda K: K.box3_act_bwd(x, y, 1, 1.4),
        "box3_actbwd": lambda K: K.box3_actbwd(x, y, 1, 1.4, C, True),
        "up2box": lambda K: K.up2box(xl, x),
        "up2box_bwd": lambda K: K.up2box_bwd(x),
        "avgpool2": lambda K: K.avgpool2(x),
        "avgpool2_bwd": lambda K: K.avgpool2_bwd(xl),
    }
    for name, fn in cases.items():
        r = [[], []]
        same = True
        for rep in range(3):
            for i, K in enumerate(Ks):
                r[i].append(t(lambda: fn(K)))
        oa, ob = fn(Ks[0]), fn(Ks[1])
        oa = oa[0] if isinstance(oa, tuple) else oa; ob = ob[0] if isinstance(ob, tuple) else ob
        same = torch.equal(oa, ob)
        a, b = min(r[0]), min(r[1])
        tot[0] += a; tot[1] += b
        print(f"{H}^2 C{C} {name:13s}: {a:7.1f} -> {b:7.1f} us ({(b / a - 1) * 100:+5.1f} %)  identical: {same}", flush=True)
print(f"sum {tot[0]:.0f} -> {tot[1]:.0f} us ({(tot[1] / tot[0] - 1) * 100:+.1f} %)")

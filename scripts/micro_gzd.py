"""What would the modulated backward launches take if the activation backward handed them d * gz (no per-sample input scale)?
Times the data-gradient (+ style-gradient epilogue) and weight-gradient launches of the generator's modulated layers with and
without `pre`.   python scripts/micro_gzd.py [lib.so]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scripts.ab_conv import kernels_for
K = kernels_for(sys.argv[1] if len(sys.argv) > 1 else "lcgan_amd/liblcgan_hip.so")
B = 32


def t(fn, n=5):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tot = [0.0, 0.0]
# (resolution of gz, O, Cin, up, launches per iteration)
for (H, O, Ci, up, n) in ((256, 128, 128, 1, 2), (128, 256, 256, 1, 1), (64, 512, 512, 1, 1), (32, 512, 512, 1, 1), (16, 512, 512, 1, 1),
                          (256, 128, 256, 2, 1), (128, 256, 512, 2, 1), (64, 512, 512, 2, 1), (32, 512, 512, 2, 1)):
    gz = torch.randn(B, H, H, O, device="cuda").bfloat16()
    Hx = H // up
    x = torch.randn(B, Hx, Hx, Ci, device="cuda").bfloat16()
    w = torch.randn(O, Ci, 3, 3, device="cuda")
    c_eq = 1 / math.sqrt(Ci * 9)
    pwT, _ = K.prep_weight(w, c_eq, True, False)
    d = torch.rand(B, O, device="cuda") + 0.5; s = torch.rand(B, Ci, device="cuda") + 0.5
    for j, pre in enumerate((d, None)):
        if up == 2:
            td = t(lambda: K.conv_fwd(gz, pwT, Ci, 3, 2, pre=pre, post=s, xs=x))
            tw = t(lambda: K.conv_wgrad(gz, x, Ci, O, 3, 2, pre_x=pre, pre_g=s))
        else:
            td = t(lambda: K.conv_bwd_data(gz, pwT, Ci, 3, 1, pre=pre, post=s, xs=x))
            tw = t(lambda: K.conv_wgrad(x, gz, O, Ci, 3, 1, pre_x=s, pre_g=pre))
        tot[j] += n * (td + tw)
        print(f"gz {H}^2 O{O} Cin{Ci} up{up} x{n}: {'pre = d ' if j == 0 else 'pre-free'}  data gradient {td:7.1f} us   weight gradient {tw:7.1f} us", flush=True)
print(f"per iteration: {tot[0] / 1e3:.2f} ms with d as a per-sample scale, {tot[1] / 1e3:.2f} ms without")

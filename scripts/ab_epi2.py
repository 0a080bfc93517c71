"""A/B of the data gradient with a half-resolution residual (EPI 2): python scripts/ab_epi2.py libA.so libB.so"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scripts.ab_conv import kernels_for
Ks = [kernels_for(p) for p in sys.argv[1:]]
B = 32
def timeit(fn, n):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (Hh, C) in [(256, 128), (128, 256), (64, 512)]:
    g = torch.randn(B, Hh, Hh, C, device="cuda").bfloat16()
    rh = torch.randn(B, Hh // 2, Hh // 2, C, device="cuda").bfloat16()
    w = torch.randn(C, C, 3, 3, device="cuda")
    outs, line = [], f"{Hh}^2 {C}: "
    for K in Ks:
        pw = K.prep_weight(w, 1 / math.sqrt(C * 9), True, False)[0]
        t0 = min(timeit(lambda: K.conv_bwd_data(g, pw, C, 3, 1), 6) for _ in range(3))
        t1 = min(timeit(lambda: K.conv_bwd_data(g, pw, C, 3, 1, residual=rh, residual_half=True), 6) for _ in range(3))
        outs.append(K.conv_bwd_data(g, pw, C, 3, 1, residual=rh, residual_half=True).float())
        line += f" plain {t0:6.1f} +res/2 {t1:6.1f} |"
    print(line, "maxdiff", float((outs[0] - outs[-1]).abs().max()), flush=True)

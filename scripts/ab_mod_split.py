"""Where the modulated forward's extra time goes: plain vs pre only vs post only vs both (one library, interleaved)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lcgan_amd.kernels import HipKernels
K = HipKernels()
B = 32
def timeit(fn, n=6):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (Hh, C) in [(256, 128), (128, 256), (64, 512)]:
    x = torch.randn(B, Hh, Hh, C, device="cuda").bfloat16()
    w = torch.randn(C, C, 3, 3, device="cuda")
    pw = K.prep_weight(w, 1 / math.sqrt(C * 9), False, False)[0]
    bias = torch.randn(C, device="cuda")
    pre, post = torch.rand(B, C, device="cuda") + 0.5, torch.rand(B, C, device="cuda") + 0.5
    cfgs = {"plain": {}, "pre": dict(pre=pre), "post": dict(post=post), "both": dict(pre=pre, post=post)}
    res = {k: [] for k in cfgs}
    for rnd in range(3):
        for k, kw in cfgs.items():
            res[k].append(timeit(lambda: K.conv_fwd(x, pw, C, 3, 1, bias=bias, act=1, gain=1.4, **kw)))
    print(f"{Hh}^2 {C}: " + "  ".join(f"{k} {min(v):6.1f}" for k, v in res.items()), flush=True)

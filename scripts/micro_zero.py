"""GPU micro-benchmark: random vs all-zero operands (clock / power effect) for the conv forward and the weight gradient."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lcgan_amd.kernels import HipKernels
H = HipKernels()
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
B = 32
for (Hh, Ci, Co) in [(128, 256, 256), (64, 512, 512), (256, 128, 128)]:
    fl = 2.0 * B * Hh * Hh * Ci * Co * 9
    for zero in (False, True, False, True):
        mk = torch.zeros if zero else torch.randn
        x = mk(B, Hh, Hh, Ci, device="cuda").bfloat16(); g = mk(B, Hh, Hh, Co, device="cuda").bfloat16()
        w = mk(Co, Ci, 3, 3, device="cuda")
        pw, _ = H.prep_weight(w, 1 / math.sqrt(Ci * 9), False, False)
        tf = timeit(lambda: H.conv_fwd(x, pw, Co, 3, 1)); tw = timeit(lambda: H.conv_wgrad(x, g, Co, Ci, 3, 1))
        print(f"{Hh}^2 {Ci}->{Co} zero={int(zero)}: fwd {fl/tf/1e9:7.0f} TF/s  wgrad {fl/tw/1e9:7.0f} TF/s", flush=True)

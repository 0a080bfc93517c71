"""GPU micro-benchmark: row-segment wgrad kernel on the big layer shapes, sweeping the split-K workgroup target."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lcgan_amd.kernels import HipKernels
H = HipKernels()
shapes = [(32, 256, 256, 128, 128, 1), (32, 128, 128, 256, 256, 1), (32, 64, 64, 512, 512, 1), (32, 256, 256, 128, 256, 2), (32, 256, 256, 128, 128, 2)]
def bench(fn, n=5):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n
for (B, Hh, W, Ci, Co, st) in shapes:
    x = torch.randn(B, Hh, W, Ci, device="cuda").bfloat16()
    g = torch.randn(B, Hh // st, W // st, Co, device="cuda").bfloat16()
    fl = 2.0 * B * (Hh // st) * (W // st) * Ci * Co * 9
    line = f"{(B,Hh,W,Ci,Co,st)}: "
    for wgs in (512, 1024, 1536, 3072, 6144):
        H.lib.lcgan_set_option(2, wgs)
        for na in (0,):
            dt = bench(lambda: H.conv_wgrad(x, g, Co, Ci, 3, st))
            line += f" wgs{wgs}{'-noatom' if na else ''}={fl/dt/1e12:.0f}TF"
    H.lib.lcgan_set_option(3, 0); H.lib.lcgan_set_option(2, 0)
    print(line, flush=True)

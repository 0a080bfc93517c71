"""GPU micro-benchmark: halo conv kernel on the big layer shapes, MFMA 32x32x16 vs 16x16x32 (interleaved rounds, one process)."""
import sys, os, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lcgan_amd.kernels import HipKernels
H = HipKernels()
shapes = [(32, 256, 256, 128, 128, 1), (32, 128, 128, 256, 256, 1), (32, 64, 64, 512, 512, 1), (32, 256, 256, 128, 256, 2)]
def bench(fn, n=5):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n
for (B, Hh, W, Ci, Co, st) in shapes:
    x = torch.randn(B, Hh, W, Ci, device="cuda").bfloat16()
    w = torch.randn(Co, Ci, 3, 3, device="cuda")
    pw, _ = H.prep_weight(w, 1 / math.sqrt(Ci * 9), False, False)
    fl = 2.0 * B * (Hh // st) * (W // st) * Ci * Co * 9
    res = {0: [], 1: []}
    for rnd in range(3):
        for m16 in (0, 1):
            H.lib.lcgan_set_option(4, m16)
            res[m16].append(fl / bench(lambda: H.conv_fwd(x, pw, Co, 3, st)) / 1e12)
    print((B, Hh, W, Ci, Co, st), "32x32x16:", " ".join(f"{v:.0f}" for v in res[0]), "TF   16x16x32:", " ".join(f"{v:.0f}" for v in res[1]), "TF", flush=True)
H.lib.lcgan_set_option(4, 1)

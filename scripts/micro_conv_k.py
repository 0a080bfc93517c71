"""GPU micro-benchmark: fixed vs per-step cost of the halo conv kernel -- the same 256x256 x 128-output tile grid with
growing reduction depth (Cin), plus 1x1 kernels (1 tap) and random vs zero data (clock effects)."""
import sys, os, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lcgan_amd.kernels import HipKernels
H = HipKernels()
def bench(fn, n=5):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n
B, Hh, W = 16, 256, 256
dbg = int(os.environ.get("DBG", "0"))
H.lib.lcgan_set_option(3, dbg)
for k in (3,):
    for Ci in (128, 256, 512):
        for zero in (False,):
            x = (torch.zeros if zero else torch.randn)(B, Hh, W, Ci, device="cuda").bfloat16()
            w = (torch.zeros if zero else torch.randn)(128, Ci, k, k, device="cuda")
            pw, _ = H.prep_weight(w, 1 / math.sqrt(Ci * k * k), False, False)
            fl = 2.0 * B * Hh * W * Ci * 128 * k * k
            t = bench(lambda: H.conv_fwd(x, pw, 128, k, 1))
            steps = k * k * (Ci // 32)
            wgs = B * 256
            print(f"k={k} Cin={Ci:4d} zero={int(zero)} steps={steps:4d}: {t*1e6:8.1f} us  {fl/t/1e12:7.0f} TF/s   per-WG {t*1e6/(wgs/512):6.2f} us  per-step {t*1e6/(wgs/512)/steps:5.3f} us", flush=True)

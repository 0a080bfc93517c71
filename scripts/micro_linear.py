import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lcgan_amd.kernels import HipKernels
H = HipKernels()
def bench(fn, n=50):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e6
for (M, I, O) in ((32, 64, 64), (32, 512, 512), (32, 512, 128), (32, 64, 512), (32, 8192, 512), (32, 512, 1), (96, 512, 512)):
    x, w, b, gy = torch.randn(M, I, device="cuda"), torch.randn(O, I, device="cuda"), torch.randn(O, device="cuda"), torch.randn(M, O, device="cuda")
    f = bench(lambda: H.linear_fwd(x, w, b, 0.1, 1.0, 0, 1.0))
    bd = bench(lambda: H.linear_bwd_data(gy, w, 0.1))
    wg = bench(lambda: H.linear_wgrad(gy, x, 0.1))
    emp = bench(lambda: torch.empty((M, O), device="cuda"))
    print(f"M{M} I{I} O{O}: fwd {f:.1f} us  bwd_data {bd:.1f} us  wgrad {wg:.1f} us   (torch.empty alone {emp:.1f} us)", flush=True)

"""GPU A/B: box filter + activation kernel between two builds (one process, interleaved)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scripts.ab_conv import kernels_for
A, Bk = kernels_for(sys.argv[1]), kernels_for(sys.argv[2])
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (R, C) in [(256, 128), (128, 256), (64, 512)]:
    x = torch.randn(32, R, R, C, device="cuda").bfloat16()
    d = float((A.box3_act(x, 1, 1.4).float() - Bk.box3_act(x, 1, 1.4).float()).abs().max())
    ra, rb = [], []
    for _ in range(3):
        ra.append(timeit(lambda: A.box3_act(x, 1, 1.4))); rb.append(timeit(lambda: Bk.box3_act(x, 1, 1.4)))
    a, b = min(ra) * 1e3, min(rb) * 1e3
    print(f"{R}^2 C={C}: A {a:7.1f} B {b:7.1f} us ({(b/a-1)*100:+.1f} %)  {2*x.numel()*2/b/1e6:5.2f} TB/s   max diff {d}", flush=True)

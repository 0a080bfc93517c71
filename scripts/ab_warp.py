"""GPU A/B: bicubic warp forward / backward between two builds of the library (one process, interleaved)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scripts.ab_conv import kernels_for  # noqa: E402  (re-uses the loader; its benchmark body runs only as __main__)
A, Bk = kernels_for(sys.argv[1]), kernels_for(sys.argv[2])
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (B, R, C) in [(32, 256, 128), (32, 128, 256), (32, 64, 512), (8, 1024, 32)]:
    x = torch.randn(B, R, R, C, device="cuda").bfloat16()
    gy = torch.randn(B, R, R, C, device="cuda").bfloat16()
    f = torch.zeros(B, R, R, 8, device="cuda")
    # a smooth flow of a few pixels' amplitude (what the generator produces), not one that folds the image
    f[..., :2] = 0.3 * torch.tanh(torch.nn.functional.interpolate(torch.randn(B, 2, 4, 4, device="cuda"), size=(R, R), mode="bicubic").permute(0, 2, 3, 1))
    f = f.bfloat16()
    ya, yb = A.warp_fwd(x, f, 0.1), Bk.warp_fwd(x, f, 0.1)
    ga, gb = A.warp_bwd(gy, x, f, 0.1), Bk.warp_bwd(gy, x, f, 0.1)
    d = [float((p.float() - q.float()).abs().max()) for p, q in ((ya, yb), (ga[0], gb[0]), (ga[1], gb[1]))]
    res = {"fa": [], "fb": [], "ba": [], "bb": []}
    for rnd in range(3):
        res["fa"].append(timeit(lambda: A.warp_fwd(x, f, 0.1))); res["fb"].append(timeit(lambda: Bk.warp_fwd(x, f, 0.1)))
        res["ba"].append(timeit(lambda: A.warp_bwd(gy, x, f, 0.1))); res["bb"].append(timeit(lambda: Bk.warp_bwd(gy, x, f, 0.1)))
    m = {k: min(v) * 1e3 for k, v in res.items()}
    print(f"B={B} {R}^2 C={C}: fwd A {m['fa']:7.1f} B {m['fb']:7.1f} us ({(m['fb']/m['fa']-1)*100:+.1f} %)   bwd A {m['ba']:7.1f} B {m['bb']:7.1f} us ({(m['bb']/m['ba']-1)*100:+.1f} %)   max diffs {d}", flush=True)

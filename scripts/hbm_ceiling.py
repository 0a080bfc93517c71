"""What the box's HBM delivers to plain streaming kernels (the ceiling the elementwise families are measured against)."""
import torch
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for shape in ((32, 256, 256, 128), (32, 128, 128, 256), (32, 64, 64, 512)):
    a = torch.randn(shape, device="cuda").bfloat16(); b = torch.randn(shape, device="cuda").bfloat16(); c = torch.empty_like(a)
    nb = a.numel() * 2
    s = t(lambda: c.copy_(a));               print(shape, f"copy   (1R+1W): {2 * nb / s / 1e12:.2f} TB/s  {s * 1e6:.0f} us")
    s = t(lambda: torch.add(a, b, out=c));   print(shape, f"add    (2R+1W): {3 * nb / s / 1e12:.2f} TB/s  {s * 1e6:.0f} us")
    s = t(lambda: torch.mul(a, 1.5, out=c)); print(shape, f"scale  (1R+1W): {2 * nb / s / 1e12:.2f} TB/s  {s * 1e6:.0f} us")
    s = t(lambda: c.zero_());                print(shape, f"fill   (1W)   : {nb / s / 1e12:.2f} TB/s  {s * 1e6:.0f} us")
    s = t(lambda: a.float().sum());          print(shape, f"(cast+sum, 2 kernels): {s * 1e6:.0f} us")

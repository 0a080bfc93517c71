import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lcgan_amd import kernels
from lcgan_amd.kernels import ACT_LRELU
H = kernels.HipKernels()
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
out = ["BOXB_RH=" + os.environ.get("LCGAN_BOXB_RH", "-")]
for shape in ((32, 32, 32, 512), (32, 16, 16, 512), (32, 64, 64, 512), (4, 256, 256, 128), (4, 128, 128, 256), (4, 64, 64, 512), (4, 32, 32, 512)):
    gy = torch.randn(shape, device="cuda").bfloat16(); y = torch.randn(shape, device="cuda").bfloat16()
    out.append(f"{shape}: {t(lambda: H.box3_actbwd(gy, y, ACT_LRELU, 1.0, shape[-1], True)):.1f} us")
print(" | ".join(out))

"""Time the activation-backward / stencil reductions on the big feature maps (one process per LCGAN_REDUCE_* setting)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lcgan_amd import kernels
from lcgan_amd.kernels import ACT_LRELU
H = kernels.HipKernels()
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
out = ["BOX_RH=" + os.environ.get("LCGAN_BOX_RH", "-")]
for shape in ((32, 256, 256, 128), (32, 128, 128, 256), (32, 64, 64, 512), (32, 32, 32, 512)):
    gy = torch.randn(shape, device="cuda").bfloat16(); y = torch.randn(shape, device="cuda").bfloat16()
    nb = gy.numel() * 2
    s1 = t(lambda: H.act_bwd_reduce(gy, y, ACT_LRELU, 1.0, shape[-1], want_gbias=True))
    s2 = t(lambda: H.box3_actbwd(gy, y, ACT_LRELU, 1.0, shape[-1], True))
    s3 = t(lambda: H.box3_act(gy, ACT_LRELU, 1.0))
    s4 = t(lambda: H.box3_act_bwd(gy, y, ACT_LRELU, 1.0))
    out.append(f"{shape[1]}^2x{shape[3]}: act_bwd {s1 * 1e6:.0f} us {3 * nb / s1 / 1e12:.2f} | box3_actbwd {s2 * 1e6:.0f} us {3 * nb / s2 / 1e12:.2f} | box3_act {s3 * 1e6:.0f} us {2 * nb / s3 / 1e12:.2f} | box3_act_bwd {s4 * 1e6:.0f} us {3 * nb / s4 / 1e12:.2f} TB/s")
print("\n".join(out))

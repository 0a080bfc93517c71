import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from lcgan_amd.kernels import HipKernels
H = HipKernels()
torch.manual_seed(0)
for (B, Hh, Ci, Co, k) in [(2, 8, 32, 32, 1), (2, 8, 32, 32, 3), (2, 8, 64, 128, 3)]:
    x = torch.randn(B, Hh, Hh, Ci, device="cuda").bfloat16()
    w = torch.randn(Co, Ci, k, k, device="cuda")
    pw, _ = H.prep_weight(w, 1 / math.sqrt(Ci * k * k), False, False)
    H.lib.lcgan_set_option(16, 0)
    ref = H.conv_fwd(x, pw, Co, k, 1).float()
    H.lib.lcgan_set_option(16, 1)
    got = H.conv_fwd(x, pw, Co, k, 1).float()
    bad = ~torch.isclose(got, ref, rtol=2e-2, atol=2e-2)
    print(f"B{B} {Hh}^2 {Ci}->{Co} k{k}: bad {int(bad.sum())} of {bad.numel()}")
    if bad.any():
        idx = bad.nonzero()
        print(" bad positions (b,y,x) unique:", idx[:, :3].unique(dim=0)[:10].tolist())
        print(" bad channels unique:", idx[:, 3].unique()[:40].tolist())
        print(" got", got[0, 0, 0, :8].tolist(), "\n ref", ref[0, 0, 0, :8].tolist())

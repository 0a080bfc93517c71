"""Stride-1 LDS-DMA convolution with two workgroups per CU (shipped) against ONE (option 3, bit 64): what a lone workgroup gets out of
the matrix pipe.   python scripts/micro_lone.py"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scripts.ab_conv import kernels_for
K = kernels_for(sys.argv[1] if len(sys.argv) > 1 else "lcgan_amd/liblcgan_hip.so")
B = 32
for (Hh, Ci, Co) in ((256, 128, 128), (128, 256, 256), (64, 512, 512)):
    x = torch.randn(B, Hh, Hh, Ci, device="cuda").bfloat16()
    w = torch.randn(Co, Ci, 3, 3, device="cuda")
    pw, _ = K.prep_weight(w, 1 / math.sqrt(Ci * 9), False, False)
    fl = 2.0 * B * Hh * Hh * Ci * Co * 9
    res = {0: [], 64: []}
    for rep in range(4):
        for opt in (0, 64):
            K.lib.lcgan_set_option(3, opt)
            for _ in range(2): K.conv_fwd(x, pw, Co, 3, 1, act=1, gain=1.4)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): K.conv_fwd(x, pw, Co, 3, 1, act=1, gain=1.4)
            e1.record(); torch.cuda.synchronize()
            res[opt].append(e0.elapsed_time(e1) / 5 * 1e3)
    K.lib.lcgan_set_option(3, 0)
    t2, t1 = np.median(res[0]), np.median(res[64])
    print(f"{Hh}^2 {Ci}->{Co}: two workgroups per CU {t2:.1f} us ({fl / t2 / 1e6:.0f} TFLOP/s), one {t1:.1f} us ({fl / t1 / 1e6:.0f} TFLOP/s): x{t1 / t2:.2f}")

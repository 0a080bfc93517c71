"""Weight-gradient launches of the big layers with the linear and the XCD-grouped workgroup order (option 15): time per launch, and --
when run under `rocprofv3 --pmc FETCH_SIZE` -- the dispatch order to read the fabric bytes per launch against.
  python scripts/micro_wgrad_xcd.py [time|pmc]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lcgan_amd.kernels import HipKernels
H = HipKernels()
mode = sys.argv[1] if len(sys.argv) > 1 else "time"
shapes = [(32, 256, 128, 128, 1), (32, 128, 256, 256, 1), (32, 64, 512, 512, 1), (32, 256, 128, 256, 2), (32, 128, 256, 512, 2)]
for (B, Hh, Ci, Co, st) in shapes:
    x = torch.randn(B, Hh, Hh, Ci, device="cuda").bfloat16()
    g = torch.randn(B, Hh // st, Hh // st, Co, device="cuda").bfloat16()
    alg = (x.numel() + g.numel()) * 2 / 1e6
    line = f"B{B} {Hh}^2 {Ci}->{Co} s{st} (operands {alg:.0f} MB): "
    for opt in (0, 1):
        H.lib.lcgan_set_option(15, opt)
        H.conv_wgrad(x, g, Co, Ci, 3, st); torch.cuda.synchronize()
        if mode == "pmc":
            H.conv_wgrad(x, g, Co, Ci, 3, st); torch.cuda.synchronize()
            continue
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                H.conv_wgrad(x, g, Co, Ci, 3, st)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 5 * 1e3)
        line += f" option15={opt}: {min(ts):7.1f} us"
    H.lib.lcgan_set_option(15, 0)
    print(line, flush=True)

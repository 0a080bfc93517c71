"""Stride-2 forward / transposed launches of the big layers under option variations: time per launch (mode time) or, under
`rocprofv3 --pmc FETCH_SIZE`, two dispatches per (shape, variant) to read the fabric bytes per launch from (mode pmc).
  python scripts/micro_s2_pmc.py time|pmc "14=0" "14=2048" ..."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lcgan_amd.kernels import HipKernels
H = HipKernels()
mode = sys.argv[1]
variants = sys.argv[2:] or [""]
shapes = [(32, 256, 128, 256), (32, 128, 256, 512), (32, 64, 512, 512)]
defaults = {}
for (B, Hh, Ci, Co) in shapes:
    x = torch.randn(B, Hh, Hh, Ci, device="cuda").bfloat16()
    g = torch.randn(B, Hh // 2, Hh // 2, Co, device="cuda").bfloat16()
    w = torch.randn(Co, Ci, 3, 3, device="cuda")
    pw, _ = H.prep_weight(w, 1 / math.sqrt(Ci * 9), False, False)
    pwt, _ = H.prep_weight(w, 1 / math.sqrt(Ci * 9), True, False)
    line = f"B{B} {Hh}^2 {Ci}->{Co} s2 (in {x.numel() * 2 / 1e6:.0f} MB, out {g.numel() * 2 / 1e6:.0f} MB): "
    for v in variants:
        olds = [(int(o), H.lib.lcgan_set_option(int(o), int(val))) for o, val in (kv.split("=") for kv in filter(None, v.split(",")))]
        fns = {"fwd": lambda: H.conv_fwd(x, pw, Co, 3, 2, act=1, gain=1.0), "tconv": lambda: H.conv_bwd_data(g, pwt, Ci, 3, 2)}
        for name, fn in fns.items():
            fn(); torch.cuda.synchronize()
            if mode == "pmc":
                fn(); torch.cuda.synchronize()
                continue
            ts = []
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(5):
                    fn()
                e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 5 * 1e3)
            line += f" [{v}] {name} {min(ts):6.1f} us"
        for o, old in olds:
            H.lib.lcgan_set_option(o, old)
    print(line, flush=True)

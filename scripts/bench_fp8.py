"""bf16 halo kernel vs the MX-fp8 kernel on the big conv shapes of the 256x256, B = 32 iteration (us per launch, TFLOP/s)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lcgan_amd.kernels import HipKernels
K = HipKernels()
B = 32


def timeit(fn, n):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (Hh, Ci, Co, st) in [(256, 128, 128, 1), (128, 256, 256, 1), (64, 512, 512, 1), (32, 512, 512, 1), (256, 128, 256, 2), (128, 256, 512, 2)]:
    x = torch.randn(B, Hh, Hh, Ci, device="cuda").bfloat16()
    g = torch.randn(B, Hh // st, Hh // st, Co, device="cuda").bfloat16()
    w = torch.randn(Co, Ci, 3, 3, device="cuda")
    bias = torch.randn(Co, device="cuda")
    sc = 1 / math.sqrt(Ci * 9)
    pw, _ = K.prep_weight(w, sc, False, False)
    pwT, _ = K.prep_weight(w, sc, True, False)
    p8, p8T = K.prep_weight_fp8(w, sc, False), K.prep_weight_fp8(w, sc, True)
    fl = 2.0 * B * (Hh // st) ** 2 * Co * Ci * 9
    r = []
    for rnd in range(3):
        r.append((timeit(lambda: K.conv_fwd(x, pw, Co, 3, st, bias=bias, act=1, gain=1.4), 6), (timeit(lambda: K.conv_fwd_fp8(x, p8, Co, 3, st, bias=bias, act=1, gain=1.4), 6) if st == 1 else float("nan")),
                  timeit(lambda: K.conv_bwd_data(g, pwT, Ci, 3, st), 6), timeit(lambda: K.conv_bwd_data_fp8(g, p8T, Ci, 3, st), 6)))
    fb, f8, db, d8 = (min(v[i] for v in r) for i in range(4))
    print(f"{Hh:4d}^2 {Ci:3d}->{Co:3d} s{st}: fwd bf16 {fb:7.1f} us ({fl / fb / 1e6:5.0f} TF/s)  fp8 {f8:7.1f} us ({fl / f8 / 1e6:5.0f})   dgrad bf16 {db:7.1f}  fp8 {d8:7.1f}", flush=True)

"""GPU micro-benchmark: weight gradient + un-prep (lcgan_conv_wgrad_fused) on the small-grid layer shapes, sweeping the number of splits
(option 21; 0 = automatic) against the round-2 launch plan (option 20 = 0).  Calls are captured in a graph: GPU time per call."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lcgan_amd.kernels import HipKernels
H = HipKernels()
shapes = [(B, R, 512, 512, k, st) for B in (4, 32) for (R, k, st) in ((16, 3, 1), (8, 3, 1), (32, 3, 2), (16, 3, 2), (16, 1, 1), (8, 1, 1))]
if os.environ.get("K1"):
    shapes = [(B, R, Ci, Co, 1, 1) for B in (4, 32) for (R, Ci, Co) in ((32, 512, 512), (64, 256, 512), (128, 128, 256), (128, 256, 24))]
if os.environ.get("SHAPES"):                                  # "B:R:Ci:Co:k:st,..."
    shapes = [tuple(int(t) for t in v.split(":")) for v in os.environ["SHAPES"].split(",")]
PARTS = [int(v) for v in os.environ.get("PARTS", "1,3,4,5,8,16").split(",")]
def bench(fn, n=20, reps=5):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        fn()
        with torch.cuda.graph(g, stream=side):
            for _ in range(n): fn()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (n * reps) * 1e3
for (B, R, Ci, Co, k, st) in shapes:
    x = torch.randn(B, R, R, Ci, device="cuda").bfloat16()
    g = torch.randn(B, R // st, R // st, Co, device="cuda").bfloat16()
    w = torch.randn(Co, Ci, k, k, device="cuda")
    scale = 1 / math.sqrt(Ci * k * k)
    fn = lambda: H.conv_wgrad_unprep(x, g, Co, Ci, k, st, scale)
    if os.environ.get("MOD"):                                 # per-sample scales + demodulation term (the modulated convolutions)
        px, pg, gwsq = torch.rand(B, Ci, device="cuda") + 0.5, torch.rand(B, Co, device="cuda") + 0.5, torch.randn(Co, Ci, device="cuda")
        fn = lambda: H.conv_wgrad_unprep(x, g, Co, Ci, k, st, scale, False, pre_x=px, pre_g=pg, w=w, gwsq=gwsq)
    H.lib.lcgan_set_option(20, 0)
    ref = fn().clone()
    line = f"{(B, R, Ci, Co, k, st)}: old {bench(fn):.1f} us |"
    H.lib.lcgan_set_option(20, 1)
    for parts in PARTS:
        H.lib.lcgan_set_option(21, parts)
        err = float((fn() - ref).abs().max() / ref.abs().max())
        line += f" p{parts} {bench(fn):.1f} (err {err:.1g})"
    H.lib.lcgan_set_option(20, 1); H.lib.lcgan_set_option(21, 0)
    line += f" | auto {bench(fn):.1f}"
    print(line, flush=True)

"""Lifetime of conv_halo_kernel workgroups (diagnostic build with -DHALO_STAMPS): per workgroup the 100 MHz timestamps at kernel
entry, main-loop start, main-loop end and after its stores were acknowledged -> prologue / loop / epilogue, and -- from the
workgroups that followed each other on one CU slot -- the gap between one workgroup's end and its successor's entry.
  python scripts/halo_life.py ab/libstamps.so"""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scripts.ab_conv import kernels_for
K = kernels_for(sys.argv[1])
K.lib.lcgan_halo_life.argtypes = [C.c_void_p]
B = 32
cases = [("fwd", 256, 128, 128, 1), ("fwd", 128, 256, 256, 1), ("fwd", 64, 512, 512, 1), ("tconv", 256, 128, 256, 2), ("fwd", 256, 128, 256, 2)]
for (kind, Hh, Ci, Co, st) in cases:
    x = torch.randn(B, Hh, Hh, Ci, device="cuda").bfloat16()
    g = torch.randn(B, Hh // st, Hh // st, Co, device="cuda").bfloat16()
    w = torch.randn(Co, Ci, 3, 3, device="cuda")
    pw, _ = K.prep_weight(w, 1 / math.sqrt(Ci * 9), kind == "tconv", False)
    fn = (lambda: K.conv_bwd_data(g, pw, Ci, 3, st)) if kind == "tconv" else (lambda: K.conv_fwd(x, pw, Co, 3, st, act=1, gain=1.4))
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    buf = (C.c_ulonglong * (16384 * 9))()
    assert K.lib.lcgan_halo_life(buf) == 0
    a = np.array(buf[:], dtype=np.float64).reshape(16384, 9)
    a = a[a[:, 0] >= a[:, 3].max() - 2 * e0.elapsed_time(e1) * 1e5]      # this launch's entries (100 MHz ticks; the buffer is never cleared)
    t0 = a[:, 0].min()
    us = (a[:, :4] - t0) / 100.0                                   # 100 MHz ticks -> us
    pro, loop, epi = us[:, 1] - us[:, 0], us[:, 2] - us[:, 1], us[:, 3] - us[:, 2]
    # successor gap: sort workgroups by (hardware slot: CU id bits of HW_ID ...) is fragile; instead: for every workgroup find the
    # earliest entry after its end among all workgroups -- with a full chip that is its successor on the freed slot (lower bound of the gap)
    ends = np.sort(us[:, 3]); starts = np.sort(us[:, 0])
    nxt = np.searchsorted(starts, ends, side="left")
    gaps = starts[np.minimum(nxt, len(starts) - 1)] - ends
    gaps = gaps[(nxt < len(starts)) & (gaps >= 0)]
    if a[:, 5].max() > 0:                                          # LDS-DMA structure: the finer stamps
        sub = (a[:, [5, 6, 7, 8]] - t0) / 100.0
        print(f"    prologue: entry -> first DMA issued {np.median(sub[:, 0] - us[:, 0]):.2f}, -> landed {np.median(sub[:, 1] - sub[:, 0]):.2f}, -> loop {np.median(us[:, 1] - sub[:, 1]):.2f};  "
              f"epilogue: loop end -> output tile in LDS {np.median(sub[:, 2] - us[:, 2]):.2f}, -> stores issued {np.median(sub[:, 3] - sub[:, 2]):.2f}, -> acknowledged {np.median(us[:, 3] - sub[:, 3]):.2f}")
    print(f"{kind} {Hh}^2 {Ci}->{Co} s{st}: kernel {e0.elapsed_time(e1) * 1e3:.0f} us, {len(a)} workgroups sampled; median us per workgroup: "
          f"prologue {np.median(pro):.2f}  main loop {np.median(loop):.2f}  epilogue (to stores acknowledged) {np.median(epi):.2f}  "
          f"life {np.median(us[:, 3] - us[:, 0]):.2f};  earliest entry after a workgroup's end: median {np.median(gaps):.2f} us (p90 {np.percentile(gaps, 90):.2f})")

"""In-kernel stamps of conv_wgrad3_dma_kernel's chunk loop (diagnostic build with -DHALO_STAMPS, never the shipped library):
where a 64-position chunk's cycles go, per wave.   python scripts/wgrad_stamps.py ab/libstamps.so ["15=1"]"""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scripts.ab_conv import kernels_for
K = kernels_for(sys.argv[1])
K.lib.lcgan_halo_stamps.argtypes = [C.c_void_p]
for kv in filter(None, (sys.argv[2] if len(sys.argv) > 2 else "").split(",")):
    K.lib.lcgan_set_option(*map(int, kv.split("=")))
B = 32
for (Hh, Ci, Co, st) in [(256, 128, 128, 1), (128, 256, 256, 1), (64, 512, 512, 1), (256, 128, 256, 2), (128, 256, 512, 2)]:
    x = torch.randn(B, Hh, Hh, Ci, device="cuda").bfloat16()
    g = torch.randn(B, Hh // st, Hh // st, Co, device="cuda").bfloat16()
    for _ in range(3):
        K.conv_wgrad(x, g, Co, Ci, 3, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); K.conv_wgrad(x, g, Co, Ci, 3, st); e1.record(); torch.cuda.synchronize()
    buf = (C.c_ulonglong * (2048 * 8 * 5))()
    assert K.lib.lcgan_halo_stamps(buf) == 0
    a = np.array(buf[:], dtype=np.float64).reshape(2048, 8, 5)
    a = a[a[:, 0, 4] > 0]
    chunks = a[:, :, 4:5]
    per = a[:, :, :4] / chunks
    med = np.median(per.reshape(-1, 4), axis=0)
    print(f"{Hh}^2 {Ci}->{Co} s{st}: {e0.elapsed_time(e1) * 1e3:.0f} us; chunks/wg {np.median(chunks):.0f}; cycles/chunk  [DMA issue] {med[0]:.0f}  [reads + MFMA issue] {med[1]:.0f}  [vmcnt wait] {med[2]:.0f}  [barrier wait] {med[3]:.0f}  total {med.sum():.0f}"
          f"   (MFMA work per SIMD and chunk with two workgroups per CU: {4 * (24 if st == 1 else 12) * 32} cycles)")

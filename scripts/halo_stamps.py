"""In-kernel stamps of conv_halo_kernel's main loop (diagnostic build with -DHALO_STAMPS, never the shipped library):
where a step's cycles go, per wave.   bash: see scripts/gpu_r2.sh `stamps`."""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scripts.ab_conv import kernels_for
K = kernels_for(sys.argv[1])
K.lib.lcgan_halo_stamps.argtypes = [C.c_void_p]
for kv in filter(None, (sys.argv[2] if len(sys.argv) > 2 else "").split(",")):     # e.g. "3=16": lcgan_set_option pairs
    K.lib.lcgan_set_option(*map(int, kv.split("=")))
B = 32
for (Hh, Ci, Co, st) in [(256, 128, 128, 1), (128, 256, 256, 1), (64, 512, 512, 1)]:
    x = torch.randn(B, Hh, Hh, Ci, device="cuda").bfloat16()
    w = torch.randn(Co, Ci, 3, 3, device="cuda")
    pw, _ = K.prep_weight(w, 1 / math.sqrt(Ci * 9), False, False)
    for _ in range(3):
        K.conv_fwd(x, pw, Co, 3, st, act=1, gain=1.4)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (2048 * 8 * 5))()
    assert K.lib.lcgan_halo_stamps(buf) == 0
    a = np.array(buf[:], dtype=np.float64).reshape(2048, 8, 5)
    steps = a[0, 0, 4]
    ntiles = B * ((Hh // st + 15) // 16) ** 2 * ((Co + 127) // 128)       # (the 1-D grid holds tiles x channel blocks)
    a = a[:min(ntiles, 2048)]                                  # (rows beyond this launch's grid hold an earlier launch)
    per = a[:, :, :4] / steps                                  # cycles per step and segment, per (workgroup, wave)
    med = np.median(per.reshape(-1, 4), axis=0)
    early, late = np.median(per[:, :4].reshape(-1, 4), axis=0), np.median(per[:, 4:].reshape(-1, 4), axis=0)
    print(f"{Hh}^2 {Ci}->{Co}: steps {steps:.0f}; cycles/step  [DMA issue + first fragments landed] {med[0]:.0f}  [MFMA issue + interleaved reads] {med[1]:.0f}  [vmcnt wait] {med[2]:.0f}  [barrier wait] {med[3]:.0f}  total {med.sum():.0f}")
    ck = (C.c_ulonglong * (2048 * 8 * 2))()
    K.lib.lcgan_halo_clock.argtypes = [C.c_void_p]
    if K.lib.lcgan_halo_clock(ck) == 0:
        c = np.array(ck[:], dtype=np.float64).reshape(2048, 8, 2)[:min(ntiles, 2048)]
        ghz = np.median(c[:, :, 0] / np.maximum(c[:, :, 1], 1.0)) * 0.1                     # s_memrealtime ticks at 100 MHz
        print(f"      shader clock during the main loop (s_memtime / s_memrealtime): {ghz:.2f} GHz; MFMA work per step and SIMD 2048 cycles -> pipe occupancy {2048 / med.sum():.2f}; "
              f"peak at this clock {ghz * 1.048576:.2f} PFLOP/s")
    print(f"      waves 0-3: {early.round()}   waves 4-7: {late.round()}")

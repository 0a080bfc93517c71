"""Do the two workgroups a CU holds run conv_halo_kernel's main loop at the same time?  With a -DHALO_STAMPS build: per CU the share of
the launch with two / one / no workgroup inside its main loop.  (The round-4 experiment that delayed the second workgroup of every CU --
`lcgan_set_option(27, tenths of a microsecond)`, profiles/r04_halo_phase_stagger.txt -- is not in the tree: a library without that
option runs the undelayed case only.)
  python scripts/halo_phase.py <lib.so> [start delays in 0.1 us ...]"""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scripts.ab_conv import kernels_for
K = kernels_for(sys.argv[1])
vals = [int(v) for v in sys.argv[2:]] or [0, 60, 120, 180, 240]
has_life = hasattr(K.lib, "lcgan_halo_life")
if has_life:
    K.lib.lcgan_halo_life.argtypes = [C.c_void_p]
B = 32
cases = [("fwd", 256, 128, 128, 1), ("fwd", 128, 256, 256, 1), ("fwd", 64, 512, 512, 1), ("tconv", 128, 128, 256, 2), ("tconv", 64, 256, 512, 2)]
for (kind, Hh, Ci, Co, st) in cases:
    x = torch.randn(B, Hh, Hh, Ci, device="cuda").bfloat16()
    g = torch.randn(B, Hh, Hh, Co, device="cuda").bfloat16()
    w = torch.randn(Co, Ci, 3, 3, device="cuda")
    pw, _ = K.prep_weight(w, 1 / math.sqrt(Ci * 9), kind == "tconv", False)
    fn = (lambda: K.conv_bwd_data(g, pw, Ci, 3, st)) if kind == "tconv" else (lambda: K.conv_fwd(x, pw, Co, 3, st, act=1, gain=1.4))
    ref = None
    for v in vals:
        if K.lib.lcgan_set_option(27, v) < 0 and v != 0:
            continue
        for _ in range(3):
            y = fn()
        torch.cuda.synchronize()
        y = y[0] if isinstance(y, tuple) else y
        if ref is None:
            ref = y.clone()
        assert torch.equal(ref, y)
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        line = f"{kind} {Hh}^2 {Ci}->{Co} s{st} stagger {v / 10:5.1f} us: launch {np.median(ts):7.1f} us (min {min(ts):7.1f})"
        buf = (C.c_ulonglong * (16384 * 9))()
        if has_life and K.lib.lcgan_halo_life(buf) == 0:
            a = np.array(buf[:], dtype=np.float64).reshape(16384, 9)
            idx = np.nonzero(a[:, 0] >= a[:, 3].max() - 2 * np.median(ts) * 100)[0]      # this launch's entries (the buffer is never cleared)
            a = a[idx]
            t0, t1 = a[:, 0].min(), a[:, 3].max()
            cu = (idx & 7) * 256 + ((a[:, 4].astype(np.int64) >> 8) & 0xff)          # 1-D launches: XCD = block id mod 8
            both = one = none = 0.0
            for c in np.unique(cu):
                r = a[cu == c]
                ev = sorted([(t, 1) for t in r[:, 1]] + [(t, -1) for t in r[:, 2]])
                lo, hi = r[:, 0].min(), r[:, 3].max()
                n, prev = 0, lo
                for t, d in ev:
                    if n >= 2: both += t - prev
                    elif n == 1: one += t - prev
                    else: none += t - prev
                    n += d; prev = t
                none += hi - prev
            tot = both + one + none
            line += f";  CU time with 2 / 1 / 0 workgroups in the main loop: {both / tot:.2f} / {one / tot:.2f} / {none / tot:.2f} ({len(np.unique(cu))} CUs seen, {len(a)} workgroups)"
        print(line, flush=True)
    K.lib.lcgan_set_option(27, 0)

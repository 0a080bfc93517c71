"""GPU A/B of lcgan_conv_fwd across an ABI change: raw ctypes calls, per-library argument lists.
  python scripts/ab_raw.py ab/libOld.so 21 ab/libNew.so 23      (number = argument count of lcgan_conv_fwd in that build)"""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
P, I, F = C.c_void_p, C.c_int, C.c_float
libs = []
for path, nargs in ((sys.argv[1], int(sys.argv[2])), (sys.argv[3], int(sys.argv[4]))):
    lib = C.CDLL(path)
    lib.lcgan_conv_weight_prep.argtypes = [P, I, I, I, F, I, P, I, P, P]
    base = [P, P, P, I, I, I, I, I, I, I, I, P, P, P, F, I, F, P]               # ... residual
    lib.lcgan_conv_fwd.argtypes = base + ([I] if nargs >= 21 else []) + ([P, P] if nargs == 23 else []) + [I, P]   # residual_half; xs, gs
    libs.append((lib, nargs))
def timeit(fn, n):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
B = 32
st = torch.cuda.current_stream().cuda_stream
for (Hh, Ci, Co, stride) in [(256, 128, 128, 1), (128, 256, 256, 1), (64, 512, 512, 1), (256, 128, 256, 2)]:
    x = torch.randn(B, Hh, Hh, Ci, device="cuda").bfloat16()
    w = torch.randn(Co, Ci, 3, 3, device="cuda")
    y = torch.empty(B, Hh // stride, Hh // stride, Co, device="cuda", dtype=torch.bfloat16)
    res = [[], []]
    wps = []
    for lib, nargs in libs:
        wp = torch.empty(9 * Co * Ci, dtype=torch.bfloat16, device="cuda")
        assert lib.lcgan_conv_weight_prep(w.data_ptr(), Co, Ci, 3, 1 / math.sqrt(Ci * 9), 0, wp.data_ptr(), 1, None, st) == 0
        wps.append(wp)
    def call(i):
        lib, nargs = libs[i]
        args = [x.data_ptr(), wps[i].data_ptr(), y.data_ptr(), B, Hh, Hh, Ci, Co, Co, 3, stride, None, None, None, 1.0, 1, 1.4, None]
        args += ([0] if nargs >= 21 else []) + ([None, None] if nargs == 23 else []) + [1, st]
        assert lib.lcgan_conv_fwd(*args) == 0
    for rnd in range(4):
        for i in (0, 1):
            res[i].append(timeit(lambda: call(i), 8))
    a, b = min(res[0]), min(res[1])
    print(f"{Hh}^2 {Ci}->{Co} s{stride}: old {a*1e3:7.1f} us  new {b*1e3:7.1f} us  ({(b/a-1)*100:+.1f} %)", flush=True)

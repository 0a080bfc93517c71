"""GPU A/B harness: two builds of liblcgan_hip.so in ONE process, interleaved per shape, on the conv shapes of the 256x256
B=32 iteration (forward, data gradient, weight gradient).  Box-to-box and thermal noise (+-2 % between gpurun calls) cancels.
  python scripts/ab_conv.py /path/libA.so /path/libB.so [batch]"""
import ctypes as C, math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lcgan_amd import _lib
from lcgan_amd.kernels import HipKernels


def kernels_for(path):
    k = HipKernels.__new__(HipKernels)
    lib = C.CDLL(path)
    for name, argtypes in _lib.SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:                       # an older build without this entry point
            continue
        fn.argtypes = argtypes
        fn.restype = C.c_int
    k.lib = lib
    from lcgan_amd.kernels import _ZeroPool
    k._zeros = _ZeroPool()
    k._prep_tables = {}
    return k


if __name__ == "__main__":
    A, Bk = kernels_for(sys.argv[1]), kernels_for(sys.argv[2])
    for K_, env in ((A, "OPT_A"), (Bk, "OPT_B")):          # e.g. OPT_B="8=0,2=1536": lcgan_set_option switches per build
        for kv in filter(None, os.environ.get(env, "").split(",")):
            o, v = kv.split("=")
            K_.lib.lcgan_set_option(int(o), int(v))
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    # (H, Cin, Cout, stride, count per iteration as fwd-like, as dgrad-like, as wgrad)   -- 256x256 generator + discriminator layers
    shapes = [(256, 128, 128, 1, 8, 6, 6), (128, 256, 256, 1, 9, 7, 5), (64, 512, 512, 1, 9, 7, 5), (32, 512, 512, 1, 12, 10, 5),
              (16, 512, 512, 1, 12, 10, 5), (256, 128, 256, 2, 6, 0, 6), (128, 256, 512, 2, 6, 0, 5), (64, 512, 512, 2, 6, 0, 5)]
    ev = lambda: torch.cuda.Event(enable_timing=True)


    def timeit(fn, n):
        fn(); torch.cuda.synchronize()
        e0, e1 = ev(), ev()
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n


    tot = {"A": 0.0, "B": 0.0}
    for (Hh, Ci, Co, st, nf, nd, nw) in shapes:
        x = torch.randn(B, Hh, Hh, Ci, device="cuda").bfloat16()
        w = torch.randn(Co, Ci, 3, 3, device="cuda")
        g = torch.randn(B, Hh // st, Hh // st, Co, device="cuda").bfloat16()
        sc = 1 / math.sqrt(Ci * 9)
        res = {}
        for tag, K in (("A", A), ("B", Bk)):
            pw, _ = K.prep_weight(w, sc, False, False)
            pwt, _ = K.prep_weight(w, sc, True, False)
            res[tag] = dict(pw=pw, pwt=pwt, K=K, f=[], d=[], w=[])
        n = 6 if Hh >= 64 else 20
        for rnd in range(3):
            for tag in ("A", "B"):
                r = res[tag]; K = r["K"]
                r["f"].append(timeit(lambda: K.conv_fwd(x, r["pw"], Co, 3, st, act=1, gain=1.4), n))
                r["d"].append(timeit(lambda: K.conv_bwd_data(g, r["pwt"], Ci, 3, st), n))
                r["w"].append(timeit(lambda: K.conv_wgrad(x, g, Co, Ci, 3, st), n))
        line = f"{Hh:4d}^2 {Ci:3d}->{Co:3d} s{st}: "
        for kind, cnt in (("f", nf), ("d", nd if st == 1 else nf), ("w", nw)):
            a, b = min(res["A"][kind]), min(res["B"][kind])
            line += f" {kind}: A {a*1e3:7.1f} B {b*1e3:7.1f} us ({(b/a-1)*100:+5.1f} %)"
            tot["A"] += a * cnt; tot["B"] += b * cnt
        print(line, flush=True)
    print(f"weighted per-iteration conv time: A {tot['A']:.2f} ms  B {tot['B']:.2f} ms  ({(tot['B']/tot['A']-1)*100:+.2f} %)")

"""GPU A/B of several option settings of ONE build in one process, interleaved per shape (forward + data gradient only).
  python scripts/ab_multi.py "10=0" "10=1" "10=1,11=1" ...      (each argument: lcgan_set_option pairs of one variant)"""
import ctypes as C, math, os, shutil, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scripts.ab_conv import kernels_for

variants = sys.argv[1:]
Ks = []
for i, v in enumerate(variants):
    path = f"/tmp/libv{i}.so"
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lcgan_amd", "liblcgan_hip.so")
    kvs = [kv.split("=") for kv in filter(None, v.split(","))]
    for o, val in kvs:
        if o == "lib":                                   # "lib=ab/libX.so,4=1": another build as this variant
            src = val
    shutil.copy(src, path)
    K = kernels_for(path)
    for o, val in kvs:
        if o != "lib":
            K.lib.lcgan_set_option(int(o), int(val))
    Ks.append(K)
B = 32
shapes = [(256, 128, 128, 1), (128, 256, 256, 1), (64, 512, 512, 1), (32, 512, 512, 1), (256, 128, 256, 2), (128, 256, 512, 2)]


def timeit(fn, n):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


print("variants:", variants)
for (Hh, Ci, Co, st) in shapes:
    x = torch.randn(B, Hh, Hh, Ci, device="cuda").bfloat16()
    w = torch.randn(Co, Ci, 3, 3, device="cuda")
    g = torch.randn(B, Hh // st, Hh // st, Co, device="cuda").bfloat16()
    bias = torch.randn(Co, device="cuda")
    sc = 1 / math.sqrt(Ci * 9)
    res = [dict(f=[], d=[], w=[], m=[]) for _ in Ks]
    pre, post = torch.rand(B, Ci, device="cuda") + 0.5, torch.rand(B, Co, device="cuda") + 0.5
    pws = [(K.prep_weight(w, sc, False, False)[0], K.prep_weight(w, sc, True, False)[0]) for K in Ks]
    n = 6 if Hh >= 64 else 20
    for rnd in range(3):
        for i, K in enumerate(Ks):
            res[i]["f"].append(timeit(lambda: K.conv_fwd(x, pws[i][0], Co, 3, st, bias=bias, act=1, gain=1.4), n))
            res[i]["d"].append(timeit(lambda: K.conv_bwd_data(g, pws[i][1], Ci, 3, st), n))
            res[i]["w"].append(timeit(lambda: K.conv_wgrad(x, g, Co, Ci, 3, st), n))
            res[i]["m"].append(timeit(lambda: K.conv_fwd(x, pws[i][0], Co, 3, st, pre=pre, post=post, bias=bias, act=1, gain=1.0), n))
    # agreement of every variant with variant 0 on the same inputs (bf16 outputs: differences beyond ~1 ulp mean a bug)
    resid = torch.randn(B, Hh // st, Hh // st, Co, device="cuda").bfloat16()
    rhalf = torch.randn(B, Hh // 2, Hh // 2, Ci, device="cuda").bfloat16() if st == 1 else None
    outs = []
    for i, K in enumerate(Ks):
        o = [K.conv_fwd(x, pws[i][0], Co, 3, st, bias=bias, act=1, gain=1.4).float(),
             K.conv_fwd(x, pws[i][0], Co, 3, st, pre=pre, post=post, bias=bias, act=1, gain=1.0, residual=None).float(),
             K.conv_fwd(x, pws[i][0], Co, 3, st, residual=resid).float(),
             K.conv_bwd_data(g, pws[i][1], Ci, 3, st).float()]
        if rhalf is not None:
            o.append(K.conv_bwd_data(g, pws[i][1], Ci, 3, st, residual=rhalf, residual_half=True).float())
        o.append(K.conv_wgrad(x, g, Co, Ci, 3, st).float().clone())               # (fp32 atomics / slabs: agreement to ~1e-6)
        o.append(K.conv_wgrad(x, g, Co, Ci, 3, st, pre_x=pre, pre_g=post).float().clone())
        outs.append(o)
    errs = [max(float((a - b).abs().max() / b.abs().max()) for a, b in zip(o, outs[0])) for o in outs]
    line = f"{Hh:4d}^2 {Ci:3d}->{Co:3d} s{st}: maxrel-vs-v0 " + " ".join(f"{e:.1e}" for e in errs)
    for kind in ("f", "d", "w", "m"):
        line += f" {kind}:" + " ".join(f"{min(r[kind]) * 1e3:7.1f}" for r in res)
    print(line, flush=True)

"""Summarise memory-side traffic per kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of the same command.

    python scripts/pmc_traffic.py <fetch_dir> <write_dir> <out.json> [launch_table.csv [iterations_profiled]]

gfx950 corrections (MI355X_MICROARCH.md, HBM / rocprofv3): the counters are in KiB; FETCH_SIZE reports half of the bytes of wide
coalesced reads -> x2.  Both count the L2's memory-side (fabric) requests, Infinity-Cache hits included.
One row per kernel INSTANTIATION (template arguments kept), then the convolution families; with a launch table of the same command
(bench.py --launch-table: shapes per launch) the ALGORITHMIC bytes of the convolution / weight-gradient families -- every operand and
result once -- are computed from the shape tags and printed beside the measured bytes as the read / total amplification."""
import collections
import csv
import glob
import json
import os
import re
import sys


def short(name: str) -> str:
    """kernel instantiation as a readable key: `conv_halo_kernel<1, false, 0, 2, false>`, `box3_act_kernel<bf16>`, `aten:add<float>`"""
    m = re.match(r"_ZN\d+_GLOBAL__N_1(\d+)", name)
    if m:                                                     # still mangled: _ZN12_GLOBAL__N_1<len><ident>I<template args>E...
        n, start = int(m.group(1)), m.end()
        ident, rest = name[start:start + n], name[start + n:]
        targ = "bf16" if rest.startswith("IDF16b") else "f32" if rest.startswith("If") else ""
        flags = re.match(r"I(?:DF16b|f)((?:Lb[01]|Li\d+)*)E", rest)
        extra = ""
        if flags and flags.group(1):
            extra = ", " + ", ".join(("true" if t[2] == "1" else "false") if t[1] == "b" else t[2:] for t in re.findall(r"L[bi]\d+", flags.group(1)))
        return f"{ident}<{targ}{extra}>" if targ else ident
    s = name.replace("(anonymous namespace)::", "").replace("void ", "").strip()
    if s.startswith("at::native::"):
        f = re.search(r"(CUDAFunctor_\w+|FillFunctor|\w+Functor\w*|func_wrapper_t|ReduceOp)<([^<>,]+)", s)
        k = re.match(r"at::native::(\w+)", s).group(1)
        return f"aten:{k}<{f.group(1).replace('CUDAFunctor_', '')}<{f.group(2)}>>" if f else f"aten:{k}"
    depth, out = 0, ""
    for ch in s:                                              # up to the argument list: the first "(" outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out += ch
    out = out.replace("bool _Accum", "bf16").replace("__bf16", "bf16")
    return out.strip() or name[:60]


def family(key: str) -> str:
    for f in ("conv_halo_narrow", "conv_halo", "conv_wgrad3", "conv_wgrad", "conv_igemm8", "conv_igemm", "slab_reduce"):
        if key.startswith(f):
            return f
    return ""


def load(d, counter):
    f = sorted(glob.glob(f"{d}/*/*counter_collection.csv"))[-1]
    out = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        out[k][0] += 1
        out[k][1] += float(r["Counter_Value"])
    return out


def algorithmic_bytes(table_csv):
    """per family, bytes of ONE iteration if every operand is read once and every result written once (bf16 feature maps, fp32 weight
    gradients), from the shape tags of bench.py --launch-table:  fwd|dgrad B32 256x256 C128->128 k3 s1 [+res|+res/2|+gs] / wgrad B32 256x256 A128 Bc128 k3 s1"""
    alg = collections.defaultdict(float)
    for r in csv.DictReader(open(table_csv)):
        tag = r["tag"]
        m = re.match(r"(fwd|dgrad) B(\d+) (\d+)x(\d+) C(\d+)->(\d+) k(\d) s(\d)(.*)", tag)
        if m:
            kind, B, H, W, Ci, Co, k, s = m.group(1), *map(int, m.groups()[1:8])
            c8 = lambda c: (c + 7) // 8 * 8
            if kind == "fwd":
                pin, pout = B * H * W, B * (H // s) * (W // s)
            else:
                pin, pout = B * H * W, B * H * s * W * s
            b = 2.0 * (pin * c8(Ci) + pout * c8(Co)) + 2.0 * k * k * Ci * Co
            rest = m.group(9)
            if "+res/2" in rest:
                b += 2.0 * pout * c8(Co) / 4
            elif "+res" in rest or "+gs" in rest:
                b += 2.0 * pout * c8(Co)
            if "+pool" in rest:
                b += 2.0 * pout * c8(Co) / 4
            alg["conv fwd/dgrad"] += b
            continue
        m = re.match(r"wgrad B(\d+) (\d+)x(\d+) A(\d+) Bc(\d+) k(\d) s(\d)", tag)
        if m:
            B, H, W, A, Bc, k, s = map(int, m.groups())
            alg["conv wgrad"] += 2.0 * B * H * W * Bc + 2.0 * B * (H // s) * (W // s) * A + 4.0 * k * k * A * Bc
    return alg


if __name__ == "__main__":
    fe, wr = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    rows = []
    for k in fe:
        n = fe[k][0]
        rd = fe[k][1] * 1024 * 2                               # KiB -> bytes, x2 gfx950 correction
        w = wr.get(k, [0, 0.0])[1] * 1024
        rows.append((rd + w, k, n, rd, w))
    rows.sort(reverse=True)
    res = {"kernels": {}, "families": {}}
    print(f"{'kernel instantiation':64s} {'launches':>8s} {'read GB':>9s} {'write GB':>9s} {'MB/launch':>10s}")
    for tot, k, n, rd, w in rows:
        res["kernels"][k] = {"launches": n, "read_bytes_corrected": rd, "write_bytes": w, "bytes_per_launch": tot / n}
        if tot >= 0.2e9:
            print(f"{k[:64]:64s} {n:8d} {rd / 1e9:9.2f} {w / 1e9:9.2f} {tot / n / 1e6:10.2f}")
    fam = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for tot, k, n, rd, w in rows:
        f = family(k)
        if f:
            fam[f][0] += n; fam[f][1] += rd; fam[f][2] += w
    print("\nconvolution families (all instantiations):")
    for f, (n, rd, w) in sorted(fam.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
        res["families"][f] = {"launches": n, "read_bytes_corrected": rd, "write_bytes": w, "bytes_per_launch": (rd + w) / n}
        print(f"{f:64s} {n:8d} {rd / 1e9:9.2f} {w / 1e9:9.2f} {(rd + w) / n / 1e6:10.2f}")
    if len(sys.argv) > 4:
        iters = float(sys.argv[5]) if len(sys.argv) > 5 else 1.0
        alg = algorithmic_bytes(sys.argv[4])
        groups = {"conv fwd/dgrad": ("conv_halo", "conv_halo_narrow", "conv_igemm8", "conv_igemm"), "conv wgrad": ("conv_wgrad3", "conv_wgrad", "slab_reduce")}
        print(f"\nalgorithmic bytes (every operand once) per iteration vs measured fabric bytes per iteration ({iters:g} iterations profiled):")
        res["amplification"] = {}
        for g, fams in groups.items():
            rd = sum(fam[f][1] for f in fams if f in fam) / iters
            w = sum(fam[f][2] for f in fams if f in fam) / iters
            a = alg.get(g, 0.0)
            if a > 0:
                res["amplification"][g] = {"algorithmic_bytes": a, "read_bytes": rd, "write_bytes": w, "total_over_algorithmic": (rd + w) / a}
                print(f"  {g:16s} algorithmic {a / 1e9:7.2f} GB   measured read {rd / 1e9:7.2f} + write {w / 1e9:7.2f} GB   total / algorithmic = {(rd + w) / a:4.2f}")
    res["_commit"] = os.environ.get("LCGAN_COMMIT")          # the build the passes ran on (set by the submitting shell: the GPU box has no .git)
    try:                                                     # ... and the hash of the kernel sources the profiled library was built from:
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from lcgan_amd.build import source_hash              # bench.py quotes `traffic` only while the running build still matches it
        res["_srchash"] = source_hash()
    except Exception:  # noqa: BLE001
        res["_srchash"] = None
    json.dump(res, open(sys.argv[3], "w"), indent=1)

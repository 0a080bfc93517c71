"""Summarise bench.py --launch-table output: per conv geometry (tag) launches, mean us, TFLOP/s.  python scripts/launch_table.py <csv>"""
import collections
import csv
import sys

names = ["conv_igemm", "conv_wgrad", "weight_prep", "stencil", "act_bwd", "warp_fwd", "warp_bwd", "rgb", "linear", "small", "optim", "layout", "scale_reduce"]
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
agg_bytes = collections.defaultdict(float)
for r in rows:
    k = (int(r["kid"]), r["tag"])
    a = agg.setdefault(k, [0, 0.0, 0.0])
    a[0] += 1; a[1] += float(r["ms"]); a[2] += float(r["flops"])
    agg_bytes[k] += float(r["bytes"])
tot = collections.defaultdict(float)
print(f"{'family':12s} {'geometry':44s} {'n':>3s} {'us/launch':>10s} {'ms':>7s} {'TFLOP/s':>8s}")
for (kid, tag), (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    tot[kid] += ms
    if ms < 0.05 or not tag:
        continue
    by = agg_bytes.get((kid, tag), 0.0)
    rate = f"{fl / ms / 1e9:8.0f}" if kid <= 1 else f"{by / ms / 1e9:6.2f}TB" if by > 0 else ""        # TFLOP/s for convolutions, algorithmic TB/s for the memory-bound families
    print(f"{names[kid]:12s} {tag:44s} {n:3d} {ms / n * 1e3:10.1f} {ms:7.3f} {rate:>8s}")
print("totals (ms):", {names[k]: round(v, 2) for k, v in sorted(tot.items())})

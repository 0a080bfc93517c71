"""Per-shape difference of two in-iteration launch tables (bench.py --launch-table): which launches a switch helps and which it hurts.
  python scripts/lt_diff.py A.csv B.csv [min_us]      prints (A - B) per tag, families conv (0) and weight gradient (1)"""
import collections, csv, sys


def load(f):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["kid"] not in ("0", "1"):
            continue
        a = agg[(r["kid"], r["tag"])]
        a[0] += 1
        a[1] += float(r["ms"]) * 1e3
    return agg


a, b = load(sys.argv[1]), load(sys.argv[2])
thr = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0
rows = [(a[t][1] - b[t][1], t, a[t][0], a[t][1] / a[t][0], b[t][1] / b[t][0]) for t in a if t in b and abs(a[t][1] - b[t][1]) > thr]
for d, t, n, x, y in sorted(rows):
    print(f"{d:+8.1f} us  {t[1]:58s} n={n:2d}  A {x:7.1f}  B {y:7.1f}")
for kid, name in (("0", "conv"), ("1", "wgrad")):
    print(name, "A", round(sum(v[1] for k, v in a.items() if k[0] == kid)), "B", round(sum(v[1] for k, v in b.items() if k[0] == kid)))

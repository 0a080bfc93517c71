"""Aggregate a rocprofv3 kernel trace CSV per kernel name: python scripts/agg_trace.py <trace.csv> [iterations] [top]"""
import collections
import csv
import re
import sys

path = sys.argv[1]
iters = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 36
rows = list(csv.DictReader(open(path)))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
    n = re.sub(r"[<(].*", "", n)
    if n.startswith("_ZN"):
        n = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", n)
        n = re.sub(r"_kernel.*", "_kernel", n)
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg[n][0] += 1
    agg[n][1] += d
print(f"launches/it {len(rows) / iters:.0f}   kernel ms/it {sum(v[1] for v in agg.values()) / iters / 1e3:.2f}")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{n[:50]:50s} n/it={c / iters:7.1f} avg={t / c:8.1f}us  ms/it={t / iters / 1e3:6.2f}")

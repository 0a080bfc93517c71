"""Summarise SQ counters per kernel family from one rocprofv3 --pmc pass: python scripts/pmc_sq.py <dir>"""
import collections
import csv
import glob
import sys

f = sorted(glob.glob(f"{sys.argv[1]}/*/*counter_collection.csv"))[-1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"]
    if "conv_halo" in name:
        key = "conv_halo<" + name.split("<")[1].split(">")[0] + "> grid " + r["Grid_Size"] if "Grid_Size" in r else "conv_halo"
    elif "wgrad3" in name:
        key = "conv_wgrad3<" + name.split("<")[1].split(">")[0] + ">"
    elif "conv_wgrad" in name:
        key = "conv_wgrad"
    elif "conv_igemm" in name:
        key = "conv_igemm"
    else:
        continue
    agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(key, r["Counter_Name"])] += 1
for k, c in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    wc = c.get("SQ_WAVE_CYCLES", 1.0)
    print(k)
    for n, v in sorted(c.items()):
        print(f"    {n:32s} {v:16.0f}  {v / wc:8.3f} of WAVE_CYCLES")

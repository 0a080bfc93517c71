"""per-dispatch FETCH_SIZE (x2, KiB -> MB) of the kernels whose name contains argv[2], in dispatch order: python scripts/pmc_per_dispatch.py <dir> <substr>"""
import csv, glob, sys
f = sorted(glob.glob(f"{sys.argv[1]}/*/*counter_collection.csv"))[-1]
rows = [r for r in csv.DictReader(open(f)) if sys.argv[2] in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
for r in rows:
    print(f'{int(r["Dispatch_Id"]):6d} grid {r["Grid_Size"]:>9s}  read {float(r["Counter_Value"]) * 2048 / 1e6:9.1f} MB  {r["Kernel_Name"][:70]}')

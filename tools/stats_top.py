"""Top kernels of a rocprofv3 --stats run:  python tools/stats_top.py OUT_DIR [n]"""
import csv
import glob
import sys

f = sorted(glob.glob(sys.argv[1] + "/**/*_kernel_stats.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print(f"total {tot / 1e6:.2f} ms over the run, {len(rows)} kernels")
for r in rows[:n]:
    print(f"{int(r['Calls']):5d} {int(r['TotalDurationNs']) / 1e6:9.2f} ms {float(r['AverageNs']) / 1e3:9.1f} us  {r['Name'][:120]}")

"""GPU box: raw start / end times of the long kernels of a rocprofv3 kernel trace (do launches of two streams overlap?).

    rocprofv3 --kernel-trace --output-format csv -d OUT -- python bench.py --legs c2_greedy_fp32 --steps 10 ...
    python tools/timeline_dump.py OUT [min_us] [last_n]
"""
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), r.get("Stream_Id", "?")))
rows.sort()
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
last = int(sys.argv[3]) if len(sys.argv) > 3 else 40
long_rows = [r for r in rows if (r[1] - r[0]) / 1e3 >= min_us and "hbm_read_probe" not in r[2]][-last:]
t0 = long_rows[0][0]
prev_end = t0
for s, e, name, q, st in long_rows:
    print(f"+{(s - t0) / 1e3:10.1f} .. +{(e - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:8.1f}  overlap_with_prev {max(0, prev_end - s) / 1e3:8.1f}  q{q} s{st}  {name[:60]}")
    prev_end = max(prev_end, e)

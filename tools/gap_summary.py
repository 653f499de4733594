"""GPU box: GPU-idle gaps of one bench leg from a rocprofv3 kernel trace (per-dispatch start / end timestamps).

    rocprofv3 --kernel-trace --output-format csv -d OUT -- python bench.py --legs c2_greedy --steps 10 ...
    python tools/gap_summary.py OUT           # prints the median timeline of one step: kernel, start offset, duration, gap before
"""
import csv
import glob
import statistics
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
anchor = sys.argv[2] if len(sys.argv) > 2 else "am_encoder"
starts = [i for i, r in enumerate(rows) if anchor in r[2]]
steps = [rows[a:b] for a, b in zip(starts[:-1], starts[1:])]
steps = steps[len(steps) // 3:]  # drop warm-up
n = statistics.mode(len(s) for s in steps)
steps = [s for s in steps if len(s) == n]
print(f"{len(steps)} steps of {n} dispatches; step period median {statistics.median(s[-1][1] - s[0][0] for s in steps) / 1e3:.1f} us (first start -> last end)")
period = statistics.median(b[0][0] - a[0][0] for a, b in zip(steps[:-1], steps[1:]))
busy = statistics.median(sum(e - s for s, e, _ in st) for st in steps)
print(f"period {period / 1e3:.1f} us, GPU busy {busy / 1e3:.1f} us, idle {100 * (1 - busy / period):.1f} %")
for k in range(n):
    off = statistics.median(st[k][0] - st[0][0] for st in steps) / 1e3
    dur = statistics.median(st[k][1] - st[k][0] for st in steps) / 1e3
    gap = statistics.median((st[k][0] - st[k - 1][1]) if k else 0 for st in steps) / 1e3
    print(f"  +{off:9.1f} us  {dur:9.1f} us  gap {gap:7.1f} us  {steps[0][k][2][:80]}")
last_gap = statistics.median(b[0][0] - a[-1][1] for a, b in zip(steps[:-1], steps[1:])) / 1e3
print(f"  gap from the last dispatch of a step to the next step's first: {last_gap:.1f} us")

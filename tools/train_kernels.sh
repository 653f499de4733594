#!/bin/bash
# Run ON THE GPU BOX (through gpurun): per-kernel time of the c4_train step (rocprofv3 --kernel-trace --stats over
# tools/train_steps.py), printed per step.   gpurun -- 'bash tools/train_kernels.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tp -- python $R/tools/train_steps.py 12 > /dev/null 2>&1
python - <<PY
import csv, glob
rows = list(csv.DictReader(open(glob.glob("/tmp/tp/*/*_kernel_stats.csv")[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel time per step (ms):", round(tot / 1e6 / 12, 3))
for r in rows[:${1:-22}]:
    print(r["Name"][:88].ljust(88), r["Calls"].rjust(5), f'{float(r["TotalDurationNs"]) / 1e6 / 12:7.3f} ms/step', f'{float(r["AverageNs"]) / 1e3:8.1f} us')
PY

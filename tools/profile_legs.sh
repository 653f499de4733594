#!/bin/bash
# Run ON THE GPU BOX (through gpurun): per-leg kernel trace (+ optional separate PMC passes) of bench.py.
#   gpurun -- 'bash tools/profile_legs.sh r02 "c2_greedy c3_greedy c5_sampling c4_train" pmc'
# Counters are collected in their own runs (no trace domains mixed with --pmc), one counter set per pass.
set -u
TAG=${1:-final}
LEGS=${2:-"c2_greedy c2_sampling c3_greedy c5_sampling c4_train"}
PMC=${3:-}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
for LEG in $LEGS; do
  O=$R/gpurun_out/$TAG/$LEG
  mkdir -p $O
  ARGS="--legs $LEG --steps 10 --warmup 2 --no-cpu-baseline --no-parity --no-large-batch --launch eager"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/bench.py $ARGS > $O/bench_trace.json 2> $O/bench_trace.err
  if [ -n "$PMC" ] && [ "$LEG" != "c4_train" ]; then
    timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $R/bench.py --legs $LEG --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-large-batch --launch eager > $O/bench_pmc_fetch.json 2> $O/bench_pmc_fetch.err
    timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $R/bench.py --legs $LEG --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-large-batch --launch eager > $O/bench_pmc_write.json 2> $O/bench_pmc_write.err
  fi
  # keep only the small summaries (the per-dispatch traces are tens of MB); the counter files lose the ~25 000 dispatches of
  # bench.py's one-second warm-up (hbm_read_probe_kernel) except the last ten — the 2 GiB calibration launches
  for f in $O/pmc_*/*/*_counter_collection.csv; do
    [ -f "$f" ] && python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
probe = [r for r in rows if "hbm_read_probe" in r["Kernel_Name"]]
keep = [r for r in rows if "hbm_read_probe" not in r["Kernel_Name"]] + probe[-10:]
with open(sys.argv[1], "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()) if rows else [])
    w.writeheader()
    w.writerows(keep)
PY
  done
  find $O -name "*_kernel_trace.csv" -delete
  find $O -name "*.db" -delete
  ls $O/trace/*/ 2>/dev/null | head
done
du -sh $R/gpurun_out/$TAG

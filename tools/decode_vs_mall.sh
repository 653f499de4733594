#!/bin/bash
# Run ON THE GPU BOX (gpurun): how much of the decode launch's "HBM" rate is Infinity-Cache (MALL, 256 MiB) hits?
#  The headline batch's planes are 315 MB and the set of rows still feasible shrinks below 256 MiB after
# ~20 steps; FETCH_SIZE counts MALL hits. Same kernel at B = 16 384: planes 1.26 GB, the feasible rows stay above 256 MiB
# for > 80 % of the steps. Per batch: kernel trace (duration), FETCH_SIZE pass, WRITE_SIZE pass (separate --pmc passes).
#   gpurun -- 'bash tools/decode_vs_mall.sh r06 "4096 16384"'  ->  gpurun_out/<tag>_mall/<B>/...; parse: tools/decode_vs_mall_parse.py
set -u
TAG=${1:-r06}
BATCHES=${2:-"4096 16384"}
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
for B in $BATCHES; do
  O=$R/gpurun_out/${TAG}_mall/$B
  mkdir -p $O
  ARGS="--legs c2_greedy --batch $B --steps 10 --warmup 2 --no-cpu-baseline --no-parity --launch eager --regions 1"
  timeout 300 python $R/bench.py $ARGS --detail $O/detail.json > $O/bench.json 2> $O/bench.err
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/bench.py $ARGS --detail $O/detail_trace.json > $O/bench_trace.json 2> $O/bench_trace.err
  timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $R/bench.py $ARGS --steps 3 --detail $O/detail_f.json > /dev/null 2> $O/pmc_fetch.err
  timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $R/bench.py $ARGS --steps 3 --detail $O/detail_w.json > /dev/null 2> $O/pmc_write.err
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
done
python3 $R/tools/decode_vs_mall_parse.py $R/gpurun_out/${TAG}_mall $R/gpurun_out/${TAG}_decode_vs_mall.json

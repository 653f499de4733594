#!/bin/bash
# Run ON THE GPU BOX (through gpurun): kernel trace + separate PMC passes of the default bench.
#   gpurun -- 'bash tools/profile_round.sh final'
# Counters are collected in their own runs (no trace domains mixed with --pmc).
set -u
TAG=${1:-final}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_trace.json 2> $O/bench_trace.err
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_pmc_fetch.json 2> $O/bench_pmc_fetch.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_pmc_write.json 2> $O/bench_pmc_write.err
ls $O

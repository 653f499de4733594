#!/bin/bash
# GPU box: training-kernel tests + the C4 training leg + headline
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05d
mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_teacher.py tests/test_gpu_norms_mlp_grad.py -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" > $O/summary.txt
timeout 600 python bench.py --legs c4_train --steps 10 --warmup 3 --no-cpu-baseline --no-parity > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?" >> $O/summary.txt
cat $O/summary.txt; tail -8 $O/tests.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05d/bench_line.json').read().strip().splitlines()[-1])
print('c2', d['ms_per_step'], d.get('region_ms_per_step'))
print({k:(v.get('ms_per_step'), v.get('ms_min_med')) for k,v in d.get('legs',{}).items() if isinstance(v,dict)})
PY

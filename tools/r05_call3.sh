#!/bin/bash
# GPU box, r05 call 3: encoder with batched constant staging / fixed-trip stores / one-instruction relu — tests, A/B, short bench
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05c
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_encoder.py tests/test_gpu_encoder_f32.py tests/test_gpu_norms_mlp_grad.py tests/test_gpu_teacher.py -x -q -m gpu -s > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt
grep -E "worst|passed|failed|Error|assert " $O/tests.log | tail -15
timeout 600 tools/enc_variants.sh run > $O/variants.log 2>&1; echo "variants rc=$?" | tee -a $O/summary.txt
grep -v "^$" $O/variants.log | sed 's/\/opt.*directory//' | paste - - | awk '{print $1, $4, $5}'
for l in old base; do echo "--- $l"; RL4CO_AMD_LIB=$R/tools/probes/_build/lib_$l.so timeout 300 python tools/enc_layers.py 2>&1 | grep layers; done | tee $O/layers.log
timeout 600 python bench.py --steps 20 --warmup 5 --legs c2_greedy,c4_train,c5_sampling --no-cpu-baseline --no-parity --detail $O/bench_detail.json > $O/bench_line.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/summary.txt
python -c "
import json; l=json.load(open('$O/bench_line.json')); print('c2', l['ms_per_step'], l.get('region_ms_per_step'), 'enc', l['encoder_roofline']['launch_ms_mean'], l['encoder_roofline']['frac'], 'dec', l['roofline']['launch_ms_mean'], {k:(v['ms_per_step'],v.get('ms_min_med')) for k,v in l['legs'].items()})"

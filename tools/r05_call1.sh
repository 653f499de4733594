#!/bin/bash
# GPU box, r05 call 1: encoder correctness on the branch-free GEMM build, variant A/B, layer sweep, PMC
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05a
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_encoder.py -x -q -m gpu > $O/test_encoder.log 2>&1; echo "test_encoder rc=$?" | tee -a $O/summary.txt
timeout 900 tools/enc_variants.sh run > $O/variants.log 2>&1; echo "variants rc=$?" | tee -a $O/summary.txt
for l in old base; do echo "--- $l"; RL4CO_AMD_LIB=$R/tools/probes/_build/lib_$l.so timeout 300 python tools/enc_layers.py; done > $O/layers.log 2>&1
timeout 900 tools/enc_pmc.sh r05a/pmc > $O/pmc.log 2>&1; echo "pmc rc=$?" | tee -a $O/summary.txt
tail -3 $O/test_encoder.log; cat $O/variants.log; cat $O/layers.log; cat $O/r05a/pmc/summary.txt 2>/dev/null || cat $R/gpurun_out/r05a/pmc/summary.txt

#!/bin/bash
# GPU box, r05 call 2: dual-instance encoder (correctness + variants), the new training parity tests (calibration run)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r05b
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_encoder.py -x -q -m gpu > $O/test_encoder.log 2>&1; echo "test_encoder rc=$?" | tee -a $O/summary.txt
tail -3 $O/test_encoder.log
timeout 900 tools/enc_variants.sh run > $O/variants.log 2>&1; echo "variants rc=$?" | tee -a $O/summary.txt
grep -v amdgpu.ids $O/variants.log | paste - - | awk '{print $1, $4, $5}'
timeout 600 python -m pytest tests/test_gpu_teacher.py -q -m gpu -s -k "oracle_cpu_encoder or per_tensor" > $O/test_train_parity.log 2>&1; echo "train parity rc=$?" | tee -a $O/summary.txt
grep -E "worst|passed|failed|Error|assert" $O/test_train_parity.log | head -40

#!/bin/bash
# GPU box: PMC passes over the REINFORCE training leg (bench.py --legs c4_train), counters only (no trace domains mixed
# in), one counter set per pass; then a kernel-trace pass of the same command. Summary: gpurun_out/<tag>/c4_train_pmc.json
#   gpurun -- 'bash tools/train_pmc.sh r03pmc'
set -u
TAG=${1:-trainpmc}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
ARGS="--legs c4_train --steps 3 --warmup 2 --no-cpu-baseline --no-parity"
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --output-format csv -d $O/p$i -- python $R/bench.py $ARGS > $O/p$i.json 2> $O/p$i.err
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/bench.py $ARGS > $O/trace.json 2> $O/trace.err
find $O -name "*_kernel_trace.csv" -delete
find $O -name "*.db" -delete
python3 $R/tools/train_pmc_parse.py $O
find $O -name "*counter_collection.csv" -size +2M -delete; du -sh $O

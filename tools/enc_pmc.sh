#!/bin/bash
# GPU box: PMC passes over the fused encoder micro-benchmark (tools/enc_bench.py); counters only, no trace domains.
set -u
#   tools/enc_pmc.sh TAG [bench script under tools/, default enc_bench.py] [kernel-name substring, default am_encoder]
TAG=${1:-encpmc}
SCRIPT=${2:-enc_bench.py}
KERNEL=${3:-am_encoder}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
i=0
for SET in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $SET --output-format csv -d $O/p$i -- python $R/tools/$SCRIPT > $O/p$i.log 2>&1
done
python3 - "$O" "$KERNEL" <<'PY' | tee $O/summary.txt
import csv,glob,sys,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1]+"/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]:
            agg[r["Counter_Name"]]["v"].append(float(r["Counter_Value"]))
for k,v in sorted(agg.items()):
    print(f"{k:36s} mean {sum(v['v'])/len(v['v']):16.1f}  n={len(v['v'])}")
PY
find $O -name "*counter_collection.csv" -size +2M -delete

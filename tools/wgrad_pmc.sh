#!/bin/bash
# GPU box: memory-side counters of the weight-gradient launches (tools/wgrad_one.py: one shape, a few launches)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/wgpmc
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "TCC_[A-Z0-9_]*\|TCP_[A-Z0-9_]*\|TA_[A-Z0-9_]*\|TD_[A-Z0-9_]*" | sort -u > $O/avail.txt
wc -l $O/avail.txt
i=0
for SET in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RD_UNCACHED_32B_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum" "TCC_EA0_RDREQ_LEVEL_sum TCC_TAG_STALL_sum TCC_BUSY_sum TCC_CYCLE_sum"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $SET --output-format csv -d $O/p$i -- python $R/tools/wgrad_one.py > $O/p$i.log 2>&1
  python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/p$i/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r.get("Kernel_Name","")
        for name in ("wgrad_bf16_kernel","linear_k128_n128","hbm_read_probe"):
            if name in k: acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    for c,vals in v.items(): print("set$i", k, c, "mean", sum(vals)/len(vals), "n", len(vals))
PY
done
find $O -name "*counter_collection.csv" -size +1M -delete

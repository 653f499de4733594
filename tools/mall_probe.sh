#!/bin/bash
# GPU box: memory-side counters of (a) a 2 GiB read, (b) a MALL-resident 128 MiB re-read, (c) the decode launch (tools/mall_probe.py)
#   gpurun -- 'bash tools/mall_probe.sh r06'  ->  gpurun_out/<tag>_mallpmc.txt
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${TAG}_mallpmc
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
i=0
for SET in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_CYCLE_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  for DT in bf16 f32; do
    timeout 300 rocprofv3 --pmc $SET --output-format csv -d $O/p${i}_$DT -- python $R/tools/mall_probe.py 4096 $DT > $O/p${i}_$DT.log 2>&1
    python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/p${i}_$DT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r.get("Kernel_Name","")
        for name in ("am_decode_kernel","hbm_read_probe"):
            if name in k: acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    for c,vals in v.items():
        if k == "hbm_read_probe":
            big=[x for x in vals[:5]]; small=[x for x in vals[5:17]]
            print("set$i $DT probe_2GiB", c, "mean", sum(big)/max(1,len(big)), "n", len(big))
            print("set$i $DT probe_128MiB_first", c, small[0] if small else None)
            print("set$i $DT probe_128MiB_rest", c, "mean", sum(small[2:])/max(1,len(small[2:])), "n", len(small[2:]))
        else:
            print("set$i $DT", k, c, "mean", sum(vals)/len(vals), "n", len(vals))
PY
  done
done | tee $R/gpurun_out/${TAG}_mallpmc.txt
find $O -name "*counter_collection.csv" -size +1M -delete
find $O -name "*.db" -delete

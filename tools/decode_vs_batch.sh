#!/bin/bash
# decode launch time against the batch size: is a half batch (planes MALL-resident: 2048 x 77 KB = 157 MB < 256 MiB) served faster than HBM?
mkdir -p gpurun_out
for b in 4096 3072 2048 1024 512; do
  python bench.py --legs c2_greedy --batch $b --steps 40 --warmup 5 --launch eager --no-parity --no-cpu-baseline --detail gpurun_out/dvb_$b.json > /dev/null 2> gpurun_out/dvb_$b.err
  python - <<PY
import json
d=json.load(open("gpurun_out/dvb_$b.json"))
r=d["roofline"]; e=d.get("encoder_roofline",{})
print("batch $b decode_ms %.4f achieved_GBs %.0f frac %.3f encoder_ms %.4f step_ms %.4f" % (r["launch_ms_mean"], r["achieved"], r["frac"], e.get("launch_ms_mean",0), d["ms_per_step"]))
PY
done

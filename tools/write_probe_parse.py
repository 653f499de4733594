#!/usr/bin/env python
"""Per-kernel WRITE_SIZE of tools/write_probe.py against the known byte counts: python tools/write_probe_parse.py gpurun_out/wp"""
import collections
import csv
import glob
import json
import sys

KNOWN = {"FillFunctor": (1 << 30, "torch fill, 1 GiB fp32"), "copyBuffer": (1 << 30, "device copy, 1 GiB"),
         "bfloat16_copy": (1 << 28, "fp32 -> bf16 cast, 256 MiB written"),
         "am_encoder": (3 * 4096 * 100 * 128 * 2 + 2 * 4096 * 100 * 128 * 4 + 4096 * 128 * 4, "fused encoder, TSP-100 x 4096")}
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/*/*_counter_collection.csv") + glob.glob(sys.argv[1] + "/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != "WRITE_SIZE":
            continue
        for key in KNOWN:
            if key in r["Kernel_Name"]:
                agg[key].append(float(r["Counter_Value"]))
out = {}
for key, vals in agg.items():
    nbytes, what = KNOWN[key]
    big = [v for v in vals if v * 1024 > 0.25 * nbytes]  # skip tiny launches of the same kernel template
    if not big:
        continue
    mean_kb = sum(big) / len(big)
    out[key] = {"what": what, "launches": len(big), "WRITE_SIZE_KB_mean": mean_kb, "algorithmic_bytes": nbytes,
                "counter_bytes_over_algorithmic": mean_kb * 1024 / nbytes}
print(json.dumps(out, indent=1))
